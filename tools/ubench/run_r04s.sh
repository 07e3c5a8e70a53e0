#!/bin/bash
# round 4, call s: ring periods of 16 and 32 slots x cache policy of the VALUE stores (base sc1; vnt = nt; vsc1nt = sc1 | nt), the
# row-index stores stay sc1: does a 32-cycle launch keep its fused-kernel gain when the values stop evicting the indices?
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04s
mkdir -p $O
cd $ROOT
LIB=cyberether_amd/lib/libjetstream_hip.so
cp $LIB /tmp/base.so
for v in base vnt vsc1nt; do
  if [ $v = base ]; then cp /tmp/base.so $LIB; else cp cyberether_amd/lib/variants/$v.so $LIB; fi
  for s in 16 32; do
    python bench.py --slots $s --no-cpu-baseline --no-host-fed --no-configs --no-alt 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v slots', $s, 'step_us', round(b['ms_per_step']*1e3,3), 'value', round(b['value']), 'kernel_us', round(b['roofline']['kernel_ms']*1e3,2), 'cycles/launch', b['roofline']['cycles_per_launch'], 'frac', round(b['roofline']['frac'],4), 'step_frac', round(b['roofline']['step_frac'],4), 'parity', b['parity']['bit_exact'])" | tee -a $O/ab.log
  done
done
cp /tmp/base.so $LIB
