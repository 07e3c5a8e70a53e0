#!/bin/bash
# round 4, call t: cache-policy variants of the fused side kernel on top of the adopted set (loads nt, value stores sc1|nt, index
# stores sc1), ring period 32: ld18 = loads sc1|nt, ld3 = loads sc0|nt, ld0 = plain loads, st19 = value stores sc0|sc1|nt,
# side18 = index stores sc1|nt.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04t
mkdir -p $O
cd $ROOT
LIB=cyberether_amd/lib/libjetstream_hip.so
cp $LIB /tmp/base.so
for v in base ld18 ld3 ld0 st19 side18 base; do
  if [ $v = base ]; then cp /tmp/base.so $LIB; else cp cyberether_amd/lib/variants/$v.so $LIB; fi
  python bench.py --no-cpu-baseline --no-host-fed --no-configs --no-alt 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'step_us', round(b['ms_per_step']*1e3,3), 'value', round(b['value']), 'kernel_us', round(b['roofline']['kernel_ms']*1e3,2), 'frac', round(b['roofline']['frac'],4), 'step_frac', round(b['roofline']['step_frac'],4), 'parity', b['parity']['bit_exact'])" | tee -a $O/ab.log
done
cp /tmp/base.so $LIB
