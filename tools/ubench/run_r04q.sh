#!/bin/bash
# round 4, call q: same-box A/B of the leapfrog order: rev = wavefronts 0..3 lifted first and 4..7 (the ones that arrive last at barrier 0) second; rev8 / fwd8 = one group lifted through the whole epilogue (4..7 / 0..3 ... see JST_EPI_LEAPFROG=8)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04q
mkdir -p $O
cd $ROOT
LIB=cyberether_amd/lib/libjetstream_hip.so
cp $LIB /tmp/base.so
for round in 1 2; do
  for v in base rev rev8 fwd8; do
    if [ $v = base ]; then cp /tmp/base.so $LIB; else cp cyberether_amd/lib/variants/$v.so $LIB; fi
    python bench.py --no-cpu-baseline --no-host-fed --no-configs 2>/dev/null | python -c "
import sys,json
b=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', 'step_us', round(b['ms_per_step']*1e3,3), 'kernel_us', round(b['roofline']['kernel_ms']*1e3,2), 'frac', round(b['roofline']['frac'],4), 'parity', b['parity']['bit_exact'], 'alt_step_us', round(b['alt_per_cycle_launch']['ms_per_step']*1e3,3))" | tee -a $O/ab.log
  done
done
cp /tmp/base.so $LIB
