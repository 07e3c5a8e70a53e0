#!/bin/bash
# round 4, call u: streaming (nt) loads of the tiled FFT kernels' once-read data (padded input, scratch image) against plain loads:
# configs 3 and 5, same box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04u
mkdir -p $O
cd $ROOT
LIB=cyberether_amd/lib/libjetstream_hip.so
cp $LIB /tmp/base.so
for v in stream plainloads stream plainloads; do
  if [ $v = stream ]; then cp /tmp/base.so $LIB; else cp cyberether_amd/lib/variants/$v.so $LIB; fi
  python tools/bench_configs.py C3 C5 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    c=json.loads(ln); print('$v', c['config'][:58], {k:round(v,4) for k,v in c.items() if k in ('us_per_cycle','ms_per_cycle')})" | tee -a $O/ab.log
done
cp /tmp/base.so $LIB
