#!/bin/bash
# Round 3, experiment t: write-through (sc1) scratch stores of the tiled columns kernel, configs 3 and 5
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03t
mkdir -p $O
cp $ROOT/cyberether_amd/lib/libjetstream_hip.so $O/base.so
cd $ROOT
run() { for i in 1 2; do python tools/bench_configs.py C5 C3 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$1', d['config'][:12], d.get('us_per_cycle') or d.get('ms_per_cycle'))"; done; }
run base
cp $ROOT/cyberether_amd/lib/variants/tiled_sc1.so $ROOT/cyberether_amd/lib/libjetstream_hip.so
run tiled_sc1
cp $O/base.so $ROOT/cyberether_amd/lib/libjetstream_hip.so
run base2
rm -f $O/base.so
