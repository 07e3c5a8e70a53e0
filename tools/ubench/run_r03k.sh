#!/bin/bash
# round-3 batch K: lane counts of the specialised tiled kernels (static plans 4-9) by rocprofv3 kernel stats
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03k
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() { # name, config, env...
  name=$1; cfg=$2; shift 2
  env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- python $ROOT/tools/bench_configs.py $cfg > $O/$name.log 2>&1
  echo "== $name ($*)"; python $ROOT/tools/kstats.py $O/$name | grep -E "fft_tile|lineplot" | head -6
}
run c5_default C5 JST_X=0
run c5_ca16 C5 JST_TILED_CA=16
run c5_ca32 C5 JST_TILED_CA=32
run c5_cb16 C5 JST_TILED_CB=16
run c3_default C3 JST_X=0
run c3_cb4 C3 JST_TILED_CB=4
run c3_ca32 C3 JST_TILED_CA=32
run c3_ca8 C3 JST_TILED_CA=8
