#!/bin/bash
# Round 5, k: full GPU suite on the quad kernel build, then the driver-form bench under rocprofv3 --kernel-trace --stats.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05k
mkdir -p $O
cd $ROOT
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 | tee $O/pytest_gpu.txt
cd /tmp && export TMPDIR=/tmp
B16="python $ROOT/bench.py --slots 16 --no-cpu-baseline --no-alt --no-parity --no-configs --no-host-fed --min-time 0.05"
mkdir -p $O/fast16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast16/trace -- $B16 > $O/fast16/trace.log 2>&1
python $ROOT/tools/kstats.py $O/fast16/trace > $O/kernel_stats_fast_period16.txt 2>&1
JST_QUAD_STATIC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast16s/trace -- $B16 > $O/fast16s/trace.log 2>&1
python $ROOT/tools/kstats.py $O/fast16s/trace > $O/kernel_stats_fast_period16_static.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
head -8 $O/kernel_stats_fast_period16.txt $O/kernel_stats_fast_period16_static.txt
