#!/bin/bash
# Round 5, n: rocprofv3 --kernel-trace --stats of the driver-form launch shape (ring period 16), dynamic and static hand-out.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05n
mkdir -p $O/fast16 $O/fast16s $O/fast32
cd /tmp && export TMPDIR=/tmp
B16="python $ROOT/bench.py --slots 16 --no-cpu-baseline --no-alt --no-parity --no-configs --no-host-fed --min-time 0.05"
B32="python $ROOT/bench.py --no-cpu-baseline --no-alt --no-parity --no-configs --no-host-fed --min-time 0.05"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast16/trace -- $B16 > $O/fast16/trace.log 2>&1
python $ROOT/tools/kstats.py $O/fast16/trace > $O/kernel_stats_fast_period16.txt 2>&1
JST_QUAD_STATIC=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast16s/trace -- $B16 > $O/fast16s/trace.log 2>&1
python $ROOT/tools/kstats.py $O/fast16s/trace > $O/kernel_stats_fast_period16_static.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast32/trace -- $B32 > $O/fast32/trace.log 2>&1
python $ROOT/tools/kstats.py $O/fast32/trace > $O/kernel_stats_fast.txt 2>&1
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
head -3 $O/kernel_stats_fast_period16.txt $O/kernel_stats_fast_period16_static.txt $O/kernel_stats_fast.txt
grep -h '^{' $O/fast16/trace.log | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l); print('under rocprof: ', round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step, kernel_ms', d['roofline']['kernel_ms'])"
