#!/bin/bash
# round 4, call k: config 3 with the fold operand requested in front of the passes (A/B against call j's timeline), workgroup
# size A/B of the blocks kernel, full GPU suite (cycle batching is the Python runtime's default now).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04k
mkdir -p $O
cd $ROOT
timeout 120 tools/ubench/bin/tiled_timeline > $O/tiled_timeline_c3.log 2>&1; tail -6 $O/tiled_timeline_c3.log
for tb in default 640 512; do
  if [ $tb = default ]; then unset JST_TILED_TB; else export JST_TILED_TB=$tb; fi
  python tools/bench_configs.py C3 2>/dev/null | python -c "
import sys,json
for ln in sys.stdin:
    c=json.loads(ln); print('TB=$tb', c['config'][:30], round(c.get('ms_per_cycle',0),4))"
done
unset JST_TILED_TB
timeout 1500 python -m pytest tests -q -m gpu > $O/pytest_gpu_full.log 2>&1; echo "full rc=$?"; tail -5 $O/pytest_gpu_full.log
cd /tmp && export TMPDIR=/tmp
rm -rf $O/cfg_C3
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg_C3 -- python $ROOT/tools/bench_configs.py C3 > $O/cfg_C3.log 2>&1
python $ROOT/tools/kstats.py $O/cfg_C3 > $O/kernel_stats_config_C3.txt 2>&1; head -8 $O/kernel_stats_config_C3.txt
find $O/cfg_C3 -name "*.csv" ! -name "*stats*" -delete
