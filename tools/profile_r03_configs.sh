#!/bin/bash
# The configs part of tools/profile_r03.sh alone (the other BASELINE configs + rocprofv3 stats of configs 3 and 5).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/prof_r03
mkdir -p $O
cd $ROOT
python tools/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err
cd /tmp && export TMPDIR=/tmp
for c in C3 C5; do
  rm -rf $O/cfg_$c
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg_$c -- python $ROOT/tools/bench_configs.py $c > $O/cfg_$c.log 2>&1
  python $ROOT/tools/kstats.py $O/cfg_$c > $O/kernel_stats_config_$c.txt 2>&1
done
head -8 $O/kernel_stats_config_C3.txt $O/kernel_stats_config_C5.txt
python - <<'PY'
import json
for ln in open('/root/repo/gpurun_out/prof_r03/bench_configs.jsonl'):
    c=json.loads(ln); print(c['config'][:40], {k:round(c[k],4) for k in c if k in ('us_per_cycle','ms_per_cycle')}, round(c.get('roofline',{}).get('frac',0),4))
PY
