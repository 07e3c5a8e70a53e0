#!/bin/bash
# Round-2 evidence run on the GPU box (via gpurun): default bench line, rocprofv3 kernel stats and three separate PMC
# passes for the exact (headline) and the fast provider, the other BASELINE configs, and the ubench A/B of the fused kernel.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/prof_r02
mkdir -p $O
cd $ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_steps20.json 2>> $O/bench_default.err
python bench.py --provider fast --no-cpu-baseline --no-alt > $O/bench_fast.json 2>> $O/bench_default.err
BENCH_ARGS="" bash tools/profile_pmc.sh prof_r02/generic > /dev/null 2>&1
BENCH_ARGS="--provider fast" bash tools/profile_pmc.sh prof_r02/fast > /dev/null 2>&1
for v in generic fast; do
  python tools/pmc_summary.py $O/$v > $O/pmc_counters_$v.txt 2>&1
  python tools/kstats.py $O/$v/trace > $O/kernel_stats_$v.txt 2>&1
  cp $(ls $O/$v/trace/*/*kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats_$v.csv
done
python tools/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err
tail -n 3 $O/kernel_stats_generic.txt $O/kernel_stats_fast.txt
