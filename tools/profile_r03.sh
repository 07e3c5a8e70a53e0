#!/bin/bash
# Round-3 evidence run on the GPU box (via gpurun): the default bench line (with parity stamp, host_fed per sample
# format, cpu_baseline incl. configs[0]), the --steps 20 and --provider fast lines, rocprofv3 kernel stats and the
# separate PMC passes for the exact (headline) and the fast provider, the HBM-traffic file with its provenance
# (tools/pmc_summary.py --traffic-json), the other BASELINE configs and rocprofv3 stats of configs 3 and 5.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/prof_r03
mkdir -p $O
cd $ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_steps20.json 2>> $O/bench_default.err
python bench.py --provider generic --no-cpu-baseline --no-alt > $O/bench_generic.json 2>> $O/bench_default.err
BENCH_ARGS="--provider generic" bash tools/profile_pmc.sh prof_r03/generic > /dev/null 2>&1
BENCH_ARGS="--provider fast" bash tools/profile_pmc.sh prof_r03/fast > /dev/null 2>&1
for v in generic fast; do
  python tools/pmc_summary.py $O/$v > $O/pmc_counters_$v.txt 2>&1
  python tools/kstats.py $O/$v/trace > $O/kernel_stats_$v.txt 2>&1
  cp $(ls $O/$v/trace/*/*kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats_$v.csv
done
rm -f $O/pmc_traffic.json
python tools/pmc_summary.py $O/generic --traffic-json $O/pmc_traffic.json --provider generic > $O/pmc_traffic.log 2>&1
python tools/pmc_summary.py $O/fast --traffic-json $O/pmc_traffic.json --provider fast >> $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json $ROOT/profiles/pmc_traffic.json   # so that the default line below quotes this run's own counters
python bench.py --no-cpu-baseline --no-alt --no-parity > $O/bench_default_with_traffic.json 2>> $O/bench_default.err
python tools/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err
cd /tmp && export TMPDIR=/tmp
for c in C3 C5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg_$c -- python $ROOT/tools/bench_configs.py $c > $O/cfg_$c.log 2>&1
  python $ROOT/tools/kstats.py $O/cfg_$c > $O/kernel_stats_config_$c.txt 2>&1
done
cd $ROOT
tail -n 4 $O/kernel_stats_generic.txt $O/kernel_stats_fast.txt $O/kernel_stats_config_C3.txt $O/kernel_stats_config_C5.txt
cat $O/pmc_traffic.log; tail -c 600 $O/bench_default.err
python - <<'PY'
import json
for f in ('bench_default','bench_steps20','bench_generic','bench_default_with_traffic'):
    try:
        d=json.loads(open(f'gpurun_out/prof_r03/{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2),'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2), 'frac', round(d['roofline']['frac'],4), 'step_frac', round(d['roofline']['step_frac'],4), 'parity', d['parity'].get('bit_exact'), 'traffic', d['roofline']['traffic'], d['config']['provider'])
    except Exception as e: print(f, 'parse failed', e)
PY
