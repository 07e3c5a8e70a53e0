#!/bin/bash
# Round-5 evidence run (via gpurun, ONE box): the default bench line (parity stamp, alt_per_cycle_launch,
# alt_provider, host_fed, cpu_baseline), the --steps 20 and --provider generic lines, rocprofv3 kernel stats and the separate
# PMC passes for both providers, the HBM-traffic file with its provenance (cycles_per_launch = 32: the ring period of the default line), the default line again
# quoting that traffic.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/prof_r05
mkdir -p $O
cd $ROOT
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/bench_steps20.json 2>> $O/bench_default.err
python bench.py --provider generic --no-cpu-baseline --no-alt > $O/bench_generic.json 2>> $O/bench_default.err
python bench.py --no-batch --no-cpu-baseline --no-alt > $O/bench_per_cycle.json 2>> $O/bench_default.err
cd /tmp && export TMPDIR=/tmp
for v in fast generic; do
  B="python $ROOT/bench.py --provider $v --no-cpu-baseline --no-alt --no-parity --min-time 0.05"
  mkdir -p $O/$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v/trace -- $B > $O/$v/trace.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/$v/pmc_fetch -- $B > $O/$v/pmc_fetch.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/$v/pmc_write -- $B > $O/$v/pmc_write.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $O/$v/pmc_sq -- $B > $O/$v/pmc_sq.log 2>&1
  python $ROOT/tools/pmc_summary.py $O/$v > $O/pmc_counters_$v.txt 2>&1
  python $ROOT/tools/kstats.py $O/$v/trace > $O/kernel_stats_$v.txt 2>&1
  cp $(ls $O/$v/trace/*/*kernel_stats.csv | head -1) $O/rocprofv3_kernel_stats_$v.csv
  grep -E '^\{' $O/$v/trace.log | tail -1 > $O/bench_under_rocprofv3_$v.json   # the line the profiled run itself printed
done
# the launch form of the driver's run (--steps 20: ring period 20, one 20-cycle launch per unit and region), provider fast;
# and the round-4 form (ring period 16) for the round-over-round comparison
B20="python $ROOT/bench.py --slots 20 --no-cpu-baseline --no-alt --no-parity --no-configs --no-host-fed --min-time 0.05"
mkdir -p $O/fast20
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast20/trace -- $B20 > $O/fast20/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fast20/pmc_fetch -- $B20 > $O/fast20/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/fast20/pmc_write -- $B20 > $O/fast20/pmc_write.log 2>&1
python $ROOT/tools/kstats.py $O/fast20/trace > $O/kernel_stats_fast_period20.txt 2>&1
B16="python $ROOT/bench.py --slots 16 --no-cpu-baseline --no-alt --no-parity --no-configs --no-host-fed --min-time 0.05"
mkdir -p $O/fast16
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast16/trace -- $B16 > $O/fast16/trace.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $O/fast16/pmc_fetch -- $B16 > $O/fast16/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $O/fast16/pmc_write -- $B16 > $O/fast16/pmc_write.log 2>&1
python $ROOT/tools/kstats.py $O/fast16/trace > $O/kernel_stats_fast_period16.txt 2>&1
# the same launch form on fft_pipe_kernel (round 4's kernel) and with the static round robin: same box, same profiler
for alt in pipe static; do
  E="JST_FFT_KERNEL=pipe"; [ $alt = static ] && E="JST_QUAD_STATIC=1"
  mkdir -p $O/fast16_$alt
  env $E timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fast16_$alt/trace -- $B16 > $O/fast16_$alt/trace.log 2>&1
  python $ROOT/tools/kstats.py $O/fast16_$alt/trace > $O/kernel_stats_fast_period16_$alt.txt 2>&1
done
cd $ROOT
rm -f $O/pmc_traffic.json
python tools/pmc_summary.py $O/fast16 --traffic-json $O/pmc_traffic.json --provider fast --cycles 16 > $O/pmc_traffic16.log 2>&1
python tools/pmc_summary.py $O/fast20 --traffic-json $O/pmc_traffic.json --provider fast --cycles 20 >> $O/pmc_traffic16.log 2>&1
python tools/pmc_summary.py $O/generic --traffic-json $O/pmc_traffic.json --provider generic --cycles 32 > $O/pmc_traffic.log 2>&1
python tools/pmc_summary.py $O/fast --traffic-json $O/pmc_traffic.json --provider fast --cycles 32 >> $O/pmc_traffic.log 2>&1
cp $O/pmc_traffic.json $ROOT/profiles/pmc_traffic.json   # so that the default line below quotes this run's own counters
python bench.py --no-cpu-baseline --no-alt --no-parity > $O/bench_default_with_traffic.json 2>> $O/bench_default.err
python bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>> $O/bench_default.err   # the driver's invocation, quoting fast@20
# the other BASELINE configs (each with its roofline object) and rocprofv3 kernel stats of configs 3 and 5 and of multi-fm.yml
python tools/bench_configs.py > $O/bench_configs.jsonl 2> $O/bench_configs.err
python tools/bench_multi_fm.py 400 > $O/multi_fm.json 2>> $O/bench_configs.err
cd /tmp
for c in C3 C5; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg_$c -- python $ROOT/tools/bench_configs.py $c > $O/cfg_$c.log 2>&1
  python $ROOT/tools/kstats.py $O/cfg_$c > $O/kernel_stats_config_$c.txt 2>&1
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mfm -- python $ROOT/tools/bench_multi_fm.py 200 > $O/mfm.log 2>&1
python $ROOT/tools/kstats.py $O/mfm > $O/kernel_stats_multi_fm.txt 2>&1
cd $ROOT
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
head -n 4 $O/kernel_stats_generic.txt $O/kernel_stats_fast.txt $O/kernel_stats_fast_period20.txt $O/kernel_stats_fast_period16.txt $O/kernel_stats_fast_period16_pipe.txt $O/kernel_stats_fast_period16_static.txt
cat $O/pmc_traffic.log; tail -c 400 $O/bench_default.err
python - <<'PY'
import json
for f in ('bench_default','bench_steps20','bench_generic','bench_per_cycle','bench_default_with_traffic','bench_driver_form'):
    try:
        d=json.loads(open(f'gpurun_out/prof_r05/{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2),'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2), 'cycles/launch', d['roofline']['cycles_per_launch'], 'frac', round(d['roofline']['frac'],4), 'step_frac', round(d['roofline']['step_frac'],4), 'parity', d['parity'].get('bit_exact'), 'traffic', d['roofline']['traffic'], d['config']['provider'])
    except Exception as e: print(f, 'parse failed', e)
PY
