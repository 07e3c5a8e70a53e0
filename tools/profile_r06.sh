#!/bin/bash
# Round-6 evidence run (via gpurun, ONE box): the default bench line (parity stamp, alt_per_cycle_launch,
# alt_provider, host_fed, cpu_baseline), the --steps 20 and --provider generic lines, rocprofv3 kernel stats and the separate
# PMC passes for both providers, the HBM-traffic file with its provenance (cycles_per_launch = 32: the ring period of the default line), the default line again
# quoting that traffic.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/prof_r06
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
cd $ROOT
rm -f $O/pmc_traffic.json
python tools/pmc_summary.py $O/fast20 --traffic-json $O/pmc_traffic.json --provider fast --cycles 20 > $O/pmc_traffic20.log 2>&1
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
# round 6: provider fast of config 3 on the matrix cores (MFMA form against the direct form, same run) + its MFMA / VALU instruction counts
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/fir -- python $ROOT/tools/bench_fir.py > $O/fir.json 2> $O/fir.err
python $ROOT/tools/kstats.py $O/fir > $O/kernel_stats_config_C3_fast.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES -d $O/fir_pmc -- python $ROOT/tools/bench_fir.py > /dev/null 2> $O/fir_pmc.err
python $ROOT/tools/pmc_kernel_means.py $O/fir_pmc fir_ > $O/pmc_counters_config_C3_fast.txt 2>&1
# round 6: config 5 with all 8 streams resident (128 transforms per cycle): both providers, every form (one JSON each), and kernel stats of ONE
# form per trace (one launch per unit and cycle; cycle-batched spans of a 4-slot ring) for provider fast and generic
python $ROOT/tools/bench_c5_streams.py generic > $O/c5_streams.json 2> $O/c5_streams.err
python $ROOT/tools/bench_c5_streams.py fast > $O/c5_streams_fast.json 2>> $O/c5_streams.err
for v in fast generic; do for f in per_cycle batched; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c5s_${v}_$f -- python $ROOT/tools/bench_c5_streams.py $v 128 $f > /dev/null 2>> $O/c5_streams.err
  python $ROOT/tools/kstats.py $O/c5s_${v}_$f > $O/kernel_stats_config_C5_streams_${v}_$f.txt 2>&1
done; done
# round 6: the REFERENCE's scheduler on DeviceType::HIP: kernel + memory-copy trace of N steady-state cycles at two values of N --
# the number of copies must not depend on N (tensors stay in HBM between modules); and its bench object
for n in 10 110; do
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $O/refdev_$n -- python $ROOT/tools/reference_device_trace.py --cycles $n > $O/refdev_$n.json 2> $O/refdev_$n.err
done
python $ROOT/tools/memcpy_count.py $O/refdev_10 $O/refdev_110 > $O/reference_device_memcpy.txt 2>&1
python $ROOT/tools/reference_driven_bench.py > $O/reference_driven.json 2> $O/reference_driven.err
cd $ROOT
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +8M -delete
head -n 4 $O/kernel_stats_generic.txt $O/kernel_stats_fast.txt $O/kernel_stats_fast_period20.txt $O/kernel_stats_fast_period16.txt $O/kernel_stats_fast_period16_pipe.txt $O/kernel_stats_fast_period16_static.txt
cat $O/pmc_traffic.log; tail -c 400 $O/bench_default.err
python - <<'PY'
import json
for f in ('bench_default','bench_steps20','bench_generic','bench_per_cycle','bench_default_with_traffic','bench_driver_form'):
    try:
        d=json.loads(open(f'gpurun_out/prof_r06/{f}.json').read().strip().splitlines()[-1])
        print(f, round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2),'us/step kernel', round(d['roofline']['kernel_ms']*1e3,2), 'cycles/launch', d['roofline']['cycles_per_launch'], 'frac', round(d['roofline']['frac'],4), 'step_frac', round(d['roofline']['step_frac'],4), 'parity', d['parity'].get('bit_exact'), 'traffic', d['roofline']['traffic'], d['config']['provider'])
    except Exception as e: print(f, 'parse failed', e)
PY
