ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r06k; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
prof() { # name, command...
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -- "$@" > $O/$name.json 2> $O/$name.err
  python $ROOT/tools/kstats.py $O/$name > $O/kstats_$name.txt 2>&1
  echo "== $name"; head -${KLINES:-5} $O/kstats_$name.txt
}
prof c5_fast_128_pc python $ROOT/tools/bench_c5_streams.py fast 128 per_cycle
prof c5_fast_128_b python $ROOT/tools/bench_c5_streams.py fast 128 batched
prof c5_fast_16_b python $ROOT/tools/bench_c5_streams.py fast 16 batched
KLINES=7 prof c3 python $ROOT/tools/bench_configs.py C3
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
