ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/c5ab; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in generic fast; do for b in 128 16; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v$b -- python $ROOT/tools/bench_c5_streams.py $v $b > $O/$v$b.json 2> $O/$v$b.err
  python $ROOT/tools/kstats.py $O/$v$b > $O/kstats_$v$b.txt 2>&1
  echo "== $v $b"; cat $O/$v$b.json; head -5 $O/kstats_$v$b.txt
done; done
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
