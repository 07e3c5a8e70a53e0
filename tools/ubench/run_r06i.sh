ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r06i; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
L=$ROOT/cyberether_amd/lib/libjetstream_hip.so
cp $L /tmp/base.so
for v in base tiled_nopersist; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp $ROOT/cyberether_amd/lib/variants/$v.so $L; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- python $ROOT/tools/bench_c5_streams.py fast 128 batched > $O/$v.json 2> $O/$v.err
  python $ROOT/tools/kstats.py $O/$v > $O/kstats_$v.txt 2>&1
  echo "== $v"; cat $O/$v.json; head -4 $O/kstats_$v.txt
done
cp /tmp/base.so $L
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
