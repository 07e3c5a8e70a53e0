ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_surfaces.py tests/test_gpu_full_sizes.py tests/test_gpu_tiled_persistent.py tests/test_gpu_batch.py tests/test_gpu_reference_flowgraphs.py -q -m gpu -x 2>&1 | tail -4
for v in fast generic; do python tools/bench_c5_streams.py $v | python -c "import sys,json; d=json.load(sys.stdin); print('  c5 $v', {k:{f:round(r['us_per_cycle'],2) for f,r in v.items()} for k,v in d.items() if k!='provider'})"; done
cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06l; mkdir -p $O
for f in per_cycle batched; do timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$f -- python $ROOT/tools/bench_c5_streams.py fast 128 $f > /dev/null 2>&1; python $ROOT/tools/kstats.py $O/$f | grep -i lineplot; done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/b16 -- python $ROOT/tools/bench_c5_streams.py fast 16 batched > /dev/null 2>&1; python $ROOT/tools/kstats.py $O/b16 | grep -i lineplot
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
