ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_fft.py tests/test_gpu_full_sizes.py tests/test_gpu_filter_modules.py tests/test_gpu_reference_flowgraphs.py tests/test_gpu_tiled_persistent.py tests/test_gpu_batch.py -q -m gpu -x 2>&1 | tail -8
for v in fast generic; do python tools/bench_c5_streams.py $v | python -c "import sys,json; d=json.load(sys.stdin); print('  c5 $v', {k:{f:round(r['us_per_cycle'],2) for f,r in v.items()} for k,v in d.items() if k!='provider'})"; done
python tools/bench_configs.py C3 2>/dev/null | python -c "import sys,json; [print('  ', d['config'][:30], round(d['ms_per_cycle'],4)) for d in map(json.loads, sys.stdin)]"
python tools/bench_multi_fm.py 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  multi-fm', {k:v for k,v in d.items() if 'us' in k or 'launch' in k})"
