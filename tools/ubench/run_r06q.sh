# lane table filled while the tile's loads are in flight + no 64-bit division for one batch axis, against the previous build; same box, alternating
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
L=cyberether_amd/lib/libjetstream_hip.so
cp $L /tmp/base.so
timeout 600 python -m pytest tests/test_gpu_fft.py tests/test_gpu_full_sizes.py tests/test_gpu_tiled_persistent.py tests/test_gpu_filter_modules.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2 3; do for v in base tiled_prev; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp cyberether_amd/lib/variants/$v.so $L; fi
  echo "== $v (run $rep)"
  python tools/bench_c5_streams.py fast | python -c "import sys,json; d=json.load(sys.stdin); print('  c5 fast', {k:{f:round(r['us_per_cycle'],2) for f,r in v.items()} for k,v in d.items() if k!='provider'})"
  python tools/bench_configs.py C3 2>/dev/null | python -c "import sys,json; [print('  ', d['config'][:30], round(d['ms_per_cycle'],4)) for d in map(json.loads, sys.stdin)]"
  python tools/bench_multi_fm.py 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  multi-fm', round(d['us_per_cycle'],2))"
done; done
cp /tmp/base.so $L
