# pad-free blocks tile (pitch CB, 16 x 4 arrival pieces) against the odd pitch, same box, alternating
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
L=cyberether_amd/lib/libjetstream_hip.so
cp $L /tmp/base.so
timeout 600 python -m pytest tests/test_gpu_fft.py tests/test_gpu_full_sizes.py tests/test_gpu_tiled_persistent.py tests/test_gpu_batch.py -q -m gpu -x 2>&1 | tail -2
for rep in 1 2; do for v in base tiled_pad; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp cyberether_amd/lib/variants/$v.so $L; fi
  echo "== $v (run $rep)"
  for pv in fast generic; do python tools/bench_c5_streams.py $pv | python -c "import sys,json; d=json.load(sys.stdin); print('  c5 $pv', {k:{f:round(r['us_per_cycle'],2) for f,r in v.items()} for k,v in d.items() if k!='provider'})"; done
done; done
cp /tmp/base.so $L
