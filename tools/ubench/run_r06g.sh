# plain against agent-scope (sc1) stores in the tiled blocks kernels' F32 epilogues, same box, alternating
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
L=cyberether_amd/lib/libjetstream_hip.so
cp $L /tmp/base.so
for rep in 1 2; do for v in base tiled_sc1; do
  if [ $v = base ]; then cp /tmp/base.so $L; else cp cyberether_amd/lib/variants/$v.so $L; fi
  echo "== $v (run $rep)"
  python tools/bench_c5_streams.py fast | python -c "import sys,json; d=json.load(sys.stdin); print('  c5 fast   ', {k:{f:round(r['us_per_cycle'],2) for f,r in v.items()} for k,v in d.items() if k!='provider'})"
  python tools/bench_c5_streams.py generic | python -c "import sys,json; d=json.load(sys.stdin); print('  c5 generic', {k:{f:round(r['us_per_cycle'],2) for f,r in v.items()} for k,v in d.items() if k!='provider'})"
done; done
cp /tmp/base.so $L
python -m pytest tests/test_gpu_fft.py tests/test_gpu_full_sizes.py tests/test_gpu_chain.py tests/test_gpu_batch.py -q -m gpu -x 2>&1 | tail -2
