# persistent blocks kernel against one workgroup per tile (both with LDS-resident block twiddles), same box, alternating:
# config 5 (fast / generic, 16 and 128 transforms), config 3 bit-exact chain, multi-fm.yml
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r06g; mkdir -p $O
cd $ROOT
L=cyberether_amd/lib/libjetstream_hip.so
cp $L /tmp/persist.so
for rep in 1 2; do for v in persist tiled_nopersist; do
  if [ $v = persist ]; then cp /tmp/persist.so $L; else cp cyberether_amd/lib/variants/$v.so $L; fi
  echo "== $v (run $rep)"
  python tools/bench_c5_streams.py fast | python -c "import sys,json; d=json.load(sys.stdin); print('  c5 fast   ', {k:{f:round(r['us_per_cycle'],2) for f,r in v.items()} for k,v in d.items() if k!='provider'})"
  python tools/bench_c5_streams.py generic | python -c "import sys,json; d=json.load(sys.stdin); print('  c5 generic', {k:{f:round(r['us_per_cycle'],2) for f,r in v.items()} for k,v in d.items() if k!='provider'})"
  python tools/bench_configs.py C3 2>/dev/null | python -c "import sys,json; [print('  ', d['config'][:30], round(d['ms_per_cycle'],4)) for d in map(json.loads, sys.stdin)]"
  python tools/bench_multi_fm.py 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  multi-fm', {k:v for k,v in d.items() if 'us' in k or 'launch' in k})"
done; done
cp /tmp/persist.so $L
