#!/bin/bash
# Round 5, v: the tiled passes' twiddles requested BEFORE the pass's first barrier (JST_TILED_EARLY_TWIDDLES=1, the tree) against
# behind the butterfly (cyberether_amd/lib/variants/early0.so): parity suites first, then C3 / C5 / multi-fm alternately, same box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05v
mkdir -p $O
cd $ROOT
{
timeout 900 python -m pytest tests/test_gpu_fft.py tests/test_gpu_filter_block.py tests/test_gpu_full_sizes.py tests/test_gpu_chain_fusions.py tests/test_gpu_reference_flowgraphs.py -x -q 2>&1 | tail -2
cp cyberether_amd/lib/libjetstream_hip.so $O/base.so
for rep in 1 2 3; do
  for v in early late; do
    if [ $v = late ]; then cp cyberether_amd/lib/variants/early0.so cyberether_amd/lib/libjetstream_hip.so; else cp $O/base.so cyberether_amd/lib/libjetstream_hip.so; fi
    c3=$(python tools/bench_configs.py C3 2>/dev/null | head -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.readline())['ms_per_cycle'],4))")
    c5=$(python tools/bench_configs.py C5 2>/dev/null | python -c "
import json,sys
print(' '.join(str(round(json.loads(l)['us_per_cycle'],2)) for l in sys.stdin if l.startswith('{')))")
    mf=$(python tools/bench_multi_fm.py 2>/dev/null | tail -1 | python -c "import json,sys; print(round(json.loads(sys.stdin.readline())['us_per_cycle'],1))")
    echo "== $v: C3 $c3 ms | C5 per-cycle / ring per-cycle / batched $c5 us | multi-fm $mf us"
  done
done
cp $O/base.so cyberether_amd/lib/libjetstream_hip.so; rm -f $O/base.so
} 2>&1 | tee $O/log.txt
