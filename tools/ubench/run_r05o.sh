#!/bin/bash
# Round 5, o: chain fusions (fft_windowed, amplitude_range, one-launch AGC): tests, then multi-fm.yml per cycle with and without
# (JST_NO_CHAIN_FUSION=1 JST_AGC_THREE_KERNELS=1), same box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r05o
mkdir -p $O
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_chain_fusions.py tests/test_gpu_reference_flowgraphs.py tests/test_gpu_ingest_modules.py tests/test_gpu_flowgraph.py -x -q 2>&1 | tail -8
for rep in 1 2; do
python tools/bench_multi_fm.py 400 > $O/multi_fm_fused.json 2> $O/err1.txt; python -c "import json;d=json.load(open('$O/multi_fm_fused.json'));print('fused   :', round(d['us_per_cycle'],1),'us per cycle,', len([u for u in d['units']]), 'units')"
JST_NO_CHAIN_FUSION=1 JST_AGC_THREE_KERNELS=1 python tools/bench_multi_fm.py 400 > $O/multi_fm_unfused.json 2> $O/err2.txt; python -c "import json;d=json.load(open('$O/multi_fm_unfused.json'));print('r04 form:', round(d['us_per_cycle'],1),'us per cycle,', len([u for u in d['units']]), 'units')"
done
tail -3 $O/err1.txt
