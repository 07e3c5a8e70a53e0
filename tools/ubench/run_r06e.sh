# LDS-resident block twiddles: timelines (config 5, config 3), the tiled-FFT parity tests, config benches
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r06f; mkdir -p $O
cd $ROOT
timeout 120 tools/ubench/bin/tiled_timeline_c5 > $O/tl_c5.log 2>&1; grep -A4 "SP 4" $O/tl_c5.log
timeout 120 tools/ubench/bin/tiled_timeline > $O/tl_c3.log 2>&1; tail -12 $O/tl_c3.log
timeout 900 python -m pytest tests/test_gpu_fft.py tests/test_gpu_full_sizes.py tests/test_gpu_filter_modules.py tests/test_gpu_reference_flowgraphs.py -q -m gpu -x 2>&1 | tail -4
python tools/bench_c5_streams.py fast > $O/c5_fast.json 2>/dev/null; cat $O/c5_fast.json
python tools/bench_c5_streams.py generic > $O/c5_generic.json 2>/dev/null; cat $O/c5_generic.json
python tools/bench_configs.py C3 2>/dev/null | cut -c1-400
