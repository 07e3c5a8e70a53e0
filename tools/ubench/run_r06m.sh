ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_fft.py tests/test_gpu_full_sizes.py tests/test_gpu_filter_modules.py tests/test_gpu_reference_flowgraphs.py tests/test_gpu_tiled_persistent.py -q -m gpu -x 2>&1 | tail -4
python tools/bench_configs.py C3 2>/dev/null | python -c "import sys,json; [print('  ', d['config'][:30], round(d['ms_per_cycle'],4), d.get('parity', d.get('max_err_vs_fft_chain_rel_peak'))) for d in map(json.loads, sys.stdin)]"
cd /tmp && export TMPDIR=/tmp
O=$ROOT/gpurun_out/r06m; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3 -- python $ROOT/tools/bench_configs.py C3 > /dev/null 2>&1; python $ROOT/tools/kstats.py $O/c3 | head -8
find $O -name "*.db" -delete; find $O -name "*kernel_trace.csv" -delete
