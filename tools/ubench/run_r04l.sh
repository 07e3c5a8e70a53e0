#!/bin/bash
# round 4, call l: the pipelined kernel with deferred stores (outputs stored at the top of the next iteration, in front of the
# prefetch) -- parity first, then the bench line.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04l
mkdir -p $O
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_chain.py tests/test_gpu_side_output.py tests/test_gpu_batch.py tests/test_gpu_exact_sweep.py tests/test_gpu_fft.py -q -x -m gpu > $O/pytest_subset.log 2>&1; echo "subset rc=$?"; tail -3 $O/pytest_subset.log
timeout 600 python bench.py --no-cpu-baseline --no-host-fed --no-configs > $O/bench_a.json 2> $O/bench_a.err; echo "bench rc=$?"
timeout 600 python bench.py --no-cpu-baseline --no-host-fed --no-configs > $O/bench_b.json 2>> $O/bench_a.err
python - <<PY
import json
for f in ("bench_a","bench_b"):
    try:
        b=json.loads(open("$O/%s.json"%f).read().strip().splitlines()[-1])
        print(f, "value", round(b["value"]), "ms_per_step", round(b["ms_per_step"]*1e3,3), "us; roofline", {k:(round(v,4) if isinstance(v,float) else v) for k,v in b["roofline"].items() if k in ("achieved","frac","kernel_us","launch_us","frac_rocprof")}, "parity", b.get("parity",{}).get("bit_exact"), "generic", b.get("value_generic_per_cycle"))
    except Exception as e: print(f, "failed", e)
PY
