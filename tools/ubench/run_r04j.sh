#!/bin/bash
# round 4, call j: config 3 diagnostics (phase timeline of the columns / blocks / blocks+fold kernels, LDS bank-conflict
# counters) and the full GPU suite with cycle batching as the Python runtime's default.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04j
mkdir -p $O
cd $ROOT
timeout 120 tools/ubench/bin/tiled_timeline > $O/tiled_timeline_c3.log 2>&1; tail -20 $O/tiled_timeline_c3.log
timeout 1500 python -m pytest tests -q -m gpu -x > $O/pytest_gpu_full.log 2>&1; echo "full rc=$?"; tail -5 $O/pytest_gpu_full.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_lds_c3 -- python $ROOT/tools/bench_configs.py C3 > $O/pmc_lds_c3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d $O/pmc_sq_c3 -- python $ROOT/tools/bench_configs.py C3 > $O/pmc_sq_c3.log 2>&1
python - <<PY
import csv, glob, collections
for d in ("pmc_lds_c3","pmc_sq_c3"):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob("$O/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"][:90]; acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
            if r["Counter_Name"] in ("SQ_WAVE_CYCLES","SQ_INSTS_VALU"): cnt[k]+=1
    for k,v in acc.items():
        if "fft_tile" in k: print(d, k, cnt[k], {c:round(x/max(cnt[k],1)) for c,x in v.items()})
PY
find $O -name "*counter_collection.csv" -size +3M -delete
