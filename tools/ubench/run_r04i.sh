#!/bin/bash
# round 4, call i (call g again after the generic pass was rewritten: l pairs per wave, H formed on the fly, fold heads on the fly): generic radix inside the tiled kernels (parity + multi-fm A/B under rocprofv3), full GPU suite after the
# header hygiene, default bench (host-fed ring push).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r04i
mkdir -p $O
cd $ROOT
python tools/bench_multi_fm.py 400 > $O/multi_fm_tiled.json 2> $O/multi_fm_tiled.err
JST_TILED_GENERIC=0 python tools/bench_multi_fm.py 400 > $O/multi_fm_passes.json 2> $O/multi_fm_passes.err
cd /tmp && export TMPDIR=/tmp
for v in tiled passes; do
  rm -rf $O/mfm_$v
  if [ $v = passes ]; then export JST_TILED_GENERIC=0; else unset JST_TILED_GENERIC; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/mfm_$v -- python $ROOT/tools/bench_multi_fm.py 200 > $O/mfm_$v.log 2>&1
  python $ROOT/tools/kstats.py $O/mfm_$v > $O/kernel_stats_multi_fm_$v.txt 2>&1
  find $O/mfm_$v -name "*.csv" ! -name "*stats*" -delete
done
unset JST_TILED_GENERIC
head -25 $O/kernel_stats_multi_fm_tiled.txt $O/kernel_stats_multi_fm_passes.txt
python - <<PY
import json
for f in ("multi_fm_tiled","multi_fm_passes"):
    try:
        d=json.loads(open("$O/%s.json"%f).read()); print(f, round(d["us_per_cycle"],2), d["fft_path_8050"], [u for u in d["units"] if "fft" in u][:6])
    except Exception as e: print(f, "failed", e)
PY
