#!/bin/bash
# Round 3, experiment l: the Spectrogram fed with one-byte row indices from the fused kernel's side output
# (JST_NO_SPECTROGRAM_SIDE=1 = the value path) and the index kernel's workgroup size; rocprofv3 per-kernel means.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03l
mkdir -p $O
cd $ROOT
timeout 900 python -m pytest tests/test_gpu_spectrogram_indices.py -q -x 2>&1 | tail -15
cd /tmp && export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  for prov in fast generic; do
    env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_${prov}_trace -- \
      python $ROOT/bench.py --provider $prov --no-cpu-baseline --no-alt --no-parity --no-host-fed > $O/${name}_$prov.json 2> $O/${name}_$prov.err
    echo "== $name $prov: $(python -c "import json,sys; d=json.loads(open('$O/${name}_$prov.json').read().strip().splitlines()[-1]); print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step')" 2>&1)"
    python $ROOT/tools/kstats.py $O/${name}_${prov}_trace | head -3
    rm -rf $O/${name}_${prov}_trace
  done
}
run values JST_NO_SPECTROGRAM_SIDE=1
run idx256 JST_SPEC_INDEX_THREADS=256
run idx512 JST_SPEC_INDEX_THREADS=512
run idx1024 JST_SPEC_INDEX_THREADS=1024
run values2 JST_NO_SPECTROGRAM_SIDE=1
run idx256b JST_SPEC_INDEX_THREADS=256
