#!/bin/bash
# Round 3, experiment m: phases of the index-fed spectrogram kernel (timing only: the DIAG variants compute nothing useful).
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03m
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  for prov in fast; do
    env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_${prov}_trace -- \
      python $ROOT/bench.py --provider $prov --no-cpu-baseline --no-alt --no-parity --no-host-fed > $O/${name}_$prov.json 2> $O/${name}_$prov.err
    echo "== $name $prov: $(python -c "import json,sys; d=json.loads(open('$O/${name}_$prov.json').read().strip().splitlines()[-1]); print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step')" 2>&1)"
    python $ROOT/tools/kstats.py $O/${name}_${prov}_trace | head -2
    rm -rf $O/${name}_${prov}_trace
  done
}
run c4 JST_SPEC_INDEX_COPIES=4
run c2 JST_SPEC_INDEX_COPIES=2
run c1 JST_SPEC_INDEX_COPIES=1
run c4b JST_SPEC_INDEX_COPIES=4
