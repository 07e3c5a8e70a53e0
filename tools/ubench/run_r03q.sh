#!/bin/bash
# Round 3, experiment q: full GPU suite + rocprofv3 per-kernel means of the bench with the index-fed Spectrogram
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=$ROOT/gpurun_out/r03q
mkdir -p $O
cd $ROOT
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
cd /tmp && export TMPDIR=/tmp
run() {  # name, env...
  name=$1; shift
  for prov in fast generic; do
    env "$@" timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_${prov}_trace -- \
      python $ROOT/bench.py --provider $prov --no-cpu-baseline --no-alt --no-parity --no-host-fed > $O/${name}_$prov.json 2> $O/${name}_$prov.err
    echo "== $name $prov: $(python -c "import json,sys; d=json.loads(open('$O/${name}_$prov.json').read().strip().splitlines()[-1]); print(round(d['value']), 'MS/s', round(d['ms_per_step']*1e3,2), 'us/step')" 2>&1)"
    python $ROOT/tools/kstats.py $O/${name}_${prov}_trace | head -2
    rm -rf $O/${name}_${prov}_trace
  done
}
run idx JST_X=1
run values JST_NO_SPECTROGRAM_SIDE=1
