#!/bin/bash
# round-3 batch A on the GPU box: the floor bisect, the wide-access variants of the fused kernel (exact / fast / trivial
# epilogue / skeleton), then the test suite and a bench line on the same box.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r03a
mkdir -p $O
cd $ROOT
for rep in 1 2; do timeout 300 $B/floor_bisect 300; done > $O/floor_bisect.log 2>&1
for rep in 1 2 3; do
  for k in e f t k; do for v in s0l0 s1l0 s0l1 s1l1; do
    timeout 120 $B/w_${k}_$v 300 w_${k}_$v 0 | grep events
  done; done
done > $O/wide.log 2>&1
cat $O/floor_bisect.log | head -60
cat $O/wide.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -5 $O/bench.err
