#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
B=$ROOT/tools/ubench/bin
O=$ROOT/gpurun_out/r03e
mkdir -p $O
for rep in 1 2 3; do
  for v in p_e s_e p_f s_f p_t s_t; do timeout 120 $B/$v 300 $v 0 | grep -E "events|stream"; done
done > $O/split.log 2>&1
for v in s_e s_f; do timeout 120 $B/$v 300 ${v}_mode1 1 | grep events; timeout 120 $B/${v/s_/p_} 300 ${v/s_/p_}_mode1 1 | grep events; done >> $O/split.log 2>&1
grep events $O/split.log
