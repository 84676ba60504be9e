#!/bin/bash
# A/B on ONE box: the round-2 build of the library (scripts/ab/libflownet2_hip_r2.so, built from git) against the in-tree one
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/ab.log
for rep in 1 2 3; do
  for L in scripts/ab/libflownet2_hip_r2.so flownet2-pytorch_amd/lib/libflownet2_hip.so; do
    echo "== $L" >> $OUT/ab.log
    timeout 300 python scripts/corr_micro.py --algos 4 --bwd 4 --lib $L 2>/dev/null | grep -v "^{" >> $OUT/ab.log
  done
done
cat $OUT/ab.log
