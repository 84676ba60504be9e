#!/bin/bash
# s_memtime timelines of the f16x2 forward (5064) and backward (6064) for the round-2 build and the in-tree one, same box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
for n in r2 HEAD; do
  L=scripts/ab/libflownet2_hip_$n.so; [ $n = HEAD ] && L=flownet2-pytorch_amd/lib/libflownet2_hip_debug.so
  timeout 300 python scripts/corr_micro.py --algos 5064 --bwd 6064 --lib $L > $OUT/timeline_$n.log 2>&1
done
paste -d'|' <(cut -c1-74 $OUT/timeline_r2.log) <(cut -c1-74 $OUT/timeline_HEAD.log) | grep -v "^{" | head -90
