#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 200 python scripts/corr_micro.py --algos 4 --iters 20 --bwd 6000,8000,8064,8065,8066,8072,8080,8074,8075 2>&1 | grep -v "^{" | tee $OUT/r4c_bwd.log | grep "^bwd\|SUMMARY"
