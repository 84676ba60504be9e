#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 300 python scripts/resample_micro.py 2>&1 | grep -v "no \|untiled\|none of" | head -40 | tee $OUT/r4b_resample.log

