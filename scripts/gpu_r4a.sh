#!/bin/bash
# gpurun trip r4a: RCCL one-rank test, Resample2d backward accumulation-window variants, LDS atomic rates
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -m pytest tests/test_one_rank_group.py -q -x 2>&1 | tail -5 | tee $OUT/r4a_rccl.log
timeout 300 python scripts/resample_micro.py 2>&1 | tee $OUT/r4a_resample.log
timeout 120 scripts/ubench/lds_atomics 2>&1 | tee $OUT/r4a_lds_atomics.log
