#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_harness.py -q -x -k "leakyrelu_cat or paths_agree or real_training or trains or fused" 2>&1 | tail -15 | tee $OUT/r4e_tests.log
timeout 300 python scripts/next_rows_micro.py 2>&1 | tail -3
