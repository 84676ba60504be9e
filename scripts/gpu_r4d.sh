#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -s -k "per_channel or huge_and_tiny or magnitude_sweep or real_training or sample_misses or out_of_range or correlation_backward" 2>&1 | grep -v "^$" | tail -25 | tee $OUT/r4d_tests.log
timeout 200 python scripts/corr_micro.py --algos 4 --iters 30 --bwd 4,6000 --check 2>&1 | grep "^bwd\|^4"
