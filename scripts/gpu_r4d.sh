#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "per_channel or correlation_backward or sample_misses or out_of_range or real_training" 2>&1 | tail -3
timeout 300 python scripts/soak_fuzz.py 120 7 2>&1 | tail -8
