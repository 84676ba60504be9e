#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 300 python scripts/wide_micro.py 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "per_channel or wide or correlation_backward or sample_misses or out_of_range" 2>&1 | tail -3
