#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "channelnorm or fuzz" 2>&1 | tail -2
bash scripts/gpu_bench_ab.sh 2>&1 | tail -4
