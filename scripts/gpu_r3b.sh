#!/bin/bash
# quick correctness (f16x2 subset) + A/B timing on one box
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "f16x2 or magnitude or real_training or accuracy or full_size or golden" -x > $OUT/r3b_pytest.log 2>&1
tail -5 $OUT/r3b_pytest.log
bash scripts/gpu_ab.sh
