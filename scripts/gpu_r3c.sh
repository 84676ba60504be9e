#!/bin/bash
# correctness subset + A/B at unit scale (k = 0: multiplies skipped) and at tiny magnitudes (scaled path)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -p no:cacheprovider -k "f16x2 or magnitude or real_training or accuracy or full_size or golden or falls_back" -x > $OUT/r3c_pytest.log 2>&1
tail -3 $OUT/r3c_pytest.log
echo "--- unit scale"; LIBS="r2" REPS=3 bash scripts/gpu_ab.sh
echo "--- inputs x 1e-3, gradOutput x 1e-6"; LIBS="r2" REPS=3 MICRO_ARGS="--in-scale 1e-3 --go-scale 1e-6" bash scripts/gpu_ab.sh
