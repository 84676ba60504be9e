#!/bin/bash
# round 5, trip a: parity of the new Resample2d backward forms, the fused N2 backward, the loss goldens; then their timings
set -u -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x -k "resample2d or warp_diff or multiscale or leakyrelu_cat_backward or flownet2_trains or native_library" 2>&1 | tail -15 | tee $OUT/r5a_pytest.log
timeout 300 python scripts/resample_r5_micro.py 2>&1 | tee $OUT/r5a_resample_micro.log
