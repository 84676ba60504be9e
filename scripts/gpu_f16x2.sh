#!/bin/bash
# gpurun trip for the f16x2 correlation forward: probe, correctness vs the direct kernel, ablations, accuracy vs fp64.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
[ "${PROBE:-1}" = 1 ] && timeout 120 scripts/ubench/tr16_probe > $OUT/tr16_probe.log 2>&1; tail -n 22 $OUT/tr16_probe.log | cut -c1-200
timeout 300 python scripts/corr_micro.py --check --algos ${ALGOS:-4,3,5001,5002,5004,5008,5016,5006,5022} > $OUT/f16x2_micro.log 2>&1; tail -14 $OUT/f16x2_micro.log
timeout 300 python scripts/corr_accuracy.py --algos 2,3,4 > $OUT/f16x2_accuracy.log 2>&1; tail -24 $OUT/f16x2_accuracy.log
