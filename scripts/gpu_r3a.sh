#!/bin/bash
# round-3 trip A: correlation parity (incl. the magnitude sweeps) + fwd/bwd micro timing of the block-scaled f16x2 kernels
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q --tb=short -rA -p no:cacheprovider -k "correlation or wrappers" -s > $OUT/r3a_pytest.log 2>&1
echo "pytest exit $?" >> $OUT/r3a_pytest.log
grep -E "passed|failed|sweep|training gradOutput|FAILED|Error|assert" $OUT/r3a_pytest.log | tail -40
timeout 300 python scripts/corr_micro.py --check --algos 4,3,2 > $OUT/r3a_fwd_micro.log 2>&1; tail -5 $OUT/r3a_fwd_micro.log
timeout 300 python scripts/corr_micro.py --check --algos 4 --bwd 4,3 > $OUT/r3a_bwd_micro.log 2>&1; grep -v "^{" $OUT/r3a_bwd_micro.log | tail -8
