#!/bin/bash
# the round-end sequence: every -m gpu test, smoke(), bench.py, bench.py --gpus 2 (self-launched ranks sharing the one GPU over gloo)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $OUT/full_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
