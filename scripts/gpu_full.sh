#!/bin/bash
# the round-end sequence: every -m gpu test, then smoke().  Exit status = the first failure's (the pipes below do not hide it).
set -u -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
rc=0
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 | tee $OUT/full_pytest.log || rc=$?
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 || rc=$?
exit $rc
