#!/bin/bash
# The closing evidence set on a FAST box of the pool: a 15-second bench first; a box of the slow class (about every second lease:
# the whole step 20 % slower) ends the call there (its line is kept as <TAG>_slowbox_probe.json), a fast one runs
# scripts/gpu_r4_final.sh.  Both classes are reported in profiles/README.md.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TAG=${TAG:-r04_final}
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --model off --pmc off > $OUT/${TAG}_probe.json 2>/dev/null
ms=$(python -c "import json,sys; print(json.loads(open('$OUT/${TAG}_probe.json').read().strip().splitlines()[-1])['ms_per_step'])")
echo "probe: $ms ms per step"
if python -c "import sys; sys.exit(0 if float('$ms') > ${LIMIT:-0.205} else 1)"; then
  cp $OUT/${TAG}_probe.json $OUT/${TAG}_slowbox_probe.json; echo "slow box: stopping"; exit 0
fi
bash scripts/gpu_r4_final.sh
