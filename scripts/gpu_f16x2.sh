#!/bin/bash
# gpurun trip for the f16x2 correlation forward: probe, correctness vs the direct kernel, ablations, accuracy vs fp64.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
if [ "${PROBE:-0}" = 1 ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 scripts/ubench/tr16_probe.hip -o /tmp/tr16_probe 2>/dev/null && timeout 120 /tmp/tr16_probe > $OUT/tr16_probe.log 2>&1
  tail -n 18 $OUT/tr16_probe.log | cut -c1-200
fi
timeout 300 python scripts/corr_micro.py --check --algos ${ALGOS:-4,3,5001,5002,5004,5008,5016,5032,5006,5024,5025,5038,5063} > $OUT/f16x2_micro.log 2>&1; tail -16 $OUT/f16x2_micro.log | head -15
[ "${ACC:-0}" = 1 ] && { timeout 300 python scripts/corr_accuracy.py --algos 2,3,4 > $OUT/f16x2_accuracy.log 2>&1; tail -24 $OUT/f16x2_accuracy.log; }
true
[ -n "${BWD:-}" ] && { timeout 300 python scripts/corr_micro.py --check --algos 4 --bwd $BWD > $OUT/f16x2_bwd_micro.log 2>&1; grep -v "^{" $OUT/f16x2_bwd_micro.log | tail -12; }
true
