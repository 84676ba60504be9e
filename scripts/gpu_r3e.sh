#!/bin/bash
# Round-3 closing evidence at HEAD: bench JSON + rocprofv3 kernel stats + correlation traffic (scripts/gpu_profile_round.sh), then
# rocprofv3 kernel stats of the wide (W > 64) and half-precision correlation kernels (scripts/wide_micro.py).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TAG=${TAG:-r03_e}
bash scripts/gpu_profile_round.sh > gpurun_out/${TAG}_profile_round.log 2>&1; tail -4 gpurun_out/${TAG}_profile_round.log | cut -c1-600
export TMPDIR=/tmp; R=$(pwd)
python scripts/wide_micro.py > gpurun_out/${TAG}_wide_micro.log 2>&1; cat gpurun_out/${TAG}_wide_micro.log | grep -v amdgpu.ids
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_wide_prof -- python $R/scripts/wide_micro.py > $R/gpurun_out/${TAG}_wide_prof.log 2>&1 )
f=$(find gpurun_out/${TAG}_wide_prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/${TAG}_wide_kernel_stats.csv && grep -i "corr_" "$f" | cut -c1-200 | head -12
