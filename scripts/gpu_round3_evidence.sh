#!/bin/bash
# Round-3 evidence at the shipped HEAD: bench JSON + rocprofv3 kernel stats of the same command, fabric traffic of all six
# kernels, L2 hit rates of the two correlation kernels, SQ counters.  Results -> gpurun_out/<TAG>_*; copied into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TAG=${TAG:-r03_a}
bash scripts/gpu_profile_round.sh > gpurun_out/${TAG}_profile_round.log 2>&1; tail -5 gpurun_out/${TAG}_profile_round.log
bash scripts/gpu_traffic_all.sh > gpurun_out/${TAG}_traffic_all.log 2>&1; tail -3 gpurun_out/${TAG}_traffic_all.log
SQ_ALGO=0 bash scripts/gpu_sq_counters.sh > gpurun_out/${TAG}_sq.log 2>&1; cp gpurun_out/sq_counters.json gpurun_out/${TAG}_sq_counters.json 2>/dev/null; tail -3 gpurun_out/${TAG}_sq.log
python scripts/corr_accuracy.py --algos 2,3,4 > gpurun_out/${TAG}_accuracy.log 2>&1; tail -12 gpurun_out/${TAG}_accuracy.log
{ /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp 2>/dev/null | head -30; } > gpurun_out/${TAG}_smi.txt 2>&1; grep -i "sclk\|mclk\|power\|temp" gpurun_out/${TAG}_smi.txt | head -8
LIBS="r2" REPS=2 bash scripts/gpu_ab.sh | tail -2
