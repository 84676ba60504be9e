#!/bin/bash
# gpurun trip: PMC counter passes (one rocprofv3 run per counter group) for one correlation-forward algo.
# usage: ALGO=356 bash scripts/gpu_pmc.sh   (groups are separate passes: counters-only runs, no tracing domains)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; R=$(pwd)
ALGO=${ALGO:-2}
# (TA_* counters are NOT collected: a pass with TA_BUSY_avr / TA_*_STALLED_* hung rocprofv3 on this pool until the gpurun limit)
G2="TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum"
G3="TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum TCC_BUSY_avr"
G4="SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"
G5="FETCH_SIZE"
G6="WRITE_SIZE GRBM_GUI_ACTIVE"
i=0
for G in "$G2" "$G3" "$G4" "$G5" "$G6"; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/$OUT/pmc_${ALGO}_$i -- python $R/scripts/corr_micro.py --algos $ALGO --iters 5 > $R/$OUT/pmc_${ALGO}_$i.log 2>&1 )
  f=$(find $OUT/pmc_${ALGO}_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "corr" not in k: continue
    for c, v in d.items(): print("%-40s n=%d mean=%.5g" % (c, len(v), sum(v) / len(v)))
PY
done
