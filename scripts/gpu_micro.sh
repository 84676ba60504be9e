#!/bin/bash
# gpurun trip: correlation-forward ablations + PMC counters for the shipped kernel.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
ALGOS=${ALGOS:-2,100,101,102,103,104,108,109,110,112,116,117,118,120,124,125,126,128}
python scripts/corr_micro.py --check --algos $ALGOS > $OUT/corr_micro.log 2>&1; tail -30 $OUT/corr_micro.log
export TMPDIR=/tmp; R=$(pwd)
PMC1="SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32"
PMC2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE"
for i in 1 2; do
  eval P=\$PMC$i
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d $R/$OUT/pmc$i -- python $R/scripts/corr_micro.py --algos ${PMCALGO:-2} --iters 5 > $R/$OUT/pmc$i.log 2>&1 )
  f=$(find $OUT/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    if "corr" not in k: continue
    print(k)
    for c, v in d.items(): print("   %-34s n=%d mean=%.4g" % (c, len(v), sum(v) / len(v)))
PY
done
tail -3 $OUT/pmc1.log
