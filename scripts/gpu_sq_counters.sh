#!/bin/bash
# gpurun trip: SQ counter passes (instruction mix, MFMA busy / co-execution, LDS bank conflicts, VMEM back-pressure) for
# the shipped correlation kernels; one rocprofv3 run per group, kernel-trace only alongside.  Summary -> gpurun_out/sq_counters.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; R=$(pwd)
P1="SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU"
P2="SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS"
P3="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR"
i=0
for G in "$P1" "$P2" "$P3"; do
  i=$((i+1))
  rm -rf $OUT/sq_$i
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/$OUT/sq_$i -- python $R/scripts/corr_micro.py --algos ${SQ_ALGO:-0} ${SQ_BWD:---bwd 0} --iters 3 > $R/$OUT/sq_$i.log 2>&1 ); echo "pass $i rc $?"
done
python - <<'PY'
import csv, glob, collections, json, os
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/sq_*/")):
    fs = sorted(glob.glob(d + "**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs: continue
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        kn = r["Kernel_Name"]
        if "corr_fwd_f16x2" in kn: k = "corr_fwd_f16x2"
        elif "corr_bwd_f16x2" in kn: k = "corr_bwd_f16x2"
        elif "corr_fwd_mfma_bf16x3" in kn: k = "corr_fwd_mfma_bf16x3"
        elif "corr_bwd_mfma_bf16x3" in kn: k = "corr_bwd_mfma_bf16x3"
        else: continue
        vals[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in vals.items():
        res[k][c] = sum(v) / len(v)
for k, d in res.items():
    if "SQ_BUSY_CYCLES" in d and "SQ_VALU_MFMA_BUSY_CYCLES" in d:
        d["mfma_busy_over_sq_busy"] = round(d["SQ_VALU_MFMA_BUSY_CYCLES"] / d["SQ_BUSY_CYCLES"], 4)
    if "SQ_LDS_IDX_ACTIVE" in d and d["SQ_LDS_IDX_ACTIVE"]:
        d["lds_bank_conflict_over_active"] = round(d["SQ_LDS_BANK_CONFLICT"] / d["SQ_LDS_IDX_ACTIVE"], 4)
    if "SQ_INSTS_MFMA" in d and d["SQ_INSTS_MFMA"]:
        d["valu_per_mfma"] = round(d["SQ_INSTS_VALU"] / d["SQ_INSTS_MFMA"], 3)   # SQ_INSTS_VALU counts the MFMAs too
        # MFMA and VALU of a SIMD issue one after the other: per-SIMD issue time if nothing else stalled
        d["issue_cycles_per_simd_est"] = round((d["SQ_INSTS_MFMA"] * 16 + (d["SQ_INSTS_VALU"] - d["SQ_INSTS_MFMA"]) * 4.25) / 1024.0, 0)
json.dump(res, open("gpurun_out/sq_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
