#!/bin/bash
# instruction-cache counters of the shipped correlation kernels (the backward's code is 80 KB: larger than the 64 KB instruction cache
# two CUs share)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; R=$(pwd)
( cd /tmp && rocprofv3 -L 2>/dev/null | grep -io "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*\|SQ_INSTS_SMEM[A-Z_]*\|SQ_WAIT_INST[A-Z_]*\|SQC_TC_INST[A-Z_]*" | sort -u | tr '\n' ' ' ) > $OUT/icache_avail.log; cat $OUT/icache_avail.log; echo
G="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
rm -rf $OUT/ic_1
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $G --output-format csv -d $R/$OUT/ic_1 -- python $R/scripts/corr_micro.py --algos 0 --bwd 0 --iters 3 > $R/$OUT/ic_1.log 2>&1 ); echo "rc $?"; tail -3 $OUT/ic_1.log
python - <<'PY'
import csv, glob, collections, json, os
fs = sorted(glob.glob("gpurun_out/ic_1/**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
vals = collections.defaultdict(list)
for r in csv.DictReader(open(fs[-1])) if fs else []:
    kn = r["Kernel_Name"]
    k = "corr_fwd_f16x2" if "corr_fwd_f16x2" in kn else "corr_bwd_f16x2" if "corr_bwd_f16x2" in kn else None
    if k: vals[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
res = collections.defaultdict(dict)
for (k, c), v in vals.items(): res[k][c] = sum(v) / len(v)
json.dump(res, open("gpurun_out/icache_counters.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
