#!/bin/bash
# gpurun trip: L2 <-> fabric traffic and L2 hit rate of the correlation-backward kernels (fp32 MFMA and bf16x3),
# one rocprofv3 pass per counter group, kernel-trace only alongside.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; R=$(pwd)
for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum"; do
  T=$(echo $C | tr ' ' '_')
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmc_bwd_$T -- python $R/scripts/corr_micro.py --algos 0 --iters 2 --bwd ${BWD:-0,104} > $R/$OUT/pmc_bwd_$T.log 2>&1 )
done
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob("gpurun_out/pmc_bwd_*/")):
    f = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not f: print(d, "no csv"); continue
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if "corr_bwd" in r["Kernel_Name"]:
            vals[(r["Kernel_Name"].split("(")[0][-40:], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k, v in sorted(vals.items()):
        print(k[0], k[1], "launches", len(v), "mean", sum(v) / len(v))
PY
