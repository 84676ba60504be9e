#!/bin/bash
# gpurun trip: L2 <-> fabric traffic and L2 hit rate of the f16x2 correlation forward (and profiling variants),
# one rocprofv3 pass per counter group, kernel-trace only alongside.  Summary -> gpurun_out/f16x2_pmc.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; R=$(pwd)
ALGOS=${ALGOS:-4}
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf $OUT/pmcf_$i
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/pmcf_$i -- python $R/scripts/corr_micro.py --algos $ALGOS --iters 3 --batch 3 > $R/$OUT/pmcf_$i.log 2>&1 ); echo "pass $i rc $?"
done
python - <<'PY'
import csv, glob, collections, json, os, re
res = collections.defaultdict(dict)
for d in sorted(glob.glob("gpurun_out/pmcf_*/")):
    fs = sorted(glob.glob(d + "**/*counter_collection.csv", recursive=True), key=os.path.getmtime)
    if not fs: continue
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        kn = r["Kernel_Name"]
        if "corr_fwd" not in kn: continue
        m = re.search(r"(corr_fwd_\w+)<([^>]*)>", kn)
        k = (m.group(1) + "<" + m.group(2) + ">") if m else kn[:50]
        vals[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in vals.items():
        res[k][c] = sum(v) / len(v)
    # kernel durations from the trace of the same pass
    kt = sorted(glob.glob(d + "**/*kernel_trace.csv", recursive=True), key=os.path.getmtime)
    if kt:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(kt[-1])):
            kn = r["Kernel_Name"]
            if "corr_fwd" not in kn: continue
            m = re.search(r"(corr_fwd_\w+)<([^>]*)>", kn)
            k = (m.group(1) + "<" + m.group(2) + ">") if m else kn[:50]
            dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        for k, v in dur.items():
            res[k].setdefault("duration_us_by_pass", []).append(round(sorted(v)[len(v) // 2], 2))
for k, d in res.items():
    if "FETCH_SIZE" in d: d["read_MB_x2"] = round(2 * d["FETCH_SIZE"] * 1024 / 1e6, 2)
    if "WRITE_SIZE" in d: d["write_MB"] = round(d["WRITE_SIZE"] * 1024 / 1e6, 2)
    if "TCC_HIT_sum" in d: d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
json.dump(res, open("gpurun_out/f16x2_pmc.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
