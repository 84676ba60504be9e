#!/bin/bash
# gpurun trip: fabric traffic (FETCH_SIZE / WRITE_SIZE, one rocprofv3 --pmc pass each, kernel-trace only alongside) of ALL six
# kernels of the bench step, per launch, next to their algorithmic bytes.  -> gpurun_out/<TAG>_all_kernels_traffic.json
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-rXX}
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; R=$(pwd)
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $OUT/${TAG}_pmcall_$C
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/${TAG}_pmcall_$C -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --model off --pmc off > $R/$OUT/${TAG}_pmcall_$C.log 2>&1 ); echo "pmc $C rc $?"
done
TAG=$TAG python - <<'PY'
import collections, csv, glob, json, os
tag = os.environ["TAG"]
names = {"corr_fwd_f16x2": "corr_fwd", "corr_bwd_f16x2": "corr_bwd", "resample_fwd_tiled": "resample_fwd", "resample_bwd_c3x": "resample_bwd", "resample_bwd_tiled": "resample_bwd",
         "chnorm_fwd_vec": "chnorm_fwd", "chnorm_bwd_vec": "chnorm_bwd"}
alg = {"corr_fwd": 93683712, "corr_bwd": 144015360, "resample_fwd": 50331648, "resample_bwd": 81788928, "chnorm_fwd": 25165824, "chnorm_bwd": 50331648}
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"gpurun_out/{tag}_pmcall_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        for k, short in names.items():
            if k in r["Kernel_Name"] and r["Counter_Name"] == c:
                vals[short].append(float(r["Counter_Value"]))
    for short, v in vals.items():
        v = v[len(v) // 3:]   # skip the first third (set-up / warm-up launches)
        res[short][c + "_KB_raw"] = sum(v) / len(v)
        res[short]["launches_averaged"] = len(v)
for k, d in res.items():
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are KB; gfx950 tallies wide (16 B per lane) coalesced reads at half their
    # bytes: x2 for these kernels, all of which read with 16-byte loads (or LDS-DMA); WRITE_SIZE as reported (uncalibrated)
    if "FETCH_SIZE_KB_raw" in d:
        d["read_bytes_per_launch"] = 2 * d["FETCH_SIZE_KB_raw"] * 1024
    if "WRITE_SIZE_KB_raw" in d:
        d["write_bytes_per_launch"] = d["WRITE_SIZE_KB_raw"] * 1024
    if "read_bytes_per_launch" in d and "write_bytes_per_launch" in d:
        d["bytes_per_launch"] = d["read_bytes_per_launch"] + d["write_bytes_per_launch"]
        d["over_algorithmic"] = round(d["bytes_per_launch"] / alg[k], 3)
    d["algorithmic_bytes"] = alg[k]
json.dump(res, open(f"gpurun_out/{tag}_all_kernels_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
