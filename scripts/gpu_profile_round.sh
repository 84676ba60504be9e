#!/bin/bash
# One gpurun trip for the round's evidence (TAG=r02_d ...): bench JSON, rocprofv3 kernel stats of the same command,
# fabric traffic + L2 hit rate of the shipped correlation forward and backward kernels (one rocprofv3 --pmc pass per
# counter group, kernel-trace only alongside), next-rows micro-benchmark.  Everything lands in gpurun_out/<TAG>_*;
# copy what is to be judged into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-rXX}
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; R=$(pwd)
python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/${TAG}_prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --model off --pmc off > $R/$OUT/${TAG}_prof.log 2>&1 )
f=$(find $OUT/${TAG}_prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_bench_kernel_stats.csv && head -12 "$f"
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  T=$(echo $C | tr ' ' '_'); rm -rf $OUT/${TAG}_pmc_$T
  ( cd /tmp && timeout 150 rocprofv3 --kernel-trace --pmc $C --output-format csv -d $R/$OUT/${TAG}_pmc_$T -- python $R/scripts/corr_micro.py --algos 0 --iters 5 --bwd 0 > $R/$OUT/${TAG}_pmc_$T.log 2>&1 ); echo "pmc $T rc $?"
done
TAG=$TAG python - <<'PY'
import collections, csv, glob, json, os
tag = os.environ["TAG"]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(f"gpurun_out/{tag}_pmc_*/")):
    fs = glob.glob(d + "**/*counter_collection.csv", recursive=True)
    if not fs:
        continue
    vals = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        kn = r["Kernel_Name"]
        key = "corr_fwd_f16x2" if "corr_fwd_f16x2" in kn else "corr_bwd_f16x2" if "corr_bwd_f16x2" in kn else None
        if key:
            vals[(key, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in vals.items():
        res[k][c] = sum(v) / len(v)
        res[k]["launches_averaged"] = len(v)
alg = {"corr_fwd_f16x2": 93683712, "corr_bwd_f16x2": 144015360}
for k, d in res.items():
    # MI355X_MICROARCH.md (HBM): FETCH_SIZE / WRITE_SIZE are KB; gfx950 tallies wide (16 B per lane) coalesced reads at half
    # their bytes -> x2 (the inputs of both kernels arrive by 16-byte buffer loads / LDS-DMA); WRITE_SIZE as reported
    if "FETCH_SIZE" in d:
        d["read_bytes_per_launch"] = 2 * d["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in d:
        d["write_bytes_per_launch"] = d["WRITE_SIZE"] * 1024
    if "read_bytes_per_launch" in d and "write_bytes_per_launch" in d:
        d["bytes_per_launch"] = d["read_bytes_per_launch"] + d["write_bytes_per_launch"]
    if "TCC_HIT_sum" in d:
        d["l2_hit_rate"] = round(d["TCC_HIT_sum"] / (d["TCC_HIT_sum"] + d["TCC_MISS_sum"]), 4)
    d["algorithmic_bytes"] = alg[k]
    d["correction"] = "FETCH_SIZE x2 (gfx950 tallies 16-byte-per-lane coalesced reads at half), WRITE_SIZE as reported (uncalibrated)"
json.dump(res, open(f"gpurun_out/{tag}_corr_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
[ -f scripts/next_rows_micro.py ] && python scripts/next_rows_micro.py 2>/dev/null | tail -1 > $OUT/${TAG}_next_rows.json && cat $OUT/${TAG}_next_rows.json
tail -c 1500 $OUT/${TAG}_bench.json
