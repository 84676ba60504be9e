#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
SECONDS=0; python bench.py > $OUT/r4g_bench.json 2> $OUT/r4g_bench.err; echo "bench exit $?"
echo "bench wall seconds: $SECONDS"; tail -3 $OUT/r4g_bench.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r4g_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "traffic_over_algorithmic", "frac_of_copy_ceiling")})
print("traffic_source", d["roofline"]["traffic_source"][:160])
print("kernels", {k: v["ms"] for k, v in d["kernels"].items()})
print("cpu", d.get("cpu_baseline", {}).get("value"), "cpu_fast", d.get("cpu_baseline_fast"))
print("flownet2c", {k: d["flownet2c"].get(k) for k in ("train_step_ms", "fwd_bwd_ms", "inference_ms", "flownet2_inference_ms_fp32", "flownet2_inference_ms_fp16")})
PY
