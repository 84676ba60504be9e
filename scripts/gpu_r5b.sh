#!/bin/bash
# round 5, trip b: the whole -m gpu suite + smoke, then bench.py (default flags) and its rocprofv3 kernel stats
set -u -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
TAG=${TAG:-r5b}
bash scripts/gpu_full.sh; echo "gpu_full rc=$?"
SECONDS=0; python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $? wall ${SECONDS}s"; tail -3 $OUT/${TAG}_bench.err
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_bench.json"))
print("value", d["value"], "ms/step", d["ms_per_step"], "modules", d["ms_per_step_autograd_modules"])
print("roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "traffic", "traffic_over_algorithmic", "frac_of_copy_ceiling")})
print("kernels", {k: v["ms"] for k, v in d["kernels"].items()})
print("per_rank", d["per_rank"], d["rank_balance_fastest_over_slowest"])
print("cpu", d.get("cpu_baseline", {}).get("value"))
print("flownet2c", {k: d["flownet2c"].get(k) for k in ("train_step_ms", "fwd_bwd_ms", "fwd_bwd_image_pairs_per_s", "inference_ms", "flownet2_inference_ms_fp32", "flownet2_inference_ms_fp16")})
PY
