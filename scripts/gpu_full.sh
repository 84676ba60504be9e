#!/bin/bash
# full GPU trip: every gpu-marked test, smoke, bench (N=1), bench --gpus 2 self-launch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/full_pytest.log 2>&1; echo "pytest exit $?" >> $OUT/full_pytest.log
tail -6 $OUT/full_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python - <<'PY'
import json
l = json.loads(open("gpurun_out/bench.json").read().strip().splitlines()[-1])
r = l["roofline"]
print("value", l["value"], "ms/step", l["ms_per_step"], "| corr fwd", r["launch_ms"], "frac", r["frac"], "copy ceiling", r["copy_ceiling_GBps"], r["copy_ceiling_kernel"], "torch", r["copy_ceiling_torch_copy_GBps"])
print({k: v["ms"] for k, v in l["kernels"].items()})
print("flownet2c", {k: v for k, v in l.get("flownet2c", {}).items() if "ms" in k or "error" in k})
print("cpu", l.get("cpu_baseline", {}).get("value"), "per_rank", l["per_rank"])
PY
tail -3 $OUT/bench.err
FN2_BENCH_SHARE_GPU=1 timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --no-cpu-baseline --model-steps 3 --model-warmup 1 > $OUT/bench_g2.json 2> $OUT/bench_g2.err; echo "bench --gpus 2 exit $?"; cut -c1-400 $OUT/bench_g2.json; tail -3 $OUT/bench_g2.err
