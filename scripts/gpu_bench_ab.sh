#!/bin/bash
# bench.py's per-kernel event timings with the in-tree library and with the round-2 build swapped in (same box)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
show() { python - "$1" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], "ms/step", l["ms_per_step"], {k: round(v["ms"] * 1e3, 1) for k, v in l["kernels"].items()})
PY
}
for rep in 1 2; do
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --model off > $OUT/bench_head.json 2>/dev/null; show $OUT/bench_head.json
cp flownet2-pytorch_amd/lib/libflownet2_hip.so /tmp/head.so
cp scripts/ab/libflownet2_hip_r2.so flownet2-pytorch_amd/lib/libflownet2_hip.so
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --model off > $OUT/bench_r2.json 2>/dev/null; show $OUT/bench_r2.json
cp /tmp/head.so flownet2-pytorch_amd/lib/libflownet2_hip.so
done
