#!/bin/bash
# bench.py's per-kernel event timings (cold operands: the whole step between two launches of a kernel) with the in-tree library
# and with other builds swapped in (scripts/ab/libflownet2_hip_<name>.so: r2 = round-2 build, ablations), same box.
# LIBS="r2 NOMUL ..." REPS=2
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
show() { python - "$1" "$2" <<'PY'
import json, sys
l = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-26s ms/step %.4f  %s" % (sys.argv[2], l["ms_per_step"], {k: round(v["ms"] * 1e3, 1) for k, v in l["kernels"].items()}))
PY
}
cp flownet2-pytorch_amd/lib/libflownet2_hip.so /tmp/head.so
for rep in $(seq 1 ${REPS:-2}); do
  for n in HEAD ${LIBS:-r2}; do
    L=scripts/ab/libflownet2_hip_$n.so; [ $n = HEAD ] && L=/tmp/head.so
    cp $L flownet2-pytorch_amd/lib/libflownet2_hip.so
    python bench.py --steps 30 --warmup 5 --no-cpu-baseline --model off > $OUT/bench_$n.json 2>/dev/null; show $OUT/bench_$n.json $n
  done
done
cp /tmp/head.so flownet2-pytorch_amd/lib/libflownet2_hip.so
