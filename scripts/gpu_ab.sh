#!/bin/bash
# A/B on ONE box: libraries under scripts/ab/ (round-2 build from git, timing ablations from scripts/build_ablations.sh)
# against the in-tree one.  LIBS="r2 NOMUL ..." selects; REPS repetitions, interleaved.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/ab.log
for rep in $(seq 1 ${REPS:-3}); do
  for n in ${LIBS:-r2} HEAD; do
    L=scripts/ab/libflownet2_hip_$n.so; [ $n = HEAD ] && L=flownet2-pytorch_amd/lib/libflownet2_hip.so
    echo "== $n" >> $OUT/ab.log
    timeout 300 python scripts/corr_micro.py --algos 4 --bwd 4 --lib $L ${MICRO_ARGS:-} 2>/dev/null | grep -v "^{" >> $OUT/ab.log
  done
done
python - <<'PY'
import re, collections
f, b = collections.defaultdict(list), collections.defaultdict(list)
cur = None
for line in open("gpurun_out/ab.log"):
    if line.startswith("=="): cur = line.split()[1]; continue
    m = re.search(r"'min_us': ([0-9.]+)", line)
    if not m: continue
    (b if line.startswith("bwd") else f)[cur].append(float(m.group(1)))
for k in f: print("%-28s fwd min %s   bwd min %s" % (k, " ".join("%.1f" % v for v in f[k]), " ".join("%.1f" % v for v in b[k])))
PY
