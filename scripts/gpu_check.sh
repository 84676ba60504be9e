#!/bin/bash
# One gpurun trip: GPU parity tests, a short bench, and a rocprofv3 kernel trace of the bench.
# Usage (from the dev container):  gpurun --timeout 1500 -- 'bash scripts/gpu_check.sh'
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out
mkdir -p $OUT
{
  echo "== host"; nproc; grep -m1 "model name" /proc/cpuinfo; free -g | head -2
  echo "== gpu"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -E "Marketing Name|gfx9" | head -4
  echo "== reference present?"; ls /root/reference 2>&1 | head -3
} > $OUT/env.txt 2>&1
python -m pytest tests -m gpu -q --tb=short -rA -p no:cacheprovider > $OUT/pytest_gpu.log 2>&1
echo "pytest exit $?" >> $OUT/pytest_gpu.log
tail -40 $OUT/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke exit $?" >> $OUT/smoke.log; tail -3 $OUT/smoke.log
python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -5 $OUT/bench.err
export TMPDIR=/tmp
ROOTDIR=$(pwd)
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof_bench -- python $ROOTDIR/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $ROOTDIR/$OUT/prof_bench.log 2>&1 )
find $OUT/prof_bench -name "*kernel_stats*" | head -3
f=$(find $OUT/prof_bench -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && head -20 "$f"
# fused "next" rows and the backward kernels' L2 traffic (cheap; skipped if the scripts are missing)
[ -f scripts/next_rows_micro.py ] && python scripts/next_rows_micro.py 2>/dev/null | tail -1 > $OUT/next_rows.json && cat $OUT/next_rows.json
