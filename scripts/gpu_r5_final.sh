#!/bin/bash
# Round-5 closing evidence, ONE lease: GPU tests, smoke, bench.py (roofline.traffic measured in that run), rocprofv3 kernel stats of
# the same step, fabric traffic of all six kernels, SQ counters of the correlation kernels, fused rows (N1-N3 incl. the N2 backward),
# wide / half kernels, Resample2d backward timeline.  Everything lands in gpurun_out/<TAG>_*; what is judged is copied into profiles/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TAG=${TAG:-r05_final}
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; R=$(pwd)
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $OUT/${TAG}_gpu_pytest.log; tail -2 $OUT/${TAG}_gpu_pytest.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/${TAG}_gpu_pytest.log
SECONDS=0
timeout 900 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $? in $SECONDS s"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/${TAG}_prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --model off --pmc off > $R/$OUT/${TAG}_prof.log 2>&1 )
f=$(find $OUT/${TAG}_prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_bench_kernel_stats.csv && head -9 "$f" | cut -c1-150
bash scripts/gpu_traffic_all.sh > $OUT/${TAG}_traffic_all.log 2>&1; tail -4 $OUT/${TAG}_traffic_all.log | cut -c1-200
SQ_ALGO=0 bash scripts/gpu_sq_counters.sh > $OUT/${TAG}_sq.log 2>&1; cp $OUT/sq_counters.json $OUT/${TAG}_sq_counters.json 2>/dev/null; grep -c "pass" $OUT/${TAG}_sq.log
timeout 200 python scripts/next_rows_micro.py 2>/dev/null | tail -1 > $OUT/${TAG}_next_rows.json; cat $OUT/${TAG}_next_rows.json
timeout 200 python scripts/wide_micro.py 2>/dev/null | tail -5 > $OUT/${TAG}_wide_micro.log; cat $OUT/${TAG}_wide_micro.log | cut -c1-220
timeout 200 python scripts/resample_r5_micro.py 2>/dev/null > $OUT/${TAG}_resample_micro.log; grep -v "no flush\|no scatter" $OUT/${TAG}_resample_micro.log | cut -c1-160
timeout 100 python scripts/resample_timeline.py 2>/dev/null > $OUT/${TAG}_resample_timeline.log; grep "workgroups start" $OUT/${TAG}_resample_timeline.log
timeout 200 python scripts/corr_micro.py --algos 4 --iters 30 --bwd 4,6064 2>/dev/null | grep -v "^{" > $OUT/${TAG}_corr_micro.log; grep "^4\|^bwd" $OUT/${TAG}_corr_micro.log | cut -c1-260
tail -c 600 $OUT/${TAG}_bench.json
