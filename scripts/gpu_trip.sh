#!/bin/bash
# ONE parametrised GPU trip (replaces the per-round gpu_r3*.sh ... gpu_r5*.sh scripts; their history is in git):
#     gpurun --timeout 2400 -- 'TAG=r06_a STAGES="tests smoke bench prof rows" bash scripts/gpu_trip.sh'
# STAGES (any subset, run in this order):
#   tests     every -m gpu test (TESTS="-k expr" / a path narrows it)          -> <TAG>_gpu_pytest.log
#   smoke     __graft_entry__.smoke()
#   bench     python bench.py (roofline.traffic measured in the run)            -> <TAG>_bench.json
#   prof      rocprofv3 --kernel-trace --stats of the same step                  -> <TAG>_bench_kernel_stats.csv
#   traffic   fabric traffic of all six kernels (PMC passes of their own)        -> <TAG>_traffic_all.log
#   sq        SQ counters of the correlation kernels                             -> <TAG>_sq_counters.json
#   rows      fused rows N1-N3 (scripts/next_rows_micro.py)                      -> <TAG>_next_rows.json
#   wide      Sintel-size / half correlation kernels (scripts/wide_micro.py)     -> <TAG>_wide_micro.log
#   resample  Resample2d micro + workgroup timeline                              -> <TAG>_resample_*.log
#   corr      correlation micro (scripts/corr_micro.py)                          -> <TAG>_corr_micro.log
#   repro     forward reproducibility of harness.FlowNet2 + fused training probe -> <TAG>_forward_reproducibility.log
#   dry8      `bench.py --gpus 8 --model on` as the shared-GPU dry run           -> <TAG>_bench_gpus8_dry_run.json
#   extra     whatever EXTRA="cmd" says (one-off experiments)                    -> <TAG>_extra.log
# Everything lands in gpurun_out/ (scratch); what is judged is copied into profiles/ by hand.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
TAG=${TAG:-trip}; STAGES=${STAGES:-"tests smoke bench"}; TESTS=${TESTS:-}
OUT=gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp; R=$(pwd)
for st in $STAGES; do
  SECONDS=0
  case $st in
    tests)   timeout ${TEST_TIMEOUT:-2400} python -m pytest tests -m gpu -q -rxs $TESTS 2>&1 | tail -${TEST_TAIL:-25} > $OUT/${TAG}_gpu_pytest.log; tail -${TEST_TAIL:-25} $OUT/${TAG}_gpu_pytest.log ;;
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee -a $OUT/${TAG}_gpu_pytest.log ;;
    bench)   timeout 900 python bench.py ${BENCH_ARGS:-} > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err; echo "bench exit $?"; tail -c 1500 $OUT/${TAG}_bench.json | head -c 1500; echo ;;
    prof)    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/${TAG}_prof -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --model off --pmc off > $R/$OUT/${TAG}_prof.log 2>&1 )
             f=$(find $OUT/${TAG}_prof -name "*kernel_stats*.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/${TAG}_bench_kernel_stats.csv && head -12 "$f" | cut -c1-160 ;;
    traffic) bash scripts/gpu_traffic_all.sh > $OUT/${TAG}_traffic_all.log 2>&1; tail -8 $OUT/${TAG}_traffic_all.log | cut -c1-220 ;;
    sq)      SQ_ALGO=${SQ_ALGO:-0} bash scripts/gpu_sq_counters.sh > $OUT/${TAG}_sq.log 2>&1; cp $OUT/sq_counters.json $OUT/${TAG}_sq_counters.json 2>/dev/null; tail -3 $OUT/${TAG}_sq.log | cut -c1-300 ;;
    rows)    timeout 300 python scripts/next_rows_micro.py 2>$OUT/${TAG}_next_rows.err | tail -1 > $OUT/${TAG}_next_rows.json; cat $OUT/${TAG}_next_rows.json; tail -3 $OUT/${TAG}_next_rows.err ;;
    wide)    timeout 300 python scripts/wide_micro.py 2>&1 | tail -8 > $OUT/${TAG}_wide_micro.log; cut -c1-240 $OUT/${TAG}_wide_micro.log ;;
    resample) timeout 200 python scripts/resample_r5_micro.py 2>/dev/null > $OUT/${TAG}_resample_micro.log; grep -v "no flush\|no scatter" $OUT/${TAG}_resample_micro.log | cut -c1-160
             timeout 100 python scripts/resample_timeline.py 2>/dev/null > $OUT/${TAG}_resample_timeline.log; grep "workgroups start" $OUT/${TAG}_resample_timeline.log ;;
    corr)    timeout 300 python scripts/corr_micro.py ${CORR_ARGS:---algos 4 --iters 30 --bwd 4} 2>/dev/null | grep -v "^{" > $OUT/${TAG}_corr_micro.log; cut -c1-260 $OUT/${TAG}_corr_micro.log | tail -12 ;;
    repro)   timeout 600 python scripts/forward_reproducibility_probe.py > $OUT/${TAG}_forward_reproducibility.log 2>&1; tail -20 $OUT/${TAG}_forward_reproducibility.log | cut -c1-250
             timeout 300 python scripts/fused_training_probe.py >> $OUT/${TAG}_forward_reproducibility.log 2>&1; tail -4 $OUT/${TAG}_forward_reproducibility.log | cut -c1-250 ;;
    dry8)    FN2_BENCH_SHARE_GPU=1 timeout 1500 python bench.py --gpus 8 --steps 5 --warmup 2 --no-cpu-baseline --model on --model-steps 3 --model-warmup 1 --model-timeout 1200 > $OUT/${TAG}_bench_gpus8_dry_run.json 2> $OUT/${TAG}_dry8.err; echo "dry8 exit $?"; tail -c 1200 $OUT/${TAG}_bench_gpus8_dry_run.json; tail -3 $OUT/${TAG}_dry8.err ;;
    extra)   bash -c "${EXTRA:-true}" > $OUT/${TAG}_extra.log 2>&1; tail -${EXTRA_TAIL:-30} $OUT/${TAG}_extra.log | cut -c1-260 ;;
    *)       echo "unknown stage $st" ;;
  esac
  echo "[$st: ${SECONDS}s]"
done
