#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out; mkdir -p $OUT
LIBS="${LIBS:-FLIPOUTER}" REPS=3 MICRO_ARGS="--iters 40 --check" bash scripts/gpu_ab.sh
grep "max_abs" $OUT/ab.log | tail -2
