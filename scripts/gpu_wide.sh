#!/bin/bash
# the wide (W > 64) f16x2 kernels: parity tests + timing against the fp32 matrix-core kernels
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "wide" 2>&1 | tail -15
timeout 300 python scripts/wide_micro.py 2>&1 | tail -20
