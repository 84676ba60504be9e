#!/bin/bash
# builds scripts/ab/libflownet2_hip_<name>.so: the in-tree library with one timing ablation of the f16x2 kernels compiled in
set -eu
cd "$(dirname "$0")/.."
P=flownet2-pytorch_amd; mkdir -p scripts/ab /tmp/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics"
OTHERS=$(ls $P/lib/*.o | grep -v "correlation_f16x2")
for name in "$@"; do
  D=$(echo $name | tr '+' ' ' | sed 's/\([A-Z_]*\)/-DFN2_ABL_\1/g')
  for f in correlation_f16x2 correlation_f16x2_bwd; do /opt/rocm/bin/hipcc $FLAGS $D -c $P/csrc/$f.hip -o /tmp/abl/${f}_$name.o 2>/dev/null & done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/ab/libflownet2_hip_$name.so $OTHERS /tmp/abl/correlation_f16x2_$name.o /tmp/abl/correlation_f16x2_bwd_$name.o
  echo built scripts/ab/libflownet2_hip_$name.so "($D)"
done
