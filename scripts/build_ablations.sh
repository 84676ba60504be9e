#!/bin/bash
# builds scripts/ab/libflownet2_hip_<name>.so: the in-tree library with timing ablations compiled in (every csrc/*.hip is
# recompiled with -DFN2_ABL_<NAME>; NAME1+NAME2 combines two)
set -eu
cd "$(dirname "$0")/.."
P=flownet2-pytorch_amd; mkdir -p scripts/ab /tmp/abl
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics"
for name in "$@"; do
  D=$(echo $name | tr '+' ' ' | sed 's/\([A-Z_0-9]*\)/-DFN2_ABL_\1/g')
  mkdir -p /tmp/abl/$name; OBJS=""
  for f in $P/csrc/*.hip; do b=$(basename $f .hip); /opt/rocm/bin/hipcc $FLAGS $D -c $f -o /tmp/abl/$name/$b.o 2>/dev/null & OBJS="$OBJS /tmp/abl/$name/$b.o"; done; wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/ab/libflownet2_hip_$name.so $OBJS
  echo built scripts/ab/libflownet2_hip_$name.so "($D)"
done
