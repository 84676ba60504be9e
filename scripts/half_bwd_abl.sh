#!/bin/bash
# timing ablations of the half-precision correlation backward kernel (csrc/correlation_f16_bwd.hip, FN2_HBH_ABL):
#   scripts/half_bwd_abl.sh build   (here: cross-compiles scripts/ab/libflownet2_hip_hbh<N>.so)
#   scripts/half_bwd_abl.sh run     (on the GPU box)
set -u
cd "$(dirname "$0")/.."
P=flownet2-pytorch_amd
VARS="0 1 2 4 8 16 32 3 12 15 47"
if [ "${1:-run}" = build ]; then
  mkdir -p scripts/ab /tmp/abl
  FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -munsafe-fp-atomics"
  OTHERS=$(ls $P/lib/*.o | grep -v "correlation_f16_bwd")
  for v in $VARS; do
    ( /opt/rocm/bin/hipcc $FLAGS -DFN2_HBH_ABL=$v -c $P/csrc/correlation_f16_bwd.hip -o /tmp/abl/hbh$v.o 2>/dev/null &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o scripts/ab/libflownet2_hip_hbh$v.so $OTHERS /tmp/abl/hbh$v.o ) &
  done
  wait; ls -la scripts/ab/ | grep hbh
else
  for v in $VARS; do
    printf "ABL %-3s " $v; timeout 120 python scripts/half_bwd_abl.py scripts/ab/libflownet2_hip_hbh$v.so 2>&1 | tail -1
  done
fi
