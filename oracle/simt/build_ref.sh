#!/bin/sh
# Builds oracle/_ref/libfn2_ref.so: the reference's own CUDA device code, compiled
# from where it lies under $1 (default /root/reference) against the CPU SIMT shim.
# The device-code slices (everything before the first host launcher of each .cu)
# go to a temporary directory outside the repository and are deleted afterwards;
# no reference source is ever written into the repo.  Dev container only:
# /root/reference does not exist on the GPU box (the built .so travels instead).
set -eu
REF=${1:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT="$HERE/../_ref"
[ -d "$REF/networks" ] || { echo "build_ref.sh: $REF/networks not found; skipping (prebuilt _ref is used if present)"; exit 0; }
TMP=$(mktemp -d /tmp/fn2_ref_slices.XXXXXX)
trap 'rm -rf "$TMP"' EXIT
mkdir -p "$OUT"
slice() { # $1 = .cu path, $2 = regex of the first host launcher line, $3 = output
    awk -v pat="$2" '$0 ~ pat { exit } { print }' "$1" > "$3"
    [ -s "$3" ] || { echo "empty slice for $1"; exit 1; }
}
slice "$REF/networks/correlation_package/correlation_cuda_kernel.cu" '^int correlation_forward_cuda_kernel' "$TMP/corr_device.inc"
slice "$REF/networks/resample2d_package/resample2d_kernel.cu" '^void resample2d_kernel_forward' "$TMP/resample_device.inc"
slice "$REF/networks/channelnorm_package/channelnorm_kernel.cu" '^void channelnorm_kernel_forward' "$TMP/chnorm_device.inc"
CXX=${CXX:-g++}
FLAGS="-O1 -g0 -fPIC -std=c++17 -ffp-contract=off -fno-fast-math -w -I$HERE/stubs -iquote $HERE/stubs -I$HERE"
$CXX $FLAGS -DFN2_REF_SLICE="\"$TMP/corr_device.inc\"" -c "$HERE/ref_corr.cpp" -o "$TMP/ref_corr.o"
$CXX $FLAGS -DFN2_REF_SLICE="\"$TMP/resample_device.inc\"" -c "$HERE/ref_resample.cpp" -o "$TMP/ref_resample.o"
$CXX $FLAGS -DFN2_REF_SLICE="\"$TMP/chnorm_device.inc\"" -c "$HERE/ref_chnorm.cpp" -o "$TMP/ref_chnorm.o"
$CXX -shared -o "$OUT/libfn2_ref.so" "$TMP/ref_corr.o" "$TMP/ref_resample.o" "$TMP/ref_chnorm.o" -lm
echo "built $OUT/libfn2_ref.so"
