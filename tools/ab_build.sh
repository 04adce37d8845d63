#!/bin/bash
# Build a second copy of the library with extra -D flags for same-box A/B timing:
#   tools/ab_build.sh jac -DHM_K3_JACOBIAN   ->  gpurun_ab/libhalo2_mi355x_jac.so
# Use it with HALO2_MI355X_LIB=<path> (see halo2-experiments_amd/_lib.py).
set -e
tag=$1; shift
cd "$(dirname "$0")/../halo2-experiments_amd/csrc"
out=../../gpurun_ab; mkdir -p $out
for f in capi multi xfer ntt poly polyops lookup graph msm msm_small; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 "$@" -c $f.hip -o $out/${f}_$tag.o & done; wait
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libhalo2_mi355x_$tag.so $out/capi_$tag.o $out/multi_$tag.o $out/xfer_$tag.o $out/ntt_$tag.o $out/poly_$tag.o $out/polyops_$tag.o $out/lookup_$tag.o $out/graph_$tag.o $out/msm_$tag.o $out/msm_small_$tag.o
rm -f $out/*_$tag.o; ls -la $out
