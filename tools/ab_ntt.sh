#!/bin/bash
# A second copy of the library that differs in ntt.hip's compile-time knobs only (the other objects are the tree's own):
#   tools/ab_ntt.sh p3 -DHM_NTT_2PASS_MAX=20   ->  gpurun_ab/libhalo2_mi355x_p3.so      (use: HALO2_MI355X_LIB=<path>, tools/ntt_plans.py)
set -e
tag=$1; shift
cd "$(dirname "$0")/../halo2-experiments_amd/csrc"
make -s libhalo2_mi355x.so
out=../../gpurun_ab; mkdir -p $out
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-result "$@" -c ntt.hip -o $out/ntt_$tag.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $out/libhalo2_mi355x_$tag.so capi.o multi.o xfer.o $out/ntt_$tag.o poly.o polyops.o lookup.o graph.o msm.o msm_small.o
rm -f $out/ntt_$tag.o; ls -la $out/libhalo2_mi355x_$tag.so
