#!/bin/bash
# Sweep the knobs of the small-MSM plan (task length, bucket-range splits, segment length, window) per size.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/small_tune.txt; : > $out
run() { echo "## $*" >> $out; env "$@" python3 tools/msm_sweep.py $K $C 2>&1 | grep "2\^" >> $out; }
for K in 11 14 16 18; do
  case $K in 11) C="6,7,8,9";; 14) C="9,10,11,12";; 16) C="12,13,14,15";; 18) C="13,14,15";; esac
  run X=1
  run HALO2_MI355X_SMALL_L=12
  run HALO2_MI355X_SMALL_L=16
  run HALO2_MI355X_SMALL_L=24
done
K=18; C=15
for H in 1 2 4 8; do run HALO2_MI355X_SMALL_H=$H; done
for S in 3 4 5 6; do run HALO2_MI355X_SMALL_SEG=$S; done
K=16; C=15
for H in 1 2 4 8; do run HALO2_MI355X_SMALL_H=$H; done
cat $out
