#!/bin/bash
# HBM traffic of the dominant kernels from the PMC counters, as MI355X_MICROARCH.md §HBM prescribes:
# FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 passes (TCC slots), counters in their own runs
# (no sys/hip/hsa trace domains), program directly after `--`.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf "$R/gpurun_out/pmc"; mkdir -p "$R/gpurun_out/pmc"     # a fresh directory: the summary averages every CSV it finds
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/gpurun_out/pmc/$ctr" -- \
    python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --replay none --no-extras --no-live-pmc > "$R/gpurun_out/pmc/$ctr.json" 2> "$R/gpurun_out/pmc/$ctr.err" || { tail -5 "$R/gpurun_out/pmc/$ctr.err"; exit 1; }
done
cd "$R" && python3 tools/pmc_summarise.py gpurun_out/pmc > gpurun_out/pmc/summary.json && cat gpurun_out/pmc/summary.json
