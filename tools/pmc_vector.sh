#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes) of the Fr vector kernels and the lookup sort at one
# size: the figures behind DESIGN.md section 5c's roofline sentences.  Usage: tools/pmc_vector.sh [log_n]
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; K="${1:-24}"
rm -rf "$R/gpurun_out/pmc_vec"; mkdir -p "$R/gpurun_out/pmc_vec"
export TMPDIR=/tmp
cd /tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_vec/$ctr" -- \
    python3 "$R/tools/polyops_time.py" "$K" > "$R/gpurun_out/pmc_vec/$ctr.txt" 2> "$R/gpurun_out/pmc_vec/$ctr.err" || { tail -5 "$R/gpurun_out/pmc_vec/$ctr.err"; exit 1; }
done
cd "$R" && python3 tools/pmc_summarise.py gpurun_out/pmc_vec > gpurun_out/pmc_vec/summary.json && cat gpurun_out/pmc_vec/summary.json
