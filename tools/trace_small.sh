#!/bin/bash
# Per-kernel averages of one small MSM size (default 2^18): where the latency of prover-sized calls goes.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; K="${1:-18}"; rm -rf "$R/gpurun_out/trace_small"; mkdir -p "$R/gpurun_out/trace_small"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/trace_small" -- python3 "$R/tools/msm_sweep.py" "$K" > "$R/gpurun_out/trace_small/out.txt" 2>&1 || { tail -5 "$R/gpurun_out/trace_small/out.txt"; exit 1; }
grep "2\^" "$R/gpurun_out/trace_small/out.txt"
python3 "$R/tools/kstats.py" "$R/gpurun_out/trace_small" 1000
