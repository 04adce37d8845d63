#!/bin/bash
# Kernel timeline of the eight-in-flight sparse / dense columns at k = 18 (tools/sparse_dense.py) under rocprofv3.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; rm -rf "$R/gpurun_out/trace_inflight"; mkdir -p "$R/gpurun_out/trace_inflight"; export TMPDIR=/tmp; cd /tmp
export GPU_MAX_HW_QUEUES="${GPU_MAX_HW_QUEUES:-16}"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$R/gpurun_out/trace_inflight" -- python3 "$R/tools/sparse_dense.py" 18 > "$R/gpurun_out/trace_inflight/out.txt" 2>&1 || { tail -5 "$R/gpurun_out/trace_inflight/out.txt"; exit 1; }
grep "2\^" "$R/gpurun_out/trace_inflight/out.txt"
python3 "$R/tools/timeline.py" "$R/gpurun_out/trace_inflight" 160
