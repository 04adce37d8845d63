#!/bin/bash
# Kernel timeline of a phase of commitments (tools/group_time.py <k> <counts>) under rocprofv3 --kernel-trace: development aid.
#   bash tools/phase_trace.sh 18 36 [tail]
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/phase_trace"; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
export GPU_MAX_HW_QUEUES="${GPU_MAX_HW_QUEUES:-16}"
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$OUT" -- python3 "$R/tools/group_time.py" "$1" "$2" > "$OUT/out.txt" 2>&1 || { tail -5 "$OUT/out.txt"; exit 1; }
grep "2\^" "$OUT/out.txt"
python3 "$R/tools/timeline.py" "$OUT" "${3:-0}"
