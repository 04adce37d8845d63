#!/bin/bash
# rocprofv3 kernel trace + PMC traffic of the headline workload only (no replay / side measurements)
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R="$PWD"; mkdir -p gpurun_out; export TMPDIR=/tmp
rm -rf gpurun_out/prof gpurun_out/pmc
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/prof" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --replay none --no-extras > "$R/gpurun_out/rocprof_bench.json" 2> "$R/gpurun_out/rocprof.err") || { tail -3 gpurun_out/rocprof.err; exit 1; }
bash tools/pmc_traffic.sh > gpurun_out/pmc_stdout.txt 2>&1 || { tail -5 gpurun_out/pmc_stdout.txt; exit 1; }
python3 - <<'PY'
import csv, glob, json
f = glob.glob('gpurun_out/prof/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n = r['Name'].split('(')[0].replace('void ', '')
    if n.startswith('hm::') and float(r['AverageNs']) > 20000:
        print(f"{n:45s} calls {r['Calls']:>3s} avg {float(r['AverageNs'])/1e6:8.3f} ms")
print(open('gpurun_out/rocprof_bench.json').read()[:400])
PY
