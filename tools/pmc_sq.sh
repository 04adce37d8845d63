#!/bin/bash
# SQ-level counters of the headline kernels (issue / stall split, effective clock): development aid
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; rm -rf "$R/gpurun_out/pmc_sq"; mkdir -p "$R/gpurun_out/pmc_sq"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$R/gpurun_out/pmc_sq" -- \
  python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --replay none --no-extras --no-live-pmc > "$R/gpurun_out/pmc_sq/out.json" 2> "$R/gpurun_out/pmc_sq/err.txt" || { tail -5 "$R/gpurun_out/pmc_sq/err.txt"; exit 1; }
cd "$R" && python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob('gpurun_out/pmc_sq/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if n == 'hm::msm_accumulate_kernel' or n.startswith('hm::ntt_pass_kernel<11'):
            a = acc[n][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for n, d in acc.items():
    print(n, {k: round(v[0] / v[1]) for k, v in d.items()})
PY
