#!/bin/bash
# SQ-level counters of the MSM sort kernels at 2^24 on the table (bench.py --steps 2 ... under rocprofv3 --pmc, two passes):
# where the two scatter levels spend their wave cycles (LDS, vector memory, waiting).  Development aid for DESIGN.md section 4.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/pmc_sort"; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
ARGS="--steps 2 --warmup 1 --no-cpu-baseline --replay none --no-extras --no-ntt --no-strong"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d "$OUT/p1" -- python3 "$R/bench.py" $ARGS > "$OUT/p1.json" 2> "$OUT/p1.err" || { tail -5 "$OUT/p1.err"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM_WR \
  --kernel-trace --output-format csv -d "$OUT/p2" -- python3 "$R/bench.py" $ARGS > "$OUT/p2.json" 2> "$OUT/p2.err" || { tail -5 "$OUT/p2.err"; exit 1; }
cd "$R" && python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob('gpurun_out/pmc_sort/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'part1' in n or 'part2' in n or 'digits' in n:
            a = acc[n][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for n, d in sorted(acc.items()):
    print(n)
    for k, v in sorted(d.items()):
        print(f"   {k:28s} {v[0] / v[1]:16.0f}")
PY
