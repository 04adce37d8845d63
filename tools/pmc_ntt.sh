#!/bin/bash
# SQ-level counters of the NTT pass kernels at 2^24 (three passes of 2^11-element tiles): issue / wait split, LDS traffic and
# bank conflicts.  Development aid for DESIGN.md section 5.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/pmc_ntt"; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
ARGS="--steps 1 --warmup 0 --log-points 18 --no-cpu-baseline --replay none --no-extras --no-strong --no-live-pmc"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d "$OUT/p1" -- python3 "$R/bench.py" $ARGS > "$OUT/p1.json" 2> "$OUT/p1.err" || { tail -5 "$OUT/p1.err"; exit 1; }
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES \
  --kernel-trace --output-format csv -d "$OUT/p2" -- python3 "$R/bench.py" $ARGS > "$OUT/p2.json" 2> "$OUT/p2.err" || { tail -5 "$OUT/p2.err"; exit 1; }
cd "$R" && python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob('gpurun_out/pmc_ntt/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'ntt_pass' in n:
            a = acc[n][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for n, d in sorted(acc.items()):
    print(n)
    for k, v in sorted(d.items()):
        print(f"   {k:28s} {v[0] / v[1]:16.0f}   ({v[1]} launches)")
PY
