#!/bin/bash
# SQ and HBM-traffic counters of hm::graph_evaluate_kernel on a circuit's whole evaluate_h program (tools/evalh_time.py): development aid.
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/pmc_evalh"; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
  d="$OUT/$(echo $set | cut -d' ' -f1)"
  timeout -k 10 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$d" -- python3 "$R/tools/evalh_time.py" "${1:-merkle_sum_tree_k18}" > "$d.txt" 2> "$d.err" || { tail -5 "$d.err"; exit 1; }
done
cd "$R" && python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in glob.glob('gpurun_out/pmc_evalh/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'graph_evaluate_kernel' in n:
            a = acc[n][r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
for n, d in sorted(acc.items()):
    print(n, {k: round(v[0] / v[1]) for k, v in d.items()})
PY
