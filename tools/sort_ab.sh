#!/bin/bash
# Same-box A/B of the MSM sort at 2^24 on the table: the default plan (9 coarse bits, 2^12 fine buckets, 16-byte runs in the
# second scatter) against 10 coarse bits without and with the carry-slot scatter (msm_part2_scatter_carry_kernel).  Per variant: the
# parity tests of tests/test_msm_gpu.py that reach the general pipeline, msm_sweep's phase timers, and the sort kernels' times from a
# kernel trace.  -> stdout (kept as profiles/r06_sort_ab.txt).
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; cd /tmp
run() {   # name, env...
  local name=$1; shift
  echo "== $name: $*"
  ( export "$@" DUMMY=1; cd "$R" && timeout -k 10 600 python -m pytest tests/test_msm_gpu.py -x -q -m gpu -k "golden or closed_form or edge or large or known or fixed or table or pipeline or 2_2" 2>&1 | tail -1 ) || return 1
  ( export "$@" DUMMY=1 PRECOMP=1; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/sortab_$name" -- python3 "$R/tools/msm_sweep.py" 24 > "$R/gpurun_out/sortab_$name.txt" 2>&1 ) || { tail -5 "$R/gpurun_out/sortab_$name.txt"; return 1; }
  grep "2^24" "$R/gpurun_out/sortab_$name.txt"
  f=$(find "$R/gpurun_out/sortab_$name" -name "*kernel_stats.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"].split("(")[0].replace("void ", "")
    if "part1" in n or "part2" in n:
        print(f"   {n:58s} calls {r['Calls']:>3s}  avg {float(r['AverageNs']) / 1e3:9.1f} us")
PY
}
run base HALO2_MI355X_CB_FIRST=9 && run cb10 HALO2_MI355X_CB_FIRST=10 && run cb10_carry HALO2_MI355X_CB_FIRST=10 HALO2_MI355X_P2_CARRY=1
