#!/bin/bash
# The NTT plan A/B in one GPU-box session: times of every build under gpurun_ab/ (tools/ntt_plans.py) and
# SQ_INSTS_VALU / GRBM_GUI_ACTIVE of the pass kernels for 8 x 2^21 per build (its own rocprofv3 --pmc run, the program after `--`).
#   usage (through gpurun): bash tools/ntt_plans.sh > gpurun_out/ntt_plans.txt
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; cd "$R"; export TMPDIR=/tmp
echo "== times"; timeout -k 10 600 python3 tools/ntt_plans.py || exit 1
for so in "" gpurun_ab/libhalo2_mi355x_*.so; do
  tag=default; [ -n "$so" ] && tag=$(basename "$so" .so | sed 's/libhalo2_mi355x_//')
  d=/tmp/ntt_plans_pmc_$tag; rm -rf "$d"
  echo "== SQ_INSTS_VALU, 8 x 2^21, build: $tag"
  if [ -n "$so" ]; then export HALO2_MI355X_LIB="$R/$so"; else unset HALO2_MI355X_LIB; fi
  (cd /tmp && timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d "$d" -- python3 "$R/tools/ntt_plans.py" --pmc-child 21 8 > /dev/null 2> "$d.err") || { tail -3 "$d.err"; exit 1; }
  python3 - "$d" <<'PY'
import csv, glob, sys
acc = {}
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
        if "ntt_pass_kernel" in name:
            a = acc.setdefault((name, row["Counter_Name"]), [0.0, 0]); a[0] += float(row["Counter_Value"]); a[1] += 1
tot = 0.0
for (name, ctr), (s, c) in sorted(acc.items()):
    print(f"  {name:60s} {ctr:16s} launches {c:3d}  per launch {s / c:.4g}")
    if ctr == "SQ_INSTS_VALU": tot += s
n_tr = 2            # two transforms of 8 x 2^21 in the child
print(f"  SQ_INSTS_VALU per element and transform: {tot / n_tr / (8 << 21):.2f} wave-instructions = {64 * tot / n_tr / (8 << 21):.0f} lane-instructions")
PY
done
