#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary python tool: per-kernel calls / average: development aid.
#   bash tools/kernel_stats.sh tools/precomp_time.py 24
set -o pipefail
R="${GRAFT_REPO_ROOT:-/root/repo}"; OUT="$R/gpurun_out/kstats"; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp; cd /tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -- python3 "$R/$1" "${@:2}" > "$OUT/out.txt" 2> "$OUT/err.txt" || { tail -5 "$OUT/err.txt"; exit 1; }
cat "$OUT/out.txt"
python3 - "$OUT" "$R" <<'PY'
import csv, sys
sys.path.insert(0, sys.argv[2] + "/tools")
import evidence
f = evidence.pick_one(sys.argv[1], "*kernel_stats.csv")      # $OUT was removed above: exactly one trace can be there
rows = sorted(csv.DictReader(open(f)), key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:40]:
    n = r['Name'].split('(')[0].replace('void ', '')
    print(f"{n:60s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e3:10.1f} us  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
