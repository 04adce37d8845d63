#!/bin/bash
# One GPU-box session that produces EVERY file a round commits under profiles/, into ONE fresh directory
# gpurun_out/evidence/<tag>/ (removed first, so nothing of an earlier session can be picked up):
#   bench.json                 the un-profiled `python bench.py --steps 20 --warmup 5` line (the driver's command) (skipped with NOBENCH=1): the compact stdout line
#   bench_extras.json          the FULL record of that run (every side measurement; bench.py --extras-out)
#   bench_under_rocprof.json   the line printed under `rocprofv3 --kernel-trace --stats` (+ bench_under_rocprof_extras.json)
#   kernel_stats.csv           that run's per-kernel summary (the ONE *kernel_stats.csv of the fresh trace directory)
#   pmc_traffic.json           FETCH_SIZE / WRITE_SIZE per launch, two separate --pmc passes (MI355X_MICROARCH.md, HBM section)
#   sq_counters.txt            SQ_INSTS_VALU / GRBM_GUI_ACTIVE of the two hot kernels
#   manifest.json              sha256 of every source the numbers depend on, taken on the box
# tools/bake_counters.py --tag <tag> then CHECKS these against each other and copies them to profiles/<tag>_*.
#   usage (through gpurun): bash tools/profile_round.sh r04_a
set -o pipefail
TAG="${1:?usage: profile_round.sh <tag>}"
R="${GRAFT_REPO_ROOT:-/root/repo}"
E="$R/gpurun_out/evidence/$TAG"
rm -rf "$E"; mkdir -p "$E/pmc"
export TMPDIR=/tmp
BENCH_ARGS="--steps 5 --warmup 1 --no-cpu-baseline --replay none --no-extras --no-live-pmc"
cd "$R"
python3 tools/evidence.py manifest "$E/manifest.json" || exit 1
if [ -z "$NOBENCH" ]; then
  echo "== bench (un-profiled)"
  timeout -k 10 900 python3 bench.py --steps 20 --warmup 5 --extras-out "$E/bench_extras.json" > "$E/bench.json" 2> "$E/bench.err" || { tail -5 "$E/bench.err"; exit 1; }
fi
echo "== rocprofv3 --kernel-trace --stats"
(cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$E/kernel_trace" -- python3 "$R/bench.py" $BENCH_ARGS --extras-out "$E/bench_under_rocprof_extras.json" \
   > "$E/bench_under_rocprof.json" 2> "$E/rocprof.err") || { tail -5 "$E/rocprof.err"; exit 1; }
python3 tools/evidence.py pick "$E/kernel_trace" '*kernel_stats.csv' "$E/kernel_stats.csv" || exit 1
echo "== rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes)"
for ctr in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout -k 10 500 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d "$E/pmc/$ctr" -- python3 "$R/bench.py" \
     --steps 2 --warmup 1 --no-cpu-baseline --replay none --no-extras --no-live-pmc --no-collective --extras-out none > "$E/pmc/$ctr.json" 2> "$E/pmc/$ctr.err") || { tail -5 "$E/pmc/$ctr.err"; exit 1; }
done
python3 tools/pmc_summarise.py "$E/pmc" > "$E/pmc_traffic.json" || exit 1
echo "== rocprofv3 --pmc SQ counters"
(cd /tmp && timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
   -d "$E/pmc_sq" -- python3 "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --replay none --no-extras --no-live-pmc --no-collective --extras-out none > "$E/pmc_sq.json" 2> "$E/pmc_sq.err") \
   || { tail -5 "$E/pmc_sq.err"; exit 1; }
python3 tools/evidence.py sq "$E/pmc_sq" > "$E/sq_counters.txt" || exit 1
rm -rf "$E/kernel_trace" "$E/pmc/FETCH_SIZE" "$E/pmc/WRITE_SIZE" "$E/pmc_sq"      # raw traces are large; the summaries stay
python3 tools/evidence.py show "$E"
