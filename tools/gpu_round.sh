#!/bin/bash
# One GPU-box session: parity tests, smoke, bench, rocprofv3 kernel trace of the same bench command.
set -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" && timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; rc=$?; tail -5 gpurun_out/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
echo "== smoke" && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; rc=$?; tail -2 gpurun_out/smoke.txt
[ $rc -ne 0 ] && exit $rc
echo "== bench" && timeout -k 10 600 python bench.py --steps 10 --warmup 2 > gpurun_out/bench.json 2> gpurun_out/bench.err; rc=$?; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
[ $rc -ne 0 ] && exit $rc
echo "== rocprofv3 kernel trace" && (cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -- python3 "$OLDPWD/bench.py" --steps 5 --warmup 1 --no-cpu-baseline --replay none --no-extras > "$OLDPWD/gpurun_out/rocprof_bench.json" 2> "$OLDPWD/gpurun_out/rocprof.err"); rc=$?
tail -2 gpurun_out/rocprof.err; find gpurun_out/prof -name "*stats*" | head
exit $rc
