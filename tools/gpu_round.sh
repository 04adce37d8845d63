#!/bin/bash
# One GPU-box session: parity tests, smoke, then the whole evidence set of a round (tools/profile_round.sh).
set -o pipefail
TAG="${1:?usage: gpu_round.sh <tag, e.g. r04_a>}"
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest -m gpu" && timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.txt 2>&1; rc=$?; tail -5 gpurun_out/pytest_gpu.txt
[ $rc -ne 0 ] && exit $rc
echo "== smoke" && timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.txt 2>&1; rc=$?; tail -2 gpurun_out/smoke.txt
[ $rc -ne 0 ] && exit $rc
echo "== evidence (bench, rocprofv3 kernel trace, PMC traffic, SQ counters) -> gpurun_out/evidence/$TAG"
bash tools/profile_round.sh "$TAG"
