"""bench.py's N > 1 path, rehearsed on the one-GPU box: two ranks launched exactly as the driver launches them
(`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 ...`), sharing device 0, the
96-byte partials travelling over gloo instead of RCCL (HALO2_BENCH_BACKEND=gloo: RCCL refuses two ranks on one device).
Checks the contract of the ONE JSON line: whole-job value, weak scaling, the folded global known answer."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(world, log_points):
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--log-points", str(log_points), "--no-cpu-baseline", "--no-ntt",
            "--replay", "none", "--no-extras", "--no-2-26"]
    env = dict(os.environ, HALO2_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # rank 0 prints ONE JSON line, the other ranks nothing
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_launched_like_the_driver_does():
    one = _bench(1, 20)
    two = _bench(2, 20)
    for line, world in ((one, 1), (two, 2)):
        assert line["n_gpus"] == world and line["steps"] == 2 and line["warmup"] == 1
        assert line["metric"] == "BN256 G1 MSM throughput" and line["unit"] == "points/s" and line["higher_is_better"] is True
        assert line["scaling"] == "weak" and line["vs_baseline"] is None and line["data"] == "synthetic"
        assert line["config"]["points_per_gpu"] == 1 << 20 and line["config"]["global_points"] == world << 20
        assert line["known_answer_ok"] is True                     # every rank's partial AND the folded global result
        assert abs(line["value"] - line["config"]["global_points"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
        assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert two["config"]["parallelism"].startswith("index-range shards x2")
