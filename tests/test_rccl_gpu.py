"""The `nccl` (= RCCL) branch of the multi-GPU path, executed for real on the ONE GPU of the test box.

Every multi-rank rehearsal before round 6 ran over gloo (RCCL refuses two ranks on one device), and with one rank the exchange
helpers of sharding.py returned before their collective -- so `dist.all_gather` on DEVICE tensors over RCCL, the route the
driver's 8-GPU run takes, had never executed anywhere.  Here a fresh child process (started before
anything in it touches the GPU) builds a ONE-RANK RCCL communicator on cuda:0 and drives all four exchange functions through
it with ``force_collective=True``, comparing with the direct results; then, with HALO2_MI355X_FORCE_COLLECTIVE=1, a whole
create_proof replay whose every commitment phase crosses the communicator and is checked against [f(s)]G.
The child also reports the shared objects it has mapped: librccl must be among them.
"""
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


def _child(out_path):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as dist

    report = {}
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    one = torch.ones(1, dtype=torch.int64, device=dev)
    dist.all_reduce(one)
    report["ranks_seen"] = int(one.item())
    report["backend"] = dist.get_backend()

    import halo2_experiments_amd as h
    from halo2_experiments_amd import sharding
    from halo2_experiments_amd.arithmetic import G1_GENERATOR, best_multiexp_batch, random_fr

    # spy on the collective: the forced route must really call dist.all_gather with device tensors
    calls = {"all_gather": 0, "on_device": 0, "bytes": 0}
    max_gather = [0]
    real_all_gather = dist.all_gather

    def spy(tensor_list, tensor, group=None, async_op=False):
        calls["all_gather"] += 1
        calls["on_device"] += int(tensor.is_cuda and all(t.is_cuda for t in tensor_list))
        calls["bytes"] += tensor.numel() * tensor.element_size()
        max_gather[0] = max(max_gather[0], tensor.numel() * tensor.element_size())
        return real_all_gather(tensor_list, tensor, group=group, async_op=async_op)

    dist.all_gather = spy
    try:
        # (1) sharded_multiexp: a plain set (2^12) and a fixed-base table set (2^18: the registration default from 2^17)
        for log_n in (12, 18):
            n = 1 << log_n
            bases = h.g1_fixed_base_mul(random_fr(n, 11 + log_n, dev), G1_GENERATOR)
            hnd = h.register_bases(bases)
            s = random_fr(n, 21 + log_n, dev)
            direct = h.best_multiexp(s, hnd)
            before = calls["all_gather"]
            forced = sharding.sharded_multiexp(s, hnd, force_collective=True)
            assert calls["all_gather"] == before + 1
            unforced = sharding.sharded_multiexp(s, hnd)                      # one rank, not forced: no collective
            assert calls["all_gather"] == before + 1
            report[f"sharded_multiexp_2_{log_n}"] = bool(np.array_equal(direct, forced) and np.array_equal(direct, unforced) and direct.any())
            if log_n == 12:
                # (2) a phase of commitments: partials of every job in ONE all-gather
                cols = [random_fr(n, 100 + i, dev) for i in range(5)]
                want = best_multiexp_batch(cols, hnd)
                before = calls["all_gather"]
                got = sharding.sharded_multiexp_batch([(c, hnd) for c in cols], force_collective=True)
                assert calls["all_gather"] == before + 1
                report["sharded_multiexp_batch"] = bool(np.array_equal(want, got) and got.shape == (5, 12))
                # (3) whole jobs dealt round-robin (one rank owns them all), ONE all-gather of the results
                before = calls["all_gather"]
                got = sharding.job_parallel_multiexp_batch([(c, hnd) for c in cols], force_collective=True)
                assert calls["all_gather"] == before + 1
                report["job_parallel_multiexp_batch"] = bool(np.array_equal(want, got))
                assert sharding.job_parallel_multiexp_batch([], force_collective=True).shape == (0, 12)
            h.release_bases(hnd)
        # (4) evaluate_h's partials by cosets: n x 32 B per coset as device tensors through the communicator
        parts = {c: random_fr(1 << 10, 500 + c, dev) for c in range(8)}
        before = calls["all_gather"]
        allp = sharding.gather_coset_partials(parts, 8, force_collective=True)
        assert calls["all_gather"] == before + 1
        report["gather_coset_partials"] = bool(len(allp) == 8 and all(p.is_cuda and torch.equal(p, parts[c]) for c, p in enumerate(allp)))
        owners = sharding.coset_owners(5, 1, spare_rank0=True)
        allp = sharding.gather_coset_partials({c: parts[c] for c in range(5)}, 5, owners=owners, force_collective=True)
        report["gather_coset_partials_min_cosets"] = bool(all(torch.equal(p, parts[c]) for c, p in enumerate(allp)))
        # (5) a whole replay with the switch in the environment: every commitment phase of the proof crosses RCCL and every
        # commitment is checked against [f(s)]G inside run_replay (a mismatch raises)
        os.environ["HALO2_MI355X_FORCE_COLLECTIVE"] = "1"
        from halo2_experiments_amd.replay import run_replay
        before = calls["all_gather"]
        rep = run_replay("poseidon_k11", device=dev, include_host_pointer_estimate=False)
        report["replay_all_gathers"] = calls["all_gather"] - before
        report["replay_commitments_checked"] = rep["verified"]["commitments_checked"]
        # ... and the extended-domain steps BY COSETS (what N > 1 ranks do from k = 14): the partials of all 8 cosets, n x 32 B each, cross
        # the communicator as device tensors in one all-gather per proof
        before, bytes_before = calls["all_gather"], calls["bytes"]
        rep = run_replay("merkle_v3_k17", device=dev, include_host_pointer_estimate=False, by_cosets=True, min_cosets=False)
        report["coset_replay_all_gathers"] = calls["all_gather"] - before
        report["coset_replay_largest_gather_bytes"] = int(max_gather[0])
        report["coset_replay_extended_domain"] = rep["extended_domain"]
        os.environ.pop("HALO2_MI355X_FORCE_COLLECTIVE")
    finally:
        dist.all_gather = real_all_gather
    report["all_gather_calls"] = calls["all_gather"]
    report["all_gather_on_device"] = calls["on_device"]
    torch.cuda.synchronize()
    with open("/proc/self/maps") as f:
        libs = sorted({os.path.basename(line.split()[-1]) for line in f if ".so" in line and ("rccl" in line or "halo2_mi355x" in line)})
    report["mapped"] = libs
    dist.destroy_process_group()
    with open(out_path, "w") as f:
        json.dump(report, f)


@pytest.mark.gpu
def test_every_exchange_function_runs_over_a_one_rank_rccl_communicator(tmp_path):
    out = tmp_path / "rccl_child.json"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("HALO2_MI355X_FORCE_COLLECTIVE", None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(out)], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=420)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    rep = json.loads(out.read_text())
    assert rep["ranks_seen"] == 1 and rep["backend"] == "nccl"
    for key in ("sharded_multiexp_2_12", "sharded_multiexp_2_18", "sharded_multiexp_batch", "job_parallel_multiexp_batch",
                "gather_coset_partials", "gather_coset_partials_min_cosets"):
        assert rep[key] is True, (key, rep)
    assert rep["all_gather_calls"] == rep["all_gather_on_device"] >= 6         # every exchange was an RCCL all-gather of DEVICE tensors
    assert rep["replay_all_gathers"] >= 3 and rep["replay_commitments_checked"] > 0
    assert rep["coset_replay_extended_domain"].startswith("by cosets, 8 of 8")
    assert rep["coset_replay_largest_gather_bytes"] == 8 * (1 << 17) * 32            # every coset's partial of the k = 17 proof in ONE all-gather
    assert any("rccl" in name for name in rep["mapped"]), rep["mapped"]
    assert any(name.startswith("libhalo2_mi355x") for name in rep["mapped"]), rep["mapped"]


@pytest.mark.gpu
def test_bench_at_one_gpu_counts_its_ranks_with_an_rccl_all_reduce(tmp_path):
    """`bench.py --gpus 1` as the driver runs it (no torchrun, backend nccl): the timed step goes through the one-rank communicator."""
    extras = tmp_path / "extras.json"
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("HALO2_BENCH_BACKEND", "MASTER_PORT", "RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--log-points", "20",
                        "--no-cpu-baseline", "--no-ntt", "--replay", "none", "--no-extras", "--no-2-26", "--no-live-pmc",
                        "--extras-out", str(extras)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    (text,) = r.stdout.splitlines()             # ONE line on stdout: RCCL's version banner (it prints to stdout) went to stderr
    line = json.loads(text)
    full = json.loads(extras.read_text())
    assert full["collective"]["on"] is True and full["collective"]["backend"] == "rccl" and full["collective"]["error"] is None
    assert line["ranks_in_collective"] == 1 == full["collective"]["ranks_seen_by_all_reduce"]
    assert line["config"]["collective"].startswith("rccl all-gather") and line["known_answer_ok"] is True
    assert full["time_budget"]["dropped"] == [] and full["time_budget"]["budget_s"] == 300.0
    # a budget that is already spent drops every side leg and keeps the headline
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--log-points", "18",
                        "--no-cpu-baseline", "--no-live-pmc", "--replay", "poseidon_k11", "--time-budget", "0.001",
                        "--extras-out", str(extras)], cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stderr[-3000:]
    (text,) = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    line = json.loads(text)
    full = json.loads(extras.read_text())
    dropped = {d["leg"] for d in full["time_budget"]["dropped"]}
    assert {"ntt", "msm_side_measurements", "replay_poseidon_k11"} <= dropped, dropped
    assert line["value"] > 0 and line["known_answer_ok"] is True and line["summary"]["legs_dropped"] == len(dropped)
    assert "ntt" not in full and full.get("create_proof_replay") in ([], None)


if __name__ == "__main__":
    if len(sys.argv) == 3 and sys.argv[1] == "--child":
        _child(sys.argv[2])
    else:
        raise SystemExit("usage: test_rccl_gpu.py --child OUT.json (run by the test above)")
