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


LINE_MAX_BYTES = 4096


class Record(dict):
    """The FULL record bench.py wrote to --extras-out, with the compact stdout line (what the driver parses) as `.line`."""
    line: dict


def _bench(world, log_points, replay="none", extra=(), extras=False):
    import tempfile
    fd, extras_out = tempfile.mkstemp(prefix="hm_bench_extras_", suffix=".json", dir="/tmp")
    os.close(fd)
    args = ["--gpus", str(world), "--steps", "2", "--warmup", "1", "--log-points", str(log_points), "--no-cpu-baseline", "--no-ntt",
            "--replay", replay, "--no-2-26", "--extras-out", extras_out] + ([] if extras else ["--no-extras"]) + list(extra)
    if "--live-pmc" in args:
        args.remove("--live-pmc")
    else:
        args.append("--no-live-pmc")
    env = dict(os.environ, HALO2_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if world == 1 or "--one-process" in extra:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py")] + args
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                     # rank 0 prints ONE JSON line, the other ranks nothing
    assert len(lines[0]) < LINE_MAX_BYTES, len(lines[0])           # ... short enough for the driver (round 4's 21 KB line: parsed null)
    assert len(out.stdout) < 2 * LINE_MAX_BYTES, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert list(line)[:8] == ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms"}
    assert all(not isinstance(v, (dict, list)) for v in line["summary"].values()) and len(line["summary"]) <= 10
    with open(extras_out) as f:
        full = Record(json.load(f))
    os.unlink(extras_out)
    for k in ("metric", "unit", "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "data", "known_answer_ok",
              "ranks_in_collective"):
        assert line[k] == full[k], k                               # the line is a projection of the record
    assert abs(line["value"] - full["value"]) < 1e-8 * full["value"]
    full.line = line
    return full


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
    assert one["ranks_in_collective"] == 1 and two["ranks_in_collective"] == 2
    # strong scaling: the SAME global size at both N (BASELINE configs[4] is a fixed 2^26; here 2^20), known answer of the fold
    for line, world in ((one, 1), (two, 2)):
        (e,) = line["strong_scaling"]
        assert e["global_log_points"] == 20 and e["n_gpus"] == world and e["points_per_rank"] == (1 << 20) // world
        assert e["known_answer_ok"] is True and e["ms_per_msm"] > 0
        assert abs(e["points_per_s"] - (1 << 20) / (e["ms_per_msm"] * 1e-3)) < 1e-6 * e["points_per_s"]
    # rank 0's one-process measurements (N > 1 only): the C-ABI split over "two devices" (device 0 listed twice here)
    assert "one_process" not in one
    op = two["one_process"]
    assert op["devices"] == [0, 0] and op["msm_split"]["known_answer_ok"] is True
    assert op["msm_split"]["base_set"]["devices"] == 2 and op["msm_split"]["from_host_array"]["same_result"] is True


@pytest.mark.gpu
def test_two_ranks_replay_the_k18_proof_and_rank0_replays_it_in_one_process():
    """The N > 1 replay (whole commitments dealt over the ranks, the extended-domain steps by cosets: replay.py job mode) launched the way the driver
    launches it, and -- in the same run -- the one-process replay by rank 0 with the other rank parked."""
    two = _bench(2, 18, replay="merkle_sum_tree_k18", extra=("--no-strong",))
    (rep,) = two["create_proof_replay"]
    assert rep["k"] == 18 and rep["n_gpus"] == 2 and rep["multi_gpu_split"].startswith("whole commitments")
    # both coset routes at N > 1: device_resident_s on all 8 cosets (the same h as N = 1's whole-array steps), the 5 that determine h beside it
    assert "8 of 8 cosets dealt over the ranks" in rep["multi_gpu_split"] and rep["extended_domain"].startswith("by cosets")
    routes = rep["extended_domain_routes_ms"]
    assert routes["by_all_cosets"]["total"] == pytest.approx(rep["device_resident_s"]["total"] * 1e3)
    assert routes["by_the_cosets_that_determine_h"]["extended_domain"].startswith("by cosets, 5 of 8")
    assert routes["by_all_cosets"]["evaluate_h"] > routes["by_the_cosets_that_determine_h"]["evaluate_h"] > 0
    assert two.line["summary"]["k18_replay_ms"] == pytest.approx(rep["device_resident_s"]["total"] * 1e3, rel=1e-3)
    assert two.line["summary"]["one_process_k18_replay_ms"] > 0
    assert rep["verified"]["commitments_checked"] >= 3 * (rep["calls"]["msm_sparse"] + rep["calls"]["msm_dense"])
    op = two["one_process"]["create_proof_replay"]
    assert op["k"] == 18 and op["multi_gpu_split"].startswith("one process")
    assert op["extended_domain"].startswith("by cosets over 2 devices")           # one host thread per listed device
    assert op["verified"]["commitments_checked"] >= 3 * (op["calls"]["msm_sparse"] + op["calls"]["msm_dense"])
    assert "strong_scaling" not in two


@pytest.mark.gpu
def test_one_gpu_line_carries_rank_shares_of_the_k18_replay():
    """N = 1: beside the replay itself, the first and the last rank's share of the 2-, 4- and 8-rank deal measured alone (what DESIGN 6's predicted curve
    is built on): less work with every doubling, the extended domain by cosets."""
    line = _bench(1, 18, replay="merkle_sum_tree_k18", extras=True)
    (rep,) = line["create_proof_replay"]
    shares = rep["rank_shares_measured_alone"]["shares"]
    assert [(s["world"], s["rank"]) for s in shares] == [(2, 0), (2, 1), (4, 0), (4, 3), (8, 0), (8, 7)] and all("error" not in s for s in shares), shares
    longest = [max(s["ms"]["total"] for s in shares if s["world"] == w) for w in (2, 4, 8)]
    assert rep["device_resident_s"]["total"] * 1e3 > longest[0] > longest[1] > longest[2] > 0
    assert all(s["extended_domain"].startswith("by cosets, 5 of 8") for s in shares)
    routes = rep["extended_domain_routes_ms"]          # the same trace with the extended-domain steps by all / by the determining cosets
    assert routes["whole_array"]["total"] == pytest.approx(rep["device_resident_s"]["total"] * 1e3)
    assert routes["by_all_cosets"]["evaluate_h"] > routes["by_the_cosets_that_determine_h"]["evaluate_h"] > 0
    one = rep["quotient_in_one_call_ms"]                  # hm_quotient_by_cosets_bn256_fr_dev on the circuit's own program, 83 columns
    assert "error" not in one and one["columns"] == 83 and one["columns_kept_on_the_cosets"] == 35
    assert one["all_cosets"]["ms"] > one["five_cosets"]["ms"] > 0 and one["five_cosets"]["cosets"] == 5


@pytest.mark.gpu
def test_one_process_form_as_the_whole_benchmark():
    """`bench.py --gpus 2 --one-process` (no torchrun): one JSON line with the same contract, the split inside the C ABI."""
    line = _bench(2, 20, extra=("--one-process",))
    assert line["n_gpus"] == 2 and line["known_answer_ok"] is True and line["scaling"] == "weak"
    assert line["config"]["global_points"] == 2 << 20 and line["config"]["devices"] == [0, 0]
    assert line["config"]["parallelism"].startswith("one process")
    assert abs(line["value"] - line["config"]["global_points"] / (line["ms_per_step"] * 1e-3)) < 1e-6 * line["value"]
    assert set(line["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}


@pytest.mark.gpu
def test_pmc_counters_are_measured_in_the_run():
    """N = 1: roofline.traffic and valu_issue come from child processes of bench.py under rocprofv3 --pmc (separate FETCH_SIZE /
    WRITE_SIZE passes), not from a committed file -- when rocprofv3 is there; without it the line says so and falls back."""
    import shutil
    line = _bench(1, 20, extra=("--live-pmc",))
    rf = line["roofline"]
    if shutil.which("rocprofv3") is None:
        assert "NOT measured in this run" in rf["traffic_note"]
        return
    assert rf["traffic_note"].startswith("MEASURED IN THIS RUN"), rf["traffic_note"]
    assert line.line["roofline"]["traffic_src"].startswith("rocprofv3 --pmc in this run") and line.line["roofline"]["traffic"] == pytest.approx(rf["traffic"])
    assert line.line["roofline"]["valu_issue_frac"] == pytest.approx(rf["valu_issue"]["frac"], rel=1e-3)
    assert rf["traffic"] > 96 * (1 << 20)                       # at least the algorithmic bytes of 2^20 points
    assert rf["valu_issue"]["stale"] is False and rf["valu_issue"]["sq_insts_valu_per_launch"] > 0
    assert 1000 < rf["valu_issue"]["wave_instr_per_64_units"] < 4000        # ~2 200 wave-instructions per 64 mixed additions
