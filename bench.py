#!/usr/bin/env python3
"""bench.py -- BN256 G1 MSM throughput on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one best_multiexp over the global array: every rank runs the MSM of its index-range
shard (2^24 points per GPU, inputs resident in HBM before the timed region), the 96-byte partial
results are all-gathered over RCCL and folded on every rank.  Weak scaling: N=1 is the north
star's 2^24 headline, N=4 is BASELINE config 5's 2^26.  Rank 0 prints ONE JSON line.
Beside the weak-scaling `value` the line carries, at every N:
  strong_scaling   BASELINE configs[4] as written -- ONE global 2^26-point MSM (and one global 2^24) split by index
                   range over the N ranks: ms per MSM (max over ranks), points/s, known answer of the folded result
  one_process      (N > 1) the form the reference's prover would use -- ONE process for the node
                   (/root/reference/src/circuits/utils.rs:22-70), hm_set_msm_devices(0..N-1): rank 0 alone, the other
                   ranks parked on the rendezvous store (CPU side) with their device memory released, splits one 2^26 MSM over the
                   N devices inside the C ABI (csrc/multi.hip) and replays the k = 18 create_proof trace with every
                   commitment phase dealt over the devices.  `python bench.py --gpus N --one-process` (no torchrun)
                   runs that form as the whole benchmark.
Inputs follow SURVEY.md §8d (uniform scalars from xoshiro256**, bases [a + i b]G); the result of the timed
steps is checked against the known answer [sum s_i (a + i b)]G outside the timed loop.

stdout carries ONE compact line (< 4 KB: the contract's keys, roofline, cpu_baseline, <= 10 scalar summaries -- compact_line());
the FULL record with every side measurement described here goes to bench_extras.json next to this script (--extras-out).
Objects of the full record:
  roofline      dominant kernel = msm_accumulate_kernel; achieved = 96 B/point (SURVEY.md §8d:
                32 B scalar + 64 B affine base, each read once) x points per launch / the
                launch's duration from HIP events on its own stream (hm_get_msm_stats).
                traffic / valu_issue: PMC counters measured IN THE RUN by three child processes of
                this script under `rocprofv3 --pmc` (N = 1; --no-live-pmc or no rocprofv3: the
                committed profiles/<tag>_baked_counters.json, flagged stale when the kernels changed).
  ntt           2^24 Fr NTT on one GPU (64 B/element algorithmic), same treatment.
  cpu_baseline  oracle/cpu_ref.c (C restatement of halo2_proofs v2023_02_02's best_multiexp) timed
                on this box's host cores on a bounded sample of the same workload (rank 0, N=1).
The oracle is used for nothing else here.
"""
import argparse
import ctypes
import json
import os
import sys
import time

PROCESS_T0 = time.perf_counter()          # the leg budget (class Legs) counts from here: the imports below are part of the run

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")   # eight MSMs in flight (see halo2-experiments_amd/_lib.py)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def to_host(t):
    """halo2_experiments_amd.arithmetic.to_host: device -> fresh numpy array through the library's lanes (never tensor.cpu() for big arrays)."""
    from halo2_experiments_amd.arithmetic import to_host as _to_host
    return _to_host(t)

# The real ceiling of both hot kernels: VALU issue.  256 CUs x 4 SIMD16 units, one
# wave64 VALU instruction per 4 cycles, at the 2.4 GHz peak engine clock (the chip holds ~2.07-2.2 GHz
# under these kernels).  Instructions per unit are a property of the build, measured with SQ_INSTS_VALU
# (tools/pmc_sq.sh) and baked -- with a hash of the kernel sources they were measured on -- by
# tools/bake_counters.py; the bench line says "stale": true when those sources have changed since.
VALU_PEAK_WAVE_INST_PER_S = 256 * 4 * 2.4e9 / 4


def newest_baked_counters_file() -> str:
    """profiles/<tag>_baked_counters.json with the greatest tag (rNN_x, written by tools/bake_counters.py only after its
    consistency checks passed); older rounds' files have no letter (r03_baked_counters.json)."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]*_baked_counters.json")))
    return os.path.relpath(found[-1], ROOT) if found else "profiles/none"


BAKED_COUNTERS_FILE = newest_baked_counters_file()


def baked_counters():
    """(figures, stale flags) of the committed counter file; stale = the kernel sources hash differently today."""
    import hashlib
    try:
        with open(os.path.join(ROOT, BAKED_COUNTERS_FILE)) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, {"k3": True, "ntt": True}
    stale = {}
    for kind in ("k3", "ntt"):
        hsh = hashlib.sha256()
        try:
            for name in d[kind]["sources"]:
                with open(os.path.join(ROOT, "halo2-experiments_amd", "csrc", name), "rb") as f:
                    hsh.update(name.encode() + b"\0" + f.read() + b"\0")
            stale[kind] = hsh.hexdigest() != d[kind]["sources_sha256"]
        except (OSError, KeyError):
            stale[kind] = True
    return d, stale


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MSM_BYTES_PER_POINT = 96
NTT_BYTES_PER_ELEM = 64


# ---- the ONE stdout line ------------------------------------------------------------------------------------------------
# The driver reads ONE short JSON line (round 4's 21 KB line came back `parsed: null`).  Everything measured goes into the
# FULL record, written to bench_extras.json next to this script (--extras-out); stdout carries compact_line(full): the
# contract's keys, roofline, cpu_baseline and at most SUMMARY_MAX scalar summaries.  Bounded: tests/test_bench_line.py.
LINE_MAX_BYTES = 4096
LINE_MAX_STRING = 120
SUMMARY_MAX = 12
EXTRAS_FILE = "bench_extras.json"


def _short(s, limit=LINE_MAX_STRING):
    s = str(s)
    return s if len(s) <= limit else s[:limit - 1] + "~"


def _r(x, digits=6):
    """Numbers of the line with `digits` significant figures (a 17-digit float is 10 bytes of noise)."""
    if isinstance(x, bool) or x is None or isinstance(x, int):
        return x
    try:
        return float(f"{float(x):.{digits}g}")
    except (TypeError, ValueError):
        return x


def _get(d, *path, default=None):
    for k in path:
        if isinstance(d, dict) and k in d:
            d = d[k]
        elif isinstance(d, (list, tuple)) and isinstance(k, int) and -len(d) <= k < len(d):
            d = d[k]
        else:
            return default
    return d


def compact_line(full: dict, extras_file: str = EXTRAS_FILE) -> dict:
    """The short line printed on stdout, from the full record.  Pure (no GPU, no files): the CPU suite bounds its size on a stub."""
    rf = full.get("roofline") or {}
    cfg = full.get("config") or {}
    line = {k: _r(full.get(k), 10) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                             "scaling", "vs_baseline", "dtype", "data")}
    line["dtype"] = _short(line["dtype"], 16)
    base_set = cfg.get("base_set")
    if isinstance(base_set, dict):
        base_set = "fixed-base table, sliced over the devices" if base_set.get("sliced") else "fixed-base table"
    line["config"] = {"workload": _short(cfg.get("workload")), "points_per_gpu": cfg.get("points_per_gpu"),
                      "global_points": cfg.get("global_points"), "base_set": _short(base_set),
                      "window_bits": cfg.get("window_bits"), "windows": cfg.get("windows"), "parallelism": _short(cfg.get("parallelism"))}
    if "devices" in cfg:
        line["config"]["devices"] = cfg["devices"]
    # what the headline's base set COSTS (registration is outside the timed region) and its sibling on the other layout -- in
    # `config` because that is an object the driver's record keeps whole
    for k_ in ("base_set_register_ms", "base_set_bytes"):
        if cfg.get(k_) is not None:
            line["config"][k_] = _r(cfg[k_], 5)
    plain = full.get("msm_plain_bases")
    if plain:
        line["config"]["plain_layout"] = {"points_per_s": _r(plain.get("points_per_s"), 4), "ms": _r(plain.get("ms"), 4),
                                          "register_ms": _r(plain.get("register_ms"), 4), "base_set_bytes": plain.get("base_set_bytes"),
                                          "note": "the same MSM on ONE copy of the points (best_multiexp's arbitrary bases per call)"}
    if full.get("collective"):
        line["config"]["collective"] = _short(_get(full, "collective", "summary"), 100)
    line["roofline"] = {"bound": rf.get("bound"), "achieved": _r(rf.get("achieved")), "peak": rf.get("peak"), "unit": rf.get("unit"),
                        "frac": _r(rf.get("frac")), "traffic": _r(rf.get("traffic"), 9), "kernel": rf.get("kernel"),
                        "kernel_ms": _r(rf.get("kernel_ms")), "algorithmic_bytes": rf.get("algorithmic_bytes"),
                        "traffic_src": _short(rf.get("traffic_src"), 64), "valu_issue_frac": _r(_get(rf, "valu_issue", "frac"), 4),
                        "note": _short(rf.get("note"))}
    cpu = full.get("cpu_baseline")
    if cpu:
        line["cpu_baseline"] = {"value": _r(cpu.get("value")), "unit": cpu.get("unit"), "cores": cpu.get("cores"), "kind": cpu.get("kind"),
                                "sample": _short(cpu.get("sample")), "agrees_with_gpu": cpu.get("agrees_with_gpu")}
    line["known_answer_ok"] = full.get("known_answer_ok")
    line["ranks_in_collective"] = full.get("ranks_in_collective")
    # at most SUMMARY_MAX scalar summaries; the objects they come from are in the extras file
    s = {}
    if full.get("msm_phase_ms"):
        s["msm_sort_ms"] = _r(_get(full, "msm_phase_ms", "sort"), 4)
    if plain:
        s["msm_plain_points_per_s"] = _r(plain.get("points_per_s"), 4)
    if full.get("ntt"):
        s[f"ntt_2_{_get(full, 'ntt', 'log_n')}_ms"] = _r(_get(full, "ntt", "ms"), 4)
        s["ntt_hbm_frac"] = _r(_get(full, "ntt", "roofline", "frac"), 4)
        if _get(full, "ntt", "cpu_baseline", "ms") is not None:
            s["ntt_cpu_ms"] = _r(_get(full, "ntt", "cpu_baseline", "ms"), 4)
    rep9 = next((r for r in full.get("create_proof_replay") or [] if r.get("k") == 9), None)
    if rep9:                          # the reference's own proving configuration (merkle_sum_tree.rs:345-358)
        s["k9_replay_ms"] = _r(1e3 * _get(rep9, "device_resident_s", "total", default=0.0), 4)
        if _get(rep9, "cpu_baseline", "total_s") is not None:
            s["k9_cpu_msm_ntt_ms"] = _r(1e3 * _get(rep9, "cpu_baseline", "total_s"), 4)
    for e in full.get("strong_scaling") or []:
        if e.get("global_log_points") == 26:
            s["msm_2_26_global_points_per_s"] = _r(e.get("points_per_s"), 4)
    reps = full.get("create_proof_replay") or []
    rep = next((r for r in reps if r.get("k") == 18), reps[-1] if reps else None)
    if rep:
        kk = f"k{rep.get('k')}"
        s[f"{kk}_replay_ms"] = _r(1e3 * _get(rep, "device_resident_s", "total", default=0.0), 4)
        if _get(rep, "cpu_baseline", "total_s") is not None:
            s[f"{kk}_cpu_msm_ntt_s"] = _r(_get(rep, "cpu_baseline", "total_s"), 4)
        if _get(rep, "total_s", "drop_in_host_pointers") is not None:
            s[f"{kk}_drop_in_ms"] = _r(1e3 * _get(rep, "total_s", "drop_in_host_pointers"), 4)
        if _get(rep, "total_s", "drop_in_with_domain_edits") is not None:
            s[f"{kk}_drop_in_domain_edits_ms"] = _r(1e3 * _get(rep, "total_s", "drop_in_with_domain_edits"), 4)
        if _get(rep, "rust_device_glue", "total_ms") is not None:     # the proof through exactly mi355x_dev.rs's entry points
            s[f"{kk}_rust_dev_glue_ms"] = _r(_get(rep, "rust_device_glue", "total_ms"), 4)
    op = full.get("one_process")
    if op:
        if "error" in op:
            s["one_process_error"] = _short(op["error"], 80)
        else:
            s[f"one_process_msm_2_{_get(op, 'msm_split', 'global_log_points', default=26)}_ms"] = _r(_get(op, "msm_split", "ms_per_msm"), 4)
            if _get(op, "create_proof_replay", "device_resident_s", "total") is not None:
                s["one_process_k18_replay_ms"] = _r(1e3 * _get(op, "create_proof_replay", "device_resident_s", "total"), 4)
    if _get(full, "time_budget", "dropped"):
        s["legs_dropped"] = len(full["time_budget"]["dropped"])
    order = ["k18_replay_ms", "msm_plain_points_per_s", "msm_2_26_global_points_per_s", "one_process_msm_2_26_ms", "one_process_k18_replay_ms",
             "one_process_error", "legs_dropped", "ntt_2_24_ms", "ntt_hbm_frac", "k9_replay_ms", "k9_cpu_msm_ntt_ms", "k18_rust_dev_glue_ms",
             "k18_drop_in_ms", "k18_cpu_msm_ntt_s", "k18_drop_in_domain_edits_ms", "msm_sort_ms", "ntt_cpu_ms"]
    rank_of = lambda k_: order.index(k_) if k_ in order else (3 if k_.startswith("one_process_msm_") else len(order))
    keys = sorted(s, key=rank_of)                                                         # the first SUMMARY_MAX, by what a scaling record needs most
    line["summary"] = {k_: s[k_] for k_ in keys[:SUMMARY_MAX]}
    line["extras_file"] = extras_file
    return line


_REAL_STDOUT_FD = None


def guard_stdout():
    """stdout carries ONE line.  RCCL prints a version banner on STDOUT when a communicator is built (five lines, seen in round 6's
    first one-rank run), gloo prints its own: file descriptor 1 is pointed at stderr for the life of the process, and emit() writes
    the line to the saved descriptor -- whatever a library prints, the driver reads exactly one JSON line."""
    global _REAL_STDOUT_FD
    if _REAL_STDOUT_FD is None:
        sys.stdout.flush()
        _REAL_STDOUT_FD = os.dup(1)
        os.dup2(2, 1)


def emit(full: dict, extras_out: str):
    """Write the full record to `extras_out`, print the compact line (the only thing this script writes to stdout)."""
    name = None
    if extras_out and extras_out != "none":
        path = extras_out if os.path.isabs(extras_out) else os.path.join(ROOT, extras_out)
        try:
            with open(path, "w") as f:
                json.dump(full, f, indent=1)
            name = os.path.relpath(path, ROOT)
        except OSError as e:
            name = f"not written ({type(e).__name__})"
    line = compact_line(full, name)
    text = json.dumps(line)
    if len(text) >= LINE_MAX_BYTES:            # never print a line the driver cannot parse: drop the summaries first,
        line["summary"] = {}
        text = json.dumps(line)
    if len(text) >= LINE_MAX_BYTES:            # then everything of config but the workload
        line["config"] = {"workload": line["config"]["workload"]}
        text = json.dumps(line)
    if _REAL_STDOUT_FD is None:
        print(text)
        sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_REAL_STDOUT_FD, (text + "\n").encode())


def rand_fr(n, seed, device):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, device)                 # uniform over the whole of [0, r)


def fq_mont_words(v):
    p = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47
    m = v * (1 << 256) % p
    return [(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)]


def cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(log_sample, device, full):
    """Time the oracle's best_multiexp (by default on the WHOLE 2^24 workload: ~10 s of CPU work on 16 cores, so no smaller
    sample is needed; --cpu-log-sample shrinks it).  oracle/ is used in this file by the three *cpu_baseline functions only."""
    import halo2_experiments_amd as h
    from oracle import cpu_ref
    cpu_ref.build()
    n = 1 << log_sample
    scalars_d, bases_d, _ = bench_inputs(n, 0, 4242, device)
    bases = to_host(bases_d)
    scalars = to_host(scalars_d)
    threads = cpu_ref.default_threads()            # scheduler affinity capped by the container's CPU quota (HALO2_CPU_THREADS overrides)
    t0 = time.perf_counter()
    ref = cpu_ref.best_multiexp(scalars, bases, threads)
    dt = time.perf_counter() - t0
    got = h.best_multiexp(scalars, bases)
    ok = bool(np.array_equal(cpu_ref.g1_to_affine(ref)[0], got[:8]))
    return {"value": n / dt, "unit": "points/s", "cores": threads, "cpus_visible": os.cpu_count(), "cpu_model": cpu_model(), "kind": "port",
            "sample": f"one 2^{log_sample}-point MSM ({'the whole timed workload' if full else 'a bounded sample'}), {dt:.2f} s wall; "
                      "oracle/cpu_ref.c, C port of halo2_proofs best_multiexp",
            "agrees_with_gpu": ok}


BENCH_SEED = 0x48324D4933353558          # "H2MI355X" 


def bench_inputs(n, first_index, seed, device):
    """scalars uniform in [0, r) from xoshiro256** (one stream per element, seeded from `seed`), bases
    P_i = [a + i b]G with a, b derived from seed + 1 (distinct, never the identity, and the MSM's answer is known:
    [sum_i s_i (a + i b)]G).  Everything is produced on the device; returns (scalars, bases, t) tensors."""
    import ctypes
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import G1_GENERATOR, _ptr, _stream_ptr
    from halo2_experiments_amd.domain import FR_MODULUS, fr_words
    lib = _lib.load()
    a = pow(seed + 1, 3, FR_MODULUS) or 1
    b = pow(seed + 1, 5, FR_MODULUS) or 1
    scalars = torch.empty((n, 4), dtype=torch.int64, device=device)
    t = torch.empty((n, 4), dtype=torch.int64, device=device)
    st = ctypes.c_void_p(_stream_ptr(scalars))
    _lib.check(lib.hm_fr_random_dev(ctypes.c_void_p(scalars.data_ptr()), n, seed + first_index, st))
    _lib.check(lib.hm_fr_affine_sequence_dev(ctypes.c_void_p(t.data_ptr()), n, _ptr(fr_words((a + first_index * b) % FR_MODULUS)),
                                              _ptr(fr_words(b)), st))
    bases = h.g1_fixed_base_mul(t, G1_GENERATOR)
    return scalars, bases, t


def known_answer(scalars, t, device):
    """[sum_i s_i t_i]G as 12 words (x, y, 1), through the device inner product and one fixed-base multiplication."""
    import ctypes
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import FQ_ONE_MONT, G1_GENERATOR, _ptr, _stream_ptr
    dot = np.zeros(4, dtype=np.uint64)
    _lib.check(_lib.load().hm_fr_dot_bn256_dev(ctypes.c_void_p(scalars.data_ptr()), ctypes.c_void_p(t.data_ptr()), scalars.shape[0],
                                                _ptr(dot), ctypes.c_void_p(_stream_ptr(scalars))))
    pt = h.g1_fixed_base_mul(torch.from_numpy(dot.view(np.int64).reshape(1, 4)).to(device), G1_GENERATOR).cpu().numpy().view(np.uint64)[0]
    out = np.zeros(12, dtype=np.uint64)
    if pt.any():
        out[:8] = pt
        out[8:] = FQ_ONE_MONT
    return out


def pmc_traffic(baked, kernel, corrected):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary (tools/pmc_traffic.sh, separate
    FETCH_SIZE / WRITE_SIZE passes of this same command).  NOT measured in this run: the bench line names
    the profiles/ file the figure comes from and whether the kernel sources have changed since.  `corrected`
    applies the gfx950 x2 on FETCH_SIZE (right for wide coalesced streams; 64-byte gathers are counted 1:1:
    tools/ubench/gather64.hip, profiles/r02_gather64_calibration.json)."""
    try:
        d = baked["traffic"][kernel]
        if corrected:
            return d["hbm_bytes_per_launch_corrected"]
        return (d["FETCH_SIZE_KiB_per_launch"] + d["WRITE_SIZE_KiB_per_launch"]) * 1024.0
    except (KeyError, TypeError):
        return None


def valu_issue(baked, stale, kind: str, units: float, kernel_ms: float):
    """The binding resource of the hot kernels, next to the contract's HBM roofline: VALU wave-instructions
    issued per second against what 1024 SIMD16 units can issue at the peak clock.  `units` = (point, bucket)
    pairs for K3, element-passes for the NTT."""
    try:
        per_unit = baked[kind]["sq_insts_valu_per_launch"] / baked[kind]["pairs_per_launch" if kind == "k3" else "elements_per_launch"]
    except (KeyError, TypeError):
        return None
    achieved = per_unit * units / (kernel_ms * 1e-3)
    return {"achieved": achieved, "peak": VALU_PEAK_WAVE_INST_PER_S, "unit": "wave-instr/s",
            "frac": achieved / VALU_PEAK_WAVE_INST_PER_S, "wave_instr_per_64_units": per_unit * 64, "stale": bool(stale[kind]),
            "note": f"NOT measured in this run: SQ_INSTS_VALU per unit from {baked[kind]['from']} via {BAKED_COUNTERS_FILE}; "
                    "stale = the kernel sources listed there hash differently now; peak = 1024 SIMDs x 2.4 GHz / 4"}


def cpu_info():
    from oracle import cpu_ref
    return {"cores": cpu_ref.default_threads(), "cpus_visible": os.cpu_count(), "cpu_model": cpu_model(), "kind": "port"}


def ntt_cpu_baseline(log_n, device):
    """oracle/cpu_ref.c:ref_best_fft (the C restatement of upstream's best_fft: serial bit reversal, serial twiddle
    table, recursive butterflies on the thread pool) on the same input as the timed GPU transform."""
    import halo2_experiments_amd as h
    from halo2_experiments_amd.domain import FR_MODULUS, FR_ROOT_OF_UNITY, fr_words
    from oracle import cpu_ref
    cpu_ref.build()
    omega = fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - log_n), FR_MODULUS))
    a_d = rand_fr(1 << log_n, 99, device)
    a = to_host(a_d)
    info = cpu_info()
    t0 = time.perf_counter()
    ref = cpu_ref.best_fft(a, omega, log_n, info["cores"])
    dt = time.perf_counter() - t0
    h.best_fft(a_d, omega, log_n)
    torch.cuda.synchronize()
    ok = bool(np.array_equal(ref, to_host(a_d)))
    info.update({"value": (1 << log_n) / dt, "unit": "elements/s", "ms": dt * 1e3,
                 "sample": f"one 2^{log_n}-element best_fft (the whole timed workload, same input), {dt:.2f} s wall incl. the copy of the "
                           "array; C restatement of halo2_proofs v2023_02_02 best_fft (not the Rust binary)",
                 "agrees_with_gpu": ok})
    return info


def replay_cpu_baseline(rep, device):
    """The CPU counterpart of the four timings the reference prints (/root/reference/src/circuits/utils.rs:66-69) for one
    replayed create_proof: each DISTINCT call shape of the trace timed once on the oracle (MSM 2^k on a sparse advice-like
    column and on a dense column; best_fft at 2^k and at 2^extended_k), times the trace's call counts.  The coset shift,
    zero padding and divisor sweeps upstream does around its FFTs, and everything of create_proof that is not an MSM or
    an NTT, are NOT in this figure (it is the counterpart of device_resident_s.msm + .ntt, not of .total)."""
    import halo2_experiments_amd as h
    from halo2_experiments_amd.arithmetic import G1_GENERATOR
    from halo2_experiments_amd.domain import FR_MODULUS, FR_ROOT_OF_UNITY, fr_words
    from halo2_experiments_amd.replay import SHAPES, _rand_fr, _sparse_column
    from oracle import cpu_ref
    cpu_ref.build()
    k, ek = rep["k"], rep["extended_k"]
    n = 1 << k
    used = SHAPES[rep["shape_key"]].used_rows
    bases = to_host(h.g1_fixed_base_mul(_rand_fr(n, 77, device), G1_GENERATOR))
    dense = to_host(_rand_fr(n, 78, device))
    sparse = to_host(_sparse_column(n, used, 79, device))
    info = cpu_info()
    T = info["cores"]

    def timed(fn):
        t0 = time.perf_counter()
        fn()
        return time.perf_counter() - t0

    # The port starts its threads per call (upstream keeps a rayon pool): below 2^14 that start-up is most of a call, so small shapes
    # are timed on ONE thread as well and the faster figure stands for a pooled runtime
    def best_threads(fn, log_size):
        t = timed(lambda: fn(T))
        if log_size < 14:
            t = min(t, timed(lambda: fn(T)), timed(lambda: fn(1)), timed(lambda: fn(1)))
        return t

    per = {"msm_sparse": best_threads(lambda th: cpu_ref.best_multiexp(sparse, bases, th), k),
           "msm_dense": best_threads(lambda th: cpu_ref.best_multiexp(dense, bases, th), k),
           "intt_n": best_threads(lambda th: cpu_ref.best_fft(dense, fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - k), FR_MODULUS)), k, th), k)}
    ext = np.zeros((1 << ek, 4), dtype=np.uint64)
    ext[:n] = dense
    per["coset_ntt_ext"] = best_threads(lambda th: cpu_ref.best_fft(ext, fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - ek), FR_MODULUS)), ek, th), ek)
    per["intt_ext"] = per["coset_ntt_ext"]
    calls = rep["calls"]
    msm = per["msm_sparse"] * calls["msm_sparse"] + per["msm_dense"] * calls["msm_dense"]
    ntt = per["intt_n"] * calls["intt_n"] + per["coset_ntt_ext"] * (calls["coset_ntt_ext"] + calls["intt_ext"])
    info.update({"per_call_s": per, "calls": calls, "msm_s": msm, "ntt_s": ntt, "total_s": msm + ntt, "unit": "s",
                 "gpu_msm_plus_ntt_s": rep["device_resident_s"]["msm"] + rep["device_resident_s"]["ntt"],
                 "sample": "each distinct call shape timed ONCE on the oracle (oracle/cpu_ref.c, C restatement of halo2_proofs "
                           "v2023_02_02 best_multiexp / best_fft; shapes below 2^14: the faster of `cores` threads and one thread, twice "
                           "each), multiplied by the trace's call counts; MSM + best_fft only"})
    return info


def rust_threshold_account(rep):
    """Which calls of a replayed trace the Rust shim's size thresholds (GPU_MIN_LOG_N_MSM / GPU_MIN_LOG_N_NTT of
    rust/halo2_proofs-patch/src/mi355x.rs, read from that file) would leave on the CPU, and what the proof's MSM + NTT calls cost
    on each route -- from per-call times MEASURED in this run: the CPU port (rep.cpu_baseline.per_call_s) and the drop-in
    host-pointer calls (rep.host_pointer_estimate_s).  Written for the reference's own k = 9 case, which sits at the crossover."""
    import re
    try:
        with open(os.path.join(ROOT, "rust", "halo2_proofs-patch", "src", "mi355x.rs")) as f:
            text = f.read()
        thr = {name: int(re.search(rf"pub const GPU_MIN_LOG_N_{name}: u32 = (\d+);", text).group(1)) for name in ("MSM", "NTT")}
        cpu, hp, calls = rep["cpu_baseline"]["per_call_s"], rep["host_pointer_estimate_s"], rep["calls"]
    except (OSError, AttributeError, KeyError, TypeError):
        return None
    k, ek = rep["k"], rep["extended_k"]
    rows = [  # (what, log size, calls, CPU port s per call, GPU drop-in s per call, threshold)
        ("best_multiexp, sparse columns", k, calls["msm_sparse"], cpu["msm_sparse"], hp["msm_each"], thr["MSM"]),
        ("best_multiexp, dense columns", k, calls["msm_dense"], cpu["msm_dense"], hp["msm_each"], thr["MSM"]),
        ("best_fft 2^k", k, calls["intt_n"], cpu["intt_n"], hp["ntt_n_each"], thr["NTT"]),
        ("best_fft 2^extended_k", ek, calls["coset_ntt_ext"] + calls["intt_ext"], cpu["coset_ntt_ext"], hp["ntt_ext_each"], thr["NTT"])]
    out_rows, tot = [], {"all_on_cpu_port": 0.0, "all_through_the_gpu_drop_in": 0.0, "as_the_shim_routes_them": 0.0}
    for what, lg, cnt, c_s, g_s, th in rows:
        on_gpu = lg >= th
        out_rows.append({"call": what, "log_n": lg, "calls": cnt, "cpu_port_ms_each": c_s * 1e3, "gpu_drop_in_ms_each": g_s * 1e3,
                         "shim_routes_to": "gpu" if on_gpu else "cpu", "gpu_wins_per_call": bool(g_s < c_s)})
        tot["all_on_cpu_port"] += cnt * c_s
        tot["all_through_the_gpu_drop_in"] += cnt * g_s
        tot["as_the_shim_routes_them"] += cnt * (g_s if on_gpu else c_s)
    return {"thresholds": {"GPU_MIN_LOG_N_MSM": thr["MSM"], "GPU_MIN_LOG_N_NTT": thr["NTT"], "from": "rust/halo2_proofs-patch/src/mi355x.rs"},
            "calls": out_rows, "msm_plus_ntt_ms": {k_: v * 1e3 for k_, v in tot.items()},
            "device_resident_msm_plus_ntt_ms": (rep["device_resident_s"]["msm"] + rep["device_resident_s"]["ntt"]) * 1e3,
            "calls_left_on_the_cpu": sum(r["calls"] for r in out_rows if r["shim_routes_to"] == "cpu"),
            "gpu_route_wins": bool(tot["as_the_shim_routes_them"] < tot["all_on_cpu_port"]),
            "cpu_threads": rep["cpu_baseline"]["cores"],
            "note": "MSM + best_fft calls only, per-call time x the trace's counts; the CPU port runs on cpu_threads threads"}


def live_pmc(log_points, log_ntt, with_ntt):
    """HBM traffic and VALU instruction counts of the two hot kernels MEASURED IN THIS RUN: three child processes, each this
    same script (two timed steps of the headline workload, nothing else) under `rocprofv3 --pmc <one counter group>
    --kernel-trace`, exactly as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in separate passes, counters in
    their own runs, the program directly after `--`).  Returns {kernel: {counter: average per launch}} or an {"error": ...}
    -- never raises: a box without rocprofv3 falls back to the committed figures."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    if shutil.which("rocprofv3") is None:
        return {"error": "rocprofv3 not on PATH"}
    out = {}
    tmp = tempfile.mkdtemp(prefix="hm_pmc_", dir="/tmp")
    try:
        child = [sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--log-points", str(log_points), "--log-ntt",
                 str(log_ntt), "--no-cpu-baseline", "--replay", "none", "--no-extras", "--no-strong", "--no-live-pmc", "--no-collective", "--extras-out", "none"] + ([] if with_ntt else ["--no-ntt"])
        env = dict(os.environ, TMPDIR="/tmp")
        for group in (["FETCH_SIZE"], ["WRITE_SIZE"], ["SQ_INSTS_VALU", "GRBM_GUI_ACTIVE"]):
            d = os.path.join(tmp, group[0])
            r = subprocess.run(["rocprofv3", "--pmc", *group, "--kernel-trace", "--output-format", "csv", "-d", d, "--", *child], cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return {"error": f"rocprofv3 --pmc {' '.join(group)} failed (rc {r.returncode}): {r.stderr[-300:]}"}
            acc = {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f) as fh:
                    for row in csv.DictReader(fh):
                        name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                        if name == "hm::msm_accumulate_kernel" or name.startswith("hm::ntt_pass_kernel<11"):
                            a = acc.setdefault((name, row["Counter_Name"]), [0.0, 0])
                            a[0] += float(row["Counter_Value"])
                            a[1] += 1
            if not acc:
                return {"error": f"no counter rows for the hot kernels in the {group[0]} pass"}
            for (name, ctr), (tot, cnt) in acc.items():
                out.setdefault(name, {})[ctr] = tot / cnt
                out[name]["launches_" + ctr] = cnt
    except Exception as e:  # noqa: BLE001
        return {"error": f"{type(e).__name__}: {e}"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return out


def cpu_side_barrier(tag, rank, world):
    """All ranks meet on the process group's rendezvous STORE (a TCP key-value store, CPU only): the ranks that have nothing to
    do while rank 0 runs the one-process measurements wait here -- an RCCL barrier would keep a spinning kernel on every device
    rank 0 is about to time, and a second (gloo) process group prints its connection banner on stdout, next to the ONE JSON
    line.  Falls back to the default group's barrier when the store is not reachable."""
    import datetime
    try:
        store = dist.distributed_c10d._get_default_store()
        store.set(f"halo2_mi355x/{tag}/{rank}", "1")
        store.wait([f"halo2_mi355x/{tag}/{r}" for r in range(world)], datetime.timedelta(minutes=30))
    except Exception:  # noqa: BLE001
        dist.barrier()


COLLECTIVE = {"on": False, "backend": None, "ranks_seen": 1, "error": None, "init_s": None}


def _free_port():
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        return s_.getsockname()[1]


def init_collective(backend, rank, world, device, comm_device, at_one_rank=True):
    """The process group of the run.  N > 1: as the driver launched it (RANK / WORLD_SIZE / MASTER_* from the environment).
    N = 1: a ONE-RANK RCCL communicator on this GPU, so that the timed step takes the same route as at N > 1 -- partial to
    a device tensor, dist.all_gather over RCCL, fold -- and `ranks_in_collective` is what an RCCL all-reduce returned, not a
    constant.  A box on which the one-rank communicator cannot be built loses the collective, never the headline: the
    error goes into the record (`collective.error`) and the step runs without the exchange."""
    t0 = time.perf_counter()
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    try:
        if world > 1:
            if backend == "nccl":
                dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
            else:
                dist.init_process_group(backend, rank=rank, world_size=world)
        elif at_one_rank:
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
            if backend == "nccl":
                dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
            else:
                dist.init_process_group(backend, rank=0, world_size=1)
        else:
            return
        one = torch.ones(1, dtype=torch.int64, device=comm_device)
        dist.all_reduce(one)                                    # how many ranks the collective really saw
        if comm_device.type == "cuda":
            torch.cuda.synchronize()
        COLLECTIVE.update(on=True, backend=("rccl" if backend == "nccl" else backend), ranks_seen=int(one.item()))
    except Exception as e:  # noqa: BLE001
        if world > 1:
            raise
        COLLECTIVE.update(on=False, error=f"{type(e).__name__}: {e}"[:300])
        try:
            if dist.is_initialized():
                dist.destroy_process_group()
        except Exception:  # noqa: BLE001
            pass
    COLLECTIVE["init_s"] = time.perf_counter() - t0


class Legs:
    """Wall-clock budget of the run.  Every leg of the script (a side measurement beside the headline) is entered through
    ``start(name, estimate_s)``: the time since the process started -- max over the ranks, so that every rank takes the same
    decision before a leg that holds a collective -- plus the leg's estimate must fit the budget, or the leg is DROPPED
    (recorded in `dropped`).  The headline (inputs, registration, warm-up, timed steps) is not a leg and is never dropped.
    Estimates: measured on one MI355X (profiles/r06_*), with the registration-bound legs scaled by the rank count where all
    ranks register at once.  The first run on eight physical devices shares the driver's limit with code whose cost there
    has never been measured; this is what bounds it."""

    def __init__(self, budget_s, world, comm_device):
        self.budget, self.world, self.comm_device = float(budget_s), world, comm_device
        self.wall, self.dropped, self._open = {}, [], None

    def elapsed(self, local=False):
        now = time.perf_counter() - PROCESS_T0
        return now if local or self.world == 1 else max_over_ranks(now, self.world, self.comm_device)

    def start(self, name, estimate_s, local=False):
        """``local``: a leg only this rank runs (no collective inside): decided on this rank's own clock."""
        now = self.elapsed(local)
        if self.budget > 0 and now + estimate_s > self.budget:
            self.dropped.append({"leg": name, "at_s": round(now, 2), "estimate_s": estimate_s})
            return False
        self._open = (name, time.perf_counter())
        return True

    def stop(self):
        if self._open:
            name, t0 = self._open
            self.wall[name] = self.wall.get(name, 0.0) + (time.perf_counter() - t0)
            self._open = None

    def record(self):
        return {"budget_s": self.budget, "legs_wall_s": {k: round(v, 3) for k, v in self.wall.items()}, "dropped": self.dropped,
                "total_s": round(time.perf_counter() - PROCESS_T0, 3),
                "note": "rank 0's wall per leg; a leg starts only if (time since process start, max over ranks) + its estimate fits the budget"}


def fold_known_answers(expected_local, world, comm_device):
    """The global expected point from every rank's [sum s_i t_i]G over its own index range."""
    if world == 1 and not COLLECTIVE["on"]:
        return expected_local
    from halo2_experiments_amd.sharding import g1_sum
    mine = torch.from_numpy(expected_local.view(np.int64).copy()).to(comm_device)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine)
    return g1_sum(torch.stack(gathered).cpu().numpy().view(np.uint64))


def max_over_ranks(x, world, comm_device):
    if world == 1 and not COLLECTIVE["on"]:
        return x
    t = torch.tensor([x], dtype=torch.float64, device=comm_device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def strong_scaling_msm(log_global, seed, rank, world, device, comm_device, reps=3):
    """BASELINE configs[4] as a STRONG-scaling shape: one global 2^log_global-point MSM split by index range over the
    ranks (every rank generates, registers and keeps its own slice; per step one all-gather of 96 B partials + fold).
    Times are max-over-ranks per MSM behind a barrier; the folded result is checked against the known answer."""
    import halo2_experiments_amd as h
    from halo2_experiments_amd.sharding import shard_range, sharded_multiexp
    n_global = 1 << log_global
    lo, hi = shard_range(n_global, rank, world)
    s, b, t = bench_inputs(hi - lo, lo, seed, device)
    t0 = time.perf_counter()
    hnd = h.register_bases(b)
    torch.cuda.synchronize()
    reg_ms = (time.perf_counter() - t0) * 1e3
    del b
    expected = fold_known_answers(known_answer(s, t, device), world, comm_device)
    del t
    res = sharded_multiexp(s, hnd)                              # warm-up: workspaces
    times = []
    for _ in range(reps):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t1 = time.perf_counter()
        res = sharded_multiexp(s, hnd)
        times.append(max_over_ranks(time.perf_counter() - t1, world, comm_device))
    st = h.msm_stats()
    h.release_bases(hnd)
    del s
    torch.cuda.empty_cache()
    dt = float(np.median(times))
    acc = st["accumulate_kernel_ms"]
    return {"global_log_points": log_global, "n_gpus": world, "points_per_rank": hi - lo, "ms_per_msm": dt * 1e3,
            "points_per_s": n_global / dt, "known_answer_ok": bool(np.array_equal(res, expected)),
            "window_bits": st["window_bits"], "windows": st["windows"], "register_ms_rank0": reg_ms,
            "rank0_kernel_ms": acc, "rank0_sort_ms": st["sort_ms"],
            "rank0_roofline": {"bound": "hbm", "achieved": MSM_BYTES_PER_POINT * (hi - lo) / (acc * 1e-3) / 1e9 if acc else None,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": MSM_BYTES_PER_POINT * (hi - lo) / (acc * 1e-3) / 1e9 / HBM_PEAK_GBS if acc else None},
            "split": f"index ranges x{world}, all-gather of 96 B partials + host fold per MSM"}


def quotient_one_call(device, k=18, reps=5):
    """hm_quotient_by_cosets_bn256_fr_dev on the MerkleSumTree circuit's own numerator program at k = 18, every one of its 83 table
    entries an array of its own: the 48 per-proof columns from coefficients, the circuit's 35 constant columns (fixed, sigmas, l_0 /
    l_last / l_active, X) kept on the cosets as a proving key would keep them.  Five cosets (what determines the quotient) and all eight."""
    from halo2_experiments_amd import circuits
    from halo2_experiments_amd.domain import EvaluationDomain, FR_MODULUS
    from halo2_experiments_amd.replay import _rand_fr
    cs = circuits.merkle_sum_tree()
    dom = EvaluationDomain(cs.degree(), k)
    g, lay = circuits.evaluate_h_program(cs, k, dom.extended_k, pow(7, 1 << 28, FR_MODULUS), per_coset=True, divide=False)
    n_cols = lay.num_fixed_entries + cs.num_advice + cs.num_instance
    prog = g.compile(lay.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=1)
    out = {"k": k, "columns": n_cols, "program_calculations": int(len(prog.calcs))}
    try:
        cols = [_rand_fr(dom.n, 7000 + i, device) for i in range(n_cols)]
        const_idx = sorted(set(range(cs.num_fixed)) | set(range(lay.sigma0, lay.sigma0 + len(cs.equality)))
                           | {lay.l0, lay.l_last, lay.l_active, lay.x_coset, lay.t_inv})
        out["columns_kept_on_the_cosets"] = len(const_idx)
        for label, q in (("five_cosets", dom.min_cosets()), ("all_cosets", dom.num_cosets())):
            cosets = list(range(q))
            kept = dom.coeff_to_cosets(torch.stack([cols[i] for i in const_idx]), cosets, internal=True)
            pre = [None] * n_cols
            for j, i in enumerate(const_idx):
                pre[i] = kept[j]
            per_proof = [None if pre[i] is not None else cols[i] for i in range(n_cols)]
            run = lambda: prog.quotient_by_cosets(dom, per_proof, cosets=cosets, beta=3, gamma=4, theta=5, y=6, on_cosets=pre)  # noqa: E731
            run()
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            out[label] = {"cosets": q, "ms": float(np.median(ts)) * 1e3}
            del kept, pre, per_proof
        out["note"] = ("coefficient columns in, the pieces of h out, one C call; upstream's steps for the same columns in the replay above: "
                       "coset transforms + evaluate_h + extended_to_coeff")
    finally:
        prog.destroy()
        torch.cuda.empty_cache()
    return out


def one_process_measurements(devs, device, log_global, replay_name, reps=3):
    """The one-process form (csrc/multi.hip under the C ABI): hm_set_msm_devices(devs), then (a) ONE 2^log_global MSM on a
    base set SLICED by index range over the devices -- scalars as one device array on devs[0] (the slices of the other
    devices cross xGMI inside the call, hipMemcpyPeer) and as one host array (every device uploads its slice over its own
    PCIe link) -- and (b) the k = 18 create_proof replay with the SRS replicated and every phase of commitments dealt over
    the devices as whole commitments.  Runs on the calling process alone."""
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    lib = _lib.load()
    arr = (ctypes.c_int * len(devs))(*devs)
    _lib.check(lib.hm_set_msm_devices(arr, len(devs)))
    out = {"devices": list(devs), "note": "one process drives every listed device (hm_set_msm_devices); what a Rust prover -- one process "
                                          "for the node, /root/reference/src/circuits/utils.rs:22-70 -- would bind"}
    try:
        n = 1 << log_global
        s, b, t = bench_inputs(n, 0, BENCH_SEED + 100 + log_global, device)
        t0 = time.perf_counter()
        hnd = h.register_bases(b)                               # a multi handle: n >= 2^22 points are sliced over the devices
        reg_ms = (time.perf_counter() - t0) * 1e3
        del b
        info = h.bases_info(hnd)
        expected = known_answer(s, t, device)
        del t
        res = h.best_multiexp(s, hnd)
        times = []
        for _ in range(reps):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            res = h.best_multiexp(s, hnd)
            times.append(time.perf_counter() - t1)
        dt = float(np.median(times))
        entry = {"global_log_points": log_global, "ms_per_msm": dt * 1e3, "points_per_s": n / dt,
                 "known_answer_ok": bool(np.array_equal(res, expected)), "register_ms": reg_ms,
                 "base_set": {"sliced": bool(info["sliced"]), "devices": info["devices"], "table_windows": info["table_windows"],
                              "device_bytes": info["device_bytes"]},
                 "scalars": f"one device array on device {devs[0]}: the other devices' slices cross xGMI inside the call"}
        hs = to_host(s)
        del s
        t1 = time.perf_counter()
        res_h = h.best_multiexp(hs, hnd)
        entry["from_host_array"] = {"ms_per_msm": (time.perf_counter() - t1) * 1e3, "same_result": bool(np.array_equal(res_h, res)),
                                    "note": "hm_msm_bn256_g1_h: pageable host scalars, every device uploads its own slice; PCIe-inclusive"}
        del hs
        h.release_bases(hnd)
        torch.cuda.empty_cache()
        out["msm_split"] = entry
        if replay_name:
            from halo2_experiments_amd.replay import run_replay
            rep = run_replay(replay_name, device=device, include_host_pointer_estimate=False, solo=True, devices=devs)
            rep["n_gpus"] = len(set(devs))
            rep["multi_gpu_split"] = ("one process: SRS replicated on every listed device, each phase of commitments dealt round-robin as whole "
                                      "commitments inside hm_msm_batch_bn256_g1_dev; the extended-domain steps by cosets over the devices, one "
                                      "host thread per device (k >= 14), n x 32 B per coset back to the first device; the rest on the first device")
            out["create_proof_replay"] = rep
    finally:
        _lib.check(lib.hm_set_msm_devices(None, 0))
    return out


def main_one_process(args):
    """`python bench.py --gpus N --one-process` (no torchrun): the whole benchmark in the one-process form.  A step is one
    hm_msm_bn256_g1_dev call on a base set sliced over the N devices (2^log_points points per device), scalars resident on
    device 0; value = global points x steps / wall time.  HALO2_BENCH_BACKEND=gloo (rehearsal on a one-GPU box) lists the
    visible devices round-robin."""
    if int(os.environ.get("WORLD_SIZE", "1")) != 1:
        raise SystemExit("--one-process is ONE process: launch it without torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for this path)")
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    ndev = torch.cuda.device_count()
    rehearsal = os.environ.get("HALO2_BENCH_BACKEND", "nccl") != "nccl"
    if ndev < args.gpus and not rehearsal:
        raise SystemExit(f"--one-process --gpus {args.gpus}: only {ndev} device(s) visible")
    devs = [r % ndev for r in range(args.gpus)]
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    lib = _lib.load()
    n_global = args.gpus << args.log_points
    _lib.check(lib.hm_set_msm_devices((ctypes.c_int * len(devs))(*devs), len(devs)))
    try:
        scalars, bases, t = bench_inputs(n_global, 0, BENCH_SEED, device)
        t0 = time.perf_counter()
        handle = h.register_bases(bases)
        reg_ms = (time.perf_counter() - t0) * 1e3
        del bases
        info = h.bases_info(handle)
        expected = known_answer(scalars, t, device)
        del t
        for _ in range(args.warmup):
            h.best_multiexp(scalars, handle)
        torch.cuda.synchronize()
        step_ms = []
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ts = time.perf_counter()
            result = h.best_multiexp(scalars, handle)
            step_ms.append((time.perf_counter() - ts) * 1e3)
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        st = h.msm_stats()                                       # device 0's part of the last MSM
        ok = bool(np.array_equal(result, expected))
        if not ok:
            raise SystemExit("bench.py --one-process: the MSM result does not match the known answer [sum s_i (a + i b)]G")
        h.release_bases(handle)
        del scalars
        torch.cuda.empty_cache()
    finally:
        _lib.check(lib.hm_set_msm_devices(None, 0))
    acc = st["accumulate_kernel_ms"]
    n_part = n_global // len(devs)
    achieved = MSM_BYTES_PER_POINT * n_part / (acc * 1e-3) / 1e9 if acc else None
    line = {"metric": "BN256 G1 MSM throughput", "value": n_global * args.steps / elapsed, "unit": "points/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": float(np.median(step_ms)),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"standalone BN256 G1 MSM, 2^{args.log_points} points per GPU (global 2^{args.log_points} x {args.gpus}), ONE process",
                       "points_per_gpu": 1 << args.log_points, "global_points": n_global, "devices": devs,
                       "base_set": {"sliced": bool(info["sliced"]), "table_windows": info["table_windows"], "device_bytes": info["device_bytes"],
                                    "register_ms": reg_ms},
                       "parallelism": f"one process, hm_set_msm_devices: index-range slices x{args.gpus} inside the C ABI (csrc/multi.hip), one host "
                                      "thread per device, 96 B partials folded on the host; scalars resident on device 0 (the other devices' "
                                      "slices cross xGMI inside every step)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS if achieved else None,
                         "traffic": None, "traffic_src": "none", "kernel": "msm_accumulate_kernel", "kernel_ms": acc,
                         "algorithmic_bytes": MSM_BYTES_PER_POINT * n_part, "note": "device 0's part of the last step"},
            "known_answer_ok": ok}
    if args.replay != "none" and not args.no_one_process:
        line["one_process"] = one_process_measurements(devs, device, args.log_points + 2 if args.log_points <= 24 and not args.no_2_26 else args.log_points,
                                                       args.replay.split(",")[-1])
    line["ranks_in_collective"] = 1
    emit(line, args.extras_out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log-points", type=int, default=24, help="log2 of the points per GPU")
    ap.add_argument("--log-ntt", type=int, default=24)
    ap.add_argument("--cpu-log-sample", type=int, default=0, help="log2 of the CPU baseline's MSM (default: the timed size, capped at 24)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ntt", action="store_true")
    ap.add_argument("--replay", default="merkle_sum_tree_k9,poseidon_k11,merkle_v3_k17,merkle_sum_tree_k18",
                    help="comma-separated create_proof MSM/NTT traces to replay after the timed MSM steps ('none' to skip)")
    ap.add_argument("--no-extras", action="store_true", help="skip the prover-like and host-pointer MSM side measurements")
    ap.add_argument("--no-2-26", action="store_true", help="skip the global 2^26-point strong-scaling measurement")
    ap.add_argument("--no-live-pmc", action="store_true",
                    help="do not measure the PMC counters of the hot kernels in child processes under rocprofv3 (N = 1 only); use the baked file")
    ap.add_argument("--no-shares", action="store_true", help="skip the one-rank shares of the N-rank k = 18 replay (N = 1 only)")
    ap.add_argument("--no-strong", action="store_true", help="skip the strong-scaling measurements (global 2^24 / 2^26 split over the ranks)")
    ap.add_argument("--no-one-process", action="store_true", help="N > 1: skip rank 0's one-process (hm_set_msm_devices) measurements")
    ap.add_argument("--extras-out", default=EXTRAS_FILE,
                    help="where the FULL record goes (every side measurement; relative to this script; 'none' = nowhere); stdout carries "
                         "only the compact line")
    ap.add_argument("--no-collective", action="store_true",
                    help="N = 1: do not build the one-rank RCCL communicator (the timed step then skips the all-gather)")
    ap.add_argument("--time-budget", type=float, default=300.0,
                    help="seconds from process start within which side legs may still START (0 = no limit); the headline is never dropped")
    ap.add_argument("--one-process", action="store_true",
                    help="ONE process drives --gpus devices through hm_set_msm_devices (launch WITHOUT torchrun): the whole benchmark in the "
                         "form the reference's single-process prover would use")
    args = ap.parse_args()
    guard_stdout()
    if args.one_process:
        return main_one_process(args)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback exists for this path)")
    # HALO2_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks share
    # devices round-robin and the 96-byte partials travel over gloo); the judged runs use nccl = RCCL.
    backend = os.environ.get("HALO2_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    comm_device = device if backend == "nccl" else torch.device("cpu")
    under_profiler = any("ROCPROF" in k.upper() for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    init_collective(backend, rank, world, device, comm_device, at_one_rank=not args.no_collective)
    ranks_seen = COLLECTIVE["ranks_seen"]
    legs = Legs(args.time_budget, world, comm_device)

    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.sharding import shard_range, sharded_multiexp
    baked, stale = baked_counters()

    n_local = 1 << args.log_points
    n_global = n_local * world
    lo, hi = shard_range(n_global, rank, world)
    assert hi - lo == n_local

    # ---- synthetic inputs, resident in HBM before the timed region -----------------
    gen = np.array(fq_mont_words(1) + fq_mont_words(2), dtype=np.uint64)        # G = (1, 2)
    scalars, bases, t_local = bench_inputs(n_local, lo, BENCH_SEED, device)     # this rank's index range of the global arrays
    torch.cuda.synchronize()
    t_reg0 = time.perf_counter()
    handle = h.register_bases(bases)                                            # device-resident SRS slice, the library's default layout
    torch.cuda.synchronize()
    headline_register_ms = (time.perf_counter() - t_reg0) * 1e3
    headline_bases_info = h.bases_info(handle)
    del bases
    expected_local = known_answer(scalars, t_local, device)                     # [sum_i s_i (a + i b)]G over this rank's range
    del t_local
    torch.cuda.synchronize()

    def step():                      # N = 1 with the one-rank communicator: the same route as N > 1 (all-gather over RCCL, fold)
        return sharded_multiexp(scalars, handle, force_collective=COLLECTIVE["on"])

    for _ in range(args.warmup):
        step()
    # known answer, once, outside the timed loop: this rank's partial, and the folded global result
    local_result = h.best_multiexp(scalars, handle)
    answer_ok = bool(np.array_equal(local_result, expected_local))
    expected_global = fold_known_answers(expected_local, world, comm_device)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    acc_ms, sort_ms, tot_ms, step_ms = [], [], [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        result = step()
        step_ms.append((time.perf_counter() - ts) * 1e3)          # a step returns its result to the host: it is synchronous
        st = h.msm_stats()
        acc_ms.append(st["accumulate_kernel_ms"])
        sort_ms.append(st["sort_ms"])
        tot_ms.append(st["total_ms"])
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=comm_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    st = h.msm_stats()
    # which layout the default registration chose shows in the plan: one shared bucket set has ceil(255 / c) < 15 windows at c > 17
    headline_mode = ("fixed-base table (registration default from 2^17 points)"
                     if st["window_bits"] > 17 else "plain (one copy of the points, one bucket set per window)")
    answer_ok = answer_ok and bool(np.array_equal(result, expected_global))     # the timed steps' own result
    if not answer_ok:
        raise SystemExit("bench.py: the MSM result does not match the known answer [sum s_i (a + i b)]G")

    # ---- the create_proof replays (configs[1..3] and the reference's own k = 9), FIRST among the side legs: with the headline's base
    # set and scalars still resident (12.5 GiB of 288: nothing is freed here).  Why first: freeing GiBs of device memory -- what the
    # legs below do between their measurements -- leaves the copy engines at two thirds of their speed for seconds afterwards (the
    # driver's work on the freed memory; tools/ntt_ext_probe.py <..> 64, profiles/r06_after_free.txt), and round 5's and round 6's first
    # drop-in totals (286 / 314-325 ms at k = 18) were taken in exactly that state.  A prover keeps its SRS registered and frees nothing
    # of that size between its calls; measured in that state the same calls total ~235 ms.
    replay = None
    if args.replay != "none":
        from halo2_experiments_amd.replay import run_replay
        # every rank takes part.  N > 1 runs the extended-domain steps by cosets; BOTH coset routes are measured at every N so that a
        # scaling record compares like with like: all E cosets = the same polynomial as N = 1's whole-array device_resident_s
        # (upstream's own steps), and the j - 1 cosets that determine the quotient of a satisfied circuit (a different
        # computation on the replay's synthetic columns: never the headline)
        # the largest shape LAST in the list but FIRST in the budget: when time runs short the small shapes are what goes
        names = args.replay.split(",")
        replay_est = {"merkle_sum_tree_k18": 9.0, "merkle_v3_k17": 5.0}
        by_name = {}
        for name in sorted(names, key=lambda nm: -replay_est.get(nm, 3.0)):
            est = replay_est.get(name, 3.0)
            if world > 1:
                if not legs.start(f"replay_{name}", est):
                    continue
                rep = run_replay(name, device=device, min_cosets=False)
                legs.stop()
                if legs.start(f"replay_{name}_min_cosets", est):
                    fewer = run_replay(name, device=device, include_host_pointer_estimate=False, min_cosets=True)
                    legs.stop()
                    rep["extended_domain_routes_ms"] = {
                        "by_all_cosets": {**{k2: v * 1e3 for k2, v in rep["device_resident_s"].items()}, "extended_domain": rep["extended_domain"]},
                        "by_the_cosets_that_determine_h": {**{k2: v * 1e3 for k2, v in fewer["device_resident_s"].items()},
                                                           "extended_domain": fewer["extended_domain"]},
                        "note": "device_resident_s = by_all_cosets: the same h as the N = 1 whole-array route"}
                by_name[name] = rep
            elif legs.start(f"replay_{name}", est + (6.0 if name.endswith("k18") else 1.0)):     # + the host-pointer (drop-in) calls of the trace
                by_name[name] = run_replay(name, device=device)
                legs.stop()
        replay = [by_name[nm] for nm in names if nm in by_name]
        # N = 1: what ONE GPU can measure of the N-GPU replay -- the first and the last rank's share of the 2-, 4- and 8-rank deal of
        # the largest shape, each alone, nothing exchanged.  DESIGN.md section 6 builds its
        # predicted curve on these.
        if world == 1 and not args.no_extras and not args.no_shares and legs.start("k18_rank_shares_and_coset_routes", 12.0):
            big = [nm for nm in args.replay.split(",") if nm == "merkle_sum_tree_k18" and nm in by_name]
            for nm in big:
                shares = []
                for w in (2, 4, 8):
                    for rk in (0, w - 1):           # rank 0 also runs the rank-0-only steps; the last rank gets the first coset
                        try:
                            r = run_replay(nm, device=device, include_host_pointer_estimate=False, share_of=(rk, w))
                            shares.append({"world": w, "rank": rk, "ms": {k2: v * 1e3 for k2, v in r["device_resident_s"].items()},
                                           "extended_domain": r["extended_domain"]})
                        except Exception as exc:  # noqa: BLE001 -- a side measurement never costs the line
                            shares.append({"world": w, "rank": rk, "error": f"{type(exc).__name__}: {exc}"})
                rep18 = next(rep for rep in replay if rep["k"] == 18)
                # the same trace with the extended-domain steps taken coset by coset on this one GPU: all E cosets (the multi-GPU
                # route's work, 5 % more than the whole array) and only the j - 1 cosets that determine the quotient
                routes = {"whole_array": {k2: v * 1e3 for k2, v in rep18["device_resident_s"].items()}}
                for label, kw in (("by_all_cosets", {"min_cosets": False}), ("by_the_cosets_that_determine_h", {"min_cosets": True})):
                    try:
                        r = run_replay(nm, device=device, include_host_pointer_estimate=False, by_cosets=True, **kw)
                        routes[label] = {k2: v * 1e3 for k2, v in r["device_resident_s"].items()}
                        routes[label]["extended_domain"] = r["extended_domain"]
                    except Exception as exc:  # noqa: BLE001
                        routes[label] = {"error": f"{type(exc).__name__}: {exc}"}
                routes["note"] = ("the quotient of a satisfied circuit has fewer than n (j - 1) coefficients: j - 1 of the E cosets determine it "
                                  "(same h word for word: tests/test_mini_prover_gpu.py); create_proof_replay.device_resident_s stays the "
                                  "whole-array route, upstream's own steps")
                rep18["extended_domain_routes_ms"] = routes
                try:
                    rep18["quotient_in_one_call_ms"] = quotient_one_call(device)
                except Exception as exc:  # noqa: BLE001 -- a side measurement never costs the line
                    rep18["quotient_in_one_call_ms"] = {"error": f"{type(exc).__name__}: {exc}"}
                rep18["rank_shares_measured_alone"] = {
                    "shares": shares, "note": "the first and the last rank's share of the N-rank replay, each run alone on this GPU (its "
                                              "commitments of every phase, its cosets of the extended domain, for rank 0 the steps only rank 0 "
                                              "runs); the step takes the longer of them plus the exchanges -- 96 B per commitment, n x 32 B "
                                              "per coset -- which are not in these times"}
            legs.stop()
        # the same proof through exactly the calls of rust/halo2_proofs-patch/src/mi355x_dev.rs (halo2-experiments_amd/rust_glue.py):
        # what a Rust prover that keeps its polynomials in DevicePolys gets -- upload once, resident steps, download -- next to the
        # per-call drop-in total
        if world == 1 and not args.no_extras and "merkle_sum_tree_k18" in by_name and legs.start("k18_rust_device_glue", 8.0):
            try:
                from halo2_experiments_amd.rust_glue import run_proof
                by_name["merkle_sum_tree_k18"]["rust_device_glue"] = run_proof("merkle_sum_tree_k18", device=device, reps=3, check=True)
            except Exception as exc:  # noqa: BLE001 -- a side measurement never costs the line
                by_name["merkle_sum_tree_k18"]["rust_device_glue"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
            legs.stop()
        if rank == 0 and world == 1 and not args.no_cpu_baseline and legs.start("replay_cpu_baselines", 10.0):
            for rep in replay:
                rep["cpu_baseline"] = replay_cpu_baseline(rep, device)
                acct = rust_threshold_account(rep)
                if acct is not None:
                    rep["rust_shim_routing"] = acct
            legs.stop()

    # ---- NTT (single GPU by design: "replicas only") -------------------------------------------
    ntt = None
    if not args.no_ntt and rank == 0 and legs.start("ntt", 6.0, local=True):
        from halo2_experiments_amd.domain import FR_MODULUS, FR_ROOT_OF_UNITY, fr_words
        k = args.log_ntt
        omega = fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - k), FR_MODULUS))
        a = rand_fr(1 << k, 99, device)
        for _ in range(2):
            h.best_fft(a, omega, k)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = max(3, args.steps)
        e0.record()                      # the NTT launches on torch's current stream (passed through the ABI)
        for _ in range(reps):
            h.best_fft(a, omega, k)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        gbs = (NTT_BYTES_PER_ELEM << k) / (ms * 1e-3) / 1e9
        ntt = {"log_n": k, "ms": ms, "elements_per_s": (1 << k) / (ms * 1e-3),
               "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                            "traffic": (pmc_traffic(baked, "hm::ntt_pass_kernel<11>", True) if k == 24 else None),
                            "traffic_note": f"NOT measured in this run: per pass launch, from {BAKED_COUNTERS_FILE}; stale = {bool(stale['ntt'])}",
                            "note": "traffic is per pass launch; 3 digit passes => 3x the algorithmic 64 B/element per transform; "
                                    "VALU-issue bound (~3.9e3 32-bit ops per element)",
                            "valu_issue": valu_issue(baked, stale, "ntt", (1 << k) * (3 if k > 21 else 2 if k > 11 else 1), ms)}}
        del a
        if world == 1 and not args.no_cpu_baseline:
            ntt["cpu_baseline"] = ntt_cpu_baseline(k, device)
        legs.stop()

    # ---- side measurements: prover-like scalars; the PCIe-inclusive drop-in call ----
    extras = None
    if not args.no_extras and world == 1 and legs.start("msm_side_measurements", 8.0):
        extras = {}
        u = torch.rand(n_local, device=device, generator=torch.Generator(device=device).manual_seed(7))
        pl = scalars.clone()
        pl[u < 0.95, 1:] = 0                                   # 5 % stay uniform
        pl[u < 0.95, 0] &= 0xFFFF                              # 5 % below 2^16 ...
        pl[u < 0.90] = 0                                       # ... 90 % zero
        # raw small words are not Montgomery form: scale by R so the canonical values are the small ones
        from halo2_experiments_amd.domain import FR_MODULUS, fr_words
        from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
        _lib.check(_lib.load().hm_fr_scale_dev(ctypes.c_void_p(pl.data_ptr()), n_local, _ptr(fr_words((1 << 256) % FR_MODULUS)),
                                               ctypes.c_void_p(_stream_ptr(pl))))
        uni = u >= 0.95
        pl[uni] = scalars[uni]
        del uni
        h.best_multiexp(pl, handle)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            h.best_multiexp(pl, handle)
        dt = (time.perf_counter() - t1) / 3
        extras["msm_prover_like"] = {"points_per_s": n_local / dt, "ms": dt * 1e3, "pairs": h.msm_stats()["pairs"],
                                     "scalars": "90 % zero, 5 % < 2^16, 5 % uniform"}
        del pl, u
        # the OTHER base-set layout: the timed steps ran on what hm_register_bases builds by default (from 2^17 points the
        # fixed-base table: 2^(offset of window j) * P_i for every window, W x the memory, one shared bucket set); here the
        # plain layout (one copy of the points, W bucket sets) -- or, below the default's threshold, the table
        _, bases2, _ = bench_inputs(n_local, lo, BENCH_SEED, device)
        other_is_plain = headline_mode.startswith("fixed-base")
        lib = _lib.load()
        if other_is_plain:
            _lib.check(lib.hm_set_fixed_base_threshold(0))
        t1 = time.perf_counter()
        hp = h.register_bases(bases2, precompute=not other_is_plain)
        torch.cuda.synchronize()
        t_reg = time.perf_counter() - t1
        if other_is_plain:
            _lib.check(lib.hm_set_fixed_base_threshold(17))
        del bases2
        ref_out = h.best_multiexp(scalars, handle)
        got_pc = h.best_multiexp(scalars, hp)
        t1 = time.perf_counter()
        for _ in range(3):
            h.best_multiexp(scalars, hp)
        dt = (time.perf_counter() - t1) / 3
        stp = h.msm_stats()
        extras["msm_plain_bases" if other_is_plain else "msm_precomputed_bases"] = {
            "points_per_s": n_local / dt, "ms": dt * 1e3, "window_bits": stp["window_bits"], "windows": stp["windows"],
            "accumulate_kernel_ms": stp["accumulate_kernel_ms"], "sort_ms": stp["sort_ms"], "register_ms": t_reg * 1e3,
            "base_set_bytes": int(h.bases_info(hp)["device_bytes"]),
            "same_result_as_headline": bool(np.array_equal(ref_out, got_pc))}
        h.release_bases(hp)
        # independent MSMs issued asynchronously, three in flight on three streams (hm_msm_submit_dev):
        # what a prover committing to several columns gets; NOT `value`, whose steps are strictly serial
        streams = [torch.cuda.Stream(device=device) for _ in range(3)]
        for st_ in streams:
            st_.wait_stream(torch.cuda.current_stream(device))
        warm = []
        for i in range(3):                                     # warm-up: the three asynchronous workspaces get allocated here
            with torch.cuda.stream(streams[i]):
                warm.append(h.best_multiexp_submit(scalars, handle))
        for t_ in warm:
            h.best_multiexp_wait(t_)
        reps, pending = 9, []
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(reps):
            if len(pending) == 3:
                h.best_multiexp_wait(pending.pop(0))
            with torch.cuda.stream(streams[i % 3]):
                pending.append(h.best_multiexp_submit(scalars, handle))
        for t_ in pending:
            last = h.best_multiexp_wait(t_)
        dt = (time.perf_counter() - t1) / reps
        extras["msm_pipelined_3_in_flight"] = {"points_per_s": n_local / dt, "ms_per_msm": dt * 1e3,
                                               "same_result": bool(np.array_equal(last, ref_out))}
        hs = to_host(scalars)
        t1 = time.perf_counter()
        h.best_multiexp(hs, handle)
        extras["msm_host_pointer"] = {"ms": (time.perf_counter() - t1) * 1e3,
                                      "note": "hm_msm_bn256_g1_h: scalars cross PCIe in the call (pageable host memory); never `value`"}
        del hs
        legs.stop()

    # ---- strong scaling (BASELINE configs[4]: "2^26 MSM ... 1/2/4/8 GPUs"): one GLOBAL MSM split over the ranks ----
    strong = None
    if not args.no_strong:
        if handle is not None:
            h.release_bases(handle)
            handle = None
        del scalars
        scalars = None
        torch.cuda.empty_cache()
        strong = []
        if world == 1:          # the headline IS the global 2^log_points MSM on one GPU
            acc1 = float(np.median(acc_ms))
            strong.append({"global_log_points": args.log_points, "n_gpus": 1, "points_per_rank": n_local,
                           "ms_per_msm": elapsed / args.steps * 1e3, "points_per_s": n_global * args.steps / elapsed,
                           "known_answer_ok": answer_ok, "window_bits": st["window_bits"], "windows": st["windows"],
                           "rank0_kernel_ms": acc1, "split": "none (the timed headline steps)"})
        elif legs.start(f"strong_2_{args.log_points}", 8.0):
            strong.append(strong_scaling_msm(args.log_points, BENCH_SEED + 24, rank, world, device, comm_device))
            legs.stop()
        # (--no-extras: the profiled command -- one MSM size in the trace); 2^26 / N points per rank: generation + registration dominate
        if args.log_points == 24 and not args.no_2_26 and not args.no_extras and legs.start("strong_2_26", 6.0 + 16.0 / world):
            strong.append(strong_scaling_msm(26, BENCH_SEED + 26, rank, world, device, comm_device))
            legs.stop()
        for e in strong:
            if not e["known_answer_ok"]:
                raise SystemExit(f"bench.py: the global 2^{e['global_log_points']} MSM does not match its known answer")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # the contract's cpu_baseline: not droppable, but SIZED to what is left (the whole 2^24 workload is ~8 s on 16 cores;
        # each halving of the sample halves it)
        log_sample = args.cpu_log_sample or min(args.log_points, 24)
        while log_sample > 16 and legs.budget > 0 and legs.elapsed() + 10.0 * (1 << log_sample) / (1 << 24) > legs.budget:
            log_sample -= 1
        legs.start("cpu_baseline", 0.0)
        cpu = cpu_baseline(log_sample, device, full=log_sample == args.log_points)
        legs.stop()

    # ---- N > 1: the ONE-PROCESS form, by rank 0 alone -------------------------------------------------------------
    one_proc = None
    if world > 1 and not args.no_one_process and legs.start("one_process", 45.0 + 2.5 * world):
        if handle is not None:
            h.release_bases(handle)
            handle = None
        scalars = None
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        if rank != 0:
            _lib.check(_lib.load().hm_shutdown())               # workspaces, tables, staging: this rank's device is rank 0's to use now
        cpu_side_barrier("released", rank, world)                # everybody has let go of its device memory ...
        if rank == 0:
            devs = [r if backend == "nccl" else r % max(ndev, 1) for r in range(world)]
            try:            # never at the expense of the line's headline: this form has not met N physical devices before the driver's run
                one_proc = one_process_measurements(devs, device, 26 if args.log_points == 24 and not args.no_2_26 else args.log_points,
                                                    None if args.replay == "none" else args.replay.split(",")[-1])
            except Exception as e:  # noqa: BLE001
                one_proc = {"devices": devs, "error": f"{type(e).__name__}: {e}"}
                try:
                    _lib.load().hm_set_msm_devices(None, 0)
                except Exception:  # noqa: BLE001
                    pass
        cpu_side_barrier("one_process_done", rank, world)        # ... and waits here, on the CPU, until rank 0 is done
        legs.stop()

    # ---- PMC counters of the two hot kernels, measured in THIS run (child processes under rocprofv3) --------------------
    live = None
    if rank == 0 and world == 1 and not args.no_live_pmc and not under_profiler and legs.start("live_pmc", 35.0, local=True):
        if handle is not None:
            h.release_bases(handle)
            handle = None
        scalars = None
        torch.cuda.empty_cache()
        live = live_pmc(args.log_points, args.log_ntt, ntt is not None)
        legs.stop()

    if rank == 0:
        acc = float(np.median(acc_ms))
        achieved = MSM_BYTES_PER_POINT * n_local / (acc * 1e-3) / 1e9
        k3_live = (live or {}).get("hm::msm_accumulate_kernel")
        ntt_live = next((v for k_, v in (live or {}).items() if k_.startswith("hm::ntt_pass_kernel<11")), None)
        pmc_how = ("MEASURED IN THIS RUN: child processes of this script (two steps of the same workload) under rocprofv3 --pmc, FETCH_SIZE and "
                   "WRITE_SIZE in separate passes, averaged per launch")
        if k3_live and "FETCH_SIZE" in k3_live and "WRITE_SIZE" in k3_live:
            k3_traffic = (k3_live["FETCH_SIZE"] + k3_live["WRITE_SIZE"]) * 1024.0      # raw: 64-byte gathers are counted 1:1 (r02_gather64_calibration.json)
            k3_traffic_src = "rocprofv3 --pmc in this run (FETCH_SIZE + WRITE_SIZE, separate passes)"
            k3_traffic_note = pmc_how + "; raw FETCH + WRITE (64-byte gathers count 1:1: profiles/r02_gather64_calibration.json); every base (or " \
                                        "its table multiple) is gathered once per window, inherent to bucketed Pippenger"
        else:
            k3_traffic = pmc_traffic(baked, "hm::msm_accumulate_kernel", False) if args.log_points == 24 else None
            k3_traffic_src = f"{BAKED_COUNTERS_FILE} (stale={bool(stale['k3'])})" if k3_traffic is not None else "none"
            k3_traffic_note = (f"NOT measured in this run ({(live or {}).get('error', 'live PMC pass not run')}): rocprofv3 --pmc figure via "
                               f"{BAKED_COUNTERS_FILE} (stale = {bool(stale['k3'])}: whether the kernel sources changed since)")
        if k3_live and "SQ_INSTS_VALU" in k3_live:
            per_unit = k3_live["SQ_INSTS_VALU"] / st["pairs"]
            a_ = per_unit * st["pairs"] / (acc * 1e-3)
            k3_valu = {"achieved": a_, "peak": VALU_PEAK_WAVE_INST_PER_S, "unit": "wave-instr/s", "frac": a_ / VALU_PEAK_WAVE_INST_PER_S,
                       "wave_instr_per_64_units": per_unit * 64, "sq_insts_valu_per_launch": k3_live["SQ_INSTS_VALU"],
                       "grbm_gui_active_per_launch": k3_live.get("GRBM_GUI_ACTIVE"), "stale": False,
                       "note": "SQ_INSTS_VALU " + pmc_how + "; peak = 1024 SIMDs x 2.4 GHz / 4"}
        else:
            k3_valu = valu_issue(baked, stale, "k3", st["pairs"], acc)
        if ntt is not None and ntt_live and "FETCH_SIZE" in ntt_live and "WRITE_SIZE" in ntt_live:
            ntt["roofline"]["traffic"] = (2.0 * ntt_live["FETCH_SIZE"] + ntt_live["WRITE_SIZE"]) * 1024.0   # gfx950: FETCH_SIZE is half of a coalesced stream
            ntt["roofline"]["traffic_note"] = pmc_how + "; per pass launch, FETCH_SIZE doubled (the gfx950 correction for wide coalesced reads)"
            if "SQ_INSTS_VALU" in ntt_live:
                passes = 3 if args.log_ntt > 21 else 2 if args.log_ntt > 11 else 1
                a_ = ntt_live["SQ_INSTS_VALU"] * passes / (ntt["ms"] * 1e-3)
                ntt["roofline"]["valu_issue"] = {"achieved": a_, "peak": VALU_PEAK_WAVE_INST_PER_S, "unit": "wave-instr/s",
                                                 "frac": a_ / VALU_PEAK_WAVE_INST_PER_S, "sq_insts_valu_per_pass_launch": ntt_live["SQ_INSTS_VALU"],
                                                 "stale": False, "note": "SQ_INSTS_VALU " + pmc_how}
        line = {
            "metric": "BN256 G1 MSM throughput",
            "value": n_global * args.steps / elapsed,
            "unit": "points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_median": float(np.median(step_ms)),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {"workload": f"standalone BN256 G1 MSM, 2^{args.log_points} points per GPU "
                                   f"(global 2^{args.log_points} x {world}; BASELINE configs[4] microbench, north-star headline size)",
                       "points_per_gpu": n_local, "global_points": n_global,
                       "scalars": "uniform in [0, r), xoshiro256** per element, seed 0x48324d4933353558 (SURVEY.md §8d)",
                       "bases": "P_i = [a + i b]G, device-resident; result checked against [sum s_i (a + i b)]G outside the timed loop",
                       "base_set": headline_mode, "base_set_register_ms": headline_register_ms,
                       "base_set_bytes": int(headline_bases_info["device_bytes"]),
                       "window_bits": st["window_bits"], "windows": st["windows"], "parallelism": f"index-range shards x{world}, "
                       "all-gather of 96 B partials (RCCL) + host fold"},
            # bound: what limits the kernel (integer VALU issue, `valu_issue`); achieved / peak / frac stay the contract's HBM
            # figures (algorithmic bytes / kernel time against the 8 TB/s peak)
            "roofline": {"bound": "valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": k3_traffic, "traffic_src": k3_traffic_src, "traffic_note": k3_traffic_note,
                         "kernel": "msm_accumulate_kernel", "kernel_ms": acc, "algorithmic_bytes": MSM_BYTES_PER_POINT * n_local,
                         "note": "integer-VALU bound: n x windows mixed additions; traffic = one 64-B table gather per point and window",
                         "valu_issue": k3_valu},
            "msm_phase_ms": {"sort": float(np.median(sort_ms)), "accumulate_kernel": acc, "device_total": float(np.median(tot_ms)),
                         "pairs": int(st["pairs"]), "tasks": int(st["tasks"])},
            "known_answer_ok": answer_ok,
        }
        line["ranks_in_collective"] = ranks_seen
        line["collective"] = {
            "on": COLLECTIVE["on"], "backend": COLLECTIVE["backend"], "ranks_seen_by_all_reduce": ranks_seen, "init_s": COLLECTIVE["init_s"],
            "error": COLLECTIVE["error"],
            "summary": (f"{COLLECTIVE['backend']} all-gather of 96 B partials in every timed step, {ranks_seen} rank(s) seen by all-reduce"
                        if COLLECTIVE["on"] else f"none in the timed step ({COLLECTIVE['error'] or 'switched off: --no-collective'})")}
        line["time_budget"] = legs.record()
        if strong is not None:
            line["strong_scaling"] = strong
        if one_proc is not None:
            line["one_process"] = one_proc
        if ntt is not None:
            line["ntt"] = ntt
        if extras is not None:
            line.update(extras)
        if replay is not None:
            line["create_proof_replay"] = replay
        if cpu is not None:
            line["cpu_baseline"] = cpu
        emit(line, args.extras_out)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
