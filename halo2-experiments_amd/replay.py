"""Replay of ``create_proof``'s MSM / NTT call trace (BASELINE.json configs 2-4).

The Rust prover cannot be built in this image, so end-to-end numbers come from replaying the
sequence of ``best_multiexp`` / ``EvaluationDomain`` calls that ``create_proof`` issues for a
circuit of a given shape (upstream ``halo2_proofs/src/plonk/prover.rs`` at the tag
pinned by /root/reference/Cargo.toml:10), on synthetic polynomials: dense columns uniform in
[0, r); sparse columns zero except ``used_rows`` small values and the last 6 (blinding) rows.
The column counts, lookups, equality columns and degree come from the constraint systems transcribed in circuits.py
(the reference's chip files); the CALL counts per column follow upstream's prover loops as recalled --
a Rust build of the shim (rust/) would replace them with counted calls (hm_get_stats).

What is replayed per proof (A advice, L lookups, P equality columns, max degree d, n = 2^k,
extended domain 2^ek with ek = k + ceil(log2(d-1))):
    commit_lagrange (MSM n)   A advice (sparse) + 2L permuted (sparse) + Zp + L grand products + 1 random
    commit          (MSM n)   (d-1) quotient pieces + 2 SHPLONK polynomials
    lagrange_to_coeff (iNTT n, scale fused)          A + 1 instance + 3L + Zp
    coeff_to_extended (coset NTT 2^ek, shift fused)  A + 1 + 3L + Zp
    extended_to_coeff (iNTT 2^ek)                    1
    evaluate_h        (GraphEvaluator over 2^ek rows)  the circuit's own program: gates, permutation argument, lookup arguments,
                                                        vanishing-polynomial division (circuits.py)
    eval_polynomial   (Horner, n coefficients)         2A + 3 Zp + 5L + (d-1) queries (estimate)
    grand products    (batch_invert + running product over n rows)   Zp + L  (permutation / lookup z columns)
    lookup permute    (permute_expression_pair: two 256-bit sorts + arrangement over n - 7 rows)   L
    multiopen         (linear combination of the committed polynomials per rotation set, kate_division per opening
                       point, final combination + division)          4 sets, 5 points (estimate)
Everything else in ``create_proof`` (witness synthesis, the transcript) stays on the CPU in the reference and is NOT part
of this number.

Like the reference's harness (prove, THEN verify: /root/reference/src/circuits/merkle_sum_tree.rs:345-358), a
replay checks what it computed: the SRS is a real one (g = [s^i]G, g_lagrange = [L_i(s)]G with a known s), and
every commitment of the timed proofs -- each distinct (column, base set) pair, through the eight-in-flight path
that produced it -- must equal [f(s)]G, where f(s) comes from an independent route through the library
(inverse NTT of the column, Horner evaluation on the device, one fixed-base multiplication).  A mismatch raises.
"""
from __future__ import annotations

import math
import time
from dataclasses import dataclass
from typing import Optional

import numpy as np

from .arithmetic import (G1_GENERATOR, to_host, batch_invert, best_multiexp, best_multiexp_submit, best_multiexp_wait, eval_polynomial,
                         g1_fixed_base_mul, grand_product_batch, kate_division, kate_division_batch, linear_combination, permute_expression_pairs, register_bases,
                         release_bases)
from .domain import EvaluationDomain, FR_MODULUS, fr_words
from .kzg import ParamsKZG
from .sharding import (_NO_GROUP, coset_owners, gather_coset_partials, job_owner, job_parallel_multiexp_batch, shard_range, sharded_multiexp,
                       sharded_multiexp_batch)

# the replay's SRS trapdoor (the reference draws it from OsRng, utils.rs:28): known here, so that every
# commitment of the replay can be checked against the KZG identity commit(f) == [f(s)]G
REPLAY_S = 0x48324D4933353558_0123456789ABCDEF_FEDCBA9876543210 % FR_MODULUS


@dataclass(frozen=True)
class CircuitShape:
    name: str
    k: int
    advice: int
    lookups: int
    equality_columns: int
    max_degree: int
    used_rows: int
    source: str


def _shapes():
    """Column counts, lookups, equality columns and the maximum degree come from the constraint systems transcribed in
    circuits.py (the reference's chip files, cited there; Pow5Chip / LtChip as recalled) -- not from estimates.  `used_rows`
    (how many rows of a 2^k column synthesis fills) stays an estimate: it is a property of the floor planner's layout."""
    from .circuits import CONSTRAINT_SYSTEMS
    # merkle_sum_tree_k9: the reference's own test_full_prover (merkle_sum_tree.rs:345-358: k = 9, depth-5 path); rows per level as at k = 18
    ks = {"poseidon_k11": (11, 40), "merkle_v3_k17": (17, 840), "merkle_sum_tree_k18": (18, 1100), "merkle_sum_tree_k9": (9, 290)}
    out = {}
    for key, make in CONSTRAINT_SYSTEMS.items():
        cs = make()
        out[key] = CircuitShape(cs.name, ks[key][0], cs.num_advice, len(cs.lookups), len(cs.equality), cs.degree(), ks[key][1],
                                cs.source + "; used_rows estimated")
    return out


SHAPES = _shapes()


def _rand_fr(n, seed, device):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, device)                 # uniform over the whole of [0, r)


def _sparse_column(n, used_rows, seed, device):
    """Zero except `used_rows` leading small values (half of them the constant 1 -- the hot-bucket
    worst case of selector / flag columns -- half random 16-bit integers) and 6 random blinding
    rows, in Montgomery form like every ``Fr`` the prover holds."""
    import ctypes

    import torch

    from . import _lib
    from .arithmetic import _ptr, _stream_ptr
    col = torch.zeros((n, 4), dtype=torch.int64, device=device)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    col[:used_rows, 0] = torch.randint(1, 1 << 16, (used_rows,), dtype=torch.int64, device=device, generator=g)
    col[: used_rows // 2, 0] = 1
    # raw word v read as a Montgomery word is the field element v / R; times R gives v, stored as v*R
    _lib.check(_lib.load().hm_fr_scale_dev(ctypes.c_void_p(col.data_ptr()), n, _ptr(fr_words((1 << 256) % FR_MODULUS)),
                                           ctypes.c_void_p(_stream_ptr(col))))
    col[n - 6:] = _rand_fr(6, seed + 1, device)
    return col


def run_replay(shape_name: str, device=None, group=None, include_host_pointer_estimate: bool = True,
               in_flight: int = 8, solo: bool = False, by_cosets=None, share_of=None, devices=None, min_cosets=None) -> dict:
    """``solo``: this process runs the replay ALONE even inside an initialised process group (bench.py's one-process form:
    the other ranks are parked; the split over devices, if any, is hm_set_msm_devices' inside the library).
    ``devices`` (with ``solo``): the device list of the one-process form (hm_set_msm_devices has been called with it).  With
    more than one entry and k >= 14 the extended-domain steps go BY COSETS OVER THE DEVICES, one host thread per device -- what
    a Rust prover with one process for the node would do with its thread pool: every device inverse-transforms all columns,
    takes its cosets, and (like the ranks of the per-process form) builds the z columns and permuted lookup columns itself;
    n x 32 B per coset come back to the first device over xGMI.  The commitments go through the multi-device handle as before.
    ``share_of=(rank, world)``: ONE rank's share of the `world`-rank replay, run alone (no process group, no exchange):
    its commitments of every phase, its cosets, and -- for rank 0 -- the steps only rank 0 runs.  What a one-GPU box can
    MEASURE of the N-GPU replay; the exchanges (96 B per commitment, n x 32 B per coset) are what it leaves out.
    ``min_cosets`` (with the coset route; default on): evaluate_h on the j - 1 cosets that DETERMINE the quotient instead of all
    E = 2^(extended_k - k) (EvaluationDomain.combine_cosets(cosets=...): h has fewer than n (j - 1) coefficients -- 5 of 8 cosets for
    the MerkleSumTree circuit); the cosets are then dealt from the last rank backwards, rank 0 gets one last.  The same h for a
    satisfied circuit, word for word (tests/test_mini_prover_gpu.py); the replay's columns are synthetic and its h is not checked.
    ``by_cosets``: the extended-domain steps (coset transforms, evaluate_h, the inverse transform of h) one coset of the
    n-th roots at a time (EvaluationDomain.coeff_to_coset).  Default: on with more than one rank from k = 14 -- the E = 2^(extended_k
    - k) cosets are dealt over the ranks, every rank transforms all columns onto ITS cosets from the coefficient arrays, runs
    the per-coset program and hands back n x 32 B per coset -- and off on one GPU (the whole-array route is the 4 % cheaper
    one there; ``by_cosets=True`` measures the other)."""
    import torch
    import torch.distributed as dist

    shape = SHAPES[shape_name]
    device = device or torch.device("cuda", torch.cuda.current_device())
    world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() and not solo else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if solo:
        group = _NO_GROUP                 # the exchange helpers of sharding.py then run their local part only
    if share_of is not None:
        if world != 1:
            raise ValueError("run_replay: share_of is measured by one process alone")
        rank, world = share_of
        if not 0 <= rank < world or world < 2 or shape.k >= 22:
            raise ValueError("run_replay: share_of = (rank, world) with world >= 2, for the prover-sized shapes")
        group = _NO_GROUP
    exchanging = world > 1 and share_of is None       # a process group is there and takes part
    k, n = shape.k, 1 << shape.k
    dom = EvaluationDomain(shape.max_degree, k)
    d = shape.max_degree
    zp = math.ceil(shape.equality_columns / (d - 2))
    A, L = shape.advice, shape.lookups
    counts = {
        "msm_sparse": A + 2 * L, "msm_dense": zp + L + 1 + (d - 1) + 2,
        "intt_n": A + 1 + 3 * L + zp, "coset_ntt_ext": A + 1 + 3 * L + zp, "intt_ext": 1,
    }

    # Multi-GPU split: prover-sized MSMs go to the ranks as WHOLE commitments (every rank holds the
    # full SRS; a 2^15-point index-range shard would be pure latency), and so do the transforms; index-range
    # shards are for n >= 2^22.
    job_mode = world > 1 and k < 22
    one_proc_devs = list(devices) if (solo and devices and len(devices) > 1 and k >= 14) else None
    if by_cosets is None:
        by_cosets = (job_mode and k >= 14) or one_proc_devs is not None   # below k = 14 the per-coset launches cost more than the split returns
    # a real SRS with a known trapdoor; in index-range mode this rank keeps its slice of g and g_lagrange resident
    lo, hi = (0, n) if job_mode else shard_range(n, rank, world)
    gen = G1_GENERATOR
    params = ParamsKZG.setup(k, REPLAY_S, device=device, keep_points=world > 1 and not job_mode)
    if world > 1 and not job_mode:
        g_h = register_bases(params.g_points[lo:hi].contiguous())
        gl_h = register_bases(params.g_lagrange_points[lo:hi].contiguous())
        params.release()
    else:
        g_h, gl_h = params.g_handle, params.g_lagrange_handle
    dense = [_rand_fr(n, 100 + i, device) for i in range(2)]
    ntt_batch = _rand_fr(8 * n, 300, device).reshape(8, n, 4)
    sparse = [_sparse_column(n, shape.used_rows, 200 + i, device) for i in range(2)]

    from .arithmetic import FQ_ONE_MONT as _FQ_ONE

    # expected commitments, by a route that shares no kernel with the MSM: f(s) by Horner on the device
    # (for a column committed in the Lagrange basis: of its inverse NTT), then [f(s)]G
    def expected_commitment(col, lagrange):
        coeffs = dom.lagrange_to_coeff(col.clone()) if lagrange else col
        fs = eval_polynomial(coeffs.reshape(1, n, 4), fr_words(REPLAY_S).reshape(1, 4))
        pt = g1_fixed_base_mul(torch.from_numpy(fs.view(np.int64)).to(device), gen).cpu().numpy().view(np.uint64)[0]
        out = np.zeros(12, dtype=np.uint64)
        if pt.any():
            out[:8] = pt
            out[8:] = _FQ_ONE
        return out

    expected = {("sparse", i): expected_commitment(sparse[i], True) for i in range(2)}
    expected.update({("dense_l", i): expected_commitment(dense[i], True) for i in range(2)})
    expected.update({("dense_c", i): expected_commitment(dense[i], False) for i in range(2)})
    checked = {"commitments": 0}

    def msm(col, handle):
        return sharded_multiexp(col[lo:hi].contiguous() if world > 1 else col, handle, group=group)

    streams = [torch.cuda.Stream(device=device) for _ in range(in_flight)]

    # evaluate_h: the circuit's own program (circuits.py: the gates of the reference's chips, the permutation argument over
    # its equality columns, its lookup arguments, divide_by_vanishing_poly as the last multiplication) over 2^extended_k rows
    from .circuits import CONSTRAINT_SYSTEMS, evaluate_h_program
    cs = CONSTRAINT_SYSTEMS[shape_name]()
    ge, lay = evaluate_h_program(cs, k, dom.extended_k, delta=pow(7, 1 << 28, FR_MODULUS))
    gate_prog = ge.compile(lay.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=1 << (dom.extended_k - k),
                           short_columns=lay.short_columns)
    t_inv_col = _rand_fr(1 << (dom.extended_k - k), 500, device)          # the 2^(extended_k - k) values of 1 / (X^n - 1) on the coset
    E = dom.num_cosets()
    if min_cosets is None:
        min_cosets = True
    use_cosets = list(range(dom.min_cosets() if min_cosets else E))       # the cosets evaluate_h runs on (positions = indices here)
    NC = len(use_cosets)
    coset_prog = None
    if by_cosets:
        # the per-coset program: the undivided numerator, rotations unscaled (1 / (X^n - 1), a constant per coset, rides on
        # the recombination matrix)
        ge1, lay1 = evaluate_h_program(cs, k, dom.extended_k, delta=pow(7, 1 << 28, FR_MODULUS), per_coset=True, divide=False)
        coset_prog = ge1.compile(lay1.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=1)
        owners = coset_owners(NC, world, spare_rank0=min_cosets) if world > 1 else [0] * NC
        my_cosets = [c for c in use_cosets if owners[c] == rank]
        stand_in = torch.zeros((n, 4), dtype=torch.int64, device=device)       # share_of: a rank without a coset still recombines
    slots = []
    if one_proc_devs:
        # one state per listed device (a device listed twice -- the one-GPU rehearsal -- gets two states): its own copy of the
        # columns, its own evaluator program (programs belong to a device context), its own outputs
        dev_owner = coset_owners(NC, len(one_proc_devs), spare_rank0=min_cosets)
        for i, dv in enumerate(one_proc_devs):
            with torch.cuda.device(dv):
                tdev = torch.device("cuda", dv)
                slots.append({
                    "index": i, "device": tdev, "cosets": [c for c in use_cosets if dev_owner[c] == i],
                    "ntt_batch": ntt_batch.to(tdev).clone(),
                    "prog": ge1.compile(lay1.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=1),
                })

    def gate_cols(e):            # every column of the table aliases one of the extended arrays (the arithmetic does not care)
        cols = [e[i % e.shape[0]] for i in range(lay.num_fixed_entries + cs.num_advice + cs.num_instance)]
        cols[lay.t_inv] = t_inv_col
        return cols

    h_values = torch.zeros((dom.extended_len(), 4), dtype=torch.int64, device=device)
    n_queries = 2 * A + 3 * zp + 5 * L + (d - 1)              # advice at ~2 rotations, permutation / lookup products at 3 ...
    eval_index = np.arange(n_queries, dtype=np.uint32) % 8
    eval_points = np.stack([fr_words(pow(REPLAY_S, 3 + q, FR_MODULUS)) for q in range(n_queries)])

    n_open_polys = A + 3 * L + zp + (d - 1)                    # advice, lookup triples, permutation products, h pieces
    open_coeffs = np.stack([fr_words(pow(REPLAY_S, 17 + j, FR_MODULUS)) for j in range(max(n_open_polys, 4))])
    open_acc = torch.empty((5, n, 4), dtype=torch.int64, device=device)
    z_factors = _rand_fr((zp + L) * n, 400, device).reshape(zp + L, n, 4)
    z_columns = torch.empty((zp + L, n, 4), dtype=torch.int64, device=device)
    # a range-check lookup: the table holds 0 .. 2^16 - 1 (repeated), the input column values of that range
    lookup_table = torch.zeros((n, 4), dtype=torch.int64, device=device)
    lookup_table[:, 0] = torch.arange(n, device=device) % min(n - 7, 1 << 16)
    lookup_table = linear_combination([lookup_table], np.stack([fr_words((1 << 256) % FR_MODULUS)]))
    lookup_input = lookup_table[torch.randperm(n, device=device)].contiguous()
    lookup_input[n - 7:] = lookup_table[:7]            # (rows beyond the usable ones are not read)
    for st in slots[1:]:                               # the devices beyond the first build the z / permuted columns themselves
        with torch.cuda.device(st["device"]):
            st["z_factors"] = z_factors.to(st["device"]).clone()
            st["z_columns"] = torch.empty_like(st["z_factors"])
            st["lookup_input"] = lookup_input.to(st["device"]).clone()
            st["lookup_table"] = lookup_table.to(st["device"]).clone()

    def msm_phase(jobs):
        """The commitments of one prover phase are independent: every rank keeps `in_flight` of its local
        MSMs in flight on as many streams (one MSM's sort / bucket reduction / host fold hides behind
        another's accumulation), then the partials of the whole phase cross xGMI in ONE all-gather."""
        keys = [key for _, _, key in jobs]
        if share_of is not None:              # this rank's commitments only (job_owner's round-robin deal), nothing exchanged
            mine = [j for j in range(len(jobs)) if job_owner(j, world) == rank]
            return (sharded_multiexp_batch([(jobs[j][0], jobs[j][1]) for j in mine], group=_NO_GROUP, streams=streams),
                    [keys[j] for j in mine])
        if job_mode:
            return job_parallel_multiexp_batch([(col, handle) for col, handle, _ in jobs], group=group, streams=streams), keys
        local = [((col[lo:hi].contiguous() if world > 1 else col), handle) for col, handle, _ in jobs]
        return sharded_multiexp_batch(local, group=group, streams=streams), keys

    def whole_array_steps(t):
        t0 = time.perf_counter()
        # each transform stays on ONE GPU (north star) and rank 0 runs them all: evaluate_h below reads every extended array,
        # and arrays transformed elsewhere would have to cross xGMI whole (round 3 dealt them and never moved them; with more
        # than one rank the scaling route is coset_steps)
        share = lambda c: c if rank == 0 else 0
        if share(counts["intt_n"]) or share(counts["coset_ntt_ext"]):
            # same-size transforms of one prover phase go through one batched call (<= 8 polynomials at a
            # time here, bounding the extended-domain buffers)
            todo = share(counts["intt_n"])
            while todo > 0:
                b = min(8, todo)
                dom.lagrange_to_coeff(ntt_batch[:b])
                todo -= b
            ext = None
            todo = share(counts["coset_ntt_ext"])
            while todo > 0:
                b = min(8, todo)
                ext = dom.coeff_to_extended(ntt_batch[:b], internal=True)
                todo -= b
            if rank == 0 and ext is not None:
                for _ in range(counts["intt_ext"]):
                    dom.extended_to_coeff(ext[0])
        torch.cuda.synchronize()
        t["ntt"] = time.perf_counter() - t0
        # evaluate_h over the extended domain (the device GraphEvaluator on the circuit's own program) and the Horner
        # evaluations at x * omega^rot
        t0 = time.perf_counter()
        if rank == 0 and ext is not None:
            # the extended arrays would come out of coeff_to_extended(internal=True) in a prover (the factor 32 rides on the
            # coset constants): the evaluator then loads every column without a conversion product
            gate_prog.evaluate(gate_cols(ext), h_values, beta=REPLAY_S + 1, gamma=REPLAY_S + 2, theta=REPLAY_S + 3, y=REPLAY_S,
                               columns_internal=True)
        torch.cuda.synchronize()
        t["evaluate_h"] = time.perf_counter() - t0

    def run_cosets(batch8, cosets, prog, dev, times):
        """The extended-domain steps of `cosets` on one device, fused: every column onto ALL of them in one transform call per 8
        columns (hm_coeff_to_cosets: the cosets of a column lie one after the other), the undivided numerator over
        len(cosets) segments of n rows in ONE launch, the partials in one call.  -> {coset: (n, 4) partial}; times["ntt"] /
        times["evaluate_h"] grow by what this took."""
        if not cosets:
            return {}
        q = len(cosets)
        t0 = time.perf_counter()
        cc = None
        todo = counts["coset_ntt_ext"]
        while todo > 0:
            b = min(8, todo)
            cc = dom.coeff_to_cosets(batch8[:b], cosets, internal=True)             # (b, q, n, 4)
            todo -= b
        torch.cuda.synchronize(dev)
        times["ntt"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        cols = [cc[i % cc.shape[0]].reshape(q * n, 4) for i in range(lay.num_fixed_entries + cs.num_advice + cs.num_instance)]
        values = torch.zeros((q, n, 4), dtype=torch.int64, device=dev)
        prog.evaluate(cols, values.reshape(q * n, 4), beta=REPLAY_S + 1, gamma=REPLAY_S + 2, theta=REPLAY_S + 3, y=REPLAY_S,
                      columns_internal=True, segments=q)
        torch.cuda.synchronize(dev)
        times["evaluate_h"] += time.perf_counter() - t0
        t0 = time.perf_counter()
        dom.cosets_to_partials(values, cosets)
        torch.cuda.synchronize(dev)
        times["ntt"] += time.perf_counter() - t0
        return {c: values[i] for i, c in enumerate(cosets)}

    def coset_steps(t):
        """The same steps with the extended domain taken by cosets.  Every rank: all inverse transforms (each rank needs every
        coefficient array: 8 MiB per column at k = 18; redundant work instead of an all-gather of them), then its cosets, fused
        (run_cosets).  Then ONE all-gather of n x 32 B per coset, and rank 0 recombines the quotient's pieces (one linear
        combination of the partials per piece, the vanishing division on the matrix)."""
        t0 = time.perf_counter()
        todo = counts["intt_n"]
        while todo > 0:
            b = min(8, todo)
            dom.lagrange_to_coeff(ntt_batch[:b])
            todo -= b
        torch.cuda.synchronize()
        t["ntt"] = time.perf_counter() - t0
        t["evaluate_h"] = 0.0
        parts = run_cosets(ntt_batch, my_cosets, coset_prog, device, t)
        t0 = time.perf_counter()
        if share_of is not None:              # no exchange: the recombination runs on this rank's partials, repeated
            # (a rank without a coset -- rank 0 of eight with five cosets -- still pays for the recombination: any array stands in)
            allp = [parts[my_cosets[c % len(my_cosets)]] if my_cosets else stand_in for c in use_cosets]
        else:
            allp = gather_coset_partials(parts, NC, group=group, shape=(n, 4), owners=owners)
        if rank == 0 and allp:
            dom.combine_cosets([p.to(device) for p in allp], cosets=use_cosets, divide_by_vanishing=True)
        torch.cuda.synchronize()
        t["ntt"] += time.perf_counter() - t0

    replicate_columns = by_cosets and world > 1

    def device_share(st):
        """One device's part of the one-process form, on its own host thread (ctypes calls release the GIL; the current device
        is per thread): all inverse transforms, its cosets, and -- beyond the first device -- the z and permuted columns."""
        torch.cuda.set_device(st["device"])
        times = {"ntt": 0.0, "evaluate_h": 0.0, "columns": 0.0}
        t0 = time.perf_counter()
        todo = counts["intt_n"]
        while todo > 0:
            b = min(8, todo)
            dom.lagrange_to_coeff(st["ntt_batch"][:b])
            todo -= b
        torch.cuda.synchronize(st["device"])
        times["ntt"] += time.perf_counter() - t0
        parts = run_cosets(st["ntt_batch"], st["cosets"], st["prog"], st["device"], times)
        if st["index"] != 0:
            t0 = time.perf_counter()
            batch_invert(st["z_factors"])
            if zp:
                grand_product_batch(list(st["z_factors"][:zp]), fr_words(1), chain_row=n - 7, outs=list(st["z_columns"][:zp]))
            if L:
                grand_product_batch(list(st["z_factors"][zp:]), fr_words(1), outs=list(st["z_columns"][zp:]))
                permute_expression_pairs([st["lookup_input"]] * L, [st["lookup_table"]] * L, n - 7, blinding_seed=1)
            torch.cuda.synchronize(st["device"])
            times["columns"] += time.perf_counter() - t0
        return parts, times

    def coset_steps_over_devices(t):
        """The one-process form: one host thread per listed device runs device_share; the partials cross xGMI to the first
        device (n x 32 B per coset), which recombines.  t["ntt"] / t["evaluate_h"]: the longest device's own times."""
        from concurrent.futures import ThreadPoolExecutor
        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=len(slots)) as pool:
            results = list(pool.map(device_share, slots))
        torch.cuda.set_device(device)
        parts = {}
        for pr, _ in results:
            parts.update(pr)
        dom.combine_cosets([parts[c].to(device) for c in use_cosets], cosets=use_cosets, divide_by_vanishing=True)
        torch.cuda.synchronize()
        block = time.perf_counter() - t0
        t["evaluate_h"] = max(tm["evaluate_h"] for _, tm in results)
        t["ntt"] = block - t["evaluate_h"]              # everything else of the block: transforms, the other devices' columns, gather, recombination

    def proof_once():
        t = {"msm": 0.0, "ntt": 0.0}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r1 = msm_phase([(sparse[i & 1], gl_h, ("sparse", i & 1)) for i in range(counts["msm_sparse"])])
        coeff_from = counts["msm_dense"] - (d - 1) - 2          # the last d + 1 commitments are in coefficient form
        r2 = msm_phase([(dense[i & 1], g_h if i >= coeff_from else gl_h, ("dense_c" if i >= coeff_from else "dense_l", i & 1))
                        for i in range(counts["msm_dense"])])
        torch.cuda.synchronize()
        t["msm"] = time.perf_counter() - t0
        t["results"] = (r1, r2)
        if one_proc_devs:
            coset_steps_over_devices(t)
        elif by_cosets:
            coset_steps(t)
        else:
            whole_array_steps(t)
        t0 = time.perf_counter()
        if rank == 0:
            eval_polynomial(ntt_batch, eval_points, poly_index=eval_index)
        t["eval_polynomial"] = time.perf_counter() - t0
        # the z columns of the permutation and lookup arguments: denominators inverted in a batch, then the running
        # product; and multiopen's witness polynomials: per rotation set one combination of the committed polynomials,
        # a division by (X - point) per opening point, then the final combination and division
        t0 = time.perf_counter()
        # (by cosets every rank needs every column's coefficients: the steps that PRODUCE columns on the device -- the z products
        # and the permuted lookup columns -- run on every rank, redundantly, instead of being broadcast: 27 x 8 MiB at k = 18)
        if rank == 0 or replicate_columns:
            batch_invert(z_factors)                 # the denominators of all z columns are independent: one call
            # permutation: each column set starts where the one before stood at the last usable row (upstream's last_z),
            # chained on the device; lookups: one independent product each.  Two launch chains in all.
            if zp:
                grand_product_batch(list(z_factors[:zp]), fr_words(1), chain_row=n - 7, outs=list(z_columns[:zp]))
            if L:
                grand_product_batch(list(z_factors[zp:]), fr_words(1), outs=list(z_columns[zp:]))
        torch.cuda.synchronize()
        t["grand_products"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if rank == 0 or replicate_columns:
            if L:                                     # the permuted input / table columns of every lookup argument, one call
                permute_expression_pairs([lookup_input] * L, [lookup_table] * L, n - 7, blinding_seed=1)
        torch.cuda.synchronize()
        t["lookup_permute"] = time.perf_counter() - t0
        t0 = time.perf_counter()
        if rank == 0:
            per_set = -(-n_open_polys // 4)
            sets = []
            for si in range(4):
                cnt = min(per_set, n_open_polys - si * per_set)
                sets.append(linear_combination([ntt_batch[j % 8] for j in range(cnt)], open_coeffs[:cnt], out=open_acc[si]))
            # the openings of one round are independent of each other: one launch chain
            kate_division_batch([sets[qi % 4] for qi in range(5)], [fr_words(pow(REPLAY_S, 5 + qi, FR_MODULUS)) for qi in range(5)])
            fin = linear_combination(sets, open_coeffs[:4], out=open_acc[4])
            kate_division(fin, fr_words(pow(REPLAY_S, 11, FR_MODULUS)))
        torch.cuda.synchronize()
        t["multiopen"] = time.perf_counter() - t0
        return t

    proof_once().pop("results")                         # warm-up: tables, workspaces
    # launch-bound work on a shared host: the best of three replays (every rank runs all three)
    wall, phases = None, None
    for _ in range(3):
        if exchanging:
            dist.barrier(group)
        t0 = time.perf_counter()
        ph = proof_once()
        if exchanging:
            dist.barrier(group)
        w = time.perf_counter() - t0
        if exchanging:
            comm_dev = device if dist.get_backend(group) == "nccl" else torch.device("cpu")
            tt = torch.tensor([w], dtype=torch.float64, device=comm_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX, group=group)
            w = float(tt.item())
        for res, keys in ph.pop("results"):              # outside the timed region: every commitment against [f(s)]G
            for got, key in zip(res, keys):
                if not np.array_equal(got, expected[key]):
                    raise RuntimeError(f"replay {shape_name}: commitment {key} does not satisfy the KZG identity")
                checked["commitments"] += 1
        if wall is None or w < wall:
            wall, phases = w, ph

    out = {
        "circuit": shape.name, "shape_key": shape_name, "k": k, "extended_k": dom.extended_k, "n_gpus": world,
        **({"share_of": {"rank": rank, "world": world, "note": "ONE rank's share of the replay, measured alone on one GPU: no exchange "
                                                               "(96 B per commitment, n x 32 B per coset) in the time"}}
           if share_of is not None else {}),
        "multi_gpu_split": ("whole commitments per rank (full SRS on every GPU); extended-domain steps by cosets: "
                            f"{NC} of {E} cosets dealt over the ranks, n x 32 B per coset gathered" if by_cosets else
                            "whole commitments / transforms per rank (full SRS on every GPU)") if job_mode else
                           ("index-range shards" if world > 1 else "none"),
        "extended_domain": (f"by cosets over {len(one_proc_devs)} devices, one host thread each ({NC} of {E} cosets)" if one_proc_devs else
                            f"by cosets, {NC} of {E} (coeff_to_coset, per-coset evaluate_h, coset_to_partial, combine_cosets"
                            + (": the j - 1 cosets that determine the quotient)" if NC < E else ")")) if by_cosets
                           else "whole array (coeff_to_extended, evaluate_h over 2^extended_k rows, extended_to_coeff)",
        "shape": {"advice": A, "lookups": L, "equality_columns": shape.equality_columns, "max_degree": d,
                  "source": shape.source},
        "calls": counts,
        "device_resident_s": {"msm": phases["msm"], "ntt": phases["ntt"], "evaluate_h": phases["evaluate_h"],
                              "eval_polynomial": phases["eval_polynomial"], "grand_products": phases["grand_products"],
                              "lookup_permute": phases["lookup_permute"], "multiopen": phases["multiopen"], "total": wall},
        "beyond_msm_ntt": {"evaluate_h": f"{len(gate_prog.calcs)} GraphEvaluator calculations per row over 2^{dom.extended_k} rows: "
                                         f"{len(cs.polynomials())} gate polynomials of {cs.source}, the permutation argument over "
                                         f"{len(cs.equality)} columns in {cs.permutation_sets()} sets, {len(cs.lookups)} lookup arguments, "
                                         "the vanishing-polynomial division",
                           "eval_polynomial": f"{n_queries} Horner evaluations of 2^{k}-coefficient polynomials (estimate)",
                           "grand_products": f"one batch inversion of {zp + L} x 2^{k} denominators, {zp + L} running products over 2^{k} rows",
                           "lookup_permute": f"{L} x permute_expression_pair over 2^{k} - 7 rows (range-check column: 256-bit bitonic sorts + arrangement)",
                           "multiopen": f"{n_open_polys} polynomials combined in 4 rotation sets, 6 divisions by X - point (estimate)"},
        "verified": {"commitments_checked": checked["commitments"], "distinct_column_base_pairs": len(expected),
                     "against": "KZG identity commit(f) == [f(s)]G, f(s) by device Horner (+ inverse NTT for Lagrange-basis columns)"},
        "note": "MSM/NTT trace replay on synthetic polynomials (no Rust toolchain here); CPU-side parts of create_proof "
                "excluded; best of three replays",
    }
    if include_host_pointer_estimate and world == 1:
        # the drop-in (host-pointer) cost of the same trace: one representative call of each kind
        import ctypes
        h_s = to_host(dense[0])

        medians = {}

        def best_of(fn, reps=5, name=None):
            """Seconds of the fastest of `reps` calls after one warm-up: a per-call figure, not a sample of whatever one call met (staging
            buffers growing, copy lanes being created).  The MEDIAN of the same calls goes to `medians[name]`: the totals are reported on
            both (round 5 reported the fastest of three only, round 4 one sample: not comparable round over round -- profiles/README.md)."""
            fn()
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                fn()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            if name:
                medians[name] = ts[len(ts) // 2]
            return ts[0]

        from .arithmetic import best_fft
        t_msm = best_of(lambda: best_multiexp(h_s, gl_h), name="msm_each")
        a_n = h_s.copy()
        t_ntt_n = best_of(lambda: best_fft(a_n, fr_words(dom.omega), k), name="ntt_n_each")
        a_e = np.zeros((dom.extended_len(), 4), dtype=np.uint64); a_e[:n] = h_s
        from . import _lib as _l0

        def _phases():
            st_ = _l0.Stats()
            _l0.check(_l0.load().hm_get_stats(ctypes.byref(st_)))
            return np.array([st_.ntt_h2d_us, st_.ntt_device_us, st_.ntt_d2h_us])

        ph0 = _phases()
        t_ntt_e = best_of(lambda: best_fft(a_e, fr_words(dom.extended_omega), dom.extended_k), name="ntt_ext_each")
        ntt_ext_phases = (_phases() - ph0) / 6.0           # mean of the six calls (warm-up + five): upload / device / download microseconds
        # the two EvaluationDomain steps through their own host-pointer forms (the optional src/poly/domain.rs edits of rust/): the zero
        # padding never goes up, the truncated tail never comes down
        ext_h = dom.coeff_to_extended(h_s)
        t_c2e = best_of(lambda: dom.coeff_to_extended(h_s, out=ext_h), name="coeff_to_extended_each")          # into pages that exist: the library's own time
        # ... into a FRESH 2^extended_k x 32 B array (what the Rust glue's new Vec is): first-touch faults on top -- the same faults
        # upstream's `resize` to the extended length takes BEFORE it calls best_fft, which the best_fft figure above does not contain
        t_c2e_fresh = best_of(lambda: dom.coeff_to_extended(h_s))

        def zero_pad():
            pad = np.empty((dom.extended_len(), 4), dtype=np.uint64)
            pad[:n] = h_s
            pad[n:] = 0
        t_resize = best_of(zero_pad)
        t_e2c = best_of(lambda: dom.extended_to_coeff(ext_h), name="extended_to_coeff_each")
        del ext_h
        # ... and the commitments of the whole proof from HOST arrays through the batch call (uploads pipelined behind the
        # other commitments' kernels): what a prover that keeps its polynomials in host vectors gets per proof
        from .arithmetic import best_multiexp_batch
        h_sparse, h_dense = to_host(sparse[0]), h_s
        for _ in range(2):                   # warm-up: the per-lane staging buffers and slot workspaces reach their sizes here
            best_multiexp_batch([h_sparse] * counts["msm_sparse"], gl_h)
            best_multiexp_batch([h_dense] * counts["msm_dense"], gl_h)
        t0 = time.perf_counter()
        best_multiexp_batch([h_sparse] * counts["msm_sparse"], gl_h)
        best_multiexp_batch([h_dense] * counts["msm_dense"], gl_h)
        t_batch_host = time.perf_counter() - t0
        hp_total = t_msm * (counts["msm_sparse"] + counts["msm_dense"]) + t_ntt_n * counts["intt_n"] \
            + t_ntt_e * (counts["coset_ntt_ext"] + counts["intt_ext"])
        # the two totals side by side: what a prover gets from the UNMODIFIED drop-in (every best_multiexp / best_fft call
        # moves its arrays over PCIe) and what it gets once its polynomials stay in HBM -- neither hidden behind the other
        batched = hp_total - t_msm * (counts["msm_sparse"] + counts["msm_dense"]) + t_batch_host
        # library time in all three totals (arrays whose pages exist).  What the HOST pays on top for memory it has never touched is listed
        # beside them, not inside: the best_fft routes need upstream's resize-to-extended-length before every call (zero-fill of fresh
        # pages), the domain edits write into a fresh output array instead (first-touch faults inside the call)
        domain_edits = batched - t_ntt_e * (counts["coset_ntt_ext"] + counts["intt_ext"]) + t_c2e * counts["coset_ntt_ext"] \
            + t_e2c * counts["intt_ext"]
        hp_total_median = medians["msm_each"] * (counts["msm_sparse"] + counts["msm_dense"]) + medians["ntt_n_each"] * counts["intt_n"] \
            + medians["ntt_ext_each"] * (counts["coset_ntt_ext"] + counts["intt_ext"])
        out["total_s"] = {"drop_in_host_pointers": hp_total, "drop_in_host_pointers_on_median_calls": hp_total_median, "device_resident": wall,
                          "drop_in_with_batched_commitments": batched, "drop_in_with_domain_edits": domain_edits,
                          "host_page_faults_on_top": {"best_fft_routes_resize_to_extended_length": t_resize * counts["coset_ntt_ext"],
                                                      "domain_edits_fresh_output_arrays": max(t_c2e_fresh - t_c2e, 0.0) * counts["coset_ntt_ext"]},
                          "note": "drop_in_host_pointers = per-call times of hm_msm_bn256_g1_h / hm_ntt_bn256_fr x the trace's counts (MSM and "
                                  "NTT only); drop_in_with_batched_commitments replaces the per-call MSMs by two hm_msm_batch_bn256_g1_h "
                                  "calls; drop_in_with_domain_edits also replaces the extended-domain best_fft calls by "
                                  "hm_coeff_to_extended_bn256_fr / hm_extended_to_coeff_bn256_fr (rust/edits.json, src/poly/domain.rs); all "
                                  "three are library time on arrays whose pages exist; host_page_faults_on_top = what never-touched host "
                                  "memory costs beside them (numpy here): the zero padding to the extended length the best_fft routes do "
                                  "before every call, resp. the fresh output array of every coeff_to_extended; "
                                  "device_resident = the whole replayed trace, polynomials in HBM (includes the non-MSM/NTT steps)"}
        from . import _lib as _l
        _st = _l.Stats()
        _l.check(_l.load().hm_get_stats(ctypes.byref(_st)))
        out["host_pointer_estimate_s"] = {
            "host_copies": {"through_the_librarys_pinned_lanes": int(_st.host_copies_staged), "handed_to_hipMemcpy": int(_st.host_copies_direct),
                            "ranges_registered": int(_st.host_ranges_registered),
                            "note": "csrc/xfer.hip: copies of this process so far, by route.  The rule is on the range, not on a timing: unregistered "
                                    "host ranges of 256 KiB or more go through the library's pinned lanes (the same cost on every box), registered "
                                    "ones (hm_host_register) and small ones straight to hipMemcpy"},
            "msm_batches_from_host_arrays": t_batch_host,
            "msm_each": t_msm, "ntt_n_each": t_ntt_n, "ntt_ext_each": t_ntt_e, "coeff_to_extended_each": t_c2e, "extended_to_coeff_each": t_e2c,
            "coeff_to_extended_into_a_fresh_array_each": t_c2e_fresh, "host_zero_padding_each": t_resize,
            "page_fault_note": "coeff_to_extended_each writes into touched pages; a fresh output array (the normal case: a new Vec) adds its "
                               "first-touch faults (the kernel zeroes 2^extended_k x 32 B either way); upstream's resize-to-extended-length "
                               "(host_zero_padding_each, numpy here) takes the same faults before best_fft",
            "total": hp_total, "median_of_five_calls": dict(medians),
            "ntt_ext_mean_phase_us": {"h2d": float(ntt_ext_phases[0]), "device": float(ntt_ext_phases[1]), "d2h": float(ntt_ext_phases[2])},
            "note": "PCIe-inclusive: every call uploads its scalars / moves its array both ways; *_each = the fastest of five calls after one "
                    "warm-up, median_of_five_calls = the median of the same five"}
    gate_prog.destroy()
    if coset_prog is not None:
        coset_prog.destroy()
    for st in slots:
        with torch.cuda.device(st["device"]):
            st["prog"].destroy()
    if world > 1 and not job_mode:
        release_bases(g_h)
        release_bases(gl_h)
    else:
        params.release()
    return out
