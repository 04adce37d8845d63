"""Multi-GPU MSM: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).

Two ways to split, chosen by size:
  * index ranges (``sharded_multiexp`` / ``sharded_multiexp_batch``): every MSM is cut into N contiguous shards,
    for n >= 2^22 or so -- BASELINE configs[4], the 2^24 .. 2^26 microbenchmark;
  * whole jobs (``job_parallel_multiexp_batch``): the independent commitments of one prover phase are dealt
    round-robin to the ranks, each rank holding the full SRS -- BASELINE configs[3] (k ~ 18), where a 2^15-point
    shard would be pure latency.

best_multiexp is a sum over i, so it shards by index range with no data-path collective: rank g
owns coeffs/bases [lo_g, hi_g), keeps its base slice resident, and produces one partial G1.  The
only exchange is an all-gather of those partials (12 words = 96 B per rank over xGMI), folded on
every rank by ``hm_g1_sum`` -- EC addition is not an RCCL reduction operator, so gather-then-fold
*is* the reduce.  NTT is not sharded ("replicas only").
"""
from __future__ import annotations

import ctypes
import os
from typing import Callable, Optional, Tuple

import numpy as np

from . import _lib
from .arithmetic import _ptr, best_multiexp


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) of rank's share of n items (first n % world ranks get one more)."""
    if not (0 <= rank < world):
        raise ValueError("rank out of range")
    q, r = divmod(n, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def g1_sum(points: np.ndarray) -> np.ndarray:
    """Fold (k, 12) G1 words into one normalised G1 (12 words)."""
    pts = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    _lib.check(_lib.load().hm_g1_sum(_ptr(pts), pts.shape[0], _ptr(out)))
    return out


def _exchanges(group, force_collective: Optional[bool]) -> bool:
    """Does this call go through the process group's collective?  Yes with more than one rank.  With ONE rank the
    exchange is an identity and is skipped -- unless ``force_collective`` (or HALO2_MI355X_FORCE_COLLECTIVE=1) asks for
    it: the all-gather over RCCL then really runs (a one-rank communicator), which is how a one-GPU box executes the
    `nccl` branch of every function here (tests/test_rccl_gpu.py, ``bench.py --gpus 1``)."""
    import torch.distributed as dist

    if group is _NO_GROUP or not dist.is_available() or not dist.is_initialized():
        return False
    if dist.get_world_size(group) > 1:
        return True
    if force_collective is None:
        force_collective = os.environ.get("HALO2_MI355X_FORCE_COLLECTIVE", "0") not in ("", "0")
    return bool(force_collective)


def _comm_device(group):
    """Where the exchanged words live: the rank's GPU for RCCL (device tensors over xGMI), the host for gloo."""
    import torch
    import torch.distributed as dist

    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def sharded_multiexp(local_coeffs, local_bases, group=None,
                     local_msm: Optional[Callable] = None, force_collective: Optional[bool] = None) -> np.ndarray:
    """Each rank passes ITS shard; every rank returns the full sum.  ``local_msm`` defaults to the
    GPU ``best_multiexp`` (tests on CPU-only hosts inject a stand-in to exercise the exchange).
    ``force_collective``: see ``_exchanges``."""
    import torch
    import torch.distributed as dist

    partial = (local_msm or best_multiexp)(local_coeffs, local_bases)
    if not _exchanges(group, force_collective):
        return g1_sum(partial.reshape(1, 12))
    world = dist.get_world_size(group)
    dev = _comm_device(group)
    mine = torch.from_numpy(partial.view(np.int64).copy()).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    pts = torch.stack(gathered).cpu().numpy().view(np.uint64)
    return g1_sum(pts)


MAX_IN_FLIGHT = 8      # asynchronous MSM slots per device (csrc/hm_internal.h: HM_MSM_SLOTS - 1)
_NO_GROUP = object()   # marker: run the local part only, even inside an initialised process group


def sharded_multiexp_batch(jobs, group=None, streams=None, local_batch: Optional[Callable] = None,
                           force_collective: Optional[bool] = None) -> np.ndarray:
    """Several independent MSMs (the commitments of one prover phase): ``jobs`` = [(local_coeffs,
    local_bases_handle), ...], every rank passing ITS shards.  Each rank keeps up to MAX_IN_FLIGHT of its
    local MSMs in flight (``hm_msm_submit_dev`` on different streams), then ALL partials travel in one
    all-gather of len(jobs) x 96 B per rank and are folded per job.  Returns (len(jobs), 12) words.
    ``local_batch`` replaces the GPU part in CPU-only tests."""
    import torch
    import torch.distributed as dist

    if local_batch is not None:
        partials = np.stack([np.asarray(p, dtype=np.uint64).reshape(12) for p in local_batch(jobs)]) if jobs else np.zeros((0, 12), np.uint64)
    else:
        partials = np.zeros((len(jobs), 12), dtype=np.uint64)
        # runs of consecutive jobs against the same base set go to the library in one call each
        # (hm_msm_batch_bn256_g1_dev keeps eight MSMs in flight on its own streams; `streams` is accepted for
        # compatibility with callers that still pass their own)
        from .arithmetic import best_multiexp_batch
        i = 0
        while i < len(jobs):
            j = i
            while j < len(jobs) and jobs[j][1] is jobs[i][1] and jobs[j][0].shape == jobs[i][0].shape:
                j += 1
            partials[i:j] = best_multiexp_batch([col for col, _ in jobs[i:j]], jobs[i][1])
            i = j
    if not _exchanges(group, force_collective):
        if local_batch is None:
            return partials             # one rank: the library's results are already normalised (x, y, 1) / zeros
        return np.stack([g1_sum(p.reshape(1, 12)) for p in partials]) if len(partials) else partials     # a stand-in's may not be
    world = dist.get_world_size(group)
    dev = _comm_device(group)
    mine = torch.from_numpy(partials.view(np.int64).copy()).to(dev)
    gathered = [torch.empty_like(mine) for _ in range(world)]
    dist.all_gather(gathered, mine, group=group)
    pts = torch.stack(gathered).cpu().numpy().view(np.uint64)          # (world, jobs, 12)
    return np.stack([g1_sum(pts[:, j, :]) for j in range(len(jobs))]) if len(jobs) else partials


def job_owner(job: int, world: int) -> int:
    """Round-robin deal of a phase's independent jobs (commitments, transforms) to the ranks."""
    return job % world


def job_parallel_multiexp_batch(jobs, group=None, streams=None, local_batch: Optional[Callable] = None,
                                force_collective: Optional[bool] = None) -> np.ndarray:
    """Job-level multi-GPU for prover-sized MSMs: ``jobs`` = [(coeffs, bases_handle), ...], the SAME list on every
    rank (every rank holds the full SRS and can see every column); rank r computes the whole MSMs j with
    j % world == r -- up to MAX_IN_FLIGHT of them in flight -- and ONE all-gather of ceil(len(jobs) / world) x 96 B
    per rank brings every result to every rank.  Returns (len(jobs), 12) words.  A 2^18-point MSM cut into eight
    2^15-point index-range shards is entirely latency-bound (a 2^16 MSM already takes 0.6 of the time of a 2^18
    one); whole jobs keep each GPU at its efficient size and need no fold at all.
    ``local_batch`` replaces the GPU part in CPU-only tests."""
    import torch
    import torch.distributed as dist

    distributed = _exchanges(group, force_collective)
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    mine_idx = [j for j in range(len(jobs)) if job_owner(j, world) == rank]
    mine = sharded_multiexp_batch([jobs[j] for j in mine_idx], group=_NO_GROUP, streams=streams, local_batch=local_batch)
    if not distributed:
        return mine
    per_rank = (len(jobs) + world - 1) // world
    buf = np.zeros((per_rank, 12), dtype=np.uint64)
    buf[: len(mine_idx)] = mine
    dev = _comm_device(group)
    t = torch.from_numpy(buf.view(np.int64).copy()).to(dev)
    gathered = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(gathered, t, group=group)
    allr = torch.stack(gathered).cpu().numpy().view(np.uint64)          # (world, per_rank, 12)
    out = np.zeros((len(jobs), 12), dtype=np.uint64)
    for j in range(len(jobs)):
        out[j] = allr[job_owner(j, world), j // world]
    return out


def coset_owner(coset: int, world: int) -> int:
    """The E cosets of the extended domain (EvaluationDomain.coeff_to_coset) are dealt round-robin like a phase's jobs."""
    return coset % world


def coset_owners(num_cosets: int, world: int, spare_rank0: bool = False):
    """rank of every coset position 0 .. num_cosets - 1.  ``spare_rank0``: dealt from the LAST rank backwards, so that with
    fewer cosets than ranks (evaluate_h on the j - 1 cosets that determine the quotient) rank 0 -- which also runs the steps
    only one rank runs -- gets one last."""
    if spare_rank0:
        return [world - 1 - (c % world) for c in range(num_cosets)]
    return [coset_owner(c, world) for c in range(num_cosets)]


def gather_coset_partials(mine, num_cosets: int, group=None, shape=None, owners=None, force_collective: Optional[bool] = None):
    """evaluate_h by cosets over the ranks: ``mine`` = {coset: (n, 4) tensor} -- what ``coset_to_partial`` left for the
    coset POSITIONS this rank owns (``owners[position]``, default ``coset_owner``: position % world).  ONE all-gather of ceil(E / world) x n x 32 B per rank; returns the list of
    the E partials in coset order (on the device for RCCL, on the host for gloo), ready for ``combine_cosets``.  Without a
    process group (or with ``_NO_GROUP``) ``mine`` must hold all E.  ``shape``: the shape of one partial -- required on a
    rank that owns no coset (more ranks than cosets)."""
    import torch
    import torch.distributed as dist

    if not _exchanges(group, force_collective):
        return [mine[c] for c in range(num_cosets)]
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    owners = list(owners) if owners is not None else coset_owners(num_cosets, world)
    if len(owners) != num_cosets or any(not 0 <= o < world for o in owners):
        raise ValueError("gather_coset_partials: one owner rank per coset position")
    owned = [c for c in range(num_cosets) if owners[c] == rank]
    if sorted(mine) != owned:
        raise ValueError(f"gather_coset_partials: rank {rank} owns cosets {owned}, got {sorted(mine)}")
    per_rank = max(1, max(owners.count(r) for r in range(world)))
    slot_of = {}
    for r in range(world):
        for slot, c in enumerate([c for c in range(num_cosets) if owners[c] == r]):
            slot_of[c] = slot
    if shape is None:
        if not mine:
            raise ValueError("gather_coset_partials: a rank that owns no coset must be told the shape of a partial")
        shape = tuple(next(iter(mine.values())).shape)
    shape = tuple(shape)
    dev = _comm_device(group)
    buf = torch.zeros((per_rank,) + shape, dtype=torch.int64, device=dev)
    for c in owned:
        buf[slot_of[c]] = mine[c].to(dev)
    gathered = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(gathered, buf, group=group)
    return [gathered[owners[c]][slot_of[c]] for c in range(num_cosets)]
