"""CPU suite: the N>1 path (index-range shards + one all-gather of 96-byte partials + host fold)
with world_size = 2 over gloo.  No GPU here, so each rank's local MSM is the oracle standing in for
the kernel; what is under test is the product's sharding / exchange / fold logic."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from halo2_experiments_amd.sharding import shard_range, sharded_multiexp
    from oracle import cpu_ref
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "msm.npz"))
    results = {}
    for name in ("n255_uniform", "n33_edge", "pmone", "n1_uniform"):
        s, b = g[f"{name}_s"], g[f"{name}_b"]
        lo, hi = shard_range(s.shape[0], rank, world)
        out = sharded_multiexp(s[lo:hi], b[lo:hi], local_msm=lambda c, p: cpu_ref.best_multiexp(c, p, 1))
        results[name] = out
    # a whole phase at once: local partials of every job, ONE all-gather, per-job fold
    from halo2_experiments_amd.sharding import sharded_multiexp_batch
    names = ("n255_uniform", "pmone", "n33_edge")
    jobs = []
    for name in names:
        s, b = g[f"{name}_s"], g[f"{name}_b"]
        lo, hi = shard_range(s.shape[0], rank, world)
        jobs.append((s[lo:hi], b[lo:hi]))
    outs = sharded_multiexp_batch(jobs, local_batch=lambda js: [cpu_ref.best_multiexp(c, p, 1) for c, p in js])
    for name, out in zip(names, outs):
        results["batch_" + name] = out
    # job-level split (prover-sized MSMs): whole commitments dealt round-robin, one all-gather of results
    from halo2_experiments_amd.sharding import job_parallel_multiexp_batch
    jnames = ("n255_uniform", "pmone", "n33_edge", "n1_uniform", "n1024_uniform")
    calls = []

    def local(js):
        calls.append(len(js))
        return [cpu_ref.best_multiexp(c, p, 1) for c, p in js]

    outs = job_parallel_multiexp_batch([(g[f"{nm}_s"], g[f"{nm}_b"]) for nm in jnames], local_batch=local)
    assert calls == [3 if rank == 0 else 2]                       # jobs 0, 2, 4 on rank 0; 1, 3 on rank 1
    for nm, out in zip(jnames, outs):
        results["jobs_" + nm] = out
    assert job_parallel_multiexp_batch([], local_batch=local).shape == (0, 12)
    # evaluate_h by cosets: each rank brings the partials of the cosets it owns, everyone ends with all E in coset order
    import torch
    from halo2_experiments_amd.sharding import coset_owner, gather_coset_partials
    for e in (8, 4, 1):
        mine = {c: torch.full((5, 4), 100 * e + c, dtype=torch.int64) for c in range(e) if coset_owner(c, world) == rank}
        allp = gather_coset_partials(mine, e, shape=(5, 4))                 # (e = 1: rank 1 owns nothing and must be told the shape)
        assert len(allp) == e and all(int(allp[c][0, 0]) == 100 * e + c and tuple(allp[c].shape) == (5, 4) for c in range(e)), (e, rank)
    # five positions dealt from the last rank backwards (evaluate_h on the cosets that determine the quotient): rank 1 gets 3, rank 0 gets 2
    from halo2_experiments_amd.sharding import coset_owners
    owners = coset_owners(5, world, spare_rank0=True)
    assert owners == [1, 0, 1, 0, 1]
    mine = {c: torch.full((5, 4), 700 + c, dtype=torch.int64) for c in range(5) if owners[c] == rank}
    allp = gather_coset_partials(mine, 5, shape=(5, 4), owners=owners)
    assert [int(p[0, 0]) for p in allp] == [700, 701, 702, 703, 704]
    assert coset_owners(5, 8, spare_rank0=True) == [7, 6, 5, 4, 3]
    try:
        gather_coset_partials({0: torch.zeros((5, 4), dtype=torch.int64)} if rank == 1 else {}, 8)
        raise AssertionError("a rank holding another rank's coset must be refused")
    except ValueError:
        pass
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, results))


def test_world_size_2_gloo(cref):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "msm.npz"))
    for rank, results in got:
        for name, out in results.items():
            exp = g[f"{name.replace('batch_', '').replace('jobs_', '')}_r"]
            if exp.any():
                assert np.array_equal(out[:8], exp), (rank, name)
            else:
                assert not out.any(), (rank, name)


def test_shard_ranges_cover_everything():
    from halo2_experiments_amd.sharding import shard_range
    for n in (0, 1, 7, 8, 1 << 20, (1 << 24) + 3):
        for world in (1, 2, 3, 8):
            spans = [shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker_many(rank, world, port, q):
    """The same exchange logic at a world size where ranks outnumber some of the things dealt (5 cosets over 4 ranks, a 1-point MSM
    over 4 ranks, 5 jobs over 4 ranks): what the driver's N = 4 / 8 runs rely on and no one-GPU box can rehearse over RCCL."""
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from halo2_experiments_amd.sharding import (coset_owners, gather_coset_partials, job_parallel_multiexp_batch, shard_range,
                                                 sharded_multiexp)
    from oracle import cpu_ref
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(ROOT, "tests", "golden", "msm.npz"))
    results = {}
    for name in ("n255_uniform", "n33_edge", "n1_uniform"):             # n = 1: three of the four ranks hold an empty shard
        s, b = g[f"{name}_s"], g[f"{name}_b"]
        lo, hi = shard_range(s.shape[0], rank, world)
        results[name] = sharded_multiexp(s[lo:hi], b[lo:hi], local_msm=lambda c, p: cpu_ref.best_multiexp(c, p, 1))
    jnames = ("n255_uniform", "pmone", "n33_edge", "n1_uniform", "n1024_uniform")
    calls = []

    def local(js):
        calls.append(len(js))
        return [cpu_ref.best_multiexp(c, p, 1) for c, p in js]

    outs = job_parallel_multiexp_batch([(g[f"{nm}_s"], g[f"{nm}_b"]) for nm in jnames], local_batch=local)
    assert calls == [2 if rank == 0 else 1]                             # five jobs round-robin over four ranks
    for nm, out in zip(jnames, outs):
        results["jobs_" + nm] = out
    for e, spare in ((8, False), (5, True), (5, False), (3, True)):     # 8 or 5 cosets over four ranks; 3: a rank without a coset
        owners = coset_owners(e, world, spare_rank0=spare)
        assert sorted(set(owners)) == sorted(set(range(world)) & set(owners)) and len(owners) == e
        if spare and e < world:
            assert 0 not in owners                                       # rank 0 (which also runs the one-rank steps) is spared
        mine = {c: torch.full((6, 4), 1000 * e + c, dtype=torch.int64) for c in range(e) if owners[c] == rank}
        allp = gather_coset_partials(mine, e, shape=(6, 4), owners=owners)
        assert [int(p[0, 0]) for p in allp] == [1000 * e + c for c in range(e)], (e, spare, rank)
    dist.barrier()
    dist.destroy_process_group()
    q.put((rank, results))


def test_world_size_4_gloo(cref):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_many, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(ROOT, "tests", "golden", "msm.npz"))
    assert sorted(r for r, _ in got) == [0, 1, 2, 3]
    for rank, results in got:
        for name, out in results.items():
            exp = g[f"{name.replace('jobs_', '')}_r"]
            if exp.any():
                assert np.array_equal(out[:8], exp), (rank, name)
            else:
                assert not out.any(), (rank, name)
