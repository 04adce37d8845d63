#!/usr/bin/env python3
"""Single-call and eight-in-flight cost of the replay's sparse and dense columns at k = 18: development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.arithmetic import G1_GENERATOR
from halo2_experiments_amd.replay import _rand_fr, _sparse_column
from halo2_experiments_amd.sharding import sharded_multiexp_batch
from halo2_experiments_amd import _lib
_lib.load().hm_msm_set_phase_timing(1)      # per-phase events also for the five-launch plan
k = int(sys.argv[1]) if len(sys.argv) > 1 else 18
n = 1 << k
dev = torch.device("cuda", 0)
pre = os.environ.get("PRECOMP", "0") == "1"
hd = h.register_bases(h.g1_fixed_base_mul(_rand_fr(n, 1, dev), G1_GENERATOR), precompute=pre)
cols = {"dense": _rand_fr(n, 2, dev), "sparse": _sparse_column(n, 1100, 3, dev)}
streams = [torch.cuda.Stream(device=dev) for _ in range(8)]
for name, col in cols.items():
    h.best_multiexp(col, hd); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): h.best_multiexp(col, hd)
    one = (time.perf_counter() - t) / 10
    st = h.msm_stats()
    sharded_multiexp_batch([(col, hd)] * 32, streams=streams); torch.cuda.synchronize()      # same count as the timed call: the slots' workspaces reach their size here
    t = time.perf_counter()
    sharded_multiexp_batch([(col, hd)] * 32, streams=streams)
    many = (time.perf_counter() - t) / 32
    print(f"2^{k} {name:6s} precomp={pre}: one call {one*1e3:.3f} ms (dev {st['total_ms']:.3f}: dig {st['digits_ms']:.3f} sort {st['sort_ms']:.3f} k3 {st['accumulate_kernel_ms']:.3f} red {st['reduce_ms']:.3f}; pairs {st['pairs']} tasks {st['tasks']} c={st['window_bits']}), 8 in flight {many*1e3:.3f} ms each")
