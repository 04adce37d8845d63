#!/usr/bin/env python3
"""MSM timing on skewed scalar columns (constant column, all ones, 0/1 flags): development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.arithmetic import G1_GENERATOR
from halo2_experiments_amd.domain import fr_words

def rand_fr(n, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    x[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return x

for k in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["18", "22"])]:
    n = 1 << k
    hd = h.register_bases(h.g1_fixed_base_mul(rand_fr(n, 11), G1_GENERATOR))
    one = torch.from_numpy(fr_words(1).view(np.int64)).cuda()
    cols = {
        "uniform": rand_fr(n, 12),
        "constant (one random value)": rand_fr(1, 13).repeat(n, 1).contiguous(),
        "all ones": one.repeat(n, 1).contiguous(),
        "0/1 flags": torch.where((torch.rand(n, device="cuda") < 0.5).unsqueeze(1), one.repeat(n, 1), torch.zeros((n, 4), dtype=torch.int64, device="cuda")).contiguous(),
    }
    for name, s in cols.items():
        h.best_multiexp(s, hd); torch.cuda.synchronize()
        t = time.perf_counter(); h.best_multiexp(s, hd); dt = time.perf_counter() - t
        st = h.msm_stats()
        print(f"2^{k} {name:30s} {dt*1e3:8.3f} ms | sort {st['sort_ms']:.3f} acc {st['accumulate_ms']:.3f} (k3 {st['accumulate_kernel_ms']:.3f}) red {st['reduce_ms']:.3f} pairs {st['pairs']} tasks {st['tasks']}")
    h.release_bases(hd)
