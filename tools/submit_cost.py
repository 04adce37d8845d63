#!/usr/bin/env python3
"""Host-side cost of enqueueing / collecting one asynchronous MSM: development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.arithmetic import G1_GENERATOR, best_multiexp_submit, best_multiexp_wait

def rand_fr(n, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    x[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return x

for k in (14, 18):
    n = 1 << k
    hd = h.register_bases(h.g1_fixed_base_mul(rand_fr(n, 1), G1_GENERATOR))
    s = rand_fr(n, 2)
    for _ in range(3):
        best_multiexp_wait(best_multiexp_submit(s, hd))
    torch.cuda.synchronize()
    sub, wait = [], []
    for _ in range(20):
        t0 = time.perf_counter(); t = best_multiexp_submit(s, hd); t1 = time.perf_counter()
        torch.cuda.synchronize(); t2 = time.perf_counter()
        best_multiexp_wait(t); t3 = time.perf_counter()
        sub.append(t1 - t0); wait.append(t3 - t2)
    print(f"2^{k}: submit (enqueue all kernels) {np.median(sub)*1e6:7.1f} us   wait after the GPU is done (host fold) {np.median(wait)*1e6:7.1f} us   gpu {(t2-t1)*1e6:7.1f} us")
    h.release_bases(hd)
