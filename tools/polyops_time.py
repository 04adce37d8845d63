#!/usr/bin/env python3
"""kate_division / grand product / batch inversion / linear combination wall time per size: development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import halo2_experiments_amd as h
from halo2_experiments_amd.domain import fr_words

def rand_fr(n, seed):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

z = fr_words(0x123456789ABCDEF123456789)
for k in [int(a) for a in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["14", "18", "21", "24"])]:
    n = 1 << k
    a = rand_fr(n, k); b = rand_fr(n, k + 1); out = torch.empty_like(a)
    q = torch.empty((n - 1, 4), dtype=torch.int64, device="cuda")
    cs = np.stack([fr_words(3 + i) for i in range(24)])
    print(f"2^{k}: kate {timed(lambda: h.kate_division(a, z)):.4f} ms | grand product {timed(lambda: h.grand_product(a, z, out=out)):.4f} ms | "
          f"batch invert {timed(lambda: h.batch_invert(b)):.4f} ms | lincomb x2 {timed(lambda: h.linear_combination([a, b], cs[:2], out=out)):.4f} ms "
          f"x24 {timed(lambda: h.linear_combination([a, b] * 12, cs, out=out)):.4f} ms", flush=True)
    if k <= 22:
        small = torch.zeros((n, 4), dtype=torch.int64, device="cuda"); small[:, 0] = torch.arange(n, device="cuda") % 65536
        tab = h.linear_combination([small], np.stack([fr_words(pow(2, 256, 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001))]))
        print(f"        lookup permute_expression_pair: range-check column {timed(lambda: h.permute_expression_pair(tab, tab, n - 7), 3):.4f} ms | "
              f"random column {timed(lambda: h.permute_expression_pair(a, a, n - 7), 3):.4f} ms", flush=True)
