#!/usr/bin/env python3
"""coeff_to_coset (table multiply fused into the first pass) against the plain batched transform of the same size: development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from halo2_experiments_amd.domain import EvaluationDomain
from halo2_experiments_amd.replay import _rand_fr
dev = torch.device("cuda", 0)
for k in (14, 16, 18, 20):
    dom = EvaluationDomain(6, k)
    a = _rand_fr(8 * dom.n, 1, dev).reshape(8, dom.n, 4)
    out = torch.empty_like(a)
    def t(fn, reps=30):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3
    plain = t(lambda: dom.lagrange_to_coeff(a))
    coset = t(lambda: dom.coeff_to_coset(a, 3, internal=True, out=out))
    inplace = t(lambda: dom.coeff_to_coset(a, 3, internal=True, out=a))
    print(f"k={k} batch 8: plain (ifft) {plain:.3f} ms, coeff_to_coset {coset:.3f} ms, in place {inplace:.3f} ms", flush=True)
