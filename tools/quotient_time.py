#!/usr/bin/env python3
"""hm_quotient_by_cosets_bn256_fr_dev on the MerkleSumTree circuit's own program at k = 17 / 18 with every column an array of its own:
all columns from coefficients, and with the fixed entries of the table (fixed columns, sigmas, l_0 / l_last / l_active, X) kept on the
cosets as a proving key would keep them.  Development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from halo2_experiments_amd import circuits
from halo2_experiments_amd.domain import EvaluationDomain, FR_MODULUS
from halo2_experiments_amd.replay import _rand_fr
dev = torch.device("cuda", 0)
cs = circuits.merkle_sum_tree()
for k in (int(a) for a in (sys.argv[1:] or ["17", "18"])):
    dom = EvaluationDomain(cs.degree(), k)
    n = dom.n
    g, lay = circuits.evaluate_h_program(cs, k, dom.extended_k, pow(7, 1 << 28, FR_MODULUS), per_coset=True, divide=False)
    n_cols = lay.num_fixed_entries + cs.num_advice + cs.num_instance
    prog = g.compile(lay.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=1)
    cols = [_rand_fr(n, 100 + i, dev) for i in range(n_cols)]
    def t(fn, reps=5):
        fn(); torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3
    for q in (dom.min_cosets(), dom.num_cosets()):
        cosets = list(range(q))
        all_ms = t(lambda: prog.quotient_by_cosets(dom, cols, cosets=cosets, beta=3, gamma=4, theta=5, y=6))
        # what a proving key holds for the whole life of the circuit: fixed columns, sigmas, l_0 / l_last / l_active, X (and the unread t_inv entry)
        const_idx = sorted(set(range(cs.num_fixed)) | set(range(lay.sigma0, lay.sigma0 + len(cs.equality)))
                           | {lay.l0, lay.l_last, lay.l_active, lay.x_coset, lay.t_inv})
        kept = dom.coeff_to_cosets(torch.stack([cols[i] for i in const_idx]), cosets, internal=True)
        pre = [None] * n_cols
        for j, i in enumerate(const_idx):
            pre[i] = kept[j]
        per_proof = [None if pre[i] is not None else cols[i] for i in range(n_cols)]
        part_ms = t(lambda: prog.quotient_by_cosets(dom, per_proof, cosets=cosets, beta=3, gamma=4, theta=5, y=6, on_cosets=pre))
        print(f"k={k} {n_cols} columns ({len(const_idx)} constant for the circuit), {q} of {dom.num_cosets()} cosets: all from coefficients "
              f"{all_ms:.2f} ms, the circuit's constant columns kept on the cosets {part_ms:.2f} ms", flush=True)
    prog.destroy()
