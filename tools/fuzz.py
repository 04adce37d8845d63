#!/usr/bin/env python3
"""Differential fuzzing against the oracle beyond what pytest runs (development aid; results quoted in DESIGN.md section 1):
    python tools/fuzz.py single SEED CASES   random n in [2^17, 1.3e6], six scalar kinds, fixed-base table AND plain layout vs oracle/cpu_ref.c
    python tools/fuzz.py batch  SEED CASES   random phases (2-23 columns: sparse, zero, constant, dense) at prover sizes: batch == single calls == oracle
    python tools/fuzz.py lookup SEED CASES   random lookup arguments (2^13 .. 2^16 rows, key widths 1 .. 200 bits, 1-9 per call): == oracle/poly_ref.py
    python tools/fuzz.py scans  SEED CASES   batched grand products (chained at a random row or not, in place or not, 1-40 columns of 1 .. 300 000
                                             rows) and batched kate divisions == the single-column calls; one column per case == oracle/poly_ref.py
    python tools/fuzz.py cosets SEED CASES   random (k, max degree, batch): every coset of coeff_to_coset == the residue class of rows of
                                             coeff_to_extended; coset_to_partial + combine_cosets == extended_to_coeff
"""
import sys
mode = sys.argv.pop(1) if len(sys.argv) > 1 else "single"
if mode == "single":
    import os, sys, time
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.domain import FR_MODULUS, fr_words
    from oracle import cpu_ref
    cpu_ref.build()
    lib = _lib.load()
    def rand_fr(n, seed):
        from halo2_experiments_amd.arithmetic import random_fr
        return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    NMAX = 1_300_000
    pool = h.g1_fixed_base_mul(rand_fr(NMAX, 1), cpu_ref.g1_generator())
    bad = 0
    t0 = time.time()
    for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        n = int(rng.integers(1 << 17, NMAX))
        off = int(rng.integers(0, NMAX - n + 1))
        bases = pool[off:off + n].contiguous()
        s = rand_fr(n, 100 + case)
        kind = int(rng.integers(0, 6))
        if kind == 1:
            s[torch.from_numpy(rng.random(n) < 0.9).cuda()] = 0
        elif kind == 2:
            s[:, 1:] = 0; s[:, 0] &= 0xFFFFFF
            s = h.linear_combination([s.contiguous()], np.stack([fr_words((1 << 256) % FR_MODULUS)]))
        elif kind == 3:
            s[::3] = s[0]
        elif kind == 4:
            s[int(rng.integers(0, n)):] = 0
        s = s.contiguous()
        want = cpu_ref.g1_to_affine(cpu_ref.best_multiexp(s.cpu().numpy().view(np.uint64), bases.cpu().numpy().view(np.uint64), 16))[0]
        for thr in (17, 0):
            lib.hm_set_fixed_base_threshold(thr)
            hd = h.register_bases(bases)
            got = h.best_multiexp(s, hd)
            h.release_bases(hd)
            ok = (not got[8:].any() and not want.any()) or np.array_equal(got[:8], want)
            if not ok:
                bad += 1
                print("MISMATCH", case, n, kind, thr, flush=True)
        if case % 10 == 9: print("case", case, "ok so far, bad =", bad, f"{time.time()-t0:.0f}s", flush=True)
    lib.hm_set_fixed_base_threshold(17)
    print("done, mismatches:", bad)
elif mode == "scans":
    import os, sys, random, time
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    import halo2_experiments_amd as h
    from halo2_experiments_amd.domain import fr_words
    from oracle import bn256_ref as o, poly_ref as pr
    def rand_fr(n, seed):
        from halo2_experiments_amd.arithmetic import random_fr
        return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)
    vals = lambda t: o.fr_from_array(t.cpu().numpy().view(np.uint64))
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    bad, t0 = 0, time.time()
    for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        n = rng.choice([1, 2, 3, 4, 5, rng.randrange(6, 600), rng.randrange(600, 70000), rng.randrange(70000, 300000)])
        count = rng.choice([1, 2, rng.randrange(3, 17), rng.randrange(17, 41)])
        if n * count > 4_000_000:
            count = max(1, 4_000_000 // n)
        cols = [rand_fr(n, 10_000 * case + j) for j in range(count)]
        if rng.random() < 0.3 and n > 2:
            cols[rng.randrange(count)][rng.randrange(n)] = 0                  # a zero factor: everything after it is zero
        start = rng.choice([1, 0, o.R - 1, rng.randrange(o.R)])
        u = rng.choice([None, 0, n - 1, rng.randrange(n)])
        inplace = rng.random() < 0.5
        work = [c.clone() for c in cols] if inplace else cols
        got = h.grand_product_batch(work, fr_words(start), chain_row=u, outs=work if inplace else None)
        st = fr_words(start)
        for j in range(count):
            one = h.grand_product(cols[j], st)
            if not bool((got[j] == one).all()):
                bad += 1
                print("MISMATCH product", case, n, count, u, j, flush=True)
                break
            if u is not None:
                st = one[u].cpu().numpy().view(np.uint64)
        if n <= 70000:
            j = rng.randrange(count)
            s_j = start if u is None or j == 0 else vals(got[j - 1][u:u + 1])[0]
            if vals(got[j]) != pr.grand_product(vals(cols[j]), s_j):
                bad += 1
                print("MISMATCH product vs oracle", case, n, count, u, j, flush=True)
        zs = [rng.choice([0, 1, o.R - 1, rng.randrange(o.R)]) for _ in range(count)]
        qs = h.kate_division_batch(cols, [fr_words(z) for z in zs])
        for j in range(count):
            if not bool((qs[j] == h.kate_division(cols[j], fr_words(zs[j]))).all()):
                bad += 1
                print("MISMATCH division", case, n, count, j, flush=True)
                break
        if 1 < n <= 70000:
            j = rng.randrange(count)
            if vals(qs[j]) != pr.kate_division(vals(cols[j]), zs[j]):
                bad += 1
                print("MISMATCH division vs oracle", case, n, count, j, flush=True)
        if case % 10 == 9: print("case", case, "bad =", bad, f"{time.time()-t0:.0f}s", flush=True)
    print("done, mismatches:", bad)
elif mode == "cosets":
    import os, sys, random, time
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    from halo2_experiments_amd.domain import EvaluationDomain
    def rand_fr(n, seed):
        from halo2_experiments_amd.arithmetic import random_fr
        return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    bad, t0 = 0, time.time()
    for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
        k = rng.randrange(1, 19)
        j = rng.randrange(3, 18)
        batch = rng.randrange(1, 6) if k <= 16 else rng.randrange(1, 3)
        dom = EvaluationDomain(j, k)
        if dom.extended_k > 24:
            continue
        n, e = dom.n, dom.num_cosets()
        a = rand_fr(batch * n, 77 * case).reshape(batch, n, 4)
        internal = rng.random() < 0.5
        ext = dom.coeff_to_extended(a, internal=internal)
        for c in rng.sample(range(e), min(e, 4)):
            if not bool((dom.coeff_to_coset(a, c, internal=internal) == ext[:, c::e]).all()):
                bad += 1
                print("MISMATCH coset", case, k, j, batch, c, internal, flush=True)
        sub = rng.sample(range(e), rng.randrange(1, min(e, 16) + 1))
        many = dom.coeff_to_cosets(a, sub, internal=internal)                     # several cosets in one launch chain
        for i, c in enumerate(sub):
            if not bool((many[:, i] == ext[:, c::e]).all()):
                bad += 1
                print("MISMATCH cosets (fused)", case, k, j, batch, sub, c, internal, flush=True)
                break
        hx = rand_fr(e * n, 77 * case + 1)
        parts = [dom.coset_to_partial(hx[c::e].contiguous(), c) for c in range(e)]
        fused = torch.stack([hx[c::e] for c in sub]).contiguous()
        dom.cosets_to_partials(fused, sub)
        if any(not bool((fused[i] == parts[c]).all()) for i, c in enumerate(sub)):
            bad += 1
            print("MISMATCH partials (fused)", case, k, j, sub, flush=True)
        whole = hx.clone()
        dom.extended_to_coeff(whole)
        if not bool((dom.combine_cosets(parts, pieces=e) == whole).all()):
            bad += 1
            print("MISMATCH recombination", case, k, j, flush=True)
        if case % 10 == 9: print("case", case, "bad =", bad, f"{time.time()-t0:.0f}s", flush=True)
    print("done, mismatches:", bad)
elif mode == "lookup":
    import os, sys, random, time
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    from oracle import poly_ref as pr, bn256_ref as o
    R = pr.R
    def to_gpu(vals): return torch.from_numpy(o.fr_array(vals).view(np.int64)).cuda()
    def from_gpu(t): return o.fr_from_array(t.cpu().numpy().view(np.uint64))
    rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    bad = 0
    t0 = time.time()
    for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
        k = rng.choice([13, 14, 15, 16])
        n = 1 << k
        rows = n - rng.choice([1, 7, 100, 1000])
        pairs = []
        for _ in range(rng.choice([1, 1, 2, 5, 9])):
            bits = rng.choice([1, 3, 8, 12, 16, 19, 20, 21, 24, 64, 200])
            span = rng.choice([1, 2, 7, 256, 5000, rows])
            base = [rng.randrange(1 << bits) % R for _ in range(span)]
            table = [base[i % span] for i in range(n)]
            mode = rng.random()
            if mode < 0.3: inp = [table[0]] * n
            elif mode < 0.6: inp = [table[rng.randrange(rows)] if rng.random() < 0.1 else table[0] for _ in range(n)]
            else: inp = [table[rng.randrange(rows)] for _ in range(n)]
            pairs.append((inp, table))
        outs = h.permute_expression_pairs([to_gpu(p[0]) for p in pairs], [to_gpu(p[1]) for p in pairs], rows, blinding_seed=case)
        for (inp, table), (a, s) in zip(pairs, outs):
            wa, ws = pr.permute_expression_pair(inp, table, rows)
            if from_gpu(a[:rows]) != wa or from_gpu(s[:rows]) != ws:
                bad += 1
                print("MISMATCH", case, k, rows, flush=True)
        if case % 5 == 4: print("case", case, "bad =", bad, f"{time.time()-t0:.0f}s", flush=True)
    print("done, mismatches:", bad)
    
else:
    import os, sys, time
    sys.path.insert(0, os.getcwd())
    import numpy as np, torch
    import halo2_experiments_amd as h
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import best_multiexp_batch
    from halo2_experiments_amd.replay import _sparse_column
    from oracle import cpu_ref
    cpu_ref.build()
    def rand_fr(n, seed):
        from halo2_experiments_amd.arithmetic import random_fr
        return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    dev = torch.device("cuda", 0)
    bad = 0
    t0 = time.time()
    for case in range(int(sys.argv[2]) if len(sys.argv) > 2 else 30):
        n = int(rng.choice([1 << 17, (1 << 17) + 5, 1 << 18, (1 << 18) - 3, 200_001, 1 << 16, 1 << 15]))
        bases = h.g1_fixed_base_mul(rand_fr(n, 5000 + case), cpu_ref.g1_generator())
        hd = h.register_bases(bases)
        count = int(rng.integers(2, 24))
        cols = []
        for i in range(count):
            k = int(rng.integers(0, 6))
            if k == 0: c = _sparse_column(n, int(rng.integers(1, 3000)), 9000 + 31 * case + i, dev)
            elif k == 1: c = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
            elif k == 2: c = rand_fr(1, 7000 + i).expand(n, 4).contiguous()
            else: c = rand_fr(n, 8000 + 17 * case + i)
            cols.append(c)
        single = np.stack([h.best_multiexp(c, hd) for c in cols])
        got = best_multiexp_batch(cols, hd)
        got_h = best_multiexp_batch([c.cpu().numpy().view(np.uint64) for c in cols], hd) if case % 3 == 0 else got
        j = int(rng.integers(0, count))
        want = cpu_ref.g1_to_affine(cpu_ref.best_multiexp(cols[j].cpu().numpy().view(np.uint64), bases.cpu().numpy().view(np.uint64), 16))[0]
        okj = (not single[j][8:].any() and not want.any()) or np.array_equal(single[j][:8], want)
        if not (np.array_equal(got, single) and np.array_equal(got_h, single) and okj):
            bad += 1
            print("MISMATCH", case, n, count, flush=True)
        h.release_bases(hd)
        if case % 10 == 9: print("case", case, "bad =", bad, f"{time.time()-t0:.0f}s", flush=True)
    print("done, mismatches:", bad)
