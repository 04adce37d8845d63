#!/usr/bin/env python3
"""Ad-hoc GPU validation + timing (development aid; the judged tests live in tests/)."""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from oracle import bn256_ref as o, cpu_ref as c

def rand_fr_gpu(n, seed):
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    x = torch.randint(-(2**63), 2**63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=g)
    x[:, 3] &= 0x0FFFFFFFFFFFFFFF   # < 2^252 < r : a valid (Montgomery-form) Fr
    return x

def main():
    quick = "--quick" in sys.argv
    print(_lib.load().hm_version().decode(), "devices", _lib.load().hm_device_count()); sys.stdout.flush()
    T = min(16, c.default_threads())
    # ---- NTT parity
    for k in [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 16, 17, 18]:
        a = o.fr_array(o.rand_scalars(1 << k, 100 + k)) if k <= 12 else rand_fr_gpu(1 << k, k).cpu().numpy().view(np.uint64)
        w = o.fr_array([o.fr_omega(k)])[0]
        exp = c.best_fft(a, w, k, T)
        got = a.copy(); h.best_fft(got, w, k)
        ok = np.array_equal(got, exp)
        print("ntt log_n", k, "OK" if ok else "MISMATCH"); sys.stdout.flush()
        if not ok:
            bad = np.nonzero((got != exp).any(axis=1))[0]
            print("  first bad rows", bad[:8], "count", len(bad)); return 1
    # ---- MSM parity
    gen = o.g1_affine_array([o.G1_GEN])[0]
    for n, kind in [(1, "uniform"), (2, "edge"), (3, "uniform"), (31, "edge"), (33, "uniform"), (255, "prover"), (1024, "uniform"),
                    (1024, "one"), (1000, "rminus1"), (4096, "small"), (1 << 14, "uniform"), (1 << 16, "uniform")]:
        if n <= 4096:
            s = o.fr_array(o.rand_scalars(n, n + 7, kind))
        else:
            s = rand_fr_gpu(n, n).cpu().numpy().view(np.uint64)
        ks = rand_fr_gpu(n, 3 * n + 1)
        bases = h.g1_fixed_base_mul(ks, gen).cpu().numpy().view(np.uint64)
        if n >= 3: bases[2] = 0                       # an identity base
        if n >= 33: bases[5] = bases[6]               # duplicate points
        exp = c.g1_to_affine(c.best_multiexp(s, bases, T))[0]
        got = h.best_multiexp(s, bases)
        ok = np.array_equal(got[:8], exp) if got[8:].any() else not exp.any()
        print("msm n", n, kind, "OK" if ok else "MISMATCH", h.msm_stats()); sys.stdout.flush()
        if not ok: return 1
    # fixed-base check against the oracle
    ks = rand_fr_gpu(8, 5); pts = h.g1_fixed_base_mul(ks, gen).cpu().numpy().view(np.uint64)
    for i in range(8):
        assert np.array_equal(pts[i], c.g1_mul(ks[i].cpu().numpy().view(np.uint64), gen)), "fixed-base mismatch"
    print("fixed-base OK"); sys.stdout.flush()
    # ---- timing
    for k in ([20] if quick else [20, 22, 24]):
        a = rand_fr_gpu(1 << k, k); w = o.fr_array([o.fr_omega(k)])[0]
        h.best_fft(a, w, k); torch.cuda.synchronize()
        t = time.time(); reps = 5
        for _ in range(reps): h.best_fft(a, w, k)
        torch.cuda.synchronize(); dt = (time.time() - t) / reps
        print(f"ntt 2^{k}: {dt*1e3:.3f} ms  {(64<<k)/dt/1e9:.1f} GB/s algorithmic"); sys.stdout.flush()
    for k in ([18] if quick else [18, 20, 22, 24]):
        n = 1 << k
        ks = rand_fr_gpu(n, 11); t = time.time(); bases = h.g1_fixed_base_mul(ks, gen); torch.cuda.synchronize()
        print(f"fixed-base 2^{k}: {(time.time()-t)*1e3:.1f} ms"); sys.stdout.flush()
        hd = h.register_bases(bases); s = rand_fr_gpu(n, 12)
        h.best_multiexp(s, hd); t = time.time(); reps = 3
        for _ in range(reps): r = h.best_multiexp(s, hd)
        dt = (time.time() - t) / reps
        print(f"msm 2^{k}: {dt*1e3:.2f} ms  {n/dt/1e6:.1f} Mpts/s", h.msm_stats()); sys.stdout.flush()
        h.release_bases(hd)
    return 0

if __name__ == "__main__":
    sys.exit(main())
