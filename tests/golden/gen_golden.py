#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the big-integer oracle (oracle/bn256_ref.py).

The reference repository holds no MSM/FFT vectors, so these fixtures are produced
by independent mathematics: naive double-and-add / known discrete logs for the MSM, the O(n^2)
DFT definition for small NTTs and the recursive restatement cross-checked against it for larger
ones.  Everything is stored in the reference's memory layout (little-endian u64 Montgomery limbs).
Run:  python tests/golden/gen_golden.py      (deterministic; ~1 minute)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import bn256_ref as o  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def fq_array(vals):
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = o.to_limbs(v % o.P * o.MONT % o.P)
    return out


def field_vectors():
    import random
    rng = random.Random(20230202)
    out = {}
    for name, mod, enc in (("fr", o.R, o.fr_array), ("fq", o.P, fq_array)):
        edge = [0, 1, 2, mod - 1, mod - 2, o.MONT % mod, (1 << 253) % mod, (1 << 128) - 1]
        a = edge + [rng.randrange(mod) for _ in range(56)]
        b = list(reversed(edge)) + [rng.randrange(mod) for _ in range(56)]
        out[f"{name}_a"], out[f"{name}_b"] = enc(a), enc(b)
        out[f"{name}_mul"] = enc([x * y % mod for x, y in zip(a, b)])
        out[f"{name}_add"] = enc([(x + y) % mod for x, y in zip(a, b)])
        out[f"{name}_sub"] = enc([(x - y) % mod for x, y in zip(a, b)])
        out[f"{name}_canon"] = np.array([o.to_limbs(x) for x in a], dtype=np.uint64)  # from_mont(a)
    np.savez_compressed(os.path.join(OUT, "field.npz"), **out)


def curve_vectors():
    import random
    rng = random.Random(196)
    ks = [1, 2, 3, 5, o.R - 1, o.R - 2] + [rng.randrange(o.R) for _ in range(10)]
    pts = [o.g1_mul(k, o.G1_GEN) for k in ks]
    pairs_a, pairs_b, sums = [], [], []
    for i in range(len(pts)):
        for j in (i, (i + 1) % len(pts), (i + 5) % len(pts)):
            pairs_a.append(pts[i]); pairs_b.append(pts[j]); sums.append(o.g1_add(pts[i], pts[j]))
        pairs_a.append(pts[i]); pairs_b.append(o.g1_neg(pts[i])); sums.append(None)
        pairs_a.append(pts[i]); pairs_b.append(None); sums.append(pts[i])
    np.savez_compressed(os.path.join(OUT, "curve.npz"), scalars=o.fr_array(ks), points=o.g1_affine_array(pts),
                        add_a=o.g1_affine_array(pairs_a), add_b=o.g1_affine_array(pairs_b), add_sum=o.g1_affine_array(sums))


def msm_vectors():
    out = {}
    names = []
    for n in (1, 2, 3, 31, 32, 33, 255, 1024):
        pts, logs = o.arith_bases(n, 1000 + n)
        for kind in ("uniform", "zero", "one", "rminus1", "prover", "edge", "small"):
            if n > 33 and kind in ("zero", "edge"):
                continue
            s = o.rand_scalars(n, 7 * n + len(kind), kind)
            exp = o.g1_mul(sum(a * b for a, b in zip(s, logs)) % o.R, o.G1_GEN)
            if n <= 33:
                assert exp == o.msm_naive(s, pts), (n, kind)
                assert exp == o.best_multiexp(s, pts, 8), (n, kind)
            key = f"n{n}_{kind}"
            names.append(key)
            out[key + "_s"], out[key + "_b"] = o.fr_array(s), o.g1_affine_array(pts)
            out[key + "_r"] = o.g1_affine_array([exp])[0]
    # structured cases: all points equal; P and -P mixed; identity bases; duplicate points
    n = 64
    pts, logs = o.arith_bases(n, 4242)
    s = o.rand_scalars(n, 4243)
    same = [pts[0]] * n
    out["same_s"], out["same_b"] = o.fr_array(s), o.g1_affine_array(same)
    out["same_r"] = o.g1_affine_array([o.g1_mul(sum(s) * logs[0] % o.R, o.G1_GEN)])[0]
    mixed = [pts[i // 2] if i % 2 == 0 else o.g1_neg(pts[i // 2]) for i in range(n)]
    mlogs = [logs[i // 2] if i % 2 == 0 else -logs[i // 2] for i in range(n)]
    out["pm_s"], out["pm_b"] = o.fr_array(s), o.g1_affine_array(mixed)
    out["pm_r"] = o.g1_affine_array([o.g1_mul(sum(a * b for a, b in zip(s, mlogs)) % o.R, o.G1_GEN)])[0]
    ones = [1] * n
    out["pmone_s"], out["pmone_b"] = o.fr_array(ones), o.g1_affine_array(mixed)
    out["pmone_r"] = o.g1_affine_array([None])[0]                      # sum of P - P pairs = identity
    withid = list(pts)
    ilogs = list(logs)
    for i in (0, 7, 63):
        withid[i] = None; ilogs[i] = 0
    out["ident_s"], out["ident_b"] = o.fr_array(s), o.g1_affine_array(withid)
    out["ident_r"] = o.g1_affine_array([o.g1_mul(sum(a * b for a, b in zip(s, ilogs)) % o.R, o.G1_GEN)])[0]
    names += ["same", "pm", "pmone", "ident"]
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "msm.npz"), **out)


def ntt_vectors():
    out = {}
    for k in range(0, 11):
        n = 1 << k
        w = o.fr_omega(k)
        v = o.rand_scalars(n, 500 + k)
        exp = o.ntt_fast(v, w)
        if k <= 6:
            assert exp == o.dft_naive(v, w)
        out[f"k{k}_in"], out[f"k{k}_out"], out[f"k{k}_omega"] = o.fr_array(v), o.fr_array(exp), o.fr_array([w])[0]
        winv = pow(w, -1, o.R)
        out[f"k{k}_omega_inv"] = o.fr_array([winv])[0]
        out[f"k{k}_ninv"] = o.fr_array([pow(n, -1, o.R)])[0]
    # delta -> all ones ; all ones -> n * delta
    k = 5
    w = o.fr_omega(k)
    out["delta_in"] = o.fr_array([1] + [0] * 31); out["delta_out"] = o.fr_array([1] * 32)
    out["ones_in"] = o.fr_array([1] * 32); out["ones_out"] = o.fr_array([32] + [0] * 31)
    # another primitive 32nd root (omega^3): best_fft must not assume omega = ROOT^(2^(28-k))
    v = o.rand_scalars(32, 77)
    w3 = pow(w, 3, o.R)
    out["w3_in"], out["w3_out"], out["w3_omega"] = o.fr_array(v), o.fr_array(o.dft_naive(v, w3)), o.fr_array([w3])[0]
    np.savez_compressed(os.path.join(OUT, "ntt.npz"), **out)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    field_vectors(); curve_vectors(); ntt_vectors(); msm_vectors()
    print("wrote", sorted(os.listdir(OUT)))
