"""Big-integer oracle for the BN256 MSM / Fr-NTT hot path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference repository holds no golden vectors, known-answer tests or
fixtures for MSM / FFT (its only boundary assertion is
``verify_proof(..).is_ok()`` at /root/reference/src/circuits/utils.rs:56-63 with an OsRng SRS),
and the code that implements the path (``halo2_proofs`` tag v2023_02_02, /root/reference/Cargo.toml:10,
and its ``halo2curves`` dependency) is absent from this container and cannot be built (no Rust).
This file therefore restates the *mathematics* the path computes (which is canonical: a group
element after affine normalisation, a vector of fully reduced field elements) and the published
algorithm of ``best_multiexp`` / ``best_fft`` as summarised in SURVEY.md §3.3 / §3.4.  It is pinned
only by (i) self-consistency between independent formulations (naive double-and-add vs. Pippenger
restatement; O(n^2) DFT vs. recursive butterflies) and (ii) public alt_bn128 known answers
(EIP-196 vectors) in tests/test_oracle.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.

Memory layout restated from SURVEY.md §8a (halo2curves bn256):
  Fr / Fq  = 4 little-endian u64 limbs, Montgomery form  (value * 2^256 mod modulus), fully reduced
  G1Affine = {x: Fq, y: Fq} = 8 u64, identity encoded as (0, 0)
  G1       = Jacobian {x, y, z: Fq} = 12 u64, identity has z = 0, affine = (X/Z^2, Y/Z^3)
"""
from __future__ import annotations

import math
import random
from typing import Iterable, List, Optional, Sequence, Tuple

import numpy as np

# ----------------------------------------------------------------------------------------------
# constants
# ----------------------------------------------------------------------------------------------
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq modulus
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr modulus (group order)
MONT = 1 << 256
B_COEFF = 3  # y^2 = x^3 + 3
G1_GEN = (1, 2)
FR_S = 28
FR_GENERATOR = 7
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (R - 1) >> FR_S, R)  # order exactly 2^28
FR_ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
FQ_INV64 = (-pow(P, -1, 1 << 64)) % (1 << 64)
FR_INV64 = (-pow(R, -1, 1 << 64)) % (1 << 64)

Affine = Optional[Tuple[int, int]]  # None = identity


# ----------------------------------------------------------------------------------------------
# limb encodings
# ----------------------------------------------------------------------------------------------
def to_limbs(v: int, n: int = 4) -> List[int]:
    return [(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(n)]


def from_limbs(limbs: Sequence[int]) -> int:
    out = 0
    for i, l in enumerate(limbs):
        out |= int(l) << (64 * i)
    return out


def fr_to_mont(v: int) -> int:
    return (v * MONT) % R


def fr_from_mont(v: int) -> int:
    return (v * pow(MONT, -1, R)) % R


def fq_to_mont(v: int) -> int:
    return (v * MONT) % P


def fq_from_mont(v: int) -> int:
    return (v * pow(MONT, -1, P)) % P


_RINV_R = pow(MONT, -1, R)
_RINV_P = pow(MONT, -1, P)


def fr_array(values: Iterable[int]) -> np.ndarray:
    """Canonical ints -> (n, 4) uint64 array of Montgomery-form limbs."""
    vals = list(values)
    out = np.zeros((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        out[i] = to_limbs((v % R) * MONT % R)
    return out


def fr_from_array(arr: np.ndarray) -> List[int]:
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 4)
    return [from_limbs(row) * _RINV_R % R for row in arr]


def g1_affine_array(points: Iterable[Affine]) -> np.ndarray:
    """Affine points (canonical ints or None) -> (n, 8) uint64 Montgomery array; identity = (0,0)."""
    pts = list(points)
    out = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, pt in enumerate(pts):
        if pt is None:
            continue
        x, y = pt
        out[i, :4] = to_limbs(x * MONT % P)
        out[i, 4:] = to_limbs(y * MONT % P)
    return out


def g1_affine_from_array(arr: np.ndarray) -> List[Affine]:
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 8)
    out: List[Affine] = []
    for row in arr:
        xm, ym = from_limbs(row[:4]), from_limbs(row[4:])
        if xm == 0 and ym == 0:
            out.append(None)
        else:
            out.append((xm * _RINV_P % P, ym * _RINV_P % P))
    return out


def g1_jacobian_from_array(arr: np.ndarray) -> List[Affine]:
    """(n, 12) uint64 Montgomery Jacobian -> canonical affine (None for z = 0)."""
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1, 12)
    out: List[Affine] = []
    for row in arr:
        x = from_limbs(row[:4]) * _RINV_P % P
        y = from_limbs(row[4:8]) * _RINV_P % P
        z = from_limbs(row[8:]) * _RINV_P % P
        if z == 0:
            out.append(None)
        else:
            zi = pow(z, -1, P)
            out.append((x * zi * zi % P, y * zi * zi * zi % P))
    return out


# ----------------------------------------------------------------------------------------------
# affine curve arithmetic (the mathematics; no implementation detail of upstream involved)
# ----------------------------------------------------------------------------------------------
def is_on_curve(pt: Affine) -> bool:
    if pt is None:
        return True
    x, y = pt
    return (y * y - x * x * x - B_COEFF) % P == 0


def g1_neg(pt: Affine) -> Affine:
    if pt is None:
        return None
    return (pt[0], (-pt[1]) % P)


def g1_add(a: Affine, b: Affine) -> Affine:
    if a is None:
        return b
    if b is None:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return None
        lam = 3 * x1 * x1 * pow(2 * y1, -1, P) % P
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def g1_mul(k: int, pt: Affine) -> Affine:
    k %= R
    acc: Affine = None
    add = pt
    while k:
        if k & 1:
            acc = g1_add(acc, add)
        add = g1_add(add, add)
        k >>= 1
    return acc


def msm_naive(scalars: Sequence[int], bases: Sequence[Affine]) -> Affine:
    """sum_i scalars[i] * bases[i] by double-and-add: the definition best_multiexp must meet."""
    assert len(scalars) == len(bases)
    acc: Affine = None
    for s, b in zip(scalars, bases):
        acc = g1_add(acc, g1_mul(s, b))
    return acc


# ----------------------------------------------------------------------------------------------
# restatement of halo2_proofs::arithmetic::best_multiexp, for cross-checking
# the C restatement's control flow on small inputs
# ----------------------------------------------------------------------------------------------
def _window_size(m: int) -> int:
    if m < 4:
        return 1
    if m < 32:
        return 3
    return int(math.ceil(math.log(m)))


def multiexp_serial(coeffs: Sequence[int], bases: Sequence[Affine], acc: Affine) -> Affine:
    """One chunk of best_multiexp: unsigned c-bit windows, MSB first, 2^c - 1 buckets,
    zero-digit skip, running-sum bucket reduction."""
    c = _window_size(len(coeffs))
    segments = 256 // c + 1
    for seg in reversed(range(segments)):
        for _ in range(c):
            acc = g1_add(acc, acc)
        buckets: List[Affine] = [None] * ((1 << c) - 1)
        for s, b in zip(coeffs, bases):
            d = (s >> (seg * c)) & ((1 << c) - 1)
            if d != 0:
                buckets[d - 1] = g1_add(buckets[d - 1], b)
        running: Affine = None
        for bk in reversed(buckets):
            running = g1_add(bk, running)
            acc = g1_add(acc, running)
    return acc


def best_multiexp(coeffs: Sequence[int], bases: Sequence[Affine], threads: int = 8) -> Affine:
    assert len(coeffs) == len(bases)
    n = len(coeffs)
    if n > threads:
        chunk = n // threads
        parts = []
        for lo in range(0, n, chunk):
            parts.append(multiexp_serial(coeffs[lo:lo + chunk], bases[lo:lo + chunk], None))
        acc: Affine = None
        for p_ in parts:
            acc = g1_add(acc, p_)
        return acc
    return multiexp_serial(coeffs, bases, None)


# ----------------------------------------------------------------------------------------------
# NTT: definition (O(n^2)) and restatement of best_fft
# ----------------------------------------------------------------------------------------------
def fr_omega(log_n: int) -> int:
    """Primitive 2^log_n-th root of unity as EvaluationDomain derives it."""
    assert 0 <= log_n <= FR_S
    return pow(FR_ROOT_OF_UNITY, 1 << (FR_S - log_n), R)


def dft_naive(a: Sequence[int], omega: int) -> List[int]:
    n = len(a)
    out = []
    for j in range(n):
        wj = pow(omega, j, R)
        acc, w = 0, 1
        for i in range(n):
            acc = (acc + a[i] * w) % R
            w = w * wj % R
        out.append(acc)
    return out


def _bitrev(k: int, bits: int) -> int:
    out = 0
    for _ in range(bits):
        out = (out << 1) | (k & 1)
        k >>= 1
    return out


def best_fft(a: List[int], omega: int, log_n: int) -> None:
    """In-place, natural order in/out, unscaled: a'[j] = sum_i a[i] omega^(ij).
    Bit-reversal, n/2 precomputed twiddles, recursive radix-2 DIT butterflies."""
    n = 1 << log_n
    assert len(a) == n
    for k in range(n):
        rk = _bitrev(k, log_n)
        if k < rk:
            a[k], a[rk] = a[rk], a[k]
    tw = [1] * max(n // 2, 1)
    for i in range(1, n // 2):
        tw[i] = tw[i - 1] * omega % R

    def rec(lo: int, m: int, tc: int) -> None:
        if m == 1:
            return
        if m == 2:
            t = a[lo + 1]
            a[lo + 1] = (a[lo] - t) % R
            a[lo] = (a[lo] + t) % R
            return
        h = m // 2
        rec(lo, h, tc * 2)
        rec(lo + h, h, tc * 2)
        for i in range(h):
            t = a[lo + h + i] * tw[i * tc] % R
            a[lo + h + i] = (a[lo + i] - t) % R
            a[lo + i] = (a[lo + i] + t) % R

    rec(0, n, 1)


def ntt_fast(a: Sequence[int], omega: int) -> List[int]:
    """Iterative NTT used to make mid-size expected values quickly (same definition)."""
    n = len(a)
    log_n = n.bit_length() - 1
    out = list(a)
    best_fft(out, omega, log_n)
    return out


def poly_eval(coeffs: Sequence[int], x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


# ----------------------------------------------------------------------------------------------
# seeded input generators shared by tests (deterministic, independent of numpy's RNG version)
# ----------------------------------------------------------------------------------------------
def rand_scalars(n: int, seed: int, kind: str = "uniform") -> List[int]:
    rng = random.Random(seed)
    if kind == "uniform":
        return [rng.randrange(R) for _ in range(n)]
    if kind == "zero":
        return [0] * n
    if kind == "one":
        return [1] * n
    if kind == "rminus1":
        return [R - 1] * n
    if kind == "prover":  # 90 % zero, 5 % < 2^16, 5 % uniform
        out = []
        for _ in range(n):
            u = rng.random()
            out.append(0 if u < 0.9 else rng.randrange(1 << 16) if u < 0.95 else rng.randrange(R))
        return out
    if kind == "small":
        return [rng.randrange(1 << 16) for _ in range(n)]
    if kind == "edge":
        pool = [0, 1, 2, R - 1, R - 2, (1 << 253), (1 << 128) - 1, (1 << 254) % R, 0xFFFF, 0x10000]
        return [pool[rng.randrange(len(pool))] for _ in range(n)]
    raise ValueError(kind)


def arith_bases(n: int, seed: int) -> Tuple[List[Affine], List[int]]:
    """P_i = [a + i*b]G by repeated affine addition; returns (points, discrete logs)."""
    rng = random.Random(seed)
    a0, b0 = rng.randrange(1, R), rng.randrange(1, R)
    step = g1_mul(b0, G1_GEN)
    cur = g1_mul(a0, G1_GEN)
    pts, logs = [], []
    for i in range(n):
        pts.append(cur)
        logs.append((a0 + i * b0) % R)
        cur = g1_add(cur, step)
    return pts, logs
