"""Host-side mirror of ``halo2_proofs::poly::kzg::commitment::ParamsKZG<Bn256>`` (upstream
``halo2_proofs/src/poly/kzg/commitment.rs`` at the tag pinned by /root/reference/Cargo.toml:10; the reference
calls ``ParamsKZG::<Bn256>::setup(k, OsRng)`` at /root/reference/src/circuits/utils.rs:28 on every run).

    setup(k, s)            g[i] = [s^i]G1, g_lagrange[i] = [L_i(s)]G1, g2 = G2, s_g2 = [s]G2
    commit(poly)           best_multiexp(poly, g[..len])           (coefficient form)
    commit_lagrange(poly)  best_multiexp(poly, g_lagrange[..len])  (evaluation form)
    write(f) / read(f)     the SRS on disk, so that it is loaded instead of regenerated every run

All G1 work runs on the GPU through the C ABI: the scalar ladder (``hm_fr_powers_dev``), the Lagrange scalars
(one scaled inverse NTT of the ladder: L_i(s) = n^-1 sum_j s^j omega^(-ij)), the 2 n fixed-base multiplications
(``hm_g1_fixed_base_mul_dev``), and the two base sets stay registered on the device.  The single G2 scalar
multiplication of ``s_g2`` is host integer arithmetic (one point, a few milliseconds).

On-disk layout (little-endian, exactly the bytes the Rust types hold -- Montgomery limbs):
    u32 k | n x 64 B g | n x 64 B g_lagrange | 128 B g2 | 128 B s_g2          (G2Affine = x.c0, x.c1, y.c0, y.c1)
which is the field order of upstream's ``ParamsKZG::write`` with raw (uncompressed Montgomery) points
[UPSTREAM-RECALLED: later PSE releases call it ``SerdeFormat::RawBytesUnchecked``; the pinned tag's own
``write`` compresses points -- a Rust-side loader converts once].
"""
from __future__ import annotations

import ctypes
import struct
from typing import BinaryIO, Optional

import numpy as np

from . import _lib
from .arithmetic import (FQ_MODULUS, G1_GENERATOR, BasesHandle, _ptr, _stream_ptr, best_multiexp, best_multiexp_submit,
                         best_multiexp_wait, g1_fixed_base_mul, register_bases, release_bases)
from .domain import FR_MODULUS, EvaluationDomain, fr_words

_P = FQ_MODULUS
# bn256::G2Affine::generator() (the alt_bn128 G2 generator of EIP-197), x = x0 + x1 u, y = y0 + y1 u, u^2 = -1
G2_GENERATOR = (
    (0x1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED,
     0x198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2),
    (0x12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA,
     0x090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B),
)


def _fq2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % _P, (a[0] * b[1] + a[1] * b[0]) % _P)


def _fq2_sub(a, b):
    return ((a[0] - b[0]) % _P, (a[1] - b[1]) % _P)


def _fq2_inv(a):
    d = pow(a[0] * a[0] + a[1] * a[1], -1, _P)
    return (a[0] * d % _P, -a[1] * d % _P)


def _g2_add(p, q):
    """Affine addition on y^2 = x^3 + 3 / (9 + u) over Fq2 (None = identity)."""
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if p[1] != q[1] or p[1] == (0, 0):
            return None
        three_x2 = _fq2_mul((3, 0), _fq2_mul(p[0], p[0]))
        lam = _fq2_mul(three_x2, _fq2_inv(_fq2_mul((2, 0), p[1])))
    else:
        lam = _fq2_mul(_fq2_sub(q[1], p[1]), _fq2_inv(_fq2_sub(q[0], p[0])))
    x3 = _fq2_sub(_fq2_sub(_fq2_mul(lam, lam), p[0]), q[0])
    return (x3, _fq2_sub(_fq2_mul(lam, _fq2_sub(p[0], x3)), p[1]))


def g2_mul(k: int, p=G2_GENERATOR):
    acc = None
    for bit in bin(k % FR_MODULUS)[2:]:
        acc = _g2_add(acc, acc)
        if bit == "1":
            acc = _g2_add(acc, p)
    return acc


def _fq_mont_bytes(v: int) -> bytes:
    return (v % _P * (1 << 256) % _P).to_bytes(32, "little")


def g2_bytes(p) -> bytes:
    """G2Affine as the 128 bytes Rust holds: x.c0, x.c1, y.c0, y.c1 (the identity is all-zero)."""
    if p is None:
        return bytes(128)
    return b"".join(_fq_mont_bytes(c) for c in (p[0][0], p[0][1], p[1][0], p[1][1]))


class ParamsKZG:
    def __init__(self, k: int, g, g_lagrange, g2: bytes, s_g2: bytes, precompute: bool = False):
        """``g`` / ``g_lagrange``: (n, 8) GPU tensors or numpy arrays of affine Montgomery words."""
        self.k, self.n = k, 1 << k
        self.g2, self.s_g2 = g2, s_g2
        self._g_h = register_bases(g, precompute=precompute)
        self._gl_h = register_bases(g_lagrange, precompute=precompute)
        if len(self._g_h) != self.n or len(self._gl_h) != self.n:
            self.release()
            raise ValueError("ParamsKZG: g and g_lagrange must hold 2^k points")

    # -- ParamsKZG::setup / unsafe_setup_with_s --------------------------------------------------
    @classmethod
    def setup(cls, k: int, s: int, device=None, precompute: bool = False, keep_points: bool = False) -> "ParamsKZG":
        """The reference draws s from OsRng (utils.rs:28); here the caller provides it (tests know it)."""
        import torch

        device = device or torch.device("cuda", torch.cuda.current_device())
        n = 1 << k
        lib = _lib.load()
        ladder = torch.empty((n, 4), dtype=torch.int64, device=device)
        _lib.check(lib.hm_fr_powers_dev(ctypes.c_void_p(ladder.data_ptr()), n, _ptr(fr_words(s)), ctypes.c_void_p(_stream_ptr(ladder))))
        g = g1_fixed_base_mul(ladder, G1_GENERATOR)
        lag = EvaluationDomain(2, k).lagrange_to_coeff(ladder)          # in place: n^-1 * NTT_{omega^-1}(ladder) = L_i(s)
        g_lagrange = g1_fixed_base_mul(lag, G1_GENERATOR)
        params = cls(k, g, g_lagrange, g2_bytes(G2_GENERATOR), g2_bytes(g2_mul(s)), precompute=precompute)
        if keep_points:
            params.g_points, params.g_lagrange_points = g, g_lagrange
        return params

    def release(self) -> None:
        for name in ("_g_h", "_gl_h"):
            hd = getattr(self, name, None)
            if hd is not None:
                release_bases(hd)
                setattr(self, name, None)

    @property
    def g_handle(self) -> BasesHandle:
        return self._g_h

    @property
    def g_lagrange_handle(self) -> BasesHandle:
        return self._gl_h

    # -- commit / commit_lagrange ------------------------------------------------------------------
    def commit(self, poly) -> np.ndarray:
        return best_multiexp(poly, self._g_h)

    def commit_lagrange(self, poly) -> np.ndarray:
        return best_multiexp(poly, self._gl_h)

    def commit_submit(self, poly) -> int:
        return best_multiexp_submit(poly, self._g_h)

    def commit_lagrange_submit(self, poly) -> int:
        return best_multiexp_submit(poly, self._gl_h)

    commit_wait = staticmethod(best_multiexp_wait)

    # -- write / read --------------------------------------------------------------------------------
    @staticmethod
    def write_points(f: BinaryIO, k: int, g: np.ndarray, g_lagrange: np.ndarray, g2: bytes, s_g2: bytes) -> None:
        n = 1 << k
        g = np.ascontiguousarray(g, dtype=np.uint64).reshape(n, 8)
        gl = np.ascontiguousarray(g_lagrange, dtype=np.uint64).reshape(n, 8)
        f.write(struct.pack("<I", k))
        f.write(g.tobytes())
        f.write(gl.tobytes())
        f.write(g2)
        f.write(s_g2)

    def write(self, f: BinaryIO) -> None:
        """Needs the affine points (setup(..., keep_points=True) or read())."""
        if not hasattr(self, "g_points"):
            raise ValueError("ParamsKZG.write: the affine points were not kept (setup(..., keep_points=True))")
        to_np = lambda t: t.cpu().numpy().view(np.uint64) if hasattr(t, "cpu") else np.asarray(t)
        self.write_points(f, self.k, to_np(self.g_points), to_np(self.g_lagrange_points), self.g2, self.s_g2)

    @classmethod
    def read(cls, f: BinaryIO, precompute: bool = False) -> "ParamsKZG":
        head = f.read(4)
        if len(head) != 4:
            raise ValueError("ParamsKZG.read: truncated header")
        (k,) = struct.unpack("<I", head)
        if k > 28:
            raise ValueError("ParamsKZG.read: k out of range")
        n = 1 << k
        raw = f.read(n * 128 + 256)
        if len(raw) != n * 128 + 256:
            raise ValueError("ParamsKZG.read: truncated file")
        pts = np.frombuffer(raw, dtype=np.uint64, count=n * 16).reshape(2, n, 8)
        params = cls(k, pts[0], pts[1], raw[n * 128: n * 128 + 128], raw[n * 128 + 128:], precompute=precompute)
        params.g_points, params.g_lagrange_points = pts[0], pts[1]
        return params
