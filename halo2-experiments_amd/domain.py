"""Host-side mirror of ``halo2_proofs::poly::EvaluationDomain`` for BN256 Fr -- the steps either
side of every ``best_fft`` call in ``create_proof`` (upstream
``halo2_proofs/src/poly/domain.rs`` at the tag pinned by /root/reference/Cargo.toml:10):

    new(j, k)             extended_k = smallest e with 2^e >= 2^k * (j - 1); omega, extended_omega,
                          g_coset = ZETA, ifft divisors
    lagrange_to_coeff(a)  ifft(a, omega_inv, k, 1/n)
    coeff_to_extended(a)  zero-pad to 2^extended_k, a[i] *= {1, zeta, zeta^2}[i % 3], best_fft(extended_omega)
    extended_to_coeff(a)  ifft(extended), a[i] *= {1, zeta^-1, zeta^-2}[i % 3], truncate to n*(j-1)

The scale-by-divisor and the coset shifts (forward and inverse) are fused into the last / first NTT
pass on the GPU.
Arrays are GPU tensors (int64 views of 4-limb Montgomery words); constants are derived here with
Python integers from the field's definition.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import _lib
from .arithmetic import _np, _ptr, _stream_ptr, _tensor_rows

FR_MODULUS = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
FR_S = 28
FR_GENERATOR = 7
FR_ROOT_OF_UNITY = pow(FR_GENERATOR, (FR_MODULUS - 1) >> FR_S, FR_MODULUS)
FR_ZETA = 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23
_MONT = 1 << 256


def fr_words(v: int) -> np.ndarray:
    """Canonical integer -> 4 Montgomery limbs (the bytes Rust's ``Fr`` holds)."""
    m = (v % FR_MODULUS) * _MONT % FR_MODULUS
    return np.array([(m >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def _vandermonde_inverse(nodes, r):
    """Inverse of V[a][t] = nodes[a]^t mod r (q x q, q <= a few dozen) by Gauss-Jordan on Python integers."""
    q = len(nodes)
    m = [[pow(x, t, r) for t in range(q)] + [1 if a == b else 0 for b in range(q)] for a, x in enumerate(nodes)]
    for col in range(q):
        piv = next(row for row in range(col, q) if m[row][col] % r)
        m[col], m[piv] = m[piv], m[col]
        inv = pow(m[col][col], -1, r)
        m[col] = [v * inv % r for v in m[col]]
        for row in range(q):
            if row != col and m[row][col]:
                f = m[row][col]
                m[row] = [(v - f * w) % r for v, w in zip(m[row], m[col])]
    return [row[q:] for row in m]


class EvaluationDomain:
    def __init__(self, j: int, k: int):
        if j < 2:
            raise ValueError("EvaluationDomain: j (max degree) must be >= 2")
        self.k = k
        self.n = 1 << k
        self.quotient_poly_degree = j - 1
        extended_k = k
        while (1 << extended_k) < self.n * self.quotient_poly_degree:
            extended_k += 1
        if extended_k > FR_S:
            raise ValueError("EvaluationDomain: extended_k exceeds the field's two-adicity")
        self.extended_k = extended_k
        r = FR_MODULUS
        self.extended_omega = pow(FR_ROOT_OF_UNITY, 1 << (FR_S - extended_k), r)
        self.extended_omega_inv = pow(self.extended_omega, -1, r)
        self.omega = pow(self.extended_omega, 1 << (extended_k - k), r)
        self.omega_inv = pow(self.omega, -1, r)
        self.g_coset = FR_ZETA
        self.g_coset_inv = FR_ZETA * FR_ZETA % r
        self.ifft_divisor = pow(self.n, -1, r)
        self.extended_ifft_divisor = pow(1 << extended_k, -1, r)

    # -- helpers ------------------------------------------------------------------------------
    def extended_len(self) -> int:
        return 1 << self.extended_k

    @staticmethod
    def _batch_of(a, n: int, name: str) -> int:
        """Rows of `a` must be a whole number of n-element polynomials: (n, 4) or (batch, n, 4)."""
        rows = _tensor_rows(a, 4, name)
        if rows == 0 or rows % n:
            raise ValueError(f"{name}: expected (batch, {n}, 4) words")
        return rows // n

    def _ifft(self, a, omega_inv: int, log_n: int, divisor: int) -> None:
        batch = self._batch_of(a, 1 << log_n, "ifft")
        _lib.check(_lib.load().hm_ntt_batch_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), batch, _ptr(fr_words(omega_inv)), log_n,
                                                          _ptr(fr_words(divisor)), None, ctypes.c_void_p(_stream_ptr(a))))

    # -- the reference's methods --------------------------------------------------------------
    def lagrange_to_coeff(self, a):
        """In place on a (n, 4) or (batch, n, 4) GPU tensor (one set of launches for the batch); returns it."""
        self._ifft(a, self.omega_inv, self.k, self.ifft_divisor)
        return a

    def coeff_to_extended(self, a, internal: bool = False, out=None):
        """(n, 4) or (batch, n, 4) coefficient tensor -> new (2^extended_k, 4) / (batch, 2^extended_k, 4)
        tensor of evaluations on the zeta-coset.  ``internal``: the evaluations come out multiplied by 32 -- the form
        ``hm_graph_evaluate_flags_dev(HM_GRAPH_COLUMNS_INTERNAL)`` loads without a conversion product; the factor rides on the
        coset constants the first NTT pass multiplies in anyway.  A ``numpy`` array takes the host-pointer form
        (``hm_coeff_to_extended_bn256_fr``: n x 32 B up, 2^extended_k x 32 B down); ``out`` (host form only) receives the result
        instead of a fresh array."""
        r = FR_MODULUS
        sc = 32 if internal else 1
        coset = np.concatenate([fr_words(sc), fr_words(sc * self.g_coset), fr_words(sc * self.g_coset * self.g_coset % r)])
        en = self.extended_len()
        if isinstance(a, np.ndarray):
            # host memory (the drop-in prover's Polynomial): (n, 4) uint64 in, new (2^extended_k, 4) uint64 out; only the n
            # coefficients cross PCIe upwards, never the zero padding upstream's resize() appends
            src = _np(a, 4, "coeff_to_extended")
            if src.shape[0] != self.n:
                raise ValueError(f"coeff_to_extended: expected ({self.n}, 4) words")
            if out is None:
                out = np.empty((en, 4), dtype=np.uint64)        # fresh pages: the copy back takes their first-touch faults
            elif not (isinstance(out, np.ndarray) and out.dtype == np.uint64 and out.shape == (en, 4) and out.flags["C_CONTIGUOUS"]):
                raise ValueError(f"coeff_to_extended: out must be a C-contiguous ({en}, 4) uint64 array")
            _lib.check(_lib.load().hm_coeff_to_extended_bn256_fr(_ptr(src), _ptr(out), _ptr(fr_words(self.extended_omega)), self.k,
                                                                 self.extended_k, _ptr(coset)))
            return out
        import torch

        batch = self._batch_of(a, self.n, "coeff_to_extended")
        a = a.contiguous()
        ext = torch.empty((batch, en, 4), dtype=a.dtype, device=a.device)      # the zero part is never materialised
        _lib.check(_lib.load().hm_coeff_to_extended_bn256_fr_dev(
            ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(ext.data_ptr()), batch, _ptr(fr_words(self.extended_omega)), self.k,
            self.extended_k, _ptr(coset), ctypes.c_void_p(_stream_ptr(ext))))
        return ext if a.dim() == 3 else ext[0]

    def divide_by_vanishing_poly(self, a):
        """In place on a (2^extended_k, 4) or (batch, 2^extended_k, 4) tensor of extended-coset evaluations: a[i] *=
        t_evaluations[i % 2^(extended_k - k)], the inverses of X^n - 1 on the coset zeta * <extended_omega> (upstream
        EvaluationDomain::divide_by_vanishing_poly)."""
        en = self.extended_len()
        self._batch_of(a, en, "divide_by_vanishing_poly")
        r = FR_MODULUS
        period = 1 << (self.extended_k - self.k)
        if period > 64:
            raise ValueError("divide_by_vanishing_poly: extension factor above 64")
        t_inv = np.stack([fr_words(pow((pow(self.g_coset * pow(self.extended_omega, i, r) % r, self.n, r) - 1) % r, -1, r))
                          for i in range(period)])
        _lib.check(_lib.load().hm_fr_mul_periodic_dev(ctypes.c_void_p(a.data_ptr()), _tensor_rows(a, 4, "a"), _ptr(t_inv), period,
                                                      ctypes.c_void_p(_stream_ptr(a))))
        return a

    # -- the extended domain one coset of <omega> at a time (later halo2_proofs: coeff_to_extended_part) ---------
    def num_cosets(self) -> int:
        return 1 << (self.extended_k - self.k)

    def coset_shift(self, j: int) -> int:
        """Row E t + j of the extended array is the value at coset_shift(j) * omega^t (E = num_cosets())."""
        return self.g_coset * pow(self.extended_omega, j, FR_MODULUS) % FR_MODULUS

    def coeff_to_coset(self, a, j: int, internal: bool = False, out=None):
        """(n, 4) or (batch, n, 4) coefficient tensor -> its evaluations on coset j of the extended domain, the same shape:
        row t equals row E t + j of ``coeff_to_extended(a)``.  ``internal`` as there."""
        import torch

        batch = self._batch_of(a, self.n, "coeff_to_coset")
        if not 0 <= j < self.num_cosets():
            raise ValueError("coeff_to_coset: no such coset")
        a = a.contiguous()
        if out is None:
            out = torch.empty_like(a)
        elif out.shape != a.shape or not out.is_contiguous():
            raise ValueError("coeff_to_coset: out must be a contiguous tensor of the input's shape")
        _lib.check(_lib.load().hm_coeff_to_coset_bn256_fr_dev(
            ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), batch, _ptr(fr_words(self.omega)), self.k,
            _ptr(fr_words(self.coset_shift(j))), 1 if internal else 0, ctypes.c_void_p(_stream_ptr(out))))
        return out

    def coeff_to_cosets(self, a, cosets, internal: bool = False):
        """(batch, n, 4) coefficient tensor -> (batch, len(cosets), n, 4): [b, i] = polynomial b on coset cosets[i], ONE launch
        chain for all of them (at most 16 cosets).  Reshaped to (batch, len(cosets) * n, 4) every row b is a column of
        ``CompiledGraph.evaluate(..., segments=len(cosets))``."""
        import torch

        batch = self._batch_of(a, self.n, "coeff_to_cosets")
        cosets = [int(c) for c in cosets]
        if not cosets or len(cosets) > 16 or any(not 0 <= c < self.num_cosets() for c in cosets):
            raise ValueError("coeff_to_cosets: 1 .. 16 coset indices below E")
        a = a.contiguous()
        out = torch.empty((batch, len(cosets), self.n, 4), dtype=a.dtype, device=a.device)
        shifts = np.stack([fr_words(self.coset_shift(c)) for c in cosets])
        _lib.check(_lib.load().hm_coeff_to_cosets_bn256_fr_dev(
            ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), batch, _ptr(fr_words(self.omega)), self.k, _ptr(shifts),
            len(cosets), 1 if internal else 0, ctypes.c_void_p(_stream_ptr(out))))
        return out

    def cosets_to_partials(self, v, cosets):
        """In place on (len(cosets), n, 4): the values ONE polynomial takes on those cosets -> its partials (``coset_to_partial``
        of each), one launch chain."""
        cosets = [int(c) for c in cosets]
        if v.dim() != 3 or v.shape[0] != len(cosets) or v.shape[1] != self.n or not v.is_contiguous():
            raise ValueError("cosets_to_partials: expected a contiguous (len(cosets), n, 4) tensor")
        inv = np.stack([fr_words(pow(self.coset_shift(c), -1, FR_MODULUS)) for c in cosets])
        _lib.check(_lib.load().hm_cosets_to_coeff_bn256_fr_dev(
            ctypes.c_void_p(v.data_ptr()), len(cosets), _ptr(fr_words(self.omega_inv)), self.k, _ptr(fr_words(self.ifft_divisor)), _ptr(inv),
            ctypes.c_void_p(_stream_ptr(v))))
        return v

    def coset_vanishing_inverse(self, j: int) -> int:
        """1 / (X^n - 1) on coset j: one constant (the entry j of divide_by_vanishing_poly's periodic pattern)."""
        r = FR_MODULUS
        return pow((pow(self.coset_shift(j), self.n, r) - 1) % r, -1, r)

    def coset_to_partial(self, v, j: int):
        """In place on the (n, 4) / (batch, n, 4) values a polynomial h takes on coset j: -> d_j[i] = sum_q h[i + q n]
        zeta^(n q) w^(j q) (w = extended_omega^n).  ``combine_cosets`` turns the E of them into h's coefficients."""
        batch = self._batch_of(v, self.n, "coset_to_partial")
        _lib.check(_lib.load().hm_coset_to_coeff_bn256_fr_dev(
            ctypes.c_void_p(v.data_ptr()), batch, _ptr(fr_words(self.omega_inv)), self.k, _ptr(fr_words(self.ifft_divisor)),
            _ptr(fr_words(pow(self.coset_shift(j), -1, FR_MODULUS))), ctypes.c_void_p(_stream_ptr(v))))
        return v

    def min_cosets(self) -> int:
        """Cosets that DETERMINE the quotient: h has fewer than n * (j - 1) coefficients, so its values on any j - 1 of the E
        cosets do (each coset contributes n equations per residue class of coefficients)."""
        return self.quotient_poly_degree

    def combine_cosets(self, partials, pieces: int = None, cosets=None, divide_by_vanishing: bool = False):
        """partials[a] = coset_to_partial(values on coset cosets[a]) -> the coefficients of the polynomial those values belong
        to.  ``cosets=None``: all E cosets in order -- any polynomial of degree < E n, the same words ``extended_to_coeff``
        returns (``pieces`` rows of n coefficients, default the quotient's j - 1).  ``cosets=[j_0, .., j_(q-1)]``: q distinct
        cosets only -- the polynomial of degree < q n through those values (q * n coefficients).  For the quotient h of a
        SATISFIED circuit (degree < n (j - 1): upstream's extended_to_coeff truncates to exactly that many coefficients)
        q = j - 1 cosets give the same words as all E; the other E - (j - 1) cosets of evaluate_h need not be computed at all.
        For an unsatisfied witness the numerator is not divisible by X^n - 1, the two differ, and neither is a proof.
        With u_j = (zeta * extended_omega^j)^n: partial_j[i] = sum_t piece_t[i] u_j^t, so piece_t = sum_a Vinv[t][a] partial_a
        with V[a][t] = u_(j_a)^t -- one linear combination of q arrays per piece (for all E cosets Vinv is the inverse DFT
        zeta^(-n t) / E * w^(-j t)).  ``divide_by_vanishing``: the partials are of the UNDIVIDED numerator; 1 / (X^n - 1) is the
        constant 1 / (u_c - 1) on coset c and rides on column c of Vinv (no pass over the data, no constant in the program)."""
        import torch
        from .arithmetic import linear_combination

        e = self.num_cosets()
        r = FR_MODULUS
        if cosets is None:
            cosets = list(range(e))
            default_pieces = self.quotient_poly_degree
        else:
            cosets = [int(c) for c in cosets]
            default_pieces = len(cosets)
            if len(set(cosets)) != len(cosets) or any(not 0 <= c < e for c in cosets):
                raise ValueError("combine_cosets: cosets must be distinct indices below E")
        q = len(cosets)
        if len(partials) != q:
            raise ValueError("combine_cosets: one partial per coset")
        pieces = default_pieces if pieces is None else pieces
        if not 0 < pieces <= q:
            raise ValueError("combine_cosets: at most one piece per coset")
        nodes = [pow(self.coset_shift(c), self.n, r) for c in cosets]
        vinv = _vandermonde_inverse(nodes, r)
        if divide_by_vanishing:
            tinv = [pow((u - 1) % r, -1, r) for u in nodes]
            vinv = [[row[a] * tinv[a] % r for a in range(q)] for row in vinv]
        out = torch.empty((pieces, self.n, 4), dtype=partials[0].dtype, device=partials[0].device)
        for t in range(pieces):
            linear_combination(list(partials), np.stack([fr_words(vinv[t][a]) for a in range(q)]), out=out[t])
        return out.reshape(pieces * self.n, 4)

    def extended_to_coeff(self, a):
        """In place on a (2^extended_k, 4) or (batch, 2^extended_k, 4) tensor; returns the first
        n*(j-1) rows of each polynomial (a view)."""
        en = self.extended_len()
        r = FR_MODULUS
        c3 = np.concatenate([fr_words(1), fr_words(self.g_coset_inv), fr_words(self.g_coset_inv * self.g_coset_inv % r)])
        if isinstance(a, np.ndarray):
            # host memory: in place on a (2^extended_k, 4) uint64 array; only the n*(j-1) coefficients upstream keeps come back
            arr = _np(a, 4, "extended_to_coeff", writable=True)
            if arr.shape[0] != en:
                raise ValueError(f"extended_to_coeff: expected ({en}, 4) words")
            keep = self.n * self.quotient_poly_degree
            _lib.check(_lib.load().hm_extended_to_coeff_bn256_fr(_ptr(arr), keep, _ptr(fr_words(self.extended_omega_inv)), self.extended_k,
                                                                 _ptr(fr_words(self.extended_ifft_divisor)), _ptr(c3)))
            return arr[:keep]
        batch = self._batch_of(a, en, "extended_to_coeff")
        # one call: the ifft divisor and the zeta^-(i % 3) pattern are folded into the last NTT pass
        _lib.check(_lib.load().hm_extended_to_coeff_bn256_fr_dev(
            ctypes.c_void_p(a.data_ptr()), batch, _ptr(fr_words(self.extended_omega_inv)), self.extended_k,
            _ptr(fr_words(self.extended_ifft_divisor)), _ptr(c3), ctypes.c_void_p(_stream_ptr(a))))
        flat = a.reshape(batch, en, 4)
        out = flat[:, : self.n * self.quotient_poly_degree]
        return out if a.dim() == 3 else out[0]
