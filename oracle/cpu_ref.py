"""ctypes front-end of oracle/cpu_ref.c (the CPU restatement).  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED -- see oracle/bn256_ref.py.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module; the product package never does.
All arrays are numpy uint64 in the layout of SURVEY.md §8a (Montgomery limbs, little-endian).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "cpu_ref.c")
_LIB = os.path.join(_HERE, "libcpu_ref.so")
_lib = None

_u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force: bool = False) -> str:
    """gcc the restatement into oracle/libcpu_ref.so (in-tree, git-ignored, travels with gpurun)."""
    if force or not os.path.exists(_LIB):
        cmd = ["gcc", "-O3", "-march=x86-64-v3", "-shared", "-fPIC", "-pthread", "-o", _LIB, _SRC, "-lm"]
        subprocess.run(cmd, check=True)
    return _LIB


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = ctypes.CDLL(_LIB)
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_u64p)


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def _cgroup_cpu_quota() -> float:
    """CPUs the container may use at once (cgroup v2 cpu.max, v1 cfs quota); inf when unlimited."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
            if quota != "max":
                return float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            quota = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = float(f.read())
        if quota > 0:
            return quota / period
    except (OSError, ValueError):
        pass
    return float("inf")


def default_threads() -> int:
    """Every core this process may actually run on: the scheduler affinity, capped by the container's CPU quota (a box
    that shows 256 CPUs but grants 16 runs 256 threads slower than 16); HALO2_CPU_THREADS overrides."""
    env = os.environ.get("HALO2_CPU_THREADS")
    if env:
        return max(1, int(env))
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:  # pragma: no cover
        n = os.cpu_count() or 1
    quota = _cgroup_cpu_quota()
    if quota != float("inf"):
        n = min(n, max(1, int(quota + 0.999)))
    return max(1, n)


def best_multiexp(scalars: np.ndarray, bases: np.ndarray, threads: int | None = None) -> np.ndarray:
    """-> (12,) Jacobian, Montgomery.  scalars (n,4), bases (n,8)."""
    s, b = _c(scalars).reshape(-1, 4), _c(bases).reshape(-1, 8)
    assert s.shape[0] == b.shape[0]
    out = np.zeros(12, dtype=np.uint64)
    lib().ref_best_multiexp(_p(s), _p(b), ctypes.c_size_t(s.shape[0]), ctypes.c_int(threads or default_threads()), _p(out))
    return out


def best_fft(a: np.ndarray, omega: np.ndarray, log_n: int, threads: int | None = None) -> np.ndarray:
    """Returns the transformed copy of a (n,4)."""
    x = _c(a).reshape(-1, 4).copy()
    assert x.shape[0] == 1 << log_n
    w = _c(omega).reshape(4)
    lib().ref_best_fft(_p(x), _p(w), ctypes.c_uint32(log_n), ctypes.c_int(threads or default_threads()))
    return x


def _binary(name: str, a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a, b = _c(a).reshape(-1, 4), _c(b).reshape(-1, 4)
    o = np.zeros_like(a)
    getattr(lib(), name)(_p(a), _p(b), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def _unary(name: str, a: np.ndarray) -> np.ndarray:
    a = _c(a).reshape(-1, 4)
    o = np.zeros_like(a)
    getattr(lib(), name)(_p(a), _p(o), ctypes.c_size_t(a.shape[0]))
    return o


def fr_mul(a, b): return _binary("ref_fr_mul", a, b)
def fq_mul(a, b): return _binary("ref_fq_mul", a, b)
def fr_add(a, b): return _binary("ref_fr_add", a, b)
def fr_sub(a, b): return _binary("ref_fr_sub", a, b)
def fr_from_mont(a): return _unary("ref_fr_from_mont", a)
def fr_to_mont(a): return _unary("ref_fr_to_mont", a)
def fr_inv(a): return _unary("ref_fr_inv", a)


def fr_horner(coeffs: np.ndarray, x: np.ndarray) -> np.ndarray:
    c, xx = _c(coeffs).reshape(-1, 4), _c(x).reshape(4)
    o = np.zeros(4, dtype=np.uint64)
    lib().ref_fr_horner(_p(c), ctypes.c_size_t(c.shape[0]), _p(xx), _p(o))
    return o


def fr_dot(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a, b = _c(a).reshape(-1, 4), _c(b).reshape(-1, 4)
    o = np.zeros(4, dtype=np.uint64)
    lib().ref_fr_dot(_p(a), _p(b), ctypes.c_size_t(a.shape[0]), _p(o))
    return o


def fr_powers(base: np.ndarray, n: int) -> np.ndarray:
    o = np.zeros((n, 4), dtype=np.uint64)
    lib().ref_fr_powers(_p(_c(base).reshape(4)), ctypes.c_size_t(n), _p(o))
    return o


def g1_mul(k_mont: np.ndarray, p_affine: np.ndarray) -> np.ndarray:
    o = np.zeros(8, dtype=np.uint64)
    lib().ref_g1_mul(_p(_c(k_mont).reshape(4)), _p(_c(p_affine).reshape(8)), _p(o))
    return o


def g1_to_affine(jac: np.ndarray) -> np.ndarray:
    j = _c(jac).reshape(-1, 12)
    o = np.zeros((j.shape[0], 8), dtype=np.uint64)
    lib().ref_g1_to_affine(_p(j), _p(o), ctypes.c_size_t(j.shape[0]))
    return o


def g1_sum(jac: np.ndarray) -> np.ndarray:
    j = _c(jac).reshape(-1, 12)
    o = np.zeros(12, dtype=np.uint64)
    lib().ref_g1_sum(_p(j), ctypes.c_size_t(j.shape[0]), _p(o))
    return o


def g1_add_affine(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    o = np.zeros(8, dtype=np.uint64)
    lib().ref_g1_add_affine(_p(_c(a).reshape(8)), _p(_c(b).reshape(8)), _p(o))
    return o


def msm_naive(scalars: np.ndarray, bases: np.ndarray) -> np.ndarray:
    s, b = _c(scalars).reshape(-1, 4), _c(bases).reshape(-1, 8)
    o = np.zeros(12, dtype=np.uint64)
    lib().ref_msm_naive(_p(s), _p(b), ctypes.c_size_t(s.shape[0]), _p(o))
    return o


def srs(s_mont: np.ndarray, n: int) -> np.ndarray:
    o = np.zeros((n, 8), dtype=np.uint64)
    lib().ref_srs(_p(_c(s_mont).reshape(4)), ctypes.c_size_t(n), _p(o))
    return o


G1_GENERATOR_MONT = None


def g1_generator() -> np.ndarray:
    """(1, 2) in Montgomery form."""
    global G1_GENERATOR_MONT
    if G1_GENERATOR_MONT is None:
        from . import bn256_ref as o  # type: ignore
        G1_GENERATOR_MONT = o.g1_affine_array([o.G1_GEN])[0]
    return G1_GENERATOR_MONT
