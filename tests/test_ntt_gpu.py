"""GPU suite: best_fft on the MI355X through the C ABI vs the oracle (bit-exact), the committed
golden vectors, and size-independent properties at the benchmark's full size."""
import ctypes

import numpy as np
import pytest

import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from halo2_experiments_amd.domain import EvaluationDomain, FR_MODULUS as FR_MODULUS_FOR_TESTS, fr_words

pytestmark = pytest.mark.gpu


def rand_fr_gpu(n, seed):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)


def test_golden_vectors(golden):
    g = golden["ntt"]
    for k in range(0, 11):
        a = g[f"k{k}_in"].copy()
        h.best_fft(a, g[f"k{k}_omega"], k)
        assert np.array_equal(a, g[f"k{k}_out"]), k
    for name in ("delta", "ones"):
        a = g[f"{name}_in"].copy()
        h.best_fft(a, g["k5_omega"], 5)
        assert np.array_equal(a, g[f"{name}_out"])
    a = g["w3_in"].copy()
    h.best_fft(a, g["w3_omega"], 5)          # omega is an argument, not assumed to be ROOT^(2^(28-k))
    assert np.array_equal(a, g["w3_out"])


@pytest.mark.parametrize("k", list(range(0, 23)))
def test_host_api_matches_oracle_every_log_n(cref, pyref, k):
    """Full-array compare with the oracle at every size up to 2^22: the 1-pass (<= 2^11), 2-pass (<= 2^21)
    and 3-pass (2^22) plans and every digit split, including config 3 / 4's extended sizes 2^20 / 2^21."""
    a = rand_fr_gpu(1 << k, 1000 + k).cpu().numpy().view(np.uint64)
    w = pyref.fr_array([pyref.fr_omega(k)])[0]
    exp = cref.best_fft(a, w, k, 8)
    got = a.copy()
    h.best_fft(got, w, k)
    assert np.array_equal(got, exp)


@pytest.mark.parametrize("k", [12, 17, 21])
def test_device_api_matches_host_api(pyref, k):
    import torch
    x = rand_fr_gpu(1 << k, 77 + k)
    w = pyref.fr_array([pyref.fr_omega(k)])[0]
    host = x.cpu().numpy().view(np.uint64).copy()
    h.best_fft(host, w, k)
    h.best_fft(x, w, k)
    torch.cuda.synchronize()
    assert np.array_equal(x.cpu().numpy().view(np.uint64), host)


@pytest.mark.parametrize("k", [20, 21, 22, 23])
def test_device_api_matches_oracle_full_array(cref, pyref, k):
    """The device-pointer form (what a device-resident prover calls) against the oracle, every element: the two-pass
    plan at its largest sizes and the three-pass plan at its two smallest."""
    import torch
    x = rand_fr_gpu(1 << k, 4100 + k)
    w = pyref.fr_array([pyref.fr_omega(k)])[0]
    exp = cref.best_fft(x.cpu().numpy().view(np.uint64), w, k)
    h.best_fft(x, w, k)
    torch.cuda.synchronize()
    assert np.array_equal(x.cpu().numpy().view(np.uint64), exp)


def _pattern3(values, rows):
    """rows x 4 words: values[i % 3] in row i."""
    base = np.stack([fr_words(v) for v in values])
    return np.tile(base, ((rows + 2) // 3, 1))[:rows]


@pytest.mark.parametrize("k", [17, 18])
def test_evaluation_domain_at_the_config_shapes_matches_oracle(cref, pyref, k):
    """BASELINE configs 3 and 4 exactly: k = 17 -> 2^20 and k = 18 -> 2^21 (j = 7, three extension bits), every element
    of lagrange_to_coeff, hm_coeff_to_extended_* (compact input, the zero part never materialised, plain and internal
    form) and hm_extended_to_coeff_* against the ORACLE's compositions (best_fft + element-wise products), not the
    library's own plain path."""
    import torch
    o = pyref
    d = EvaluationDomain(j=7, k=k)
    n, en, R = d.n, d.extended_len(), o.R
    assert d.extended_k == k + 3
    a = rand_fr_gpu(2 * n, 5200 + k).reshape(2, n, 4)
    ah = a.cpu().numpy().view(np.uint64)
    exp_coeff = [cref.fr_mul(cref.best_fft(ah[b], fr_words(d.omega_inv), k), np.tile(fr_words(d.ifft_divisor), (n, 1))) for b in range(2)]
    coeff = d.lagrange_to_coeff(a.clone())
    for b in range(2):
        assert np.array_equal(coeff[b].cpu().numpy().view(np.uint64), exp_coeff[b]), b
    zeta = _pattern3([1, d.g_coset, d.g_coset * d.g_coset % R], en)
    exp_ext = []
    for b in range(2):
        pad = np.zeros((en, 4), dtype=np.uint64)
        pad[:n] = exp_coeff[b]
        exp_ext.append(cref.best_fft(cref.fr_mul(pad, zeta), fr_words(d.extended_omega), d.extended_k))
    ext = d.coeff_to_extended(coeff)
    for b in range(2):
        assert np.array_equal(ext[b].cpu().numpy().view(np.uint64), exp_ext[b]), b
    ext32 = d.coeff_to_extended(coeff[0], internal=True)
    assert np.array_equal(ext32.cpu().numpy().view(np.uint64), cref.fr_mul(exp_ext[0], np.tile(fr_words(32), (en, 1))))
    # extended_to_coeff on an ARBITRARY extended array (h(X) is not band-limited to n coefficients): every element
    x = rand_fr_gpu(en, 5300 + k)
    xh = x.cpu().numpy().view(np.uint64)
    inv = cref.fr_mul(cref.best_fft(xh, fr_words(d.extended_omega_inv), d.extended_k), np.tile(fr_words(d.extended_ifft_divisor), (en, 1)))
    exp_back = cref.fr_mul(inv, _pattern3([1, d.g_coset_inv, d.g_coset_inv * d.g_coset_inv % R], en))
    got = x.clone()
    out = d.extended_to_coeff(got)
    torch.cuda.synchronize()
    assert np.array_equal(got.cpu().numpy().view(np.uint64), exp_back)
    assert out.shape[-2] == n * 6


@pytest.fixture(params=["direct", "lanes"])
def host_copies(request):
    """Both ways the host-pointer forms move their arrays (csrc/xfer.hip): the runtime's pageable path and the library's pinned lanes."""
    lib = _lib.load()
    _lib.check(lib.hm_set_host_copies(2 if request.param == "direct" else 1))
    yield request.param
    _lib.check(lib.hm_set_host_copies(0))


@pytest.mark.parametrize("k", [17, 18])
def test_host_pointer_domain_steps_match_oracle(cref, pyref, k, host_copies):
    """hm_coeff_to_extended_bn256_fr / hm_extended_to_coeff_bn256_fr (host arrays: what the patched EvaluationDomain::coeff_to_extended /
    extended_to_coeff of the drop-in prover call) at configs 3 and 4's shapes, j = 7: every element against the ORACLE's compositions
    (zero-pad, distribute_powers_zeta, best_fft; best_fft with the inverse root, divisor, inverse zeta powers, truncate), and the
    PCIe byte counters show that neither the zero padding nor the truncated tail travelled."""
    import ctypes
    from halo2_experiments_amd import _lib
    o = pyref
    d = EvaluationDomain(j=7, k=k)
    n, en, R = d.n, d.extended_len(), o.R
    coeffs = rand_fr_gpu(n, 6100 + k).cpu().numpy().view(np.uint64)
    before = coeffs.copy()
    zeta = _pattern3([1, d.g_coset, d.g_coset * d.g_coset % R], en)
    pad = np.zeros((en, 4), dtype=np.uint64)
    pad[:n] = coeffs
    exp_ext = cref.best_fft(cref.fr_mul(pad, zeta), fr_words(d.extended_omega), d.extended_k)
    lib = _lib.load()
    def pcie_bytes():
        st = _lib.Stats()
        _lib.check(lib.hm_get_stats(ctypes.byref(st)))
        return st.h2d_bytes, st.d2h_bytes
    up0, down0 = pcie_bytes()
    ext = d.coeff_to_extended(coeffs)
    assert isinstance(ext, np.ndarray) and ext.shape == (en, 4) and np.array_equal(coeffs, before)
    assert np.array_equal(ext, exp_ext)
    # extended_to_coeff on an ARBITRARY extended array, in place; the first n * 6 coefficients come back, the tail keeps its contents
    x = rand_fr_gpu(en, 6200 + k).cpu().numpy().view(np.uint64)
    inv = cref.fr_mul(cref.best_fft(x, fr_words(d.extended_omega_inv), d.extended_k), np.tile(fr_words(d.extended_ifft_divisor), (en, 1)))
    exp_back = cref.fr_mul(inv, _pattern3([1, d.g_coset_inv, d.g_coset_inv * d.g_coset_inv % R], en))
    got = x.copy()
    out = d.extended_to_coeff(got)
    keep = n * 6
    assert out.shape == (keep, 4) and np.array_equal(out, exp_back[:keep])
    assert np.array_equal(got[keep:], x[keep:])                       # never downloaded: upstream truncates it away
    up1, down1 = pcie_bytes()
    assert up1 - up0 == 32 * (n + en) and down1 - down0 == 32 * (en + keep)      # 8 + 64 MiB up, 64 + 48 MiB down at k = 18
    # round trip of a band-limited polynomial: coeff -> extended -> coeff returns the coefficients, zeros above n
    back = d.extended_to_coeff(d.coeff_to_extended(coeffs))
    assert np.array_equal(back[:n], coeffs) and not back[n:].any()
    # argument checks: the reference's assert_eq!(a.len(), 1 << k) becomes ValueError; NULL pointers an error code, not a crash
    with pytest.raises(ValueError):
        d.coeff_to_extended(coeffs[:-1])
    with pytest.raises(ValueError):
        d.extended_to_coeff(np.zeros((en // 2, 4), dtype=np.uint64))
    assert lib.hm_coeff_to_extended_bn256_fr(None, None, None, k, k + 3, None) == -1
    assert lib.hm_extended_to_coeff_bn256_fr(None, 0, None, k, None, None) == -1


def test_host_pointer_domain_steps_small_and_unextended(cref, pyref, host_copies):
    """The host forms on the plans the big shapes do not reach: single-pass sizes, log_ext == log_n, one extension bit, keep = 0 and
    keep = 2^log_ext, output aliasing the input allocation (upstream resizes its Vec in place)."""
    import ctypes
    from halo2_experiments_amd import _lib
    o = pyref
    lib = _lib.load()
    R = o.R
    u64p = ctypes.POINTER(ctypes.c_uint64)
    for k, j in ((3, 2), (5, 3), (9, 4), (11, 7), (12, 5), (13, 9)):
        d = EvaluationDomain(j=j, k=k)
        n, en = d.n, d.extended_len()
        coeffs = rand_fr_gpu(n, 6300 + k).cpu().numpy().view(np.uint64)
        pad = np.zeros((en, 4), dtype=np.uint64)
        pad[:n] = coeffs
        exp = cref.best_fft(cref.fr_mul(pad, _pattern3([1, d.g_coset, d.g_coset * d.g_coset % R], en)), fr_words(d.extended_omega), d.extended_k)
        assert np.array_equal(d.coeff_to_extended(coeffs), exp), (k, j)
        # in one allocation: coefficients at the front of the buffer the evaluations are written to
        buf = np.zeros((en, 4), dtype=np.uint64)
        buf[:n] = coeffs
        coset = np.concatenate([fr_words(1), fr_words(d.g_coset), fr_words(d.g_coset * d.g_coset % R)])
        p = buf.ctypes.data_as(u64p)
        _lib.check(lib.hm_coeff_to_extended_bn256_fr(p, p, fr_words(d.extended_omega).ctypes.data_as(u64p), k, d.extended_k, coset.ctypes.data_as(u64p)))
        assert np.array_equal(buf, exp), (k, j)
        back = d.extended_to_coeff(exp.copy())
        keep = n * (j - 1)
        assert back.shape == (keep, 4) and np.array_equal(back[:n], coeffs) and not back[n:].any(), (k, j)
        c3 = np.concatenate([fr_words(1), fr_words(d.g_coset_inv), fr_words(d.g_coset_inv * d.g_coset_inv % R)])
        for keep in (0, en):
            y = exp.copy()
            _lib.check(lib.hm_extended_to_coeff_bn256_fr(y.ctypes.data_as(u64p), keep, fr_words(d.extended_omega_inv).ctypes.data_as(u64p), d.extended_k,
                                                         fr_words(d.extended_ifft_divisor).ctypes.data_as(u64p), c3.ctypes.data_as(u64p)))
            if keep == 0:
                assert np.array_equal(y, exp)
            else:
                assert np.array_equal(y[:n], coeffs) and not y[n:].any()
        assert lib.hm_extended_to_coeff_bn256_fr(exp.ctypes.data_as(u64p), en + 1, fr_words(d.extended_omega_inv).ctypes.data_as(u64p), d.extended_k,
                                                 fr_words(d.extended_ifft_divisor).ctypes.data_as(u64p), c3.ctypes.data_as(u64p)) == -1
    d = EvaluationDomain(j=2, k=6)                                 # quotient degree 1: the extended domain IS the domain
    assert d.extended_k == 6
    coeffs = rand_fr_gpu(64, 6400).cpu().numpy().view(np.uint64)
    exp = cref.best_fft(cref.fr_mul(coeffs, _pattern3([1, d.g_coset, d.g_coset * d.g_coset % R], 64)), fr_words(d.extended_omega), 6)
    assert np.array_equal(d.coeff_to_extended(coeffs), exp)


def test_device_memory_entry_points_carry_a_prover_without_a_hip_binding(cref, host_copies):
    """hm_device_malloc / hm_copy_to_device / hm_copy_to_host / hm_device_synchronize / hm_device_free: what rust/.../mi355x_dev.rs's
    DevicePoly is made of.  Upload, the EvaluationDomain steps on the device-resident array (default stream = NULL), download: the
    same words as the host-pointer forms and as the oracle; both copy paths."""
    from halo2_experiments_amd.arithmetic import _ptr
    lib = _lib.load()
    d = EvaluationDomain(6, 14)
    n, en = d.n, d.extended_len()
    coeffs = rand_fr_gpu(n, 6500).cpu().numpy().view(np.uint64).copy()
    want = d.coeff_to_extended(coeffs)                                        # host-pointer form (oracle-checked above)
    d_in, d_out = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.check(lib.hm_device_malloc(n * 32, ctypes.byref(d_in)))
    _lib.check(lib.hm_device_malloc(en * 32, ctypes.byref(d_out)))
    try:
        assert d_in.value and d_out.value
        _lib.check(lib.hm_copy_to_device(d_in, coeffs.ctypes.data_as(ctypes.c_void_p), n * 32))
        R = FR_MODULUS_FOR_TESTS
        coset = np.concatenate([fr_words(1), fr_words(d.g_coset), fr_words(d.g_coset * d.g_coset % R)])
        _lib.check(lib.hm_coeff_to_extended_bn256_fr_dev(d_in, d_out, 1, _ptr(fr_words(d.extended_omega)), d.k, d.extended_k, _ptr(coset), None))
        _lib.check(lib.hm_device_synchronize())
        got = np.empty((en, 4), dtype=np.uint64)
        _lib.check(lib.hm_copy_to_host(got.ctypes.data_as(ctypes.c_void_p), d_out, en * 32))
        assert np.array_equal(got, want)
        c3 = np.concatenate([fr_words(1), fr_words(d.g_coset_inv), fr_words(d.g_coset_inv * d.g_coset_inv % R)])
        _lib.check(lib.hm_extended_to_coeff_bn256_fr_dev(d_out, 1, _ptr(fr_words(d.extended_omega_inv)), d.extended_k,
                                                         _ptr(fr_words(d.extended_ifft_divisor)), _ptr(c3), None))
        _lib.check(lib.hm_device_synchronize())
        back = np.empty((n, 4), dtype=np.uint64)
        _lib.check(lib.hm_copy_to_host(back.ctypes.data_as(ctypes.c_void_p), d_out, n * 32))     # a prefix of the device array
        assert np.array_equal(back, coeffs)
        assert lib.hm_copy_to_host(back.ctypes.data_as(ctypes.c_void_p), d_out, 0) == 0
    finally:
        _lib.check(lib.hm_device_free(d_in))
        _lib.check(lib.hm_device_free(d_out))
    assert lib.hm_device_free(None) == 0
    z = ctypes.c_void_p(0x1)
    assert lib.hm_device_malloc(0, ctypes.byref(z)) == 0 and z.value is None


def test_the_copy_route_is_a_rule_on_the_range_and_both_routes_are_ordered_behind_the_default_stream(cref):
    """csrc/xfer.hip, round 6: no timing decides anything.  Under the default policy an array the caller has NOT registered goes through
    the library's pinned lanes, a range registered with hm_host_register straight to hipMemcpy (a DMA from registered memory) -- counted
    by hm_get_stats, the same words either way.  And hm_copy_to_host waits for the default stream whichever way the bytes go: a transform
    queued on stream NULL just before it, no hm_device_synchronize in between (on the lanes' non-blocking streams that read
    stale data)."""
    from halo2_experiments_amd.arithmetic import _ptr
    lib = _lib.load()
    _lib.check(lib.hm_set_host_copies(0))
    k = 20                                                                     # 32 MiB: every lane
    n = 1 << k
    omega = fr_words(pow(7, (FR_MODULUS_FOR_TESTS - 1) >> k, FR_MODULUS_FOR_TESTS))
    a0 = rand_fr_gpu(n, 6600).cpu().numpy().view(np.uint64).copy()
    want = cref.best_fft(a0, omega, k, 8)

    def stats():
        st = _lib.Stats()
        _lib.check(lib.hm_get_stats(ctypes.byref(st)))
        return int(st.host_copies_direct), int(st.host_copies_staged), int(st.host_ranges_registered)

    d0, s0, r0 = stats()
    a = a0.copy()
    h.best_fft(a, omega, k)                                                    # unregistered: up and down through the lanes
    d1, s1, _ = stats()
    assert np.array_equal(a, want) and s1 == s0 + 2 and d1 == d0
    reg = a0.copy()
    p = reg.ctypes.data_as(ctypes.c_void_p)
    _lib.check(lib.hm_host_register(p, reg.nbytes))
    try:
        assert stats()[2] == r0 + 1
        assert lib.hm_host_register(ctypes.c_void_p(p.value + 4096), 4096) == -1            # overlaps a registered range
        assert lib.hm_host_unregister(ctypes.c_void_p(p.value + 4096)) == -1                # not the start of one
        h.best_fft(reg, omega, k)                                              # registered: both copies direct
        d2, s2, _ = stats()
        assert np.array_equal(reg, want) and d2 == d1 + 2 and s2 == s1
        half = reg[: n // 2]                                                   # a sub-range of a registered range is registered
        h.best_fft(half, fr_words(pow(7, (FR_MODULUS_FOR_TESTS - 1) >> (k - 1), FR_MODULUS_FOR_TESTS)), k - 1)
        assert stats()[0] == d2 + 2
    finally:
        _lib.check(lib.hm_host_unregister(p))
    assert stats()[2] == r0 and lib.hm_host_unregister(p) == -1 and lib.hm_host_register(None, 8) == -1
    # ordering: queue the transform on the default stream, read back at once
    d_a = ctypes.c_void_p()
    _lib.check(lib.hm_device_malloc(n * 32, ctypes.byref(d_a)))
    try:
        for mode in (0, 2):                                                    # the lanes (default policy), then hipMemcpy
            _lib.check(lib.hm_set_host_copies(mode))
            _lib.check(lib.hm_copy_to_device(d_a, a0.ctypes.data_as(ctypes.c_void_p), n * 32))
            for _ in range(3):                                                 # forward, forward, forward: ~1 ms of queued work
                _lib.check(lib.hm_ntt_batch_bn256_fr_dev(d_a, 1, _ptr(omega), k, None, None, None))
            got = np.empty((n, 4), dtype=np.uint64)
            _lib.check(lib.hm_copy_to_host(got.ctypes.data_as(ctypes.c_void_p), d_a, n * 32))        # no synchronize before it
            exp = want
            for _ in range(2):
                exp = cref.best_fft(exp, omega, k, 8)
            assert np.array_equal(got, exp), mode
    finally:
        _lib.check(lib.hm_set_host_copies(0))
        _lib.check(lib.hm_device_free(d_a))


def test_many_arrays_travel_as_one_transfer(host_copies):
    """hm_copy_many_to_device / _to_host (what DevicePoly::upload_many_at / to_vecs_ranges call): arrays of odd sizes -- below a page, not a
    multiple of a slot, several MiB --, scattered into one device buffer and read back from it in another order; both copy policies; the
    byte counters see the sum."""
    lib = _lib.load()
    sizes = [32, 4096 + 32, 100003 * 32, (1 << 18) * 32, 7 * 32, (3 << 20) + 64, 2 << 20]
    rng = np.random.default_rng(77)
    arrays = [rng.integers(0, 1 << 63, size=sz // 8, dtype=np.uint64) for sz in sizes]
    total = sum(sizes)
    d = ctypes.c_void_p()
    _lib.check(lib.hm_device_malloc(total, ctypes.byref(d)))
    try:
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        k = len(sizes)
        st0 = _lib.Stats()
        _lib.check(lib.hm_get_stats(ctypes.byref(st0)))
        dsts = (ctypes.c_void_p * k)(*[d.value + int(o) for o in offs])
        srcs = (ctypes.c_void_p * k)(*[a.ctypes.data for a in arrays])
        nb = (ctypes.c_size_t * k)(*sizes)
        _lib.check(lib.hm_copy_many_to_device(dsts, srcs, nb, k))
        whole = np.empty(total // 8, dtype=np.uint64)
        _lib.check(lib.hm_copy_to_host(whole.ctypes.data_as(ctypes.c_void_p), d, total))
        assert np.array_equal(whole, np.concatenate(arrays))
        order = [3, 0, 6, 2, 5, 1, 4]
        outs = [np.zeros(sizes[i] // 8, dtype=np.uint64) for i in order]
        _lib.check(lib.hm_copy_many_to_host((ctypes.c_void_p * k)(*[o.ctypes.data for o in outs]),
                                            (ctypes.c_void_p * k)(*[d.value + int(offs[i]) for i in order]),
                                            (ctypes.c_size_t * k)(*[sizes[i] for i in order]), k))
        assert all(np.array_equal(o, arrays[i]) for o, i in zip(outs, order))
        st1 = _lib.Stats()
        _lib.check(lib.hm_get_stats(ctypes.byref(st1)))
        assert st1.h2d_bytes - st0.h2d_bytes == total and st1.d2h_bytes - st0.d2h_bytes == 2 * total
        assert lib.hm_copy_many_to_device(dsts, srcs, nb, 0) == 0 and lib.hm_copy_many_to_device(None, srcs, nb, k) == -1
        bad = (ctypes.c_void_p * k)(*([None] + [a.ctypes.data for a in arrays[1:]]))
        assert lib.hm_copy_many_to_device(dsts, bad, nb, k) == -1
    finally:
        _lib.check(lib.hm_device_free(d))


def test_edge_values(cref, pyref):
    o = pyref
    k = 6
    vals = [0, 1, o.R - 1, o.R - 2, 2, (1 << 253) % o.R] * 11
    a = o.fr_array(vals[:64])
    w = o.fr_array([o.fr_omega(k)])[0]
    exp = cref.best_fft(a, w, k, 1)
    h.best_fft(a, w, k)
    assert np.array_equal(a, exp)
    z = np.zeros((1 << 13, 4), dtype=np.uint64)
    h.best_fft(z, o.fr_array([o.fr_omega(13)])[0], 13)
    assert not z.any()


@pytest.mark.parametrize("k", [20, 21, 22, 23, 24, 25, 26])
def test_full_size_properties(cref, pyref, k):
    """At sizes no oracle brute-forces: inverse(forward(x)) * n^-1 == x bit-exactly, linearity, and
    spot checks of single outputs by Horner evaluation X_j = f(omega^j)."""
    import torch
    o = pyref
    n = 1 << k
    w, winv = o.fr_omega(k), pow(o.fr_omega(k), -1, o.R)
    x = rand_fr_gpu(n, 5)
    y = x.clone()
    h.best_fft(y, o.fr_array([w])[0], k)
    # spot checks against the oracle's Horner evaluation (bounded: 2 points)
    xh = x.cpu().numpy().view(np.uint64)
    yh = y.cpu().numpy().view(np.uint64)
    if k <= 23:
        for j in (1, n - 3):
            pt = o.fr_array([pow(w, j, o.R)])[0]
            assert np.array_equal(cref.fr_horner(xh, pt), yh[j]), j
    # linearity: NTT(x + x') == NTT(x) + NTT(x')  (checked on a slice with the oracle's field add)
    x2 = rand_fr_gpu(n, 6)
    y2 = x2.clone()
    h.best_fft(y2, o.fr_array([w])[0], k)
    sl = slice(12345, 12345 + 4096)
    xs = torch.from_numpy(cref.fr_add(xh, x2.cpu().numpy().view(np.uint64)).view(np.int64)).cuda()
    h.best_fft(xs, o.fr_array([w])[0], k)
    assert np.array_equal(xs.cpu().numpy().view(np.uint64)[sl], cref.fr_add(yh[sl], y2.cpu().numpy().view(np.uint64)[sl]))
    # round trip through the fused inverse (EvaluationDomain::ifft)
    from halo2_experiments_amd import _lib
    import ctypes
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    _lib.check(_lib.load().hm_ifft_bn256_fr_dev(ctypes.c_void_p(y.data_ptr()), _ptr(fr_words(winv)), k,
                                                 _ptr(fr_words(pow(n, -1, o.R))), ctypes.c_void_p(_stream_ptr(y))))
    torch.cuda.synchronize()
    assert torch.equal(y, x)


@pytest.mark.parametrize("k", [11, 16, 21, 22, 24, 26])
def test_geometric_series_known_answer_needs_no_oracle(pyref, k):
    """A known answer that neither oracle takes part in: for a_i = c^i the transform is the geometric series
    X_j = sum_i (c w^j)^i = (c^n - 1) / (c w^j - 1), checked with Python integers at 256 random output indices (and the
    library-made input at 64 random input indices) -- one-, two- and three-pass plans up to 2^26."""
    import ctypes
    import torch
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    R = pyref.R
    n = 1 << k
    w = pyref.fr_omega(k)
    c = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % R
    a = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    _lib.check(_lib.load().hm_fr_powers_dev(ctypes.c_void_p(a.data_ptr()), n, _ptr(fr_words(c)), ctypes.c_void_p(_stream_ptr(a))))
    rng = np.random.default_rng(k)
    idx_in = torch.from_numpy(rng.integers(0, n, 64)).cuda()
    got_in = a[idx_in].cpu().numpy().view(np.uint64)
    for i, row in zip(idx_in.cpu().tolist(), got_in):
        assert np.array_equal(row, fr_words(pow(c, i, R))), i
    h.best_fft(a, fr_words(w), k)
    idx = np.concatenate([[0, 1, n - 1, n // 2], rng.integers(0, n, 252)])
    got = a[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint64)
    top = (pow(c, n, R) - 1) % R
    for j, row in zip(idx.tolist(), got):
        want = top * pow((c * pow(w, j, R) - 1) % R, -1, R) % R
        assert np.array_equal(row, fr_words(want)), (k, j)


@pytest.mark.parametrize("k", [11, 17, 18])
def test_evaluation_domain_known_answers_need_no_oracle(pyref, k):
    """The same closed form for the EvaluationDomain steps at the circuits' shapes (k = 11 / 17 / 18, j = 7): the coefficients
    c^i evaluated on the zeta-coset of the extended domain are ((c x)^n - 1) / (c x - 1) at x = zeta * w_ext^j
    (coeff_to_extended), their values on the n-th roots of unity come back to c^i (lagrange_to_coeff), and
    extended_to_coeff returns c^i followed by zeros -- Python integers only, 128 random indices each."""
    import ctypes
    import torch
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    R = pyref.R
    d = EvaluationDomain(j=7, k=k)
    n, en = d.n, d.extended_len()
    c = 0xFEDCBA9876543210FEDCBA9876543210FEDCBA98 % R
    coeff = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    _lib.check(_lib.load().hm_fr_powers_dev(ctypes.c_void_p(coeff.data_ptr()), n, _ptr(fr_words(c)), ctypes.c_void_p(_stream_ptr(coeff))))
    rng = np.random.default_rng(100 + k)
    top = lambda x: (pow(c * x % R, n, R) - 1) * pow((c * x - 1) % R, -1, R) % R        # f(x) = sum_i (c x)^i
    ext = d.coeff_to_extended(coeff)
    idx = np.concatenate([[0, 1, en - 1], rng.integers(0, en, 125)])
    got = ext[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint64)
    for j, row in zip(idx.tolist(), got):
        assert np.array_equal(row, fr_words(top(d.g_coset * pow(d.extended_omega, j, R) % R))), (k, j)
    # values on the n-th roots of unity -> lagrange_to_coeff -> c^i
    evals = coeff.clone()
    h.best_fft(evals, fr_words(d.omega), k)                                                # f(w^j), checked by the test above
    back = d.lagrange_to_coeff(evals)
    idx = rng.integers(0, n, 128)
    got = back[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint64)
    for i, row in zip(idx.tolist(), got):
        assert np.array_equal(row, fr_words(pow(c, i, R))), (k, i)
    # extended_to_coeff of the coset evaluations: the coefficients again, zeros above n
    out = d.extended_to_coeff(ext.clone())
    idx = rng.integers(0, out.shape[0], 128)
    got = out[torch.from_numpy(idx).cuda()].cpu().numpy().view(np.uint64)
    for i, row in zip(idx.tolist(), got):
        assert np.array_equal(row, fr_words(pow(c, i, R) if i < n else 0)), (k, i)


def test_evaluation_domain_steps(cref, pyref):
    """lagrange_to_coeff / coeff_to_extended / extended_to_coeff against the oracle's compositions."""
    import torch
    o = pyref
    d = EvaluationDomain(j=7, k=10)
    n, en = d.n, d.extended_len()
    a = rand_fr_gpu(n, 42)
    ah = a.cpu().numpy().view(np.uint64)
    # lagrange_to_coeff = best_fft(omega_inv) then * n^-1
    exp = cref.fr_mul(cref.best_fft(ah, fr_words(d.omega_inv), d.k, 4), np.tile(fr_words(d.ifft_divisor), (n, 1)))
    coeff = d.lagrange_to_coeff(a.clone())
    assert np.array_equal(coeff.cpu().numpy().view(np.uint64), exp)
    # coeff_to_extended = zero-pad, a[i] *= zeta^(i%3), best_fft(extended_omega)
    pad = np.zeros((en, 4), dtype=np.uint64)
    pad[:n] = exp
    zpow = [1, d.g_coset, d.g_coset * d.g_coset % o.R]
    pat = np.stack([fr_words(zpow[i % 3]) for i in range(en)])
    exp_ext = cref.best_fft(cref.fr_mul(pad, pat), fr_words(d.extended_omega), d.extended_k, 4)
    ext = d.coeff_to_extended(coeff)
    assert np.array_equal(ext.cpu().numpy().view(np.uint64), exp_ext)
    # internal=True: the same evaluations times 32 (the form hm_graph_evaluate_flags_dev loads without a conversion product)
    ext32 = d.coeff_to_extended(coeff, internal=True)
    assert np.array_equal(ext32.cpu().numpy().view(np.uint64), cref.fr_mul(exp_ext, np.tile(fr_words(32), (en, 1))))
    # extended_to_coeff inverts it (and truncates to n * (j - 1) rows)
    back = d.extended_to_coeff(ext.clone())
    bh = back.cpu().numpy().view(np.uint64)
    assert bh.shape[0] == n * 6
    assert np.array_equal(bh[:n], exp) and not bh[n:].any()


@pytest.mark.parametrize("j,k", [(3, 9), (4, 8), (7, 10), (9, 7)])
def test_divide_by_vanishing_poly(cref, pyref, j, k):
    """EvaluationDomain::divide_by_vanishing_poly: a[i] *= 1 / ((zeta * omega_ext^i)^n - 1), a pattern of period
    2^(extended_k - k) -- elementwise against the oracle, single and batched; and the identity it exists for: the
    extended evaluations of g(X) * (X^n - 1), divided, are the extended evaluations of g."""
    import torch
    o = pyref
    d = EvaluationDomain(j=j, k=k)
    n, en, R = d.n, d.extended_len(), o.R
    period = en // n
    t_inv = [pow((pow(d.g_coset * pow(d.extended_omega, i, R) % R, n, R) - 1) % R, -1, R) for i in range(period)]
    a = rand_fr_gpu(2 * en, 77 + j).reshape(2, en, 4)
    ah = a.cpu().numpy().view(np.uint64).reshape(2 * en, 4)
    pat = np.stack([fr_words(t_inv[i % period]) for i in range(2 * en)])
    d.divide_by_vanishing_poly(a)
    assert np.array_equal(a.cpu().numpy().view(np.uint64).reshape(2 * en, 4), cref.fr_mul(ah, pat))
    # g of degree < n; f = g * (X^n - 1) evaluated on the coset point by point is g(x) * (x^n - 1)
    g = rand_fr_gpu(n, 99 + k)
    g_ext = d.coeff_to_extended(g)
    vanish = np.stack([fr_words((pow(d.g_coset * pow(d.extended_omega, i, R) % R, n, R) - 1) % R) for i in range(en)])
    f_ext = torch.from_numpy(cref.fr_mul(g_ext.cpu().numpy().view(np.uint64), vanish).view(np.int64)).cuda()
    d.divide_by_vanishing_poly(f_ext)
    assert torch.equal(f_ext, g_ext)


def test_batched_transforms_equal_single_ones(pyref):
    """(batch, n, 4) tensors go through one set of launches and must equal per-polynomial calls."""
    import torch
    d = EvaluationDomain(j=7, k=12)
    a = rand_fr_gpu(5 * d.n, 4242).reshape(5, d.n, 4)
    single = torch.stack([d.lagrange_to_coeff(a[i].clone()) for i in range(5)])
    batched = d.lagrange_to_coeff(a.clone())
    assert torch.equal(single, batched)
    ext_single = torch.stack([d.coeff_to_extended(single[i]) for i in range(5)])
    ext_batched = d.coeff_to_extended(batched)
    assert torch.equal(ext_single, ext_batched)
    back = d.extended_to_coeff(ext_batched.clone())
    assert torch.equal(back[:, : d.n], batched) and not back[:, d.n:].any()


@pytest.mark.parametrize("k,j", [(3, 7), (6, 3), (8, 7), (9, 2), (11, 4), (13, 10), (14, 7), (18, 7), (19, 7), (20, 3)])
def test_coeff_to_extended_never_materialises_the_zero_part(k, j):
    """hm_coeff_to_extended_bn256_fr_dev (compact input, first log_z butterfly stages skipped) against the
    plain composition it replaces: zero-pad, coset shift fused into an ordinary in-place NTT.  Covers the
    single-pass fallback, two- and three-pass plans, log_z = 0 .. 4, batches, and the coset-less form."""
    import ctypes
    import torch
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    from halo2_experiments_amd.domain import FR_MODULUS
    d = EvaluationDomain(j=j, k=k)
    n, en = d.n, d.extended_len()
    batch = 3 if k <= 14 else 2
    a = rand_fr_gpu(batch * n, 7000 + 31 * k + j).reshape(batch, n, 4)
    lib = _lib.load()
    coset = np.concatenate([fr_words(1), fr_words(d.g_coset), fr_words(d.g_coset * d.g_coset % FR_MODULUS)])
    for with_coset in (True, False):
        padded = torch.zeros((batch, en, 4), dtype=a.dtype, device=a.device)
        padded[:, :n] = a
        _lib.check(lib.hm_ntt_batch_bn256_fr_dev(ctypes.c_void_p(padded.data_ptr()), batch, _ptr(fr_words(d.extended_omega)),
                                                  d.extended_k, None, _ptr(coset) if with_coset else None,
                                                  ctypes.c_void_p(_stream_ptr(padded))))
        ext = torch.full((batch, en, 4), -1, dtype=a.dtype, device=a.device)      # garbage: the call must overwrite all of it
        _lib.check(lib.hm_coeff_to_extended_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(ext.data_ptr()), batch,
                                                          _ptr(fr_words(d.extended_omega)), d.k, d.extended_k,
                                                          _ptr(coset) if with_coset else None, ctypes.c_void_p(_stream_ptr(ext))))
        torch.cuda.synchronize()
        assert torch.equal(ext, padded), (k, j, with_coset)
    if k == 8:
        assert lib.hm_coeff_to_extended_bn256_fr_dev(None, None, 1, _ptr(fr_words(d.extended_omega)), 8, 11, None, None) == -1
        assert lib.hm_coeff_to_extended_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(ext.data_ptr()), 1,
                                                     _ptr(fr_words(d.extended_omega)), 12, 11, None, None) == -1


@pytest.mark.parametrize("k,j", [(8, 7), (13, 7), (19, 7), (18, 3)])
def test_extended_to_coeff_fused_into_the_last_pass(cref, k, j):
    """hm_extended_to_coeff_bn256_fr_dev (ifft divisor x zeta^-(i % 3) folded into the last NTT pass) against
    the composition it replaces -- scaled inverse NTT, then hm_fr_distribute_powers_dev per polynomial -- on
    arbitrary (not band-limited) extended arrays, one-, two- and three-pass plans, batch 2; and against the
    oracle's composition at the smallest size."""
    import ctypes
    import torch
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    from halo2_experiments_amd.domain import FR_MODULUS
    d = EvaluationDomain(j=j, k=k)
    en, ek = d.extended_len(), d.extended_k
    lib = _lib.load()
    x = rand_fr_gpu(2 * en, 9100 + k).reshape(2, en, 4)
    c3 = np.concatenate([fr_words(1), fr_words(d.g_coset_inv), fr_words(d.g_coset_inv * d.g_coset_inv % FR_MODULUS)])
    ref = x.clone()
    _lib.check(lib.hm_ntt_batch_bn256_fr_dev(ctypes.c_void_p(ref.data_ptr()), 2, _ptr(fr_words(d.extended_omega_inv)), ek,
                                              _ptr(fr_words(d.extended_ifft_divisor)), None, ctypes.c_void_p(_stream_ptr(ref))))
    for b in range(2):
        _lib.check(lib.hm_fr_distribute_powers_dev(ctypes.c_void_p(ref[b].data_ptr()), en, _ptr(c3), ctypes.c_void_p(_stream_ptr(ref))))
    got = x.clone()
    out = d.extended_to_coeff(got)
    torch.cuda.synchronize()
    assert torch.equal(got, ref)
    assert out.shape[1] == d.n * (j - 1)
    if k == 8:
        xh = x[1].cpu().numpy().view(np.uint64)
        inv = cref.fr_mul(cref.best_fft(xh, fr_words(d.extended_omega_inv), ek, 4), np.tile(fr_words(d.extended_ifft_divisor), (en, 1)))
        zi = [1, d.g_coset_inv, d.g_coset_inv * d.g_coset_inv % FR_MODULUS]
        exp = cref.fr_mul(inv, np.stack([fr_words(zi[i % 3]) for i in range(en)]))
        assert np.array_equal(got[1].cpu().numpy().view(np.uint64), exp)


def test_concurrent_transforms_on_two_streams(pyref):
    """Every *_dev NTT entry point is asynchronous on the caller's stream and may run concurrently with
    another call on another stream: multi-pass transforms of different sizes and flavours (fused inverse,
    coset NTT, extending transform) are interleaved on two streams -- the first use of each twiddle table
    included -- and must equal the same calls issued one after the other."""
    import ctypes
    import torch
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import _ptr
    from halo2_experiments_amd.domain import FR_MODULUS
    lib = _lib.load()
    da, db = EvaluationDomain(j=7, k=16), EvaluationDomain(j=4, k=13)       # extended: 2^19 (2 passes), 2^15 (2 passes)
    big = EvaluationDomain(j=3, k=22)                                       # 2^22: three passes
    xa = rand_fr_gpu(4 * da.n, 1).reshape(4, da.n, 4)
    xb = rand_fr_gpu(6 * db.n, 2).reshape(6, db.n, 4)
    xc = rand_fr_gpu(big.n, 3)

    def work_a(t):         # lagrange_to_coeff -> coeff_to_extended -> extended_to_coeff
        c = da.lagrange_to_coeff(t.clone())
        e = da.coeff_to_extended(c)
        return c, e, da.extended_to_coeff(e.clone()).clone()

    def work_b(t):
        c = db.lagrange_to_coeff(t.clone())
        e = db.coeff_to_extended(c)
        return c, e, db.extended_to_coeff(e.clone()).clone()

    def work_c(t):
        y = t.clone()
        h.best_fft(y, fr_words(big.omega), big.k)
        return (y,)

    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    conc = []
    for it in range(3):                       # round 0 builds every twiddle table while the other stream runs
        with torch.cuda.stream(sa):
            ra = work_a(xa)
        with torch.cuda.stream(sb):
            rb = work_b(xb)
        with torch.cuda.stream(sa):
            rc = work_c(xc)
        with torch.cuda.stream(sb):
            ra2 = work_a(xa)
        conc.append((ra, rb, rc, ra2))
    torch.cuda.synchronize()
    ea, eb, ec = work_a(xa), work_b(xb), work_c(xc)
    torch.cuda.synchronize()
    for ra, rb, rc, ra2 in conc:
        for got, exp in ((ra, ea), (rb, eb), (rc, ec), (ra2, ea)):
            for g_, e_ in zip(got, exp):
                assert torch.equal(g_, e_)


# ---- the extended domain one coset of <omega> at a time (hm_coeff_to_coset / hm_coset_to_coeff) ----------------------
@pytest.mark.parametrize("k,j", [(3, 3), (5, 6), (9, 6), (11, 6), (12, 4), (13, 6), (17, 6), (18, 6)])
def test_coeff_to_coset_is_a_residue_class_of_rows_of_coeff_to_extended(pyref, k, j):
    """Row t of coset c equals row E t + c of the extended array, bit for bit (one-pass, two-pass plans; batch; the
    evaluator's internal column form; in place), and at the smallest size the oracle's direct evaluation."""
    import torch
    from halo2_experiments_amd.domain import EvaluationDomain
    dom = EvaluationDomain(j, k)
    n, e = dom.n, dom.num_cosets()
    a = rand_fr_gpu(3 * n, 4100 + k).reshape(3, n, 4)
    for internal in (False, True):
        ext = dom.coeff_to_extended(a, internal=internal)
        for c in range(e):
            got = dom.coeff_to_coset(a, c, internal=internal)
            assert bool((got == ext[:, c::e]).all()), (k, c, internal)
    one = dom.coeff_to_coset(a[1], e - 1)
    assert bool((one == dom.coeff_to_extended(a[1])[e - 1::e]).all())
    b = a.clone()
    assert dom.coeff_to_coset(b, 1, out=b) is b and bool((b == dom.coeff_to_extended(a)[:, 1::e]).all())     # in place
    if k == 3:
        coeffs = pyref.fr_from_array(a[0].cpu().numpy().view(np.uint64))
        for c in range(e):
            sh = dom.coset_shift(c)
            want = [sum(v * pow(sh * pow(dom.omega, t, pyref.R) % pyref.R, i, pyref.R) for i, v in enumerate(coeffs)) % pyref.R for t in range(n)]
            assert pyref.fr_from_array(dom.coeff_to_coset(a[0], c).cpu().numpy().view(np.uint64)) == want


@pytest.mark.parametrize("k,j", [(3, 3), (5, 6), (11, 6), (12, 5), (16, 6), (18, 6)])
def test_cosets_recombine_to_what_extended_to_coeff_returns(k, j):
    """Values of a polynomial of degree < E n on the E cosets -> hm_coset_to_coeff each -> E-term linear combinations: the same
    words as extended_to_coeff of the whole array (all E pieces compared, not only the quotient's j - 1)."""
    import torch
    from halo2_experiments_amd.domain import EvaluationDomain
    dom = EvaluationDomain(j, k)
    n, e = dom.n, dom.num_cosets()
    h_ext = rand_fr_gpu(2 * e * n, 4300 + k).reshape(2, e * n, 4)
    parts = [dom.coset_to_partial(h_ext[:, c::e].contiguous(), c) for c in range(e)]
    whole = h_ext.clone()
    dom.extended_to_coeff(whole)                                          # in place: `whole` now holds all E n coefficients
    for b in range(2):
        got = dom.combine_cosets([p[b] for p in parts], pieces=e)
        assert bool((got == whole[b]).all()), (k, b)
    short = dom.combine_cosets([p[0] for p in parts])
    assert short.shape[0] == n * (j - 1) and bool((short == whole[0][: n * (j - 1)]).all())
    with pytest.raises(ValueError):
        dom.combine_cosets(parts[:-1])
    with pytest.raises(ValueError):
        dom.coeff_to_coset(h_ext[0, :n].contiguous(), e)


def test_coset_tables_survive_many_shifts_and_two_streams():
    """More distinct shifts than the table cache holds (the oldest goes, behind a device synchronisation), and the same
    shift first used on one stream, then on another (published behind its event)."""
    import torch
    from halo2_experiments_amd.domain import EvaluationDomain
    lib = _lib.load()
    k = 10
    dom = EvaluationDomain(6, k)
    a = rand_fr_gpu(dom.n, 4500)
    want = dom.coeff_to_extended(a)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    torch.cuda.synchronize()
    with torch.cuda.stream(s1):
        x1 = dom.coeff_to_coset(a, 5)
    with torch.cuda.stream(s2):
        x2 = dom.coeff_to_coset(a, 5)
    torch.cuda.synchronize()
    assert bool((x1 == want[5::8]).all()) and bool((x2 == want[5::8]).all())
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    from halo2_experiments_amd.domain import fr_words
    out = torch.empty_like(a)
    for i in range(60):                                                    # 60 shifts > the 48 tables kept
        _lib.check(lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), 1, _ptr(fr_words(dom.omega)),
                                                      k, _ptr(fr_words(1000 + i)), 0, ctypes.c_void_p(_stream_ptr(a))))
    assert bool((dom.coeff_to_coset(a, 5) == want[5::8]).all())             # rebuilt after eviction


def test_coset_table_cache_is_bounded_by_bytes():
    """The power tables of the coset transforms are kept between calls (LRU): at most 48 of them AND at most 2 GiB -- 48 tables of
    2^24 elements would hold 24 GiB of HBM until hm_shutdown.  hm_get_stats reports what is held.  One call's own working
    set may exceed the cap (16 cosets of 2^23 = 4 GiB: the call's tables are pinned while it gathers them); the next call trims it."""
    import torch
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    lib = _lib.load()

    def held():
        st = _lib.Stats()
        _lib.check(lib.hm_get_stats(ctypes.byref(st)))
        return st.coset_table_bytes, st.coset_tables

    cap = 2 << 30
    k = 22                                                                  # 128 MiB per table: the byte cap binds at 16 tables
    from halo2_experiments_amd.domain import FR_MODULUS, FR_ROOT_OF_UNITY
    a = rand_fr_gpu(1 << k, 4800)
    omega = fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - k), FR_MODULUS))
    out = torch.empty_like(a)
    first = None
    for i in range(20):
        _lib.check(lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), 1, _ptr(omega), k,
                                                      _ptr(fr_words(2000 + i)), 0, ctypes.c_void_p(_stream_ptr(a))))
        if i == 0:
            first = out.clone()
        b, cnt = held()
        assert b <= cap and cnt <= 48 and b >= (32 << k), (i, b, cnt)
    assert held()[0] == cap                                                 # 16 tables of 128 MiB: full, the four oldest went
    _lib.check(lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), 1, _ptr(omega), k,
                                                  _ptr(fr_words(2000)), 0, ctypes.c_void_p(_stream_ptr(a))))
    assert bool((out == first).all())                                       # shift 2000 was evicted and rebuilt: same values
    # one call that needs more than the cap by itself: 16 cosets at 2^23 (16 x 256 MiB)
    k2 = 23
    a2 = rand_fr_gpu(1 << k2, 4801)
    omega2 = fr_words(pow(FR_ROOT_OF_UNITY, 1 << (28 - k2), FR_MODULUS))
    shifts = np.stack([fr_words(3000 + i) for i in range(16)])
    out2 = torch.empty((16, 1 << k2, 4), dtype=torch.int64, device="cuda")
    _lib.check(lib.hm_coeff_to_cosets_bn256_fr_dev(ctypes.c_void_p(a2.data_ptr()), ctypes.c_void_p(out2.data_ptr()), 1, _ptr(omega2), k2,
                                                   _ptr(shifts), 16, 0, ctypes.c_void_p(_stream_ptr(a2))))
    assert held()[0] >= 16 * (32 << k2)                                     # soft for the call's own tables
    one = torch.empty_like(a2)
    for i in (0, 15):                                                       # ... whose results are those of the one-coset call
        _lib.check(lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(a2.data_ptr()), ctypes.c_void_p(one.data_ptr()), 1, _ptr(omega2), k2,
                                                      _ptr(fr_words(3000 + i)), 0, ctypes.c_void_p(_stream_ptr(a2))))
        assert bool((one == out2[i]).all()), i
    _lib.check(lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), 1, _ptr(omega), k,
                                                  _ptr(fr_words(2000)), 0, ctypes.c_void_p(_stream_ptr(a))))
    assert held()[0] <= cap and bool((out == first).all())                  # the next call that builds a table trims the cache


def test_twiddle_table_cache_is_bounded(cref):
    """One set of twiddle tables per (omega, log_n), kept between calls: with the direct inter-pass table (75 MB at 2^21) a caller that walks
    through many roots of unity would pile them up -- LRU, at most 64 sets and 1 GiB; results after an eviction are those before it."""
    from halo2_experiments_amd.domain import FR_MODULUS, FR_ROOT_OF_UNITY
    lib = _lib.load()

    def held():
        st = _lib.Stats()
        _lib.check(lib.hm_get_stats(ctypes.byref(st)))
        return st.ntt_table_bytes, st.ntt_tables

    k = 21
    w = pow(FR_ROOT_OF_UNITY, 1 << (28 - k), FR_MODULUS)
    a0 = rand_fr_gpu(1 << k, 4900)
    first = a0.clone()
    h.best_fft(first, fr_words(w), k)
    for e in range(3, 3 + 2 * 18, 2):                                       # 18 more primitive 2^21-th roots: w^odd
        a = a0.clone()
        h.best_fft(a, fr_words(pow(w, e, FR_MODULUS)), k)
        b, cnt = held()
        assert 0 < b <= (1 << 30) and cnt <= 64, (e, b, cnt)
    assert held()[0] > (1 << 30) - (80 << 20)                               # thirteen 75 MB sets fit 1 GiB; the first ones went
    again = a0.clone()
    h.best_fft(again, fr_words(w), k)                                       # rebuilt
    assert bool((again == first).all())
    small = a0[: 1 << 12].clone()
    exp = cref.best_fft(small.cpu().numpy().view(np.uint64), fr_words(pow(w, 1 << 9, FR_MODULUS)), 12)
    h.best_fft(small, fr_words(pow(w, 1 << 9, FR_MODULUS)), 12)
    assert np.array_equal(small.cpu().numpy().view(np.uint64), exp)


def test_coset_transforms_at_the_smallest_sizes(pyref):
    """log_n = 0 and 1 through the C entry points themselves (EvaluationDomain starts at k = 1): one element is its own
    transform on any coset; two elements a0 + a1 X at shift and -shift."""
    import torch
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    lib = _lib.load()
    R = pyref.R
    shift = 0x1234567890ABCDEF1234567890ABCDEF % R
    a = rand_fr_gpu(2, 4700)
    av = pyref.fr_from_array(a.cpu().numpy().view(np.uint64))
    out = torch.empty_like(a)
    one = fr_words(1)
    _lib.check(lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), 2, _ptr(one), 0, _ptr(fr_words(shift)),
                                                  0, ctypes.c_void_p(_stream_ptr(a))))
    assert bool((out == a).all())                                           # two 1-element arrays
    _lib.check(lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), 1, _ptr(fr_words(R - 1)), 1,
                                                  _ptr(fr_words(shift)), 0, ctypes.c_void_p(_stream_ptr(a))))
    assert pyref.fr_from_array(out.cpu().numpy().view(np.uint64)) == [(av[0] + av[1] * shift) % R, (av[0] - av[1] * shift) % R]
    back = out.clone()
    _lib.check(lib.hm_coset_to_coeff_bn256_fr_dev(ctypes.c_void_p(back.data_ptr()), 1, _ptr(fr_words(R - 1)), 1, _ptr(fr_words(pow(2, -1, R))),
                                                  _ptr(fr_words(pow(shift, -1, R))), ctypes.c_void_p(_stream_ptr(back))))
    assert bool((back == a).all())                                          # the inverse route returns the coefficients
    assert lib.hm_coeff_to_coset_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(out.data_ptr()), 0, _ptr(one), 1, _ptr(one), 0, None) == 0


@pytest.mark.parametrize("k,j", [(4, 6), (10, 6), (13, 7), (16, 4), (18, 6)])
def test_a_polynomial_below_q_n_is_recovered_from_any_q_cosets(pyref, k, j):
    """The quotient of a satisfied circuit has fewer than n (j - 1) coefficients, so its values on j - 1 of the E cosets
    determine it: combine_cosets(cosets=[...]) solves the (j - 1) x (j - 1) Vandermonde system per residue class.  A random
    polynomial with q n coefficients (as q pieces of n) is evaluated on the cosets piece by piece -- value on coset c =
    sum_t u_c^t * coeff_to_coset(piece_t, c), u_c = shift_c^n -- and comes back word for word from several subsets; with one coset
    more than needed the extra piece is zero; all E cosets agree with extended_to_coeff (the existing test)."""
    import torch
    dom = EvaluationDomain(j, k)
    n, e, q = dom.n, dom.num_cosets(), dom.min_cosets()
    R = pyref.R
    assert q == j - 1 and q <= e
    pieces = rand_fr_gpu(q * n, 5100 + k).reshape(q, n, 4)

    def values_on(c):
        u = pow(dom.coset_shift(c), n, R)
        terms = [dom.coeff_to_coset(pieces[t], c) for t in range(q)]
        return h.linear_combination(terms, np.stack([fr_words(pow(u, t, R)) for t in range(q)]))

    rng = np.random.default_rng(k)
    subsets = [list(range(q)), sorted(rng.choice(e, q, replace=False).tolist(), reverse=True)]
    if q < e:
        subsets.append(list(range(e - q, e)))
    for cosets in subsets:
        parts = [dom.coset_to_partial(values_on(c), c) for c in cosets]
        got = dom.combine_cosets(parts, cosets=cosets)
        assert got.shape == (q * n, 4) and bool((got == pieces.reshape(q * n, 4)).all()), (k, cosets)
    if q < e:
        cosets = list(range(q + 1))
        parts = [dom.coset_to_partial(values_on(c), c) for c in cosets]
        got = dom.combine_cosets(parts, cosets=cosets)
        assert bool((got[: q * n] == pieces.reshape(q * n, 4)).all()) and not got[q * n:].any()
    with pytest.raises(ValueError):
        dom.combine_cosets([pieces[0]], cosets=[e])
    with pytest.raises(ValueError):
        dom.combine_cosets([pieces[0], pieces[1]], cosets=[1, 1])
