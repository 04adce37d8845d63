"""GPU suite: best_multiexp on the MI355X through the C ABI vs the oracle (bit-exact after affine
normalisation), the committed golden vectors, the edge cases of the domain, and known-answer /
additivity properties at the benchmark's full size."""
import ctypes

import numpy as np
import pytest

import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from conftest import g1_equal

pytestmark = pytest.mark.gpu


def rand_fr_gpu(n, seed):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)


def test_golden_vectors(golden):
    g = golden["msm"]
    for name in g["names"]:
        got = h.best_multiexp(g[f"{name}_s"], g[f"{name}_b"])
        assert g1_equal(got, g[f"{name}_r"]), name


def test_empty_and_identity_results(golden):
    g = golden["msm"]
    out = h.best_multiexp(np.zeros((0, 4), dtype=np.uint64), np.zeros((0, 8), dtype=np.uint64))
    assert not out.any()
    assert not h.best_multiexp(g["n33_zero_s"], g["n33_zero_b"]).any()
    xy = np.ones(8, dtype=np.uint64)
    is_id = ctypes.c_int(0)
    P64 = ctypes.POINTER(ctypes.c_uint64)
    s, b = np.ascontiguousarray(g["pmone_s"]), np.ascontiguousarray(g["pmone_b"])
    _lib.check(_lib.load().hm_msm_bn256_g1(s.ctypes.data_as(P64), b.ctypes.data_as(P64), s.shape[0], xy.ctypes.data_as(P64),
                                           ctypes.byref(is_id)))
    assert is_id.value == 1 and not xy.any()


@pytest.mark.parametrize("window", [0, 4, 7, 11, 13, 16, 17, 19, 22])
def test_every_window_size(golden, window):
    g = golden["msm"]
    lib = _lib.load()
    _lib.check(lib.hm_msm_set_window(window))
    try:
        for name in ("n255_uniform", "n1024_prover", "n33_edge", "same", "pm", "ident", "n1024_rminus1"):
            assert g1_equal(h.best_multiexp(g[f"{name}_s"], g[f"{name}_b"]), g[f"{name}_r"]), (window, name)
    finally:
        lib.hm_msm_set_window(0)


@pytest.mark.parametrize("log_n,kind", [(12, "uniform"), (14, "prover"), (16, "uniform"), (16, "small"), (17, "one")])
def test_matches_oracle_mid_sizes(cref, pyref, log_n, kind):
    """Random bases [k_i]G from the fixed-base kernel; scalars of several distributions, with an
    identity base and duplicate points planted; oracle = the C restatement (8 threads)."""
    o = pyref
    n = 1 << log_n
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 3 * log_n), gen).cpu().numpy().view(np.uint64).copy()
    bases[2] = 0
    bases[5] = bases[6]
    if kind == "uniform":
        s = rand_fr_gpu(n, log_n).cpu().numpy().view(np.uint64)
    else:
        base = o.fr_array(o.rand_scalars(4096, log_n, kind))
        s = np.tile(base, (n // 4096, 1))
    exp = cref.g1_to_affine(cref.best_multiexp(s, bases, 8))[0]
    assert g1_equal(h.best_multiexp(s, bases), exp)


def test_fixed_base_kernel_matches_oracle(cref):
    gen = cref.g1_generator()
    ks = rand_fr_gpu(16, 5)
    ks[3] = 0
    pts = h.g1_fixed_base_mul(ks, gen).cpu().numpy().view(np.uint64)
    kh = ks.cpu().numpy().view(np.uint64)
    for i in range(16):
        assert np.array_equal(pts[i], cref.g1_mul(kh[i], gen)), i
    assert not pts[3].any()


def test_registered_bases_offsets_and_device_scalars(cref, golden):
    import torch
    g = golden["msm"]
    s, b, = g["n1024_uniform_s"], g["n1024_uniform_b"]
    hd = h.register_bases(b)
    try:
        assert g1_equal(h.best_multiexp(s, hd), g["n1024_uniform_r"])
        lo, cnt = 100, 500
        exp = cref.g1_to_affine(cref.best_multiexp(s[lo:lo + cnt], b[lo:lo + cnt], 4))[0]
        assert g1_equal(h.best_multiexp(s[lo:lo + cnt], hd, offset=lo), exp)
        sd = torch.from_numpy(s[lo:lo + cnt].view(np.int64).copy()).cuda()
        assert g1_equal(h.best_multiexp(sd, hd, offset=lo), exp)
        with pytest.raises(_lib.Halo2Mi355xError):
            h.best_multiexp(s, hd, offset=1)            # offset + n exceeds the set
    finally:
        h.release_bases(hd)
    with pytest.raises(_lib.Halo2Mi355xError):
        h.best_multiexp(s, hd)                          # released handle


def test_hot_bucket_columns_are_split(cref):
    """Constant columns (every scalar equal) put all points in one bucket per window: must stay
    correct and must not serialise (the tasks statistic shows the split)."""
    n = 1 << 16
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 9), gen)
    hd = h.register_bases(bases)
    try:
        one = rand_fr_gpu(1, 10).repeat(n, 1).contiguous()
        got = h.best_multiexp(one, hd)
        st = h.msm_stats()
        assert st["tasks"] > 1000
        bh = bases.cpu().numpy().view(np.uint64)
        # sum of all bases times the scalar, via the oracle
        ones = np.tile(cref.fr_to_mont(np.array([[1, 0, 0, 0]], dtype=np.uint64)), (n, 1))
        tot = cref.best_multiexp(ones, bh, 8)
        exp = cref.g1_mul(one[0].cpu().numpy().view(np.uint64), cref.g1_to_affine(tot)[0])
        assert g1_equal(got, exp)
    finally:
        h.release_bases(hd)


@pytest.mark.parametrize("log_n", [20, 24])
def test_full_size_kzg_known_answer_and_additivity(cref, pyref, log_n):
    """The KZG consistency check of SURVEY.md §8c at the benchmark's size: with the SRS g_i = [x^i]G,
    MSM(coeffs(f), g) == [f(x)]G; plus MSM(whole) == MSM(first half) + MSM(second half)."""
    import torch
    from halo2_experiments_amd.sharding import g1_sum
    o = pyref
    n = 1 << log_n
    x = o.fr_array([0x48324D4933353558])[0]
    powers = torch.from_numpy(cref.fr_powers(x, n).view(np.int64)).cuda()       # x^i, oracle-side Fr work
    srs = h.g1_fixed_base_mul(powers, cref.g1_generator())                     # ParamsKZG::setup's g
    hd = h.register_bases(srs)
    try:
        f = rand_fr_gpu(n, 2024)
        got = h.best_multiexp(f, hd)
        fx = cref.fr_horner(f.cpu().numpy().view(np.uint64), x)
        assert g1_equal(got, cref.g1_mul(fx, cref.g1_generator()))
        half = n // 2
        lo = h.best_multiexp(f[:half].contiguous(), hd)
        hi = h.best_multiexp(f[half:].contiguous(), hd, offset=half)
        assert np.array_equal(g1_sum(np.stack([lo, hi])), got)
    finally:
        h.release_bases(hd)


def test_commit_lagrange_equals_commit(cref, pyref):
    """MSM and NTT together, as ParamsKZG::commit / commit_lagrange use them: committing to the
    evaluations with g_lagrange equals committing to the coefficients with g, where
    g_lagrange = n^-1 * NTT_{omega^-1}-transform of g in the exponent -- here checked through
    scalars: MSM(evals, [L_i(x)]G) == MSM(coeffs, [x^i]G)."""
    import torch
    o = pyref
    k, n = 10, 1 << 10
    xv = 987654321
    w = o.fr_omega(k)
    f = o.rand_scalars(n, 31337)
    evals = o.ntt_fast(f, w)                                                  # f(omega^j)
    # Lagrange basis at x: L_j(x) = (x^n - 1) / n * omega^j / (x - omega^j)
    xn1 = (pow(xv, n, o.R) - 1) * pow(n, -1, o.R) % o.R
    lag = [xn1 * pow(w, j, o.R) % o.R * pow((xv - pow(w, j, o.R)) % o.R, -1, o.R) % o.R for j in range(n)]
    gen = cref.g1_generator()
    g = h.g1_fixed_base_mul(torch.from_numpy(o.fr_array([pow(xv, i, o.R) for i in range(n)]).view(np.int64)).cuda(), gen)
    gl = h.g1_fixed_base_mul(torch.from_numpy(o.fr_array(lag).view(np.int64)).cuda(), gen)
    a = torch.from_numpy(o.fr_array(f).view(np.int64)).cuda()
    c1 = h.best_multiexp(a, h.register_bases(g))
    ev = a.clone()
    h.best_fft(ev, o.fr_array([w])[0], k)
    assert np.array_equal(ev.cpu().numpy().view(np.uint64), o.fr_array(evals))
    c2 = h.best_multiexp(ev, h.register_bases(gl))
    assert np.array_equal(c1, c2) and c1[8:].any()
    assert g1_equal(c1, cref.g1_mul(o.fr_array([o.poly_eval(f, xv)])[0], gen))


# ---- fixed-base mode: precomputed 2^(c*j) * P_i, one shared bucket set --------------------------

def test_precomputed_bases_golden_and_slices(cref, golden):
    g = golden["msm"]
    for name in ("n1024_uniform", "n1024_prover", "n1024_one", "n1024_rminus1", "n255_uniform", "n1024_small"):
        s, b = g[f"{name}_s"], g[f"{name}_b"]
        hd = h.register_bases(b, precompute=True)
        try:
            assert g1_equal(h.best_multiexp(s, hd), g[f"{name}_r"]), name
            if s.shape[0] == 1024:                 # a slice of the set falls back to the plain path on the same handle
                exp = cref.g1_to_affine(cref.best_multiexp(s[10:700], b[10:700], 4))[0]
                assert g1_equal(h.best_multiexp(s[10:700], hd, offset=10), exp), name
        finally:
            h.release_bases(hd)


@pytest.mark.parametrize("log_n,kind", [(12, "uniform"), (15, "prover"), (16, "uniform"), (17, "one")])
def test_precomputed_bases_match_oracle(cref, pyref, log_n, kind):
    o = pyref
    n = 1 << log_n
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 5 * log_n), gen).cpu().numpy().view(np.uint64).copy()
    bases[2] = 0                      # identity base
    bases[5] = bases[6]               # duplicate points
    bases[9] = bases[8].copy()
    if kind == "uniform":
        s = rand_fr_gpu(n, log_n + 100).cpu().numpy().view(np.uint64)
    else:
        s = np.tile(o.fr_array(o.rand_scalars(4096, log_n, kind)), (n // 4096, 1))
    exp = cref.g1_to_affine(cref.best_multiexp(s, bases, 8))[0]
    hd = h.register_bases(bases, precompute=True)
    try:
        assert g1_equal(h.best_multiexp(s, hd), exp)
        plain = h.best_multiexp(s, bases)
        assert g1_equal(plain, exp)
    finally:
        h.release_bases(hd)


def test_precomputed_full_size_kzg_known_answer(cref, pyref):
    import torch
    o = pyref
    log_n = 22
    n = 1 << log_n
    x = o.fr_array([0x48324D4933353558])[0]
    powers = torch.from_numpy(cref.fr_powers(x, n).view(np.int64)).cuda()
    srs = h.g1_fixed_base_mul(powers, cref.g1_generator())
    hd = h.register_bases(srs, precompute=True)
    try:
        f = rand_fr_gpu(n, 777)
        got = h.best_multiexp(f, hd)
        st = h.msm_stats()
        assert st["window_bits"] > 16                       # the shared-bucket-set plan was used
        fx = cref.fr_horner(f.cpu().numpy().view(np.uint64), x)
        assert g1_equal(got, cref.g1_mul(fx, cref.g1_generator()))
    finally:
        h.release_bases(hd)


def test_fixed_base_table_is_the_registration_default_from_its_threshold(cref, pyref):
    """hm_register_bases* builds the fixed-base table by itself from 2^threshold points (default 17; here raised to 2^20
    so that the oracle can check the result): one bucket set, fewer windows, the positional 32-bit sort items (n W = 2^23.7
    items, 2^19 buckets), the balanced window split with doubled narrow digits -- against the oracle on uniform scalars and
    on a column with a hot bucket; one point below the threshold, and with the threshold off, the plain layout."""
    import torch
    lib = _lib.load()
    o = pyref
    n = 1 << 20
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 4100), cref.g1_generator())
    bh = bases.cpu().numpy().view(np.uint64)
    s = rand_fr_gpu(n, 4101)
    hot = s.clone()
    hot[: n // 2] = torch.from_numpy(o.fr_array([5]).view(np.int64)).cuda()          # half the column is the constant 5
    exp = [cref.g1_to_affine(cref.best_multiexp(c.cpu().numpy().view(np.uint64), bh, 8))[0] for c in (s, hot)]
    assert lib.hm_set_fixed_base_threshold(3) == -1 and lib.hm_set_fixed_base_threshold(40) == -1
    _lib.check(lib.hm_set_fixed_base_threshold(20))
    try:
        hd = h.register_bases(bases)
        try:
            assert g1_equal(h.best_multiexp(s, hd), exp[0])
            st = h.msm_stats()
            assert st["windows"] == 13 and st["window_bits"] == 20, st          # 255 bits = 8 x 20 + 5 x 19 over ONE bucket set
            assert g1_equal(h.best_multiexp(hot, hd), exp[1])
            part = h.best_multiexp(s[: n // 2].contiguous(), hd, offset=n // 4)      # a slice: the plain path on the same handle
            assert g1_equal(part, cref.g1_to_affine(cref.best_multiexp(s[: n // 2].cpu().numpy().view(np.uint64), bh[n // 4: 3 * n // 4], 8))[0])
        finally:
            h.release_bases(hd)
        hd = h.register_bases(bases[: n - 1].contiguous())                            # one point short of the threshold
        try:
            assert g1_equal(h.best_multiexp(s[: n - 1].contiguous(), hd),
                            cref.g1_to_affine(cref.best_multiexp(s[: n - 1].cpu().numpy().view(np.uint64), bh[: n - 1], 8))[0])
            assert h.msm_stats()["windows"] >= 15                                     # c = 16 or 17: the plain layout
        finally:
            h.release_bases(hd)
        _lib.check(lib.hm_set_fixed_base_threshold(0))
        hd = h.register_bases(bases)
        try:
            assert g1_equal(h.best_multiexp(s, hd), exp[0]) and h.msm_stats()["windows"] == 15
        finally:
            h.release_bases(hd)
    finally:
        _lib.check(lib.hm_set_fixed_base_threshold(17))


def test_transient_sets_take_the_plain_layout_and_released_tables_are_not_parked(cref):
    """the tensor form best_multiexp(coeffs_tensor, bases_tensor) registers its bases for ONE MSM -- it must not
    build the fixed-base table (ten MSMs' worth of work and W copies of the points).  hm_register_bases_plain* is the
    entry for that; hm_get_bases_info tells which layout a handle has.  A released table goes back to the allocator
    (parked buffers are capped at 2 GiB: a 2^22-point table is 3.2 GiB), a released plain set is recycled."""
    n = 1 << 18
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 4300), cref.g1_generator())
    s = rand_fr_gpu(n, 4301)
    exp = cref.g1_to_affine(cref.best_multiexp(s.cpu().numpy().view(np.uint64), bases.cpu().numpy().view(np.uint64), 8))[0]
    assert g1_equal(h.best_multiexp(s, bases), exp)                      # the per-call form
    st = h.msm_stats()
    assert st["windows"] == 17 and st["window_bits"] == 15, st           # the plain layout's plan at 2^18 (the table's: 15 windows of 17 bits)
    hp = h.register_bases(bases, plain=True)
    hd = h.register_bases(bases)                                         # the default: a table from 2^17 points
    ht = h.register_bases(bases[: 1 << 12].contiguous(), precompute=True)
    try:
        ip, idf, it = h.bases_info(hp), h.bases_info(hd), h.bases_info(ht)
        assert ip["table_windows"] == 0 and ip["n"] == n and ip["device_bytes"] == n * 65 and ip["devices"] == 1
        assert idf["table_windows"] == 15 and idf["table_window_bits"] == 17 and idf["device_bytes"] == n * (15 * 64 + 1)
        assert it["table_windows"] > 0 and it["n"] == 1 << 12
        assert g1_equal(h.best_multiexp(s, hp), exp) and h.msm_stats()["windows"] == 17
        assert g1_equal(h.best_multiexp(s, hd), exp) and h.msm_stats()["windows"] == 15
        with pytest.raises(ValueError):
            h.register_bases(bases, precompute=True, plain=True)
    finally:
        for x in (hp, hd, ht):
            h.release_bases(x)
    with pytest.raises(_lib.Halo2Mi355xError):
        h.bases_info(hd)
    # parking is by bytes: a 2^22-point table (12 windows x 256 MiB + flags) is freed, not kept
    big = h.g1_fixed_base_mul(rand_fr_gpu(1 << 22, 4302), cref.g1_generator())
    hb = h.register_bases(big, precompute=True)
    assert h.bases_info(hb)["device_bytes"] > (2 << 30)
    h.release_bases(hb)
    hq = h.register_bases(bases, plain=True)
    try:
        assert h.bases_info(hq)["parked_bytes"] <= (2 << 30)
    finally:
        h.release_bases(hq)


def test_phase_with_columns_of_every_density_picks_a_window_per_chain(cref, pyref):
    """One hm_msm_batch_bn256_g1_dev call at 2^17 over a table set: columns with 1, 300, 1 100 and 5 000 used rows (+ six
    blinding rows at the end), an all-zero column, rows used only in the LAST block, and dense columns in between.  The
    planner counts surviving 256-row blocks per column, groups the sparse ones into chains of the five-launch plan whose
    WINDOW is sized for the rows that survive (c = 4 .. 10 here, not the 15 of 2^17 points), and sends the dense ones to the
    table.  Every result against the oracle, and equal to the one-at-a-time calls."""
    import torch
    n = 1 << 17
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 9100), gen)
    bh = bases.cpu().numpy().view(np.uint64)
    full = rand_fr_gpu(n, 9101)

    def column(used, seed, tail=6, head=True):
        c = torch.zeros_like(full)
        r = rand_fr_gpu(used + tail, seed)
        if head:
            c[:used] = r[:used]
        else:
            c[n - 256: n - 256 + used] = r[:used]
        if tail:
            c[n - tail:] = r[used:]
        return c

    cols = [column(1, 1), full, column(300, 2), column(1100, 3), torch.zeros_like(full), column(5000, 4), column(40, 5, tail=0, head=False),
            full.flip(0).contiguous(), column(1100, 6), column(2, 7), column(300, 8), column(700, 9), column(1100, 10)]
    exp = [cref.g1_to_affine(cref.best_multiexp(c.cpu().numpy().view(np.uint64), bh, 8))[0] for c in cols]
    hd = h.register_bases(bases)
    try:
        got = h.best_multiexp_batch(cols, hd)
        for i in range(len(cols)):
            assert g1_equal(got[i], exp[i]), i
            assert g1_equal(h.best_multiexp(cols[i], hd), exp[i]), i
        again = h.best_multiexp_batch(cols[::-1], hd)                   # another order: other groups, other windows per chain
        for i in range(len(cols)):
            assert g1_equal(again[len(cols) - 1 - i], exp[i]), i
    finally:
        h.release_bases(hd)


@pytest.mark.parametrize("n", [16385, 65537, (1 << 17) + 5, 100003])
def test_sizes_off_the_power_of_two_grid(cref, pyref, n):
    """Chunk / vector-load boundaries of the sort and ragged last chunks: n is arbitrary for an MSM."""
    o = pyref
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, n % 1000), gen).cpu().numpy().view(np.uint64).copy()
    tile = o.fr_array(o.rand_scalars(4099, n, "prover"))
    s = np.concatenate([np.tile(tile, (n // 4099, 1)), tile[: n % 4099]])
    s[::7] = rand_fr_gpu((n + 6) // 7, n).cpu().numpy().view(np.uint64)
    exp = cref.g1_to_affine(cref.best_multiexp(s, bases, 8))[0]
    assert g1_equal(h.best_multiexp(s, bases), exp)
    hd = h.register_bases(bases, precompute=True)
    try:
        assert g1_equal(h.best_multiexp(s, hd), exp)
    finally:
        h.release_bases(hd)


def test_host_pointer_form_never_returns_a_stale_result(cref, golden):
    """The drop-in call takes both arrays by host pointer and reuses converted bases only on a match of a digest over the
    WHOLE array (never by pointer): a buffer that is mutated in place -- at any index -- or re-used at the same address
    with other contents (a Rust Vec freed and re-allocated, as the verifier's MSMs do) must give the result of its CURRENT
    contents; the same contents at another address may hit.  Checked with the reuse on and off."""
    g = golden["msm"]
    lib = _lib.load()
    for mode in (1, 0):
        _lib.check(lib.hm_set_host_base_cache(mode))
        try:
            _stale_result_checks(cref, g)
        finally:
            _lib.check(lib.hm_set_host_base_cache(1))
    # the digest covers the 2^16 x 8 words of a larger array in four host threads: one flipped bit anywhere is a miss
    n = 1 << 17
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 4711), gen).cpu().numpy().view(np.uint64).copy()
    s = np.zeros((n, 4), dtype=np.uint64)
    one = cref.fr_to_mont(np.array([[1, 0, 0, 0]], dtype=np.uint64))[0]
    for i in (0, 1, n // 4 + 3, n // 2, n - 2, n - 1):
        s[:] = 0
        s[i] = one
        first = h.best_multiexp(s, bases)
        assert np.array_equal(first[:8], bases[i]), i                       # 1 * P_i
        bases[i] = bases[(i + 5) % n]                                       # change ONE point in place (any thread's share)
        assert np.array_equal(h.best_multiexp(s, bases)[:8], bases[i]), i   # the changed array must be seen


def _stale_result_checks(cref, g):
    s, b = g["n1024_uniform_s"].copy(), g["n1024_uniform_b"].copy()
    assert g1_equal(h.best_multiexp(s, b), g["n1024_uniform_r"])
    assert g1_equal(h.best_multiexp(s, b), g["n1024_uniform_r"])
    for i, j in ((0, 512), (1, 513), (7, 1000), (1022, 1023)):          # sampled and un-sampled rows alike
        b[i], b[j] = b[j].copy(), b[i].copy()
        exp = cref.g1_to_affine(cref.best_multiexp(s, b, 4))[0]
        assert g1_equal(h.best_multiexp(s, b), exp), (i, j)
    addr = b.ctypes.data
    b[:] = g["n1024_prover_b"]                                          # same address, same length, other points
    assert b.ctypes.data == addr
    exp = cref.g1_to_affine(cref.best_multiexp(s, b, 4))[0]
    assert g1_equal(h.best_multiexp(s, b), exp)
    b[3] = 0                                                            # one base becomes the identity
    exp = cref.g1_to_affine(cref.best_multiexp(s, b, 4))[0]
    assert g1_equal(h.best_multiexp(s, b), exp)


def _raw_to_mont(t):
    """(n, 4) int64 GPU tensor of canonical integers -> Montgomery words (one device product by 2^256)."""
    from halo2_experiments_amd.domain import FR_MODULUS, fr_words
    return h.linear_combination([t.contiguous()], np.stack([fr_words((1 << 256) % FR_MODULUS)]))


def _replay_sparse_column(n, used_rows, seed):
    from halo2_experiments_amd.replay import _sparse_column
    import torch
    return _sparse_column(n, used_rows, seed, torch.device("cuda", torch.cuda.current_device()))


@pytest.mark.parametrize("log_n,used_rows", [(18, 1100), (11, 40)])
def test_baseline_config_sizes_plain_and_eight_in_flight(cref, log_n, used_rows):
    """BASELINE configs[3] (MerkleSumTree, k = 18) and configs[1] (Poseidon, k = 11) at EXACTLY their MSM
    sizes: a uniform column and the create_proof replay's sparse column (used_rows small values, half of
    them the constant 1, six blinding rows) against the C oracle -- through the synchronous call and
    through hm_msm_submit_dev with all eight asynchronous slots in flight on four streams."""
    import torch
    n = 1 << log_n
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 1800 + log_n), gen)
    bh = bases.cpu().numpy().view(np.uint64)
    cols = [rand_fr_gpu(n, 1801 + log_n), _replay_sparse_column(n, used_rows, 1802 + log_n)]
    exps = [cref.g1_to_affine(cref.best_multiexp(c.cpu().numpy().view(np.uint64), bh, 8))[0] for c in cols]
    assert g1_equal(h.best_multiexp(cols[0].cpu().numpy().view(np.uint64), bh), exps[0])    # host-pointer drop-in form
    hd = h.register_bases(bases)
    try:
        for c, e in zip(cols, exps):
            assert g1_equal(h.best_multiexp(c, hd), e)
        st = h.msm_stats()
        assert st["windows"] * st["window_bits"] >= 255
        streams = [torch.cuda.Stream() for _ in range(4)]
        for st_ in streams:
            st_.wait_stream(torch.cuda.current_stream())
        for round_ in range(2):
            tickets = []
            for i in range(8):
                with torch.cuda.stream(streams[i % 4]):
                    tickets.append(h.best_multiexp_submit(cols[(i + round_) & 1], hd))
            for i, t in enumerate(tickets):
                assert g1_equal(h.best_multiexp_wait(t), exps[(i + round_) & 1]), (round_, i)
    finally:
        h.release_bases(hd)


def test_release_with_a_ticket_in_flight_and_buffer_recycling(cref, golden):
    """hm_release_bases never waits for the device: a set released while a ticket still reads it stays
    alive until that ticket is awaited, and the next registration of the same size recycles the buffers
    of a released set (other contents => other result)."""
    import torch
    g = golden["msm"]
    s = torch.from_numpy(g["n1024_uniform_s"].view(np.int64).copy()).cuda()
    hd = h.register_bases(g["n1024_uniform_b"])
    t = h.best_multiexp_submit(s, hd)
    h.release_bases(hd)                                     # ticket still in flight
    hd2 = h.register_bases(g["n1024_prover_b"])             # must NOT take over the buffers the ticket reads
    t2 = h.best_multiexp_submit(torch.from_numpy(g["n1024_prover_s"].view(np.int64).copy()).cuda(), hd2)
    assert g1_equal(h.best_multiexp_wait(t), g["n1024_uniform_r"])
    assert g1_equal(h.best_multiexp_wait(t2), g["n1024_prover_r"])
    h.release_bases(hd2)
    for name in ("n1024_small", "n1024_uniform", "n1024_one"):           # recycled buffers, new contents each time
        hd3 = h.register_bases(g[f"{name}_b"])
        assert g1_equal(h.best_multiexp(g[f"{name}_s"], hd3), g[f"{name}_r"]), name
        h.release_bases(hd3)
    with pytest.raises(_lib.Halo2Mi355xError):
        h.release_bases(hd3)


def test_call_counters(golden):
    """hm_get_stats: the measured call trace a build of the Rust shim reads after create_proof."""
    import torch
    lib = _lib.load()
    g = golden["msm"]
    _lib.check(lib.hm_set_host_base_cache(0))          # count the drop-in call's full upload (a digest hit would skip the bases)
    _lib.check(lib.hm_reset_stats())
    st = _lib.Stats()
    _lib.check(lib.hm_get_stats(ctypes.byref(st)))
    assert st.msm_calls == 0 and st.ntt_calls == 0 and st.h2d_bytes == 0
    h.best_multiexp(g["n1024_uniform_s"], g["n1024_uniform_b"])          # host-pointer form: 96 B per point cross PCIe
    hd = h.register_bases(g["n255_uniform_b"])
    h.best_multiexp(g["n255_uniform_s"], hd)                               # handle form: 32 B per point
    t = h.best_multiexp_submit(torch.from_numpy(g["n255_uniform_s"].view(np.int64).copy()).cuda(), hd)
    h.best_multiexp_wait(t)
    h.release_bases(hd)
    a = golden["ntt"]["k7_in"].copy()
    h.best_fft(a, golden["ntt"]["k7_omega"], 7)
    d = torch.from_numpy(golden["ntt"]["k10_in"].view(np.int64).copy()).cuda().reshape(1, 1024, 4).repeat(3, 1, 1).contiguous()
    from halo2_experiments_amd.domain import EvaluationDomain
    EvaluationDomain(3, 10).lagrange_to_coeff(d)
    torch.cuda.synchronize()
    _lib.check(lib.hm_get_stats(ctypes.byref(st)))
    assert st.msm_calls == 3 and st.msm_points == 1024 + 255 + 255
    assert st.msm_calls_by_log2[10] == 1 and st.msm_calls_by_log2[7] == 2
    assert st.ntt_calls == 4 and st.ntt_elements == 128 + 3 * 1024
    assert st.ntt_calls_by_log2[7] == 1 and st.ntt_calls_by_log2[10] == 3
    assert st.h2d_bytes == 1024 * 96 + 255 * 32 + 128 * 32 and st.d2h_bytes == 128 * 32
    assert st.msm_device_us > 0 and st.msm_h2d_us > 0 and st.ntt_device_us > 0
    _lib.check(lib.hm_set_host_base_cache(1))
    _lib.check(lib.hm_reset_stats())
    h.best_multiexp(g["n1024_small_s"], g["n1024_small_b"])               # miss: 96 B per point
    h.best_multiexp(g["n1024_small_s"], g["n1024_small_b"].copy())        # same contents at another address: 32 B per point
    _lib.check(lib.hm_get_stats(ctypes.byref(st)))
    assert st.h2d_bytes == 1024 * 96 + 1024 * 32


def test_async_submit_wait_overlapping_streams(cref, golden):
    """Eight MSMs in flight (every asynchronous slot) on three streams; results identical to the
    synchronous calls, a ninth submit is refused until a ticket has been awaited, unknown tickets are
    errors.  Two rounds: every slot is used again with other scalar buffers."""
    import torch
    from halo2_experiments_amd.sharding import MAX_IN_FLIGHT
    g = golden["msm"]
    names = [["n1024_uniform", "n1024_prover", "n1024_small"][i % 3] for i in range(MAX_IN_FLIGHT)]
    hd = h.register_bases(g["n1024_uniform_b"])
    try:
        streams = [torch.cuda.Stream() for _ in range(3)]
        for round_ in range(2):
            tickets, keep = [], []
            for i, name in enumerate(names):
                with torch.cuda.stream(streams[i % 3]):
                    s_dev = torch.from_numpy(g[f"{name}_s"].view(np.int64).copy()).cuda()
                    keep.append(s_dev)
                    tickets.append(h.best_multiexp_submit(s_dev, hd))
            with pytest.raises(_lib.Halo2Mi355xError):
                h.best_multiexp_submit(keep[0], hd)                  # every asynchronous slot is busy
            assert g1_equal(h.best_multiexp(g["n255_uniform_s"], g["n255_uniform_b"]), g["n255_uniform_r"])   # sync path still free
            for name, t in reversed(list(zip(names, tickets))):     # await out of order
                assert g1_equal(h.best_multiexp_wait(t), g[f"{name}_r"]), (round_, name)
            with pytest.raises(_lib.Halo2Mi355xError):
                h.best_multiexp_wait(tickets[0])
        t = h.best_multiexp_submit(keep[0], hd)
        assert g1_equal(h.best_multiexp_wait(t), g["n1024_uniform_r"])
    finally:
        h.release_bases(hd)


def test_batched_commitments_equal_single_calls(cref, golden):
    """hm_msm_batch_bn256_g1_dev: a phase of commitments in one call (more of them than asynchronous slots, dense and
    sparse columns mixed, a slice of the base set) equals the one-at-a-time results; runs twice (slots are reused)."""
    import torch
    n = 1 << 13
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n + 64, 8100), gen)
    hd = h.register_bases(bases)
    try:
        cols = [rand_fr_gpu(n, 8200 + i) if i % 3 else _replay_sparse_column(n, 300, 8300 + i) for i in range(19)]
        single = np.stack([h.best_multiexp(c, hd, offset=32) for c in cols])
        bh = bases.cpu().numpy().view(np.uint64)[32:32 + n]
        exp0 = cref.g1_to_affine(cref.best_multiexp(cols[1].cpu().numpy().view(np.uint64), bh, 4))[0]
        assert g1_equal(single[1], exp0)
        from halo2_experiments_amd.arithmetic import best_multiexp_batch
        for _ in range(2):
            got = best_multiexp_batch(cols, hd, offset=32)
            assert np.array_equal(got, single)
        stacked = torch.stack(cols[:5])
        assert np.array_equal(best_multiexp_batch(stacked, hd, offset=32), single[:5])
        assert best_multiexp_batch([], hd).shape == (0, 12)
        with pytest.raises(_lib.Halo2Mi355xError):
            best_multiexp_batch(cols[:2], hd, offset=100)           # offset + n exceeds the set: error, no ticket left behind
        assert np.array_equal(best_multiexp_batch(cols[:9], hd, offset=32), single[:9])
    finally:
        h.release_bases(hd)


@pytest.mark.parametrize("log_n", [17, 18])
def test_dense_columns_of_a_phase_share_chains_on_the_table(cref, log_n):
    """Round 4: the DENSE columns of a phase on a fixed-base table set go through the general pipeline several per chain --
    every column one bucket set of the same sort, K3 and reduction launches.  Eleven dense columns of every kind (uniform,
    16-bit values, one constant -- a hot bucket per window inside a group --, all ones, 0 / 1 flags), sparse ones in between
    (those keep the five-launch plan), one all-zero column: the batch equals the one-at-a-time results, three of which are
    checked against the oracle; from host arrays too."""
    import torch
    from halo2_experiments_amd.arithmetic import best_multiexp_batch
    n = 1 << log_n
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 8600 + log_n), cref.g1_generator())
    bases[5] = 0                                                         # an identity base
    bases[7] = bases[6]
    one = torch.from_numpy(cref.fr_to_mont(np.array([[1, 0, 0, 0]], dtype=np.uint64)).view(np.int64)).cuda()
    cols = []
    for i in range(11):
        c = rand_fr_gpu(n, 8700 + 13 * i + log_n)
        if i == 2:
            c[:, 1:] = 0
            c[:, 0] &= 0xFFFF
            c = _raw_to_mont(c)
        elif i == 4:
            c = c[:1].expand(n, 4).contiguous()
        elif i == 6:
            c = one.expand(n, 4).contiguous()
        elif i == 8:
            c = torch.where((torch.arange(n, device="cuda") % 3 == 0)[:, None], one.expand(n, 4), torch.zeros_like(c)).contiguous()
        cols.append(c)
    cols.insert(3, _replay_sparse_column(n, 900, 8800))
    cols.insert(9, torch.zeros((n, 4), dtype=torch.int64, device="cuda"))
    cols.insert(10, _replay_sparse_column(n, 40, 8801))
    hd = h.register_bases(bases)
    try:
        assert h.bases_info(hd)["table_windows"] != 0
        single = np.stack([h.best_multiexp(c, hd) for c in cols])
        bh = bases.cpu().numpy().view(np.uint64)
        for i in (0, 5, 7):
            assert g1_equal(single[i], cref.g1_to_affine(cref.best_multiexp(cols[i].cpu().numpy().view(np.uint64), bh, 8))[0]), i
        assert not single[9].any()
        for _ in range(2):
            assert np.array_equal(best_multiexp_batch(cols, hd), single)
        assert np.array_equal(best_multiexp_batch([c.cpu().numpy().view(np.uint64) for c in cols], hd), single)
        assert np.array_equal(best_multiexp_batch(cols[:2], hd), single[:2])
    finally:
        h.release_bases(hd)


def test_grouped_chains_with_empty_ragged_and_identity_rows(cref, pyref):
    """The group form of the five-launch plan (one chain carries several commitments) on a length that is not a multiple
    of the 256-row compaction block: an all-zero column (no surviving block at all), a column whose only non-zero scalar
    sits in the last, partial block, a column that only meets identity bases, dense columns -- against the oracle."""
    import torch
    n = 3001
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 8400), gen)
    bases[5] = 0                                             # identity bases (0, 0)
    bases[n - 1] = 0
    hd = h.register_bases(bases)
    try:
        zero = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
        last = zero.clone()
        last[n - 2] = rand_fr_gpu(1, 8401)[0]
        only_identity = zero.clone()
        only_identity[5] = rand_fr_gpu(1, 8402)[0]
        only_identity[n - 1] = rand_fr_gpu(1, 8403)[0]
        cols = [zero, last, rand_fr_gpu(n, 8404), only_identity, rand_fr_gpu(n, 8405), zero, last, rand_fr_gpu(n, 8406), zero]
        bh = bases.cpu().numpy().view(np.uint64)
        want = [cref.g1_to_affine(cref.best_multiexp(c.cpu().numpy().view(np.uint64), bh, 4))[0] for c in cols]
        from halo2_experiments_amd.arithmetic import best_multiexp_batch
        for _ in range(2):
            got = best_multiexp_batch(cols, hd)
            for g, w in zip(got, want):
                assert g1_equal(g, w)
        assert not got[0][8:].any() and not got[3][8:].any()  # the identity
        for c, w in zip(cols[:4], want[:4]):                  # and one at a time
            assert g1_equal(h.best_multiexp(c, hd), w)
    finally:
        h.release_bases(hd)


@pytest.mark.parametrize("log_n", [12, 17])
def test_two_threads_batch_at_once(cref, log_n):
    """Two host threads each commit a phase through hm_msm_batch_bn256_g1_dev at the same time (the library is advertised
    as thread-safe; the eight asynchronous slots are shared, so each call has to wait for the other's tickets).  At 2^17
    both calls also count surviving blocks (one shared buffer, one call at a time), group their sparse columns and send
    the dense ones through the fixed-base table."""
    import threading
    from halo2_experiments_amd.arithmetic import best_multiexp_batch
    n = 1 << log_n
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 8500), cref.g1_generator())
    hd = h.register_bases(bases)
    try:
        count = 21 if log_n == 12 else 11
        cols = [[rand_fr_gpu(n, 8600 + 40 * t + i) if i % 3 else _replay_sparse_column(n, 300, 8700 + 40 * t + i) for i in range(count)]
                for t in range(2)]
        want = [np.stack([h.best_multiexp(c, hd) for c in cs]) for cs in cols]
        got, errs = [None, None], []

        def work(t):
            try:
                for _ in range(3):
                    got[t] = best_multiexp_batch(cols[t], hd)
            except Exception as e:      # noqa: BLE001
                errs.append(e)

        th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
        for x in th:
            x.start()
        for x in th:
            x.join()
        assert not errs, errs
        for t in range(2):
            assert np.array_equal(got[t], want[t])
    finally:
        h.release_bases(hd)


def test_shutdown_then_reuse(cref, golden):
    """hm_shutdown releases every device resource of the context (workspaces, tables, streams); the next call builds them
    again -- same results, sparse and dense, through the synchronous path and a batch, with a general-pipeline MSM in
    between (which scribbles over the slot workspace the five-launch plan keeps its block counters in)."""
    from halo2_experiments_amd.arithmetic import best_multiexp_batch
    lib = _lib.load()
    g = golden["msm"]
    n = 1 << 12
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 8700), cref.g1_generator())
    cols = [rand_fr_gpu(n, 8701), _replay_sparse_column(n, 100, 8702)]
    big_b = h.g1_fixed_base_mul(rand_fr_gpu(1 << 19, 8703), cref.g1_generator())
    big_s = rand_fr_gpu(1 << 19, 8704)
    results = []
    for round_ in range(3):
        hd, hb = h.register_bases(bases), h.register_bases(big_b)
        one = [h.best_multiexp(c, hd) for c in cols]
        big = h.best_multiexp(big_s, hb)                       # n = 2^19: the general pipeline, same slot 0
        again = [h.best_multiexp(c, hd) for c in cols]
        batch = best_multiexp_batch(cols * 5, hd)
        assert all(np.array_equal(a, b) for a, b in zip(one, again))
        assert all(np.array_equal(batch[i], one[i % 2]) for i in range(10))
        results.append((one, big))
        assert g1_equal(h.best_multiexp(g["n255_uniform_s"], g["n255_uniform_b"]), g["n255_uniform_r"])
        _lib.check(lib.hm_shutdown())                           # handles die with the context
    for one, big in results[1:]:
        assert all(np.array_equal(a, b) for a, b in zip(one, results[0][0])) and np.array_equal(big, results[0][1])


def test_randomized_prover_size_sweep(cref):
    """Forty random shapes of prover-sized MSMs against the C oracle, each through the synchronous call and through a
    batch (groups below 2^16 + 1, separate chains above): random length (ragged, around the 256-row compaction blocks and
    the window-table steps), random fraction of zero scalars (down to a single survivor), small / full-width / r - 1
    scalars, repeated and identity bases."""
    import torch
    from halo2_experiments_amd.arithmetic import best_multiexp_batch
    rng = np.random.default_rng(20241004)
    gen = cref.g1_generator()
    pool = h.g1_fixed_base_mul(rand_fr_gpu(1 << 17, 9000), gen)
    r_minus_1 = torch.from_numpy(cref.fr_to_mont(np.array([[0x43E1F593F0000000, 0x2833E84879B97091, 0xB85045B68181585D, 0x30644E72E131A029]],
                                                          dtype=np.uint64)).view(np.int64)).cuda()
    for case in range(40):
        lg = int(rng.integers(0, 17))
        n = int(rng.integers(max(1, (1 << lg) // 2), (1 << lg) + 1)) if case % 4 else int(rng.choice([1, 2, 255, 256, 257, 511, 513, 4096, 65535, 65536, 65537]))
        off = int(rng.integers(0, (1 << 17) - n + 1))
        bases = pool[off:off + n].clone()
        if n > 4:
            bases[int(rng.integers(0, n))] = 0                                       # an identity base
            bases[int(rng.integers(0, n))] = bases[int(rng.integers(0, n))]          # a repeated base
        hd = h.register_bases(bases)
        try:
            cols = []
            for v in range(3):
                s = rand_fr_gpu(n, 9100 + 7 * case + v)
                kind = int(rng.integers(0, 4))
                if kind == 1:                                                        # small values (the replay's sparse columns)
                    s[:, 1:] = 0
                    s[:, 0] &= 0xFFFF
                    s = _raw_to_mont(s)
                elif kind == 2 and n > 1:
                    s[int(rng.integers(0, n))] = r_minus_1[0]
                keep = float(rng.choice([1.0, 0.5, 0.05, 0.0]))
                if keep < 1.0:
                    mask = torch.from_numpy(rng.random(n) >= keep).cuda()
                    if keep == 0.0:
                        mask[int(rng.integers(0, n))] = False                        # a single survivor
                    s[mask] = 0
                cols.append(s.contiguous())
            bh = bases.cpu().numpy().view(np.uint64)
            want = [cref.g1_to_affine(cref.best_multiexp(c.cpu().numpy().view(np.uint64), bh, 4))[0] for c in cols]
            for c, w in zip(cols, want):
                assert g1_equal(h.best_multiexp(c, hd), w), (case, n)
            got = best_multiexp_batch(cols * 3, hd)
            for i, g_ in enumerate(got):
                assert g1_equal(g_, want[i % 3]), (case, n, i)
        finally:
            h.release_bases(hd)


def test_randomized_table_sized_sweep(cref):
    """Twelve random shapes at the sizes where registration builds the fixed-base table by itself (2^17 .. 2^19.6, off the
    power-of-two grid): positional sort items, the balanced window split, 16 384-item sort tiles, the two-launch bucket
    reduction.  Uniform / small / 0-1 / constant / mostly-zero scalars, identity and repeated bases.  Every result three
    ways: the table, the plain layout of the same points (hm_set_fixed_base_threshold(0)), and the C oracle."""
    import torch
    lib = _lib.load()
    rng = np.random.default_rng(20261004)
    gen = cref.g1_generator()
    pool = h.g1_fixed_base_mul(rand_fr_gpu(800_000, 9500), gen)
    one = torch.from_numpy(cref.fr_to_mont(np.array([[1, 0, 0, 0]], dtype=np.uint64)).view(np.int64)).cuda()
    try:
        for case in range(12):
            n = int(rng.choice([1 << 17, (1 << 17) + 1, (1 << 18) - 1, 1 << 19])) if case < 4 else int(rng.integers(1 << 17, 800_000))
            off = int(rng.integers(0, 800_000 - n + 1))
            bases = pool[off:off + n].clone()
            for _ in range(3):
                bases[int(rng.integers(0, n))] = 0
                bases[int(rng.integers(0, n))] = bases[int(rng.integers(0, n))]
            s = rand_fr_gpu(n, 9600 + case)
            kind = case % 6
            if kind == 1:                                       # 16-bit values
                s[:, 1:] = 0
                s[:, 0] &= 0xFFFF
                s = _raw_to_mont(s)
            elif kind == 2:                                     # 0 / 1 flags
                flags = torch.from_numpy(rng.random(n) < 0.3).cuda()
                s = torch.where(flags[:, None], one.expand(n, 4), torch.zeros_like(s))
            elif kind == 3:                                     # one constant: every digit of a window lands in ONE bucket
                s = s[:1].expand(n, 4).contiguous()
            elif kind == 4:                                     # 2 % survivors
                s[torch.from_numpy(rng.random(n) >= 0.02).cuda()] = 0
            s = s.contiguous()
            bh = bases.cpu().numpy().view(np.uint64)
            want = cref.g1_to_affine(cref.best_multiexp(s.cpu().numpy().view(np.uint64), bh, 8))[0]
            _lib.check(lib.hm_set_fixed_base_threshold(17))
            hd = h.register_bases(bases)
            try:
                got = h.best_multiexp(s, hd)
                assert h.msm_stats()["windows"] <= 15, (case, n)                # the shared bucket set
                assert g1_equal(got, want), (case, n, kind)
            finally:
                h.release_bases(hd)
            _lib.check(lib.hm_set_fixed_base_threshold(0))
            hd = h.register_bases(bases)
            try:
                assert g1_equal(h.best_multiexp(s, hd), want), (case, n, kind, "plain")
            finally:
                h.release_bases(hd)
    finally:
        _lib.check(lib.hm_set_fixed_base_threshold(17))


@pytest.mark.parametrize("n", [(1 << 19) + 1, (1 << 19) + 3, (1 << 20) - 1])
def test_vector_load_sort_at_odd_sizes_and_with_zero_digits(cref, n):
    """The round-4 sort scatters read four items per 16-byte load: a window's digit array then starts at an address that is
    16-byte aligned only when n is a multiple of four (plain layout: window w starts at w * n words), tiles begin up to three
    positions before their chunk, and zero digits leave holes in a lane's four items.  Odd n at the general pipeline's
    sizes, both layouts, uniform scalars and a column that is half zeros with small values in between."""
    import torch
    lib = _lib.load()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 9800 + (n & 7)), cref.g1_generator())
    bh = bases.cpu().numpy().view(np.uint64)
    uni = rand_fr_gpu(n, 9810)
    holes = rand_fr_gpu(n, 9811)
    holes[::2] = 0                                                     # every other scalar zero: every 16-byte load has two holes
    holes[1::4, 1:] = 0                                                # a quarter small (most windows zero)
    holes[1::4, 0] &= 0xFFFFF
    holes = _raw_to_mont(holes)
    want = [cref.g1_to_affine(cref.best_multiexp(c.cpu().numpy().view(np.uint64), bh, 8))[0] for c in (uni, holes)]
    try:
        for threshold in (17, 0):                                      # the table's single bucket set, then the plain layout's 15
            _lib.check(lib.hm_set_fixed_base_threshold(threshold))
            hd = h.register_bases(bases)
            try:
                assert (h.bases_info(hd)["table_windows"] != 0) == (threshold != 0)
                for c, w in zip((uni, holes), want):
                    assert g1_equal(h.best_multiexp(c, hd), w), (n, threshold)
            finally:
                h.release_bases(hd)
    finally:
        _lib.check(lib.hm_set_fixed_base_threshold(17))


@pytest.mark.parametrize("log_n", [12, 18, 22, 24, 27])
def test_geometric_known_answer_needs_neither_oracle_nor_the_dot_product_kernel(pyref, log_n):
    """A full-size known answer in which the only shared code is the affine group law of Python integers: scalars s_i = c^i,
    bases P_i = [d^i]G, so sum_i s_i P_i = [((c d)^n - 1) / (c d - 1)]G.  The expected point is one Python double-and-add of the
    closed form; eight random bases made by the library's fixed-base kernel are checked against Python's [d^i]G first.  (The
    KZG known answers elsewhere take their expected scalar from the library's own inner-product kernel.)"""
    import ctypes
    import torch
    from halo2_experiments_amd.arithmetic import _ptr, _stream_ptr
    from halo2_experiments_amd.domain import fr_words
    o = pyref
    n = 1 << log_n
    c, d = 0x1D1D1D1D2B2B2B2B3C3C3C3C4D4D4D4D5E5E5E5E % o.R, 0x0F0E0D0C0B0A09080706050403020100FFEEDDCC % o.R
    lib = _lib.load()
    s = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    t = torch.empty((n, 4), dtype=torch.int64, device="cuda")
    _lib.check(lib.hm_fr_powers_dev(ctypes.c_void_p(s.data_ptr()), n, _ptr(fr_words(c)), ctypes.c_void_p(_stream_ptr(s))))
    _lib.check(lib.hm_fr_powers_dev(ctypes.c_void_p(t.data_ptr()), n, _ptr(fr_words(d)), ctypes.c_void_p(_stream_ptr(t))))
    gen = o.g1_affine_array([o.G1_GEN])[0]
    bases = h.g1_fixed_base_mul(t, gen)
    rng = np.random.default_rng(log_n)
    for i in [0, 1, n - 1] + rng.integers(0, n, 5).tolist():
        want = o.g1_affine_array([o.g1_mul(pow(d, i, o.R), o.G1_GEN)])[0]
        assert np.array_equal(bases[i].cpu().numpy().view(np.uint64), want), i
        assert np.array_equal(s[i].cpu().numpy().view(np.uint64), fr_words(pow(c, i, o.R))), i
    cd = c * d % o.R
    total = (pow(cd, n, o.R) - 1) * pow(cd - 1, -1, o.R) % o.R
    want = o.g1_affine_array([o.g1_mul(total, o.G1_GEN)])[0]
    # (2^27 is the largest call the library takes: n * windows < 2^31 item slots, include/halo2_mi355x.h; 15 windows plain, 12 table)
    assert g1_equal(h.best_multiexp(s, bases), want)                     # a transient set: the plain layout
    if log_n >= 17:
        hd = h.register_bases(bases)                                     # ... and the fixed-base table
        try:
            assert h.bases_info(hd)["table_windows"] != 0
            assert g1_equal(h.best_multiexp(s, hd), want)
        finally:
            h.release_bases(hd)


def test_go_ethereum_precompile_vectors_through_the_hip_path(pyref):
    """The third-party known answers of tests/test_oracle.py (go-ethereum's EIP-196 precompile test data, "chfast1..3")
    computed by the HIP path itself: [k]P as an MSM of one point, P + Q as an MSM of two points with scalars one, and all
    of them at once inside a larger MSM whose other scalars are zero -- no oracle of this repository in between."""
    from known_answers import GETH_ADD_CHFAST1, GETH_MUL_CHFAST
    o = pyref
    a, b, s = GETH_ADD_CHFAST1
    got = h.best_multiexp(o.fr_array([1, 1]), o.g1_affine_array([a, b]))
    assert o.g1_jacobian_from_array(got.reshape(1, 12))[0] == s
    for pt, k, res in GETH_MUL_CHFAST:
        got = h.best_multiexp(o.fr_array([k % o.R]), o.g1_affine_array([pt]))
        assert o.g1_jacobian_from_array(got.reshape(1, 12))[0] == res, hex(k)
    # sum_i [k_i] P_i with the three cases planted among 5000 zero scalars: equals the sum of the three published results
    n = 5000
    pts = [o.G1_GEN] * n
    sc = [0] * n
    want = None
    for slot, (pt, k, res) in zip((7, 2500, 4999), GETH_MUL_CHFAST):
        pts[slot], sc[slot] = pt, k % o.R
        want = res if want is None else o.g1_add(want, res)
    got = h.best_multiexp(o.fr_array(sc), o.g1_affine_array(pts))
    assert o.g1_jacobian_from_array(got.reshape(1, 12))[0] == want


@pytest.mark.parametrize("log_n", [12, 17])
def test_batch_from_host_arrays_equals_the_device_batch(cref, log_n):
    """hm_msm_batch_bn256_g1_h: a phase of commitments whose scalar arrays live in host memory (uploads on the chains' own
    streams) -- grouped chains at 2^12, one chain per commitment at 2^17, more commitments than lanes, a base-set slice."""
    from halo2_experiments_amd.arithmetic import best_multiexp_batch
    n = 1 << log_n
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n + 16, 8800 + log_n), cref.g1_generator())
    hd = h.register_bases(bases)
    try:
        count = 19 if log_n == 12 else 11
        cols = [rand_fr_gpu(n, 8900 + i) if i % 3 else _replay_sparse_column(n, 200, 8950 + i) for i in range(count)]
        dev = best_multiexp_batch(cols, hd, offset=8)
        host_cols = [c.cpu().numpy().view(np.uint64) for c in cols]
        for _ in range(2):
            assert np.array_equal(best_multiexp_batch(host_cols, hd, offset=8), dev)
        assert np.array_equal(best_multiexp_batch(host_cols[:1], hd, offset=8), dev[:1])
        one = h.best_multiexp(cols[1], hd, offset=8)
        assert np.array_equal(dev[1], one)
    finally:
        h.release_bases(hd)


def test_a_call_above_the_item_limit_is_refused_not_truncated():
    """n * windows must stay below 2^31 (the sort's 32-bit item slots): 2^28 points are refused with HM_ERR_BAD_ARG and a message,
    in the synchronous and the asynchronous form -- never a wrapped index.  (Bases and scalars are zeros: identity points are legal
    bases, and nothing is computed.)  A larger commitment is the sum of several calls (hm_g1_sum) or a split over devices."""
    import ctypes
    import torch
    lib = _lib.load()
    n = 1 << 28
    free, _ = torch.cuda.mem_get_info()
    if free < 40 << 30:
        pytest.skip("needs 26 GiB of device memory for the operands alone")
    bases = torch.zeros((n, 8), dtype=torch.int64, device="cuda")
    scal = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    hd = h.register_bases(bases)
    try:
        assert h.bases_info(hd)["table_windows"] == 0                   # the default table would not fit the limit either: plain layout
        with pytest.raises(RuntimeError, match=r"2\^31"):
            h.best_multiexp(scal, hd)
        with pytest.raises(RuntimeError, match=r"2\^31"):
            h.best_multiexp_wait(h.best_multiexp_submit(scal, hd))
        half = h.best_multiexp(scal[: n // 2], hd)                      # 2^27 of them are fine (all identity: the identity)
        assert not half.any()
    finally:
        h.release_bases(hd)
        del bases, scal
        torch.cuda.empty_cache()


def test_config5_size_2_26_fits_one_gpu_and_is_additive():
    """BASELINE config 5's 2^26-point MSM on ONE GPU (2^30 (point, bucket) pairs, 4 + 4 GiB of
    bases, ~20 GiB of workspace): the whole equals the sum of its four 2^24 quarters."""
    import torch
    from halo2_experiments_amd.arithmetic import G1_GENERATOR
    from halo2_experiments_amd.sharding import g1_sum
    n = 1 << 26
    hd = h.register_bases(h.g1_fixed_base_mul(rand_fr_gpu(n, 2601), G1_GENERATOR))
    try:
        s = rand_fr_gpu(n, 2602)
        whole = h.best_multiexp(s, hd)
        assert whole[8:].any()
        q = n // 4
        parts = np.stack([h.best_multiexp(s[i * q:(i + 1) * q].contiguous(), hd, offset=i * q) for i in range(4)])
        assert np.array_equal(g1_sum(parts), whole)
    finally:
        h.release_bases(hd)
        torch.cuda.empty_cache()


@pytest.mark.parametrize("kind", ["constant", "ones", "flags", "two_values"])
def test_skewed_columns_use_the_cooperative_sort(cref, kind):
    """Selector / flag / constant columns send a whole window to one sort region and one bucket: the
    region is then sorted by many workgroups (work list + global per-bucket cursors) and the bucket is
    accumulated as many tasks.  Results must still match the oracle."""
    import torch
    n = 1 << 19
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 1900), gen)
    one = cref.fr_to_mont(np.array([[1, 0, 0, 0]], dtype=np.uint64))
    if kind == "constant":
        s = rand_fr_gpu(1, 1901).cpu().numpy().view(np.uint64).repeat(n, axis=0)
    elif kind == "ones":
        s = one.repeat(n, axis=0)
    elif kind == "flags":
        s = np.where((np.arange(n) % 3 == 0)[:, None], one.repeat(n, axis=0), np.zeros((n, 4), dtype=np.uint64))
    else:
        two = rand_fr_gpu(2, 1902).cpu().numpy().view(np.uint64)
        s = two[(np.arange(n) * 2654435761 >> 7) & 1]
    s = np.ascontiguousarray(s)
    bh = bases.cpu().numpy().view(np.uint64)
    exp = cref.g1_to_affine(cref.best_multiexp(s, bh, 8))[0]
    hd = h.register_bases(bases)
    try:
        assert g1_equal(h.best_multiexp(torch.from_numpy(s.view(np.int64)).cuda(), hd), exp)
        assert h.msm_stats()["tasks"] > 1000
    finally:
        h.release_bases(hd)


def test_single_process_multi_device_split(cref, pyref, golden):
    """hm_set_msm_devices: the one-process, many-GPU form of best_multiexp.  With one card on the box the
    list names device 0 three times -- same splitting, threads, per-slice uploads and host fold as on
    a node with three cards (the slices then simply queue on one device)."""
    lib = _lib.load()
    o = pyref
    n = 3 * (1 << 14) + 7                                   # just above the split threshold, ragged thirds
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 31), gen).cpu().numpy().view(np.uint64).copy()
    s = rand_fr_gpu(n, 32).cpu().numpy().view(np.uint64).copy()
    single = h.best_multiexp(s, bases)[:8]
    assert g1_equal(np.concatenate([single, np.ones(4, dtype=np.uint64)]), cref.g1_to_affine(cref.best_multiexp(s, bases, 8))[0])
    devs = (ctypes.c_int * 3)(0, 0, 0)
    _lib.check(lib.hm_set_msm_devices(devs, 3))
    try:
        assert g1_equal(h.best_multiexp(s, bases), single)
        assert g1_equal(h.best_multiexp(s, bases), single)              # second call: same buffers, same answer
        g = golden["msm"]                                               # small input: goes whole to devices[0]
        assert g1_equal(h.best_multiexp(g["n1024_uniform_s"], g["n1024_uniform_b"]), g["n1024_uniform_r"])
        s[: n // 3] = 0                                                 # first slice sums to the identity
        exp = cref.g1_to_affine(cref.best_multiexp(s, bases, 8))[0]
        assert g1_equal(h.best_multiexp(s, bases), exp)
        bad = (ctypes.c_int * 2)(0, 99)
        assert lib.hm_set_msm_devices(bad, 2) == -1                     # out of range: list unchanged
        assert g1_equal(h.best_multiexp(s, bases), exp)
    finally:
        _lib.check(lib.hm_set_msm_devices(None, 0))
    assert g1_equal(h.best_multiexp(s, bases), exp)


@pytest.mark.parametrize("pattern", ["P,P,-P,P", "P,-P,P,-P", "P,P,P,P,P,-P,-P,-P"])
def test_repeated_and_opposite_bases_inside_bucket_chains(cref, pyref, pattern):
    """The bucket chain's fast law has no equal-x case: a lane that meets one (same point again =>
    doubling, its negative => identity, then a restart) finishes its chain with the general law.
    Runs of one point and its negative under one scalar land next to each other in every bucket."""
    o = pyref
    signs = [1 if t == "P" else -1 for t in pattern.split(",")]
    m, groups = len(signs), 1 << 12
    n = m * groups
    gen = cref.g1_generator()
    pts = h.g1_fixed_base_mul(rand_fr_gpu(groups, 77), gen).cpu().numpy().view(np.uint64).copy()
    neg = pts.copy()
    P = o.P
    for i in range(groups):
        y = o.from_limbs(pts[i, 4:])
        neg[i, 4:] = o.to_limbs((P - y) % P)
    bases = np.empty((n, 8), dtype=np.uint64)
    for j, sg in enumerate(signs):
        bases[j::m] = pts if sg > 0 else neg
    sc = rand_fr_gpu(groups, 78).cpu().numpy().view(np.uint64)
    s = np.repeat(sc, m, axis=0)
    exp = cref.g1_to_affine(cref.best_multiexp(s, bases, 8))[0]
    assert g1_equal(h.best_multiexp(s, bases), exp)
    hd = h.register_bases(bases, precompute=True)
    try:
        assert g1_equal(h.best_multiexp(s, hd), exp)
    finally:
        h.release_bases(hd)


@pytest.mark.parametrize("alternate", [False, True])
def test_shared_set_reduction_meets_equal_and_opposite_bucket_values(cref, pyref, alternate):
    """Every base the same point G (or G, -G, G, ...): the table's entries of one window are all equal, so the buckets of
    the shared set hold small multiples of the same few points and the row / column sums and bit sums of the reduction
    (msm_reduce_rowcol_kernel, msm_reduce_bits_kernel) keep adding a value to itself or to its negative -- the doubling
    and the cancellation branch of the general law inside plain sums.  Known answer: [sum (+-) s_i] G."""
    import torch
    o = pyref
    n = 1 << 17
    gen = cref.g1_generator()
    bases = np.tile(gen, (n, 1))
    x = o.fr_array([1])[0]
    if alternate:
        neg = gen.copy()
        neg[4:] = o.to_limbs((o.P - o.from_limbs(gen[4:])) % o.P)
        bases[1::2] = neg
        x = o.fr_array([o.R - 1])[0]
    f = rand_fr_gpu(n, 4242 + int(alternate))
    exp = cref.g1_mul(cref.fr_horner(f.cpu().numpy().view(np.uint64), x), gen)
    hd = h.register_bases(np.ascontiguousarray(bases))
    try:
        got = h.best_multiexp(f, hd)
        assert h.msm_stats()["window_bits"] >= 16 and h.msm_stats()["windows"] <= 16      # the shared bucket set
        assert g1_equal(got, exp)
    finally:
        h.release_bases(hd)


@pytest.mark.parametrize("n", [257, 600, 1500, 2049, 4097, 6000, 8191, 12000, 20000, 40000, 70000])
def test_ragged_small_sizes_across_the_window_table(cref, pyref, n):
    """One size inside every entry of the window table below 2^17 (and the bucket-reduction segment
    rule that goes with it), none of them a power of two; uniform and prover-like scalars."""
    o = pyref
    gen = cref.g1_generator()
    bases = h.g1_fixed_base_mul(rand_fr_gpu(n, 4000 + n % 97), gen).cpu().numpy().view(np.uint64).copy()
    uni = rand_fr_gpu(n, 5000 + n % 89).cpu().numpy().view(np.uint64)
    tile = o.fr_array(o.rand_scalars(257, n, "prover"))
    pro = np.concatenate([np.tile(tile, (n // 257, 1)), tile[: n % 257]])
    for s in (uni, pro):
        exp = cref.g1_to_affine(cref.best_multiexp(s, bases, 4))[0]
        assert g1_equal(h.best_multiexp(s, bases), exp)


@pytest.mark.parametrize("mode", ["lanes", "direct"])
def test_host_pointer_forms_through_both_copy_paths(cref, mode):
    """The host-pointer forms move their arrays either through the runtime's pageable path or through the library's pinned staging lanes
    (csrc/xfer.hip; the default policy sends everything the caller has not registered through the lanes): the same results either way --
    the drop-in pointer form (bases + scalars), the handle form, a phase of commitments from host arrays, and best_fft, at sizes that
    are not a whole number of lanes, slots or pages."""
    import ctypes
    from halo2_experiments_amd import _lib
    from halo2_experiments_amd.arithmetic import G1_GENERATOR, best_multiexp_batch
    lib = _lib.load()
    _lib.check(lib.hm_set_host_copies(1 if mode == "lanes" else 2))
    st0 = _lib.Stats()
    _lib.check(lib.hm_get_stats(ctypes.byref(st0)))
    try:
        for n in (8193, 100003, (1 << 18) + 5):                          # 256 KiB + 32 B, 3.05 MiB, 8 MiB + 160 B of scalars
            s_d = rand_fr_gpu(n, 7700 + n % 97)
            b_d = h.g1_fixed_base_mul(rand_fr_gpu(n, 7800 + n % 89), G1_GENERATOR)
            s, b = s_d.cpu().numpy().view(np.uint64).copy(), b_d.cpu().numpy().view(np.uint64).copy()
            want = cref.g1_to_affine(cref.best_multiexp(s, b, 8))[0]
            assert np.array_equal(h.best_multiexp(s, b)[:8], want), (mode, n)          # pointer form: both arrays cross
            hd = h.register_bases(b)                                                   # host registration: the bases cross
            try:
                assert np.array_equal(h.best_multiexp(s, hd)[:8], want), (mode, n)
                cols = [s, np.roll(s, 1, axis=0).copy(), s.copy()]
                got = best_multiexp_batch(cols, hd)
                assert np.array_equal(got[0][:8], want) and np.array_equal(got[2], got[0]), (mode, n)
                assert np.array_equal(got[1], h.best_multiexp(cols[1], hd)), (mode, n)
            finally:
                h.release_bases(hd)
        st1 = _lib.Stats()
        _lib.check(lib.hm_get_stats(ctypes.byref(st1)))
        # every copy of 256 KiB or more went the way the mode says (smaller ones always go straight to hipMemcpy)
        if mode == "lanes":
            assert st1.host_copies_staged > st0.host_copies_staged
        else:
            assert st1.host_copies_staged == st0.host_copies_staged and st1.host_copies_direct > st0.host_copies_direct
    finally:
        _lib.check(lib.hm_set_host_copies(0))
