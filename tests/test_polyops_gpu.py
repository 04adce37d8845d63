"""GPU suite: kate_division, grand product, batch inversion, linear combination through the C ABI, bit-exact against
oracle/poly_ref.py (sizes the oracle finishes in seconds) and through size-independent identities at 2^20 / 2^22."""
import ctypes
import random

import numpy as np
import pytest

import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from halo2_experiments_amd.domain import fr_words
from oracle import poly_ref as pr

pytestmark = pytest.mark.gpu
R = pr.R


def to_gpu(pyref, values):
    import torch
    return torch.from_numpy(pyref.fr_array(values).view(np.int64)).cuda()


def from_gpu(pyref, t):
    return pyref.fr_from_array(t.cpu().numpy().view(np.uint64))


def rand_fr_gpu(n, seed):
    from halo2_experiments_amd.arithmetic import random_fr
    return random_fr(n, seed, "cuda")                 # uniform over the whole of [0, r)


SIZES = [1, 2, 3, 4, 5, 255, 256, 257, 1023, 1024, 1025, 4099, (1 << 14) - 1, 1 << 16, (1 << 16) + 1, (1 << 18) + 7]


@pytest.mark.parametrize("n", SIZES)
def test_kate_division_matches_oracle(pyref, n):
    """Every plan boundary: one lane, one workgroup, several workgroups, ragged last chunks (B = 4 up to 2^18, 5 beyond)."""
    rng = random.Random(n)
    a = [rng.randrange(R) for _ in range(n)]
    if n > 8:
        a[n - 1] = 0                      # a zero leading coefficient
        a[3] = R - 1
    d = to_gpu(pyref, a)
    for z in ([0, 1, R - 1, rng.randrange(R)] if n <= 4099 else [rng.randrange(R)]):
        q = h.kate_division(d, fr_words(z))
        assert q.shape == (n - 1, 4)
        assert from_gpu(pyref, q) == pr.kate_division(a, z), (n, z)


def test_kate_division_large_identity(pyref, cref):
    """n = 2^22 (B = 64): q(X) (X - z) + a(z) = a(X), checked at random points with the Horner kernel."""
    import torch
    n = 1 << 22
    a = rand_fr_gpu(n, 42)
    z = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF % R
    q = h.kate_division(a, fr_words(z))
    torch.cuda.synchronize()
    az = pyref.fr_from_array(h.eval_polynomial(a.reshape(1, n, 4), np.stack([fr_words(z)])))[0]
    for x in (5, 0xDEADBEEFCAFEBABE, R - 2):
        ax = pyref.fr_from_array(h.eval_polynomial(a.reshape(1, n, 4), np.stack([fr_words(x)])))[0]
        qx = pyref.fr_from_array(h.eval_polynomial(q.reshape(1, n - 1, 4), np.stack([fr_words(x)])))[0]
        assert (qx * (x - z) + az) % R == ax


def test_kate_division_errors(pyref):
    import torch
    a = rand_fr_gpu(16, 1)
    lib = _lib.load()
    rc = lib.hm_kate_division_bn256_fr_dev(ctypes.c_void_p(a.data_ptr()), 16, fr_words(3).ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)),
                                           ctypes.c_void_p(a.data_ptr() + 32), None)
    assert rc != 0 and b"overlaps" in lib.hm_last_error()
    with pytest.raises(ValueError):
        h.kate_division(torch.empty((0, 4), dtype=torch.int64, device="cuda"), fr_words(3))
    assert h.kate_division(a[:1], fr_words(3)).shape == (0, 4)


@pytest.mark.parametrize("n", SIZES)
def test_grand_product_matches_oracle(pyref, n):
    rng = random.Random(1000 + n)
    m = [rng.randrange(R) for _ in range(n)]
    if n > 300:
        m[77] = 1
        m[n // 2] = R - 1
    start = rng.randrange(1, R)
    d = to_gpu(pyref, m)
    z = h.grand_product(d, fr_words(start))
    assert from_gpu(pyref, z) == pr.grand_product(m, start), n
    assert from_gpu(pyref, d) == m                                   # input untouched
    h.grand_product(d, fr_words(start), out=d)                       # in place
    assert from_gpu(pyref, d) == pr.grand_product(m, start), n


def test_grand_product_with_a_zero_factor(pyref):
    rng = random.Random(5)
    m = [rng.randrange(1, R) for _ in range(3000)]
    m[1234] = 0
    z = from_gpu(pyref, h.grand_product(to_gpu(pyref, m), fr_words(1)))
    assert z == pr.grand_product(m, 1) and z[1234] != 0 and all(v == 0 for v in z[1235:])


@pytest.mark.parametrize("n", [1, 7, 8, 9, 511, 512, 513, 4099, (1 << 14) + 3, 1 << 16])
def test_batch_invert_matches_oracle(pyref, n):
    rng = random.Random(2000 + n)
    v = [rng.randrange(R) for _ in range(n)]
    for i in (0, 5, n - 1, n // 2):                                  # zeros: first, last, inside a chunk
        if n > 16 or i == 0 and n > 1:
            v[i % n] = 0
    if n > 20:
        v[17] = 1
        v[18] = R - 1
        for i in range(64, min(n, 72)):                              # one lane's whole chunk zero
            v[i] = 0
    d = to_gpu(pyref, v)
    h.batch_invert(d)
    assert from_gpu(pyref, d) == pr.batch_invert(v), n


@pytest.mark.parametrize("n", [1 << 20, (1 << 21) + 5, 11 << 18, 65536 * 60 + 3, (1 << 22) + 17])
def test_batch_invert_large_is_an_involution(pyref, n):
    """2^20 (8 elements per lane), 2^21 + 5 (32 per lane, ragged tail), the 11 x 2^18 denominators of the k = 18 proof (48 per
    lane: one round of waves), 60 x 65536 + 3 (64 per lane) and 2^22 + 17 (32 again): x * x^-1 = 1 at sampled rows, planted
    zeros stay zero -- at every position of a 64-element lane chunk --, and inverting twice returns the input."""
    import torch
    a = rand_fr_gpu(n, 77)
    zeros = [3, 31, 32, 33, 47, 48, 63, 64, 64 * 32 - 1, 64 * 48 - 1, 64 * 64 - 1, n // 2, n - 1]
    a[zeros] = 0
    b = a.clone()
    h.batch_invert(b)
    rows = [0, 1, 7, 8, 30, 34, 46, 49, 62, 65, 4094, n // 3, n - 9, n - 2]
    av, bv = from_gpu(pyref, a[rows]), from_gpu(pyref, b[rows])
    for x, y in zip(av, bv):
        assert x * y % R == 1
    assert not b[zeros].any()
    h.batch_invert(b)
    assert torch.equal(a, b)


@pytest.mark.parametrize("count", [0, 1, 2, 4, 5, 23, 24, 25, 49, 60])
def test_linear_combination_matches_oracle(pyref, count):
    """1 .. 60 terms: one launch carries 24; later launches re-read the running sum."""
    import torch
    n = 777
    rng = random.Random(3000 + count)
    polys = [[rng.randrange(R) for _ in range(n)] for _ in range(count)]
    cs = [rng.randrange(R) for _ in range(count)]
    if count > 2:
        cs[1] = 0
        cs[2] = R - 1
    d = [to_gpu(pyref, p) for p in polys]
    if count == 0:
        out = torch.full((n, 4), 5, dtype=torch.int64, device="cuda")
        h.linear_combination([], [], out=out)
        assert not out.any()
        return
    got = h.linear_combination(d, np.stack([fr_words(c) for c in cs]))
    assert from_gpu(pyref, got) == pr.linear_combination(polys, cs, n)
    # accumulate into one of the inputs: the first, or (more than one launch) one that a later launch would have read
    tgt = count - 1 if count > 24 else 0
    h.linear_combination(d, np.stack([fr_words(c) for c in cs]), out=d[tgt])
    assert from_gpu(pyref, d[tgt]) == pr.linear_combination(polys, cs, n)
    with pytest.raises(ValueError):
        h.linear_combination(d, np.stack([fr_words(c) for c in cs])[:-1] if count > 1 else np.zeros((2, 4), dtype=np.uint64))


def test_permutation_product_column(pyref):
    """The z column of the permutation argument for one chunk of three columns, the way upstream's
    permutation/prover.rs builds it: denominators, batch inversion, numerators, running product -- the factor sweep
    by linear combinations on the device, the last two steps by the kernels under test."""
    k, n = 10, 1 << 10
    rng = random.Random(31)
    omega = pyref.fr_omega(k)
    delta = pow(7, 1 << 28, R)
    beta, gamma = rng.randrange(R), rng.randrange(R)
    cols = [[rng.randrange(R) for _ in range(n)] for _ in range(3)]
    perm = list(range(3 * n))
    rng.shuffle(perm)                                               # a random permutation of the 3n cells
    ident = [pow(delta, j, R) * pow(omega, i, R) % R for j in range(3) for i in range(n)]
    sig = [[ident[perm[j * n + i]] for i in range(n)] for j in range(3)]
    want_mv = pr.permutation_factors(cols, sig, omega, delta, beta, gamma)
    want_z = pr.grand_product(want_mv, 1)
    # device: den_j = beta * sigma_j + gamma + v_j, num_j = beta * id_j + gamma + v_j as linear combinations
    ones = to_gpu(pyref, [1] * n)
    dc, ds = [to_gpu(pyref, c) for c in cols], [to_gpu(pyref, s) for s in sig]
    di = [to_gpu(pyref, ident[j * n:(j + 1) * n]) for j in range(3)]
    w = lambda *v: np.stack([fr_words(x) for x in v])
    den = [h.linear_combination([ds[j], ones, dc[j]], w(beta, gamma, 1)) for j in range(3)]
    num = [h.linear_combination([di[j], ones, dc[j]], w(beta, gamma, 1)) for j in range(3)]
    mul = lambda a, b: to_gpu(pyref, [x * y % R for x, y in zip(from_gpu(pyref, a), from_gpu(pyref, b))])   # host glue of the test
    d_all = mul(mul(den[0], den[1]), den[2])
    h.batch_invert(d_all)
    mv = mul(mul(mul(d_all, num[0]), num[1]), num[2])
    assert from_gpu(pyref, mv) == want_mv
    z = h.grand_product(mv, fr_words(1))
    assert from_gpu(pyref, z) == want_z
    # a permutation of equal... the product over all rows telescopes to 1 when the columns satisfy the permutation;
    # with random columns it does not, but z[0] = 1 always
    assert from_gpu(pyref, z[:1]) == [1]


def test_call_counters_cover_the_vector_entry_points(pyref):
    """hm_get_stats counts the entry points beyond best_multiexp / best_fft by kind (the measured call trace of SURVEY.md §5)."""
    lib = _lib.load()
    _lib.check(lib.hm_reset_stats())
    a = rand_fr_gpu(1000, 1)
    h.kate_division(a, fr_words(3))
    h.grand_product(a, fr_words(1))
    h.batch_invert(a.clone())
    h.linear_combination([a, a], np.stack([fr_words(2), fr_words(3)]))
    h.eval_polynomial(a.reshape(1, 1000, 4), np.stack([fr_words(5), fr_words(6)]), poly_index=[0, 0])
    h.permute_expression_pairs([a, a], [a, a], 990)
    st = _lib.Stats()
    _lib.check(lib.hm_get_stats(ctypes.byref(st)))
    got = {k: (st.vector_calls[i], st.vector_elements[i]) for i, k in enumerate(_lib.Stats.KINDS)}
    assert got == {"eval_polynomial": (2, 2000), "graph_evaluate": (0, 0), "kate_division": (1, 1000), "grand_product": (1, 1000),
                   "batch_invert": (1, 1000), "linear_combination": (1, 2000), "lookup_permute": (2, 1980)}


# ---- the batched forms: several columns in one launch chain -------------------------------------------------------
@pytest.mark.parametrize("n,count", [(1, 3), (5, 2), (257, 5), (4099, 16), (4099, 17), ((1 << 16) + 1, 4), ((1 << 18) + 7, 3), (1000, 35)])
def test_kate_division_batch_equals_the_calls_one_by_one(pyref, n, count):
    """Same words as hm_kate_division_bn256_fr_dev column by column (itself checked against the oracle above), and
    against the oracle directly where that takes seconds; counts above 16 take several launch groups."""
    rng = random.Random(n * 31 + count)
    polys = [rand_fr_gpu(n, 7000 + 13 * j + n) for j in range(count)]
    zs = [rng.choice([0, 1, R - 1, rng.randrange(R)]) for _ in range(count)]
    got = h.kate_division_batch(polys, [fr_words(z) for z in zs])
    assert len(got) == count
    for j in range(count):
        one = h.kate_division(polys[j], fr_words(zs[j]))
        assert got[j].shape == (n - 1, 4) and bool((got[j] == one).all()), (n, j)
    if n <= 4099:
        for j in (0, count - 1):
            assert from_gpu(pyref, got[j]) == pr.kate_division(from_gpu(pyref, polys[j]), zs[j])


@pytest.mark.parametrize("n,count", [(1, 3), (4, 2), (5, 2), (257, 5), (4099, 16), (4099, 18), ((1 << 16) + 1, 4), ((1 << 18) + 7, 3), (1000, 33)])
def test_grand_product_batch_unchained(pyref, n, count):
    rng = random.Random(n * 17 + count)
    start = rng.randrange(1, R)
    cols = [rand_fr_gpu(n, 9000 + 7 * j + n) for j in range(count)]
    got = h.grand_product_batch(cols, fr_words(start))
    for j in range(count):
        assert bool((got[j] == h.grand_product(cols[j], fr_words(start))).all()), (n, j)
    if n <= 4099:
        assert from_gpu(pyref, got[-1]) == pr.grand_product(from_gpu(pyref, cols[-1]), start)
    keep = [c.clone() for c in cols]
    h.grand_product_batch(cols, fr_words(start), outs=cols)             # in place
    for j in range(count):
        assert bool((cols[j] == got[j]).all()) and not bool((keep[j] == cols[j]).all()) or n == 1


@pytest.mark.parametrize("n,count,u", [(1, 3, 0), (8, 4, 0), (8, 4, 7), (300, 5, 293), (4099, 16, 4092), (4099, 20, 1), (4099, 37, 4098),
                                       ((1 << 16) + 1, 3, 1 << 16), ((1 << 18), 11, (1 << 18) - 6), ((1 << 18) + 7, 2, 255 * 1028 + 3)])
def test_grand_product_batch_chained_is_the_last_z_recurrence(pyref, n, count, u):
    """Upstream's permutation prover: z_{j+1}[0] = z_j[u], u = n - (blinding_factors + 1).  Checked against the single-column
    form fed with the previous column's element read back, and against the oracle at small sizes."""
    rng = random.Random(n + count + u)
    start = rng.randrange(1, R)
    cols = [rand_fr_gpu(n, 11000 + 5 * j + n) for j in range(count)]
    got = h.grand_product_batch(cols, fr_words(start), chain_row=u)
    st = fr_words(start)
    for j in range(count):
        one = h.grand_product(cols[j], st)
        assert bool((got[j] == one).all()), (n, j)
        st = one[u].cpu().numpy().view(np.uint64)
    if n <= 4099:
        s, want = start, None
        for j in range(count):
            want = pr.grand_product(from_gpu(pyref, cols[j]), s)
            s = want[u]
        assert from_gpu(pyref, got[-1]) == want
    outs = h.grand_product_batch(cols, fr_words(start), chain_row=u, outs=cols)   # in place: the chain reads factors, not outputs
    for j in range(count):
        assert bool((outs[j] == got[j]).all())


def test_batched_scans_reject_bad_calls(pyref):
    import torch
    lib = _lib.load()
    a, b = rand_fr_gpu(64, 1), rand_fr_gpu(64, 2)
    with pytest.raises(ValueError):
        h.grand_product_batch([a, b[:32]], fr_words(1))
    with pytest.raises(ValueError):
        h.grand_product_batch([a, b], fr_words(1), chain_row=64)
    with pytest.raises(ValueError):
        h.kate_division_batch([a, b], [fr_words(1)])
    assert h.kate_division_batch([], []) == [] and h.grand_product_batch([], fr_words(1)) == []
    # one column's output over another column's factors
    ptrs = (ctypes.c_void_p * 2)(a.data_ptr(), b.data_ptr())
    outs = (ctypes.c_void_p * 2)(b.data_ptr(), a.data_ptr())
    one = fr_words(1).ctypes.data_as(ctypes.POINTER(ctypes.c_uint64))
    assert lib.hm_fr_grand_product_batch_dev(ptrs, 64, one, _lib.NO_CHAIN, outs, 2, None) == -1
    zz = np.stack([fr_words(3), fr_words(4)])
    assert lib.hm_kate_division_batch_bn256_fr_dev(ptrs, 64, zz.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), ptrs, 2, None) == -1
    q = torch.empty((126, 4), dtype=torch.int64, device="cuda")
    both = (ctypes.c_void_p * 2)(q.data_ptr(), q.data_ptr() + 32)      # two quotients over each other
    assert lib.hm_kate_division_batch_bn256_fr_dev(ptrs, 64, zz.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), both, 2, None) == -1
    assert lib.hm_fr_grand_product_batch_dev(None, 64, one, _lib.NO_CHAIN, outs, 2, None) == -1
    assert lib.hm_fr_grand_product_batch_dev(ptrs, 0, one, _lib.NO_CHAIN, outs, 2, None) == 0     # nothing to do
