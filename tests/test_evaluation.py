"""evaluate_h's gate arithmetic: the GraphEvaluator mirror against the oracle's restatement and
against direct evaluation of the expression trees -- builder rules on the CPU, the device interpreter on the GPU."""
import random

import numpy as np
import pytest

from halo2_experiments_amd import evaluation as ev
from halo2_experiments_amd.domain import FR_MODULUS, fr_words

R = FR_MODULUS


def poseidon_like_gates(width=3):
    """Gate polynomials of the shape the reference's Poseidon chip gives halo2 (s * (x^5 + rc) mixed by an MDS row minus the
    next state; /root/reference/src/chips/poseidon/hash.rs:50-57 configures Pow5Chip): selectors are fixed columns."""
    s_full, s_partial = ev.Fixed(0), ev.Fixed(1)
    rc = [ev.Fixed(2 + i) for i in range(width)]
    state = [ev.Advice(i) for i in range(width)]
    nxt = [ev.Advice(i, 1) for i in range(width)]
    mds = [[(3 * i + 7 * j + 11) % R for j in range(width)] for i in range(width)]

    def pow5(x):
        x2 = x * x
        return x2 * x2 * x

    polys = []
    for i in range(width):
        acc = None
        for j in range(width):
            term = pow5(state[j] + rc[j]) * mds[i][j]
            acc = term if acc is None else acc + term
        polys.append(s_full * (acc - nxt[i]))
    polys.append(s_partial * (pow5(state[0] + rc[0]) - ev.Advice(0, -1)))
    polys.append(s_full * (1 - s_full))                                   # boolean selector
    polys.append(ev.Challenge(0) * (ev.Instance(0) - state[1]) * s_partial)
    return polys, 2 + width, width, 1


def test_builder_rules_and_cse():
    """The peephole rules of add_expression and the deduplication of calculations (upstream evaluation.rs)."""
    g = ev.GraphEvaluator()
    a, b = ev.Advice(0), ev.Advice(1, 1)
    r1 = g.add_expression(a * b)
    r2 = g.add_expression(b * a)                      # commutative operands are ordered: the same calculation
    assert r1 == r2
    n = len(g.calculations)
    assert g.add_expression(a - b)[0] == "Intermediate" and g.calculations[-1][0][0] == "Sub"
    assert g.add_expression(a * a)[0] == "Intermediate" and g.calculations[-1][0][0] == "Square"
    assert g.add_expression(a * ev.Constant(2)) == g.add_expression(ev.Constant(2) * a) and g.calculations[-1][0][0] == "Double"
    assert g.add_expression(a * ev.Constant(1)) == g.add_expression(a)
    assert g.add_expression(a * ev.Constant(0)) == ("Constant", 0)
    assert g.add_expression(ev.Constant(0) + a) == g.add_expression(a)
    assert g.add_expression(-ev.Constant(5)) == ("Constant", g.constants.index(R - 5))
    assert g.add_expression(ev.Scaled(a, 1)) == g.add_expression(a)
    assert g.rotations == [0, 1] and g.constants[:3] == [0, 1, 2]
    assert len(g.calculations) > n
    before = len(g.calculations)
    g.add_expression(a * b)                            # already there
    assert len(g.calculations) == before


def test_graph_equals_direct_expression_evaluation(pyref):
    """Oracle's graph interpreter == direct evaluation of the expression trees, folded by Horner in y."""
    from oracle import graph_ref
    rng = random.Random(7)
    polys, nf, na, ni = poseidon_like_gates()
    g = ev.GraphEvaluator()
    g.add_custom_gates(polys)
    isize, rot_scale = 16, 2
    cols = lambda c: [[rng.randrange(R) for _ in range(isize)] for _ in range(c)]
    fixed, advice, instance = cols(nf), cols(na), cols(ni)
    ch, y = [rng.randrange(R)], rng.randrange(R)
    prev = [rng.randrange(R) for _ in range(isize)]
    got = graph_ref.evaluate_graph(g.calculations, g.constants, g.rotations, fixed, advice, instance, ch, 0, 0, 0, y, prev, rot_scale, isize)
    for idx in range(isize):
        v = prev[idx]
        for p in polys:
            v = (v * y + graph_ref.evaluate_expression(p, fixed, advice, instance, ch, idx, rot_scale, isize)) % R
        assert got[idx] == v, idx


def _random_expression(rng, depth, nf, na, ni):
    if depth == 0 or rng.random() < 0.2:
        k = rng.randrange(5)
        if k == 0:
            return ev.Constant(rng.choice([0, 1, 2, 3, R - 1, rng.randrange(R)]))
        if k == 1:
            return ev.Fixed(rng.randrange(nf), rng.choice([0, 0, 1, -1]))
        if k == 2:
            return ev.Instance(rng.randrange(ni), rng.choice([0, 2]))
        if k == 3:
            return ev.Challenge(0)
        return ev.Advice(rng.randrange(na), rng.choice([0, 0, 1, -1, 3, -2]))
    k = rng.randrange(5)
    a = _random_expression(rng, depth - 1, nf, na, ni)
    if k == 0:
        return -a
    if k == 1:
        return ev.Scaled(a, rng.choice([0, 1, 2, rng.randrange(R)]))
    b = _random_expression(rng, depth - 1, nf, na, ni)
    return a + b if k == 2 else (a - b if k == 3 else a * b)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["poseidon", "random0", "random1", "random2", "empty"])
def test_device_graph_matches_oracle(pyref, case):
    """The device interpreter (lowered program, liveness-allocated slots, rotations with wrap-around, PreviousValue,
    per-call challenge / y constants) against the oracle's restatement of GraphEvaluator::evaluate."""
    import torch
    from oracle import graph_ref
    rng = random.Random(hash(case) & 0xFFFF)
    if case == "poseidon":
        polys, nf, na, ni = poseidon_like_gates()
    elif case == "empty":
        polys, nf, na, ni = [], 1, 1, 1
    else:
        nf, na, ni = 2, 3, 1
        polys = [_random_expression(rng, 5, nf, na, ni) for _ in range(6)]
    g = ev.GraphEvaluator()
    g.add_custom_gates(polys)
    k, ek = 5, 7
    isize, rot_scale = 1 << ek, 1 << (ek - k)
    mk = lambda c: [[rng.randrange(R) for _ in range(isize)] for _ in range(c)]
    fixed, advice, instance = mk(nf), mk(na), mk(ni)
    ch, y = [rng.randrange(R)], rng.randrange(R)
    prev = [rng.randrange(R) for _ in range(isize)]
    exp = graph_ref.evaluate_graph(g.calculations, g.constants, g.rotations, fixed, advice, instance, ch, 0, 0, 0, y, prev, rot_scale, isize)
    to_dev = lambda col: torch.from_numpy(pyref.fr_array(col).view(np.int64)).cuda()
    cols = [to_dev(c) for c in fixed + advice + instance]
    values = to_dev(prev)
    prog = g.compile(nf, na, ni, num_challenges=1, rot_scale=rot_scale)
    try:
        prog.evaluate(cols, values, challenges=ch, y=y)
        torch.cuda.synchronize()
        got = values.cpu().numpy().view(np.uint64)
        assert np.array_equal(got, pyref.fr_array(exp)), case
        # internal-form columns (32 x the value): the other lowering of the same program
        v32 = to_dev(prev)
        prog.evaluate([to_dev([32 * v % R for v in c]) for c in fixed + advice + instance], v32, challenges=ch, y=y, columns_internal=True)
        torch.cuda.synchronize()
        assert np.array_equal(v32.cpu().numpy().view(np.uint64), pyref.fr_array(exp)), case + " (internal-form columns)"
        # a second proof: other challenges on the same compiled program, chained onto the first result (PreviousValue)
        ch2, y2 = [rng.randrange(R)], rng.randrange(R)
        exp2 = graph_ref.evaluate_graph(g.calculations, g.constants, g.rotations, fixed, advice, instance, ch2, 0, 0, 0, y2, exp, rot_scale, isize)
        prog.evaluate(cols, values, challenges=ch2, y=y2)
        torch.cuda.synchronize()
        assert np.array_equal(values.cpu().numpy().view(np.uint64), pyref.fr_array(exp2)), case
    finally:
        prog.destroy()


@pytest.mark.gpu
def test_device_graph_at_prover_size_and_argument_checks(pyref):
    """k = 16 extended to 2^19 rows (grid-stride loop, more rows than lanes): spot rows against the oracle; bad programs
    and mismatched calls are errors, not faults."""
    import ctypes
    import torch
    from halo2_experiments_amd import _lib
    from oracle import graph_ref
    rng = random.Random(99)
    polys, nf, na, ni = poseidon_like_gates()
    g = ev.GraphEvaluator()
    g.add_custom_gates(polys)
    k, ek = 16, 19
    isize, rot_scale = 1 << ek, 1 << (ek - k)
    from halo2_experiments_amd.arithmetic import random_fr
    seeds = iter(range(500, 10 ** 6))
    def rand_col():
        return random_fr(isize, next(seeds), "cuda")           # uniform over the whole of [0, r)
    cols = [rand_col() for _ in range(nf + na + ni)]
    values = rand_col()
    prev_h = values.cpu().numpy().view(np.uint64).copy()
    ch, y = [rng.randrange(R)], rng.randrange(R)
    prog = g.compile(nf, na, ni, num_challenges=1, rot_scale=rot_scale)
    try:
        prog.evaluate(cols, values, challenges=ch, y=y)
        torch.cuda.synchronize()
        got = values.cpu().numpy().view(np.uint64)
        hosts = [c.cpu().numpy().view(np.uint64) for c in cols]
        mont = lambda w: pyref.from_limbs(w) * pow(1 << 256, -1, R) % R
        for idx in (0, 1, rot_scale - 1, 131071, 131072, isize - 1):
            # the oracle on a 1-row window: columns as sparse dicts around idx
            def view(h):
                return {j % isize: mont(h[j % isize]) for r in g.rotations for j in [idx + r * rot_scale]}
            fx = [view(h) for h in hosts[:nf]]
            ad = [view(h) for h in hosts[nf:nf + na]]
            ins = [view(h) for h in hosts[nf + na:]]
            v = mont(prev_h[idx])
            for p in polys:
                v = (v * y + graph_ref.evaluate_expression(p, fx, ad, ins, ch, idx, rot_scale, isize)) % R
            assert np.array_equal(got[idx], fr_words(v)), idx
        with pytest.raises(ValueError):
            prog.evaluate(cols[:-1], values, challenges=ch, y=y)
    finally:
        prog.destroy()
    lib = _lib.load()
    h = ctypes.c_uint64(0)
    bad = np.array([[2, (1 << 30) | 5, 0, 0, 0]], dtype=np.uint32)            # reads intermediate 5 before any write
    assert lib.hm_graph_create(bad.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 1, None, 0, 0, None, 0, 0, 8, ctypes.byref(h)) == -1
    bad = np.array([[9, 0, 0, 0, 0]], dtype=np.uint32)                        # unknown operation
    assert lib.hm_graph_create(bad.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), 1, None, 0, 0, None, 0, 0, 8, ctypes.byref(h)) == -1


@pytest.mark.gpu
def test_permutation_and_lookup_arguments_and_vanishing_division(pyref):
    """The whole of evaluate_h after the gates: the permutation argument (two sets: chunked columns, the delta chain across
    sets, the rotation to the last usable row), one lookup argument (two-expression input and table compressed by theta),
    and divide_by_vanishing_poly as a periodically read column -- built as expression trees, run by the device evaluator,
    against the oracle's restatement of upstream's row loops."""
    import torch
    from oracle import graph_ref
    rng = random.Random(2024)
    k, ek = 4, 6
    n, isize, rot_scale = 1 << k, 1 << ek, 1 << (ek - k)
    w_ext = pyref.fr_omega(ek)
    zeta, delta = pyref.FR_ZETA if hasattr(pyref, "FR_ZETA") else 0x30644E72E131A029048B6E193FD84104CC37A73FEC2BC5E9B8CA0B2D36636F23, pow(7, 1 << 28, R)
    col = lambda: [rng.randrange(R) for _ in range(isize)]
    advice, fixed = [col() for _ in range(3)], [col() for _ in range(2)]
    sigmas = [col() for _ in range(5)]
    zs = [col(), col()]                                        # two permutation sets (chunk_len 3: 3 + 2 columns)
    l0, l_last, l_active = col(), col(), col()
    x_coset = [zeta * pow(w_ext, i, R) % R for i in range(isize)]
    lk_z, lk_a, lk_s = col(), col(), col()
    t_inv = [pow((pow(zeta * pow(w_ext, i, R) % R, n, R) - 1) % R, -1, R) for i in range(rot_scale)]
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    prev = col()
    chunk_len, last_rotation = 3, -6
    # column table: fixed = [f0, f1, sigma0..4, z0, z1, l0, l_last, l_active, x_coset, lk_z, lk_a, lk_s, t_inv]; advice = 3
    F = ev.Fixed
    fx_all = fixed + sigmas + zs + [l0, l_last, l_active, x_coset, lk_z, lk_a, lk_s, t_inv]
    i_sig, i_z, i_l0, i_ll, i_la, i_x, i_lz, i_lka, i_lks, i_t = 2, 7, 9, 10, 11, 12, 13, 14, 15, 16
    perm_cols_e = [ev.Advice(0), ev.Advice(1), ev.Advice(2), F(0), F(1)]
    perm_cols_v = advice + fixed
    polys = ev.permutation_expressions(perm_cols_e, [F(i_sig + j) for j in range(5)], [lambda r, i=i: F(i_z + i, r) for i in range(2)],
                                       F(i_l0), F(i_ll), F(i_la), F(i_x), chunk_len, delta, last_rotation)
    inputs_e = [ev.Advice(0) * ev.Advice(1), ev.Advice(2) + F(0)]
    tables_e = [F(1), F(0) * 3]
    polys += ev.lookup_expressions(inputs_e, tables_e, lambda r: F(i_lz, r), lambda r: F(i_lka, r), lambda r: F(i_lks, r),
                                   F(i_l0), F(i_ll), F(i_la))
    g = ev.GraphEvaluator()
    g.add_custom_gates(polys)
    g.add_vanishing_division(F(i_t))
    inputs_v = [[advice[0][i] * advice[1][i] % R for i in range(isize)], [(advice[2][i] + fixed[0][i]) % R for i in range(isize)]]
    tables_v = [fixed[1], [fixed[0][i] * 3 % R for i in range(isize)]]
    exp = graph_ref.evaluate_h_permutation_and_lookups(prev, y, beta, gamma, theta, isize, rot_scale, w_ext, zeta, delta, perm_cols_v,
                                                       sigmas, zs, chunk_len, last_rotation, l0, l_last, l_active,
                                                       [(inputs_v, tables_v, lk_z, lk_a, lk_s)], t_inverse=t_inv)
    to_dev = lambda c: torch.from_numpy(pyref.fr_array(c).view(np.int64)).cuda()
    cols = [to_dev(c) for c in fx_all + advice]
    values = to_dev(prev)
    prog = g.compile(len(fx_all), 3, 0, num_challenges=0, rot_scale=rot_scale, short_columns={i_t: ek - k})
    try:
        prog.evaluate(cols, values, beta=beta, gamma=gamma, theta=theta, y=y)
        torch.cuda.synchronize()
        assert np.array_equal(values.cpu().numpy().view(np.uint64), pyref.fr_array(exp))
    finally:
        prog.destroy()


@pytest.mark.gpu
def test_device_graph_with_the_largest_argument_block(pyref):
    """The per-call argument block at its limits: 256 columns (the table's maximum), several of them short and read
    periodically with different periods, 12 challenges + beta / gamma / theta / y = 16 per-call constants.  Rounds 1-2
    passed that block by value in the kernel-argument segment and one variant of it aborted at run time; it now travels
    through a device buffer (csrc/graph.hip), and this shape is the regression test for it."""
    import torch
    from oracle import graph_ref
    rng = random.Random(256)
    nf, na, ni, nch = 200, 50, 6, 12
    ek, k = 8, 6
    isize, rot_scale = 1 << ek, 1 << (ek - k)
    short = {3: 2, 77: 4, 199: 1, 255: 3}                    # column table index -> log2(rows)
    polys = []
    for j in range(0, nf + na + ni, 5):                      # every region of the table is read, the last entry included
        col = ev.Fixed(j) if j < nf else (ev.Advice(j - nf, rng.choice([-1, 0, 1])) if j < nf + na else ev.Instance(j - nf - na))
        polys.append(col * ev.Challenge(j % nch) + ev.Fixed(3) * ev.Fixed(77, 1))
    polys.append(ev.Fixed(199) * ev.Instance(5, -1) - ev.Advice(49, 1) * ev.Challenge(11))
    polys.append(ev.Instance(5) * ev.Fixed(199, 1))           # table index 255 is Instance(5)
    g = ev.GraphEvaluator()
    g.add_custom_gates(polys)
    rows = lambda i: 1 << short[i] if i in short else isize
    table = [[rng.randrange(R) for _ in range(rows(i))] for i in range(nf + na + ni)]
    # the oracle reads full-size columns: a short column is its pattern repeated
    full = [c * (isize // len(c)) for c in table]
    ch = [rng.randrange(R) for _ in range(nch)]
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    prev = [rng.randrange(R) for _ in range(isize)]
    exp = graph_ref.evaluate_graph(g.calculations, g.constants, g.rotations, full[:nf], full[nf:nf + na], full[nf + na:], ch, beta, gamma,
                                   theta, y, prev, rot_scale, isize)
    to_dev = lambda col: torch.from_numpy(pyref.fr_array(col).view(np.int64)).cuda()
    cols = [to_dev(c) for c in table]
    values = to_dev(prev)
    prog = g.compile(nf, na, ni, num_challenges=nch, rot_scale=rot_scale, short_columns=short)
    try:
        assert prog.n_columns == 256 and prog.n_dynamic == 16
        for _ in range(2):                                    # the second call reuses the stream's argument buffer
            values.copy_(to_dev(prev))
            prog.evaluate(cols, values, challenges=ch, beta=beta, gamma=gamma, theta=theta, y=y)
            torch.cuda.synchronize()
            assert np.array_equal(values.cpu().numpy().view(np.uint64), pyref.fr_array(exp))
        side = torch.cuda.Stream()                            # another stream: its own AuxSlot, its own argument buffer
        with torch.cuda.stream(side):
            v2 = to_dev(prev)
            prog.evaluate(cols, v2, challenges=ch, beta=beta, gamma=gamma, theta=theta, y=y)
        side.synchronize()
        assert np.array_equal(v2.cpu().numpy().view(np.uint64), pyref.fr_array(exp))
    finally:
        prog.destroy()
