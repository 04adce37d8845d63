"""Oracle evidence AT THE SHAPES THAT ARE TIMED (missing 2 / weak 2): bench.py and replay.py run the MerkleSumTree circuit's
own evaluate_h program at k = 18 (2^21 rows x 83 columns) and MerkleTreeV3's at k = 17 (2^20 rows) -- whole-array, on 5 / 8 cosets
as segments of one launch, and through the one-call quotient -- while the oracle comparisons of tests/test_circuits.py stop at
k = 4 .. 7.  Here the same programs run at the timed sizes and are held to:

  * oracle/graph_ref.py (the restatement of upstream's GraphEvaluator::evaluate) on SPOT ROWS: the interpreter reads the columns
    through a sparse window gathered around each row -- first / last rows, the rows where a rotation wraps, rows either side of
    2^19 and 2^20 (a 32-bit or grid-stride slip would show beyond 2^19 rows x 83 columns), random rows;
  * each other on EVERY row, compared on the device: ordinary vs internal-form columns, whole-array vs per-coset segments
    (x 1 / (X^n - 1)), multi-coset transform vs rows of the whole-array transform, one-call quotient on all E cosets vs
    coeff_to_extended -> program -> extended_to_coeff, the 5-coset quotient evaluated back on its cosets;
  * a constructed divisible numerator with a known quotient through hm_quotient_combine at k = 18 (Horner on Python integers
    ties the coset values to the polynomial).

Reference for the gates: /root/reference/src/chips/merkle_sum_tree.rs:44-138, merkle_v3.rs:30-82; BASELINE.json configs[2-3]."""
import random

import numpy as np
import pytest

import halo2_experiments_amd as h
from halo2_experiments_amd import circuits, evaluation as ev
from halo2_experiments_amd.domain import EvaluationDomain, FR_MODULUS, fr_words

R = FR_MODULUS


def test_row_sparse_interpreter_equals_the_whole_domain_one():
    """oracle/graph_ref.evaluate_graph_rows (sparse windows) is evaluate_graph restricted to the listed rows -- on the MerkleSumTree
    program itself, with its short column and every rotation (CPU only)."""
    from oracle import graph_ref
    cs = circuits.merkle_sum_tree()
    k, ek = 3, 6
    isize, rot_scale = 1 << ek, 1 << (ek - k)
    g, lay = circuits.evaluate_h_program(cs, k, ek, pow(7, 1 << 28, R))
    rng = random.Random(3)
    col = lambda rows=isize: [rng.randrange(R) for _ in range(rows)]
    fixed = [col() for _ in range(lay.num_fixed_entries)]
    fixed[lay.t_inv] = col(rot_scale) * (isize // rot_scale)
    advice, instance = [col() for _ in range(cs.num_advice)], [col() for _ in range(cs.num_instance)]
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    prev = col()
    full = graph_ref.evaluate_graph(g.calculations, g.constants, g.rotations, fixed, advice, instance, [], beta, gamma, theta, y, prev, rot_scale, isize)
    rows = [0, 1, rot_scale - 1, isize // 2, isize - 1]
    need = graph_ref.cells_needed(g.rotations, rows, rot_scale, isize)
    table = {"Fixed": fixed, "Advice": advice, "Instance": instance}
    window = {(kind, c, r): table[kind][c][r] for kind in table for c in range(len(table[kind])) for r in need}
    got = graph_ref.evaluate_graph_rows(g.calculations, g.constants, g.rotations, lambda kind, c, r: window[(kind, c, r)], [], beta, gamma, theta, y,
                                        {i: prev[i] for i in rows}, rows, rot_scale, isize)
    assert got == [full[i] for i in rows]
    assert set(need) >= {(i + r * rot_scale) % isize for i in rows for r in g.rotations}


def _spot_rows(n, e, rng):
    isize = n * e
    rows = {0, 1, e - 1, e, isize // 2 - 1, isize // 2, isize // 2 + 1, isize - 1, isize - e, isize - e - 1}
    rows |= {b + d for b in (1 << 19, 1 << 20) if b < isize for d in (-1, 0, 1)}         # either side of 2^19 / 2^20 rows
    rows |= {rng.randrange(isize) for _ in range(6)}
    return sorted(r for r in rows if 0 <= r < isize)


def _gather_ints(pyref, tensor, rows):
    """tensor: (columns, isize, 4) ordinary Montgomery words on the device -> {(column, row): canonical integer} for the listed rows."""
    import torch
    idx = torch.tensor(rows, dtype=torch.int64, device=tensor.device)
    sub = tensor[:, idx].contiguous().cpu().numpy().view(np.uint64)                   # (columns, len(rows), 4)
    out = {}
    for c in range(sub.shape[0]):
        for r, v in zip(rows, pyref.fr_from_array(sub[c])):
            out[(c, r)] = v
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("name,k", [("merkle_v3_k17", 17), ("merkle_sum_tree_k18", 18)])
def test_evaluate_h_of_the_timed_circuits_at_the_timed_sizes(pyref, name, k):
    import torch
    from oracle import graph_ref
    cs = circuits.CONSTRAINT_SYSTEMS[name]()
    dom = EvaluationDomain(cs.degree(), k)
    ek, n, e, q = dom.extended_k, dom.n, dom.num_cosets(), dom.min_cosets()
    assert (ek, e, q) == (k + 3, 8, 5)                                               # BASELINE configs[2] / [3]: 2^20 / 2^21 rows
    isize = 1 << ek
    delta = pow(7, 1 << 28, R)
    g_all, lay = circuits.evaluate_h_program(cs, k, ek, delta)
    g_num, lay_n = circuits.evaluate_h_program(cs, k, ek, delta, per_coset=True, divide=False)
    nf, na, ni = lay.num_fixed_entries, cs.num_advice, cs.num_instance
    n_cols = nf + na + ni
    coeffs = h.random_fr(n_cols * n, 9000 + k, "cuda", shape=(n_cols, n, 4))          # uniform over the whole of [0, r)
    coeffs[lay.x_coset] = 0
    coeffs[lay.x_coset, 1] = torch.from_numpy(fr_words(1).view(np.int64)).cuda()     # the identity polynomial X
    rng = random.Random(1000 + k)
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    to_dev = lambda vals: torch.from_numpy(pyref.fr_array(vals).view(np.int64)).cuda()
    tinv = [dom.coset_vanishing_inverse(c) for c in range(e)]
    prog_all = g_all.compile(nf, na, ni, num_challenges=0, rot_scale=e, short_columns=lay.short_columns)
    prog_num = g_num.compile(lay_n.num_fixed_entries, na, ni, num_challenges=0, rot_scale=1)
    assert n_cols == (83 if k == 18 else n_cols) and len(prog_all.calcs) == (700 if k == 18 else 236)     # the programs bench.py times
    try:
        # ---- whole-array launch: ordinary and internal-form columns, every row the same; spot rows against the oracle -------------
        ext = dom.coeff_to_extended(coeffs)                                            # (n_cols, 2^ek, 4) ordinary words
        prev = h.random_fr(isize, 9100 + k, "cuda")                                    # PreviousValue: random, not zero
        cols = [ext[i] for i in range(n_cols)]
        cols[lay.t_inv] = to_dev(tinv)
        h_plain = prev.clone()
        prog_all.evaluate(cols, h_plain, beta=beta, gamma=gamma, theta=theta, y=y)
        rows = _spot_rows(n, e, rng)
        need = graph_ref.cells_needed(g_all.rotations, rows, e, isize)
        window = _gather_ints(pyref, ext, need)
        prev_at = dict(zip(rows, pyref.fr_from_array(prev[torch.tensor(rows, device="cuda")].cpu().numpy().view(np.uint64))))

        def cell(kind, c, r):
            i = c if kind == "Fixed" else (nf + c if kind == "Advice" else nf + na + c)
            return tinv[r % e] if i == lay.t_inv else window[(i, r)]

        want = graph_ref.evaluate_graph_rows(g_all.calculations, g_all.constants, g_all.rotations, cell, [], beta, gamma, theta, y, prev_at, rows, e, isize)
        got = pyref.fr_from_array(h_plain[torch.tensor(rows, device="cuda")].cpu().numpy().view(np.uint64))
        assert got == want, [r for r, a, b in zip(rows, got, want) if a != b]
        # the undivided per-coset program on the oracle at the same rows: row E t + c of the extended array is row t of coset c
        num_want = {}
        for r in rows:
            c, t = r % e, r // e
            cell_c = lambda kind, col, tt, c=c: cell(kind, col, (e * tt + c) % isize)
            (num_want[r],) = graph_ref.evaluate_graph_rows(g_num.calculations, g_num.constants, g_num.rotations, cell_c, [], beta, gamma, theta, y,
                                                           {t: prev_at[r]}, [t], 1, n)
            assert num_want[r] * tinv[c] % R == want[rows.index(r)], r                # the two programs agree on the oracle itself
        del cols, window
        ext32 = dom.coeff_to_extended(coeffs, internal=True)                          # the form the replay feeds: 32 x the values
        cols32 = [ext32[i] for i in range(n_cols)]
        cols32[lay.t_inv] = to_dev([32 * v % R for v in tinv])
        h_int = prev.clone()
        prog_all.evaluate(cols32, h_int, beta=beta, gamma=gamma, theta=theta, y=y, columns_internal=True)
        assert bool((h_int == h_plain).all())
        del ext, cols32
        # ---- the q = 5 cosets that determine h, and all E, as segments of ONE launch ------------------------------------------------
        for cosets in (list(range(q)), list(range(e))):
            m = len(cosets)
            cc = dom.coeff_to_cosets(coeffs, cosets, internal=True)                    # (n_cols, m, n, 4): one transform call per batch
            for i, c in enumerate(cosets):                                             # = the rows E t + c of the whole-array transform
                assert bool((cc[:, i] == ext32[:, c::e]).all()), (name, c)
            v = torch.stack([prev[c::e] for c in cosets]).contiguous()                 # PreviousValue of coset c = rows E t + c
            prog_num.evaluate([cc[j].reshape(m * n, 4) for j in range(n_cols)], v.reshape(m * n, 4), beta=beta, gamma=gamma, theta=theta, y=y,
                              columns_internal=True, segments=m)
            for r in rows:                                                             # the segment launch against the oracle, row by row
                if r % e in cosets:
                    word = v[cosets.index(r % e), r // e].cpu().numpy().view(np.uint64)
                    assert np.array_equal(word, fr_words(num_want[r])), (name, m, r)
            for i, c in enumerate(cosets):                                             # ... and against the whole-array launch, every row
                scaled = h.linear_combination([v[i]], np.stack([fr_words(tinv[c])]))
                assert bool((scaled == h_plain[c::e]).all()), (name, m, c)
            del cc, v
        # ---- the one-call quotient (hm_quotient_by_cosets): all E cosets == upstream's steps, word for word ----------------------------
        cols32 = [ext32[i] for i in range(n_cols)]
        cols32[lay.t_inv] = to_dev([32 * v % R for v in tinv])
        h0 = torch.zeros((isize, 4), dtype=torch.int64, device="cuda")
        prog_all.evaluate(cols32, h0, beta=beta, gamma=gamma, theta=theta, y=y, columns_internal=True)
        on_ext = h0.clone()
        dom.extended_to_coeff(h0)                                                      # in place: all E n coefficients
        del cols32, ext32
        columns = [coeffs[i] for i in range(n_cols)]
        got_all = prog_num.quotient_by_cosets(dom, columns, cosets=list(range(e)), beta=beta, gamma=gamma, theta=theta, y=y)
        assert got_all.shape == (e * n, 4) and bool((got_all == h0).all()), name
        del got_all
        # the five cosets alone (the default route with N > 1 GPUs): random columns give a numerator that X^n - 1 does not divide, so
        # the result is not h0 -- it is THE polynomial of degree < 5 n that takes numerator / (X^n - 1) on those cosets: evaluate it back
        five = list(range(q))
        got5 = prog_num.quotient_by_cosets(dom, columns, cosets=five, beta=beta, gamma=gamma, theta=theta, y=y).reshape(q, n, 4)
        for c in five:
            u = pow(dom.coset_shift(c), n, R)
            back = h.linear_combination([dom.coeff_to_coset(got5[t], c) for t in range(q)], np.stack([fr_words(pow(u, t, R)) for t in range(q)]))
            assert bool((back == on_ext[c::e]).all()), (name, c)
        # ... and in two steps, as two devices would run it
        pa = prog_num.quotient_partials(dom, columns, five[:2], beta=beta, gamma=gamma, theta=theta, y=y)
        pb = prog_num.quotient_partials(dom, columns, five[2:], beta=beta, gamma=gamma, theta=theta, y=y)
        both = ev.quotient_combine(dom, [pa[i] for i in range(2)] + [pb[i] for i in range(q - 2)], five)
        assert bool((both.reshape(q, n, 4) == got5).all()), name
    finally:
        prog_all.destroy()
        prog_num.destroy()
        torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("k", [17, 18])
def test_quotient_combine_returns_a_constructed_quotient(pyref, k):
    """A numerator that X^n - 1 DOES divide, with a known quotient: pick h with 5 n random coefficients, hand hm_quotient_combine the
    partials of N = h (X^n - 1) on five cosets (N on coset c = (u_c - 1) x h on coset c, u_c = shift_c^n) -- h comes back word for
    word from several subsets of the eight cosets.  Two values of h on a coset are tied to the coefficients by Horner on Python
    integers (1.3 M terms at k = 18)."""
    import torch
    dom = EvaluationDomain(6, k)
    n, e, q = dom.n, dom.num_cosets(), dom.min_cosets()
    assert (e, q) == (8, 5)
    pieces = h.random_fr(q * n, 9300 + k, "cuda", shape=(q, n, 4))

    def values_on(c):
        u = pow(dom.coset_shift(c), n, R)
        return h.linear_combination([dom.coeff_to_coset(pieces[t], c) for t in range(q)], np.stack([fr_words(pow(u, t, R)) for t in range(q)]))

    coeffs_int = pyref.fr_from_array(pieces.reshape(q * n, 4).cpu().numpy().view(np.uint64))
    vals3 = values_on(3)
    for t in (0, n - 1):
        x = dom.coset_shift(3) * pow(dom.omega, t, R) % R
        acc = 0
        for cf in reversed(coeffs_int):
            acc = (acc * x + cf) % R
        assert np.array_equal(vals3[t].cpu().numpy().view(np.uint64), fr_words(acc)), t
    for cosets in ([0, 1, 2, 3, 4], [7, 5, 3, 2, 0], [3, 4, 5, 6, 7]):
        parts = []
        for c in cosets:
            u = pow(dom.coset_shift(c), n, R)
            numerator = h.linear_combination([values_on(c)], np.stack([fr_words((u - 1) % R)]))
            parts.append(dom.coset_to_partial(numerator, c))
        got = ev.quotient_combine(dom, parts, cosets)
        assert got.shape == (q * n, 4) and bool((got == pieces.reshape(q * n, 4)).all()), (k, cosets)
    # one coset more than needed: the extra piece is zero
    cosets = [0, 1, 2, 3, 4, 6]
    parts = []
    for c in cosets:
        u = pow(dom.coset_shift(c), n, R)
        parts.append(dom.coset_to_partial(h.linear_combination([values_on(c)], np.stack([fr_words((u - 1) % R)])), c))
    got = ev.quotient_combine(dom, parts, cosets)
    assert bool((got[: q * n] == pieces.reshape(q * n, 4)).all()) and not got[q * n:].any()
