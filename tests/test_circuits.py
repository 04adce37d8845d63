"""The constraint systems of the reference's circuits as evaluate_h programs (halo2-experiments_amd/circuits.py).

CPU: the transcription against what the reference's chip files say (column counts, equality columns, gate counts, lookups;
/root/reference/src/chips/merkle_sum_tree.rs:32-138, merkle_v3.rs:30-82, poseidon/hash.rs:45-72) and the builder against
direct evaluation of the expression trees.  GPU: the device interpreter on each whole program -- gates, permutation
argument, lookup arguments, vanishing-polynomial division -- against the oracle's restatement of upstream's loops."""
import random

import numpy as np
import pytest

import halo2_experiments_amd as h
from halo2_experiments_amd import circuits, evaluation as ev
from halo2_experiments_amd.domain import FR_MODULUS, fr_words

R = FR_MODULUS


def test_merkle_sum_tree_constraint_system_matches_the_chip():
    cs = circuits.merkle_sum_tree()
    # merkle_sum_tree.rs: 5 chip columns (:27-31 of the circuit), 5 hash inputs (:103), PoseidonChip's partial_sbox
    # (poseidon/hash.rs:50), LtChip's lt + 8 diff bytes (LtConfig<F, 8>, :21)
    assert cs.num_advice == 5 + 5 + 1 + 1 + 8 == 20
    assert cs.num_instance == 1
    # 4 chip selectors (:44-47), rc_a + rc_b (hash.rs:51-52: 2 x WIDTH), Pow5Chip's 3 selectors, LtChip's u8 table
    assert cs.num_fixed == 4 + 10 + 3 + 1
    # equality: a..e and the instance column (:53-58), the 5 hash inputs (hash.rs:54-56), rc_b[0] (enable_constant, hash.rs:57)
    assert cs.equality == [("advice", i) for i in range(5)] + [("instance", 0)] + [("advice", 5 + i) for i in range(5)] + [("fixed", 9)]
    names = [n for n, _ in cs.gates]
    assert names == ["bool constraint", "swap constraint", "sum constraint", "full round", "partial rounds", "pad-and-add", "lt gate",
                     "check == is_lt"]
    counts = {n: len(p) for n, p in cs.gates}
    assert counts == {"bool constraint": 1, "swap constraint": 2, "sum constraint": 1, "full round": 5, "partial rounds": 6,
                      "pad-and-add": 5, "lt gate": 2, "check == is_lt": 1}
    assert len(cs.lookups) == 8 and all(len(i) == 1 and len(t) == 1 for i, t in cs.lookups)
    assert cs.degree() == 6                       # selector x (state + rc)^5
    assert cs.permutation_chunk_len() == 4 and cs.permutation_sets() == 3


def test_the_transcribed_gates_hold_on_a_satisfying_row():
    """The swap / sum / bool gates of merkle_sum_tree.rs:62-101 evaluated directly on a row that satisfies them, and on a
    tampered one (the reference's negative tests change one witness value and expect failure: merkle_sum_tree.rs:214-343)."""
    from oracle import graph_ref
    cs = circuits.merkle_sum_tree()
    gates = dict(cs.gates)
    rng = random.Random(5)
    a, b, c, d = (rng.randrange(R) for _ in range(4))
    for e in (0, 1):
        l1, l2, r1, r2 = (c, d, a, b) if e else (a, b, c, d)
        adv = [{0: a, 1: l1}, {0: b, 1: l2}, {0: c, 1: r1}, {0: d, 1: r2}, {0: e}] + [{0: 0, 1: 0, 7: 0} for _ in range(15)]
        fixed = [{0: 1, 1: 1} for _ in range(cs.num_fixed)]
        for p in gates["swap constraint"] + gates["bool constraint"]:
            assert graph_ref.evaluate_expression(p, fixed, adv, [{0: 0}], [], 0, 1, 8) == 0
        adv[0][1] = (l1 + 1) % R                  # left output tampered
        assert graph_ref.evaluate_expression(gates["swap constraint"][0], fixed, adv, [{0: 0}], [], 0, 1, 8) != 0
    adv = [{0: 0}, {0: 30}, {0: 0}, {0: 12}, {0: 42}] + [{0: 0} for _ in range(15)]
    fixed = [{0: 1} for _ in range(cs.num_fixed)]
    assert graph_ref.evaluate_expression(gates["sum constraint"][0], fixed, adv, [{0: 0}], [], 0, 1, 8) == 0
    adv[4][0] = 43
    assert graph_ref.evaluate_expression(gates["sum constraint"][0], fixed, adv, [{0: 0}], [], 0, 1, 8) != 0


def test_the_other_two_circuits():
    v3, po = circuits.merkle_v3(), circuits.poseidon()
    assert (v3.num_advice, v3.num_fixed, v3.num_instance) == (3 + 3 + 1, 2 + 6 + 3, 1)          # merkle_v3.rs:30-82
    assert [n for n, _ in v3.gates] == ["bool constraint", "swap constraint", "full round", "partial rounds", "pad-and-add"]
    assert len(v3.polynomials()) == 1 + 1 + 3 + 4 + 3 and not v3.lookups and v3.degree() == 6
    assert (po.num_advice, po.num_fixed, po.num_instance) == (5 + 1, 10 + 3, 1)                    # circuits/poseidon.rs:36-41
    assert len(po.polynomials()) == 5 + 6 + 5 and len(po.equality) == 5 + 1 + 1


@pytest.mark.parametrize("name", sorted(circuits.CONSTRAINT_SYSTEMS))
def test_builder_equals_direct_evaluation_of_every_gate(name):
    """GraphEvaluator::add_expression's rules and CSE on the circuit's own gate polynomials: the straight-line program's
    value equals Horner in y over the expression trees evaluated directly (both on the oracle's integers)."""
    from oracle import graph_ref
    cs = circuits.CONSTRAINT_SYSTEMS[name]()
    g, _ = circuits.evaluate_h_program(cs, 3, 6, 7, with_arguments=False)
    rng = random.Random(len(name))
    isize, rot_scale = 1 << 6, 1 << 3
    mk = lambda cnt: [[rng.randrange(R) for _ in range(isize)] for _ in range(cnt)]
    fixed, advice, instance = mk(cs.num_fixed), mk(cs.num_advice), mk(cs.num_instance)
    y = rng.randrange(R)
    prev = [rng.randrange(R) for _ in range(isize)]
    got = graph_ref.evaluate_graph(g.calculations, g.constants, g.rotations, fixed, advice, instance, [], 0, 0, 0, y, prev, rot_scale, isize)
    for idx in (0, 5, isize - 1):
        v = prev[idx]
        for p in cs.polynomials():
            v = (v * y + graph_ref.evaluate_expression(p, fixed, advice, instance, [], idx, rot_scale, isize)) % R
        assert got[idx] == v, (name, idx)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(circuits.CONSTRAINT_SYSTEMS))
def test_device_evaluate_h_of_each_circuit_matches_the_oracle(pyref, name):
    """The WHOLE evaluate_h program of each configuration (what the k = 11 / 17 / 18 replays time at 2^(k+3) rows) at a small
    k: device interpreter vs the oracle's interpreter of the same straight-line program, and -- for the argument terms --
    vs the oracle's restatement of upstream's hand-written row loops (evaluate_h_permutation_and_lookups)."""
    import torch
    from oracle import graph_ref
    cs = circuits.CONSTRAINT_SYSTEMS[name]()
    k, ek = 4, 7
    n, isize, rot_scale = 1 << k, 1 << ek, 1 << (ek - k)
    delta = pow(7, 1 << 28, R)
    g, lay = circuits.evaluate_h_program(cs, k, ek, delta)
    rng = random.Random(7 + len(name))
    col = lambda rows=isize: [rng.randrange(R) for _ in range(rows)]
    w_ext, zeta = pyref.fr_omega(ek), pyref.FR_ZETA
    fixed = [col() for _ in range(lay.num_fixed_entries)]
    fixed[lay.x_coset] = [zeta * pow(w_ext, i, R) % R for i in range(isize)]
    t_inv = [pow((pow(zeta * pow(w_ext, i, R) % R, n, R) - 1) % R, -1, R) for i in range(rot_scale)]
    fixed[lay.t_inv] = t_inv
    advice, instance = [col() for _ in range(cs.num_advice)], [col() for _ in range(cs.num_instance)]
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    prev = col()
    # 1. the straight-line program on the oracle's interpreter (short column: its pattern repeated)
    full_fixed = [c * (isize // len(c)) for c in fixed]
    exp = graph_ref.evaluate_graph(g.calculations, g.constants, g.rotations, full_fixed, advice, instance, [], beta, gamma, theta, y, prev,
                                   rot_scale, isize)
    # 2. the argument terms by upstream's row loops, started from the gates' running value
    g_gates, _ = circuits.evaluate_h_program(cs, k, ek, delta, with_arguments=False)
    after_gates = graph_ref.evaluate_graph(g_gates.calculations, g_gates.constants, g_gates.rotations, full_fixed[:cs.num_fixed], advice,
                                           instance, [], beta, gamma, theta, y, prev, rot_scale, isize)
    table = {"advice": advice, "fixed": full_fixed, "instance": instance}
    perm_cols = [table[kd][i] for kd, i in cs.equality]
    P, nsets = len(cs.equality), cs.permutation_sets()
    lookups = []
    for j, (ins, tabs) in enumerate(cs.lookups):
        b = lay.lookup0 + 3 * j
        ev_col = lambda e: [graph_ref.evaluate_expression(e, full_fixed, advice, instance, [], i, rot_scale, isize) for i in range(isize)]
        lookups.append(([ev_col(e) for e in ins], [ev_col(e) for e in tabs], full_fixed[b], full_fixed[b + 1], full_fixed[b + 2]))
    exp2 = graph_ref.evaluate_h_permutation_and_lookups(after_gates, y, beta, gamma, theta, isize, rot_scale, w_ext, zeta, delta, perm_cols,
                                                        full_fixed[lay.sigma0:lay.sigma0 + P], full_fixed[lay.z0:lay.z0 + nsets],
                                                        cs.permutation_chunk_len(), -(cs.blinding_factors + 1), full_fixed[lay.l0],
                                                        full_fixed[lay.l_last], full_fixed[lay.l_active], lookups, t_inverse=t_inv)
    assert exp == exp2, name
    # 3. the device
    to_dev = lambda c: torch.from_numpy(pyref.fr_array(c).view(np.int64)).cuda()
    cols = [to_dev(c) for c in fixed + advice + instance]
    values = to_dev(prev)
    prog = g.compile(lay.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=rot_scale, short_columns=lay.short_columns)
    try:
        prog.evaluate(cols, values, beta=beta, gamma=gamma, theta=theta, y=y)
        torch.cuda.synchronize()
        assert np.array_equal(values.cpu().numpy().view(np.uint64), pyref.fr_array(exp)), name
        # 4. the same program on INTERNAL-form columns (every column word = 32 x the value: HM_GRAPH_COLUMNS_INTERNAL) -- the
        # lowering drops the Stores of columns and every conversion product; PreviousValue and the result stay ordinary words
        cols32 = [to_dev([32 * v % R for v in c]) for c in fixed + advice + instance]
        values = to_dev(prev)
        prog.evaluate(cols32, values, beta=beta, gamma=gamma, theta=theta, y=y, columns_internal=True)
        torch.cuda.synchronize()
        assert np.array_equal(values.cpu().numpy().view(np.uint64), pyref.fr_array(exp)), name + " (internal-form columns)"
    finally:
        prog.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(circuits.CONSTRAINT_SYSTEMS))
def test_evaluate_h_coset_by_coset_gives_the_same_quotient(pyref, name):
    """The multi-GPU route of DESIGN 6 on one device: every column as n coefficients; (A) coeff_to_extended, the whole
    program over 2^extended_k rows, extended_to_coeff; (B) for each of the E cosets coeff_to_coset, the per-coset program
    over 2^k rows (rotations unscaled, 1 / (X^n - 1) a constant), coset_to_partial -- then combine_cosets.  The per-coset
    values are the residue classes of rows of (A)'s array and the quotient coefficients agree word for word."""
    import torch
    from halo2_experiments_amd.domain import EvaluationDomain
    cs = circuits.CONSTRAINT_SYSTEMS[name]()
    k = 6
    dom = EvaluationDomain(cs.degree(), k)
    ek, n, e = dom.extended_k, dom.n, dom.num_cosets()
    assert e >= 4
    delta = pow(7, 1 << 28, R)
    g_all, lay = circuits.evaluate_h_program(cs, k, ek, delta)
    g_one, lay1 = circuits.evaluate_h_program(cs, k, ek, delta, per_coset=True)
    n_cols = lay.num_fixed_entries + cs.num_advice + cs.num_instance
    coeffs = h.random_fr(n_cols * n, 77 + len(name), "cuda", shape=(n_cols, n, 4))          # uniform over the whole of [0, r)
    x_poly = [0, 1] + [0] * (n - 2)                                         # the identity polynomial: its evaluations are the points
    coeffs[lay.x_coset] = torch.from_numpy(pyref.fr_array(x_poly).view(np.int64)).cuda()
    rng = random.Random(5)
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    to_dev = lambda vals: torch.from_numpy(pyref.fr_array(vals).view(np.int64)).cuda()
    prog_all = g_all.compile(lay.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=e, short_columns=lay.short_columns)
    prog_one = g_one.compile(lay1.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=1, rot_scale=1)
    try:
        # (A)
        ext = dom.coeff_to_extended(coeffs, internal=True)
        cols = [ext[i] for i in range(n_cols)]
        cols[lay.t_inv] = to_dev([32 * dom.coset_vanishing_inverse(c) % R for c in range(e)])
        h_ext = torch.zeros((1 << ek, 4), dtype=torch.int64, device="cuda")
        prog_all.evaluate(cols, h_ext, beta=beta, gamma=gamma, theta=theta, y=y, columns_internal=True)
        h_a = dom.extended_to_coeff(h_ext.clone()).clone()
        # (B)
        parts = []
        for c in range(e):
            cc = dom.coeff_to_coset(coeffs, c, internal=True)
            v = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
            prog_one.evaluate([cc[i] for i in range(n_cols)], v, challenges=[dom.coset_vanishing_inverse(c)], beta=beta, gamma=gamma, theta=theta,
                              y=y, columns_internal=True)
            assert bool((v == h_ext[c::e]).all()), (name, c)
            parts.append(dom.coset_to_partial(v, c))
        h_b = dom.combine_cosets(parts)
        assert h_b.shape == h_a.shape and bool((h_a == h_b).all()), name
    finally:
        prog_all.destroy()
        prog_one.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(circuits.CONSTRAINT_SYSTEMS))
def test_evaluate_h_on_several_cosets_in_one_launch(pyref, name):
    """The fused form of the coset route: every column onto q cosets in ONE transform call (hm_coeff_to_cosets), the UNDIVIDED
    numerator over q segments of n rows in ONE launch (rotations wrap inside a segment), the partials in one call, and
    1 / (X^n - 1) folded into the recombination matrix.  With all E cosets: word for word the quotient of the whole-array
    route (random columns: any numerator); the per-segment values equal the per-coset program's, coset by coset."""
    import torch
    from halo2_experiments_amd.domain import EvaluationDomain
    cs = circuits.CONSTRAINT_SYSTEMS[name]()
    k = 6
    dom = EvaluationDomain(cs.degree(), k)
    ek, n, e = dom.extended_k, dom.n, dom.num_cosets()
    delta = pow(7, 1 << 28, R)
    g_all, lay = circuits.evaluate_h_program(cs, k, ek, delta)
    g_num, lay_n = circuits.evaluate_h_program(cs, k, ek, delta, per_coset=True, divide=False)
    g_one, lay1 = circuits.evaluate_h_program(cs, k, ek, delta, per_coset=True)
    n_cols = lay.num_fixed_entries + cs.num_advice + cs.num_instance
    coeffs = h.random_fr(n_cols * n, 99 + len(name), "cuda", shape=(n_cols, n, 4))          # uniform over the whole of [0, r)
    coeffs[lay.x_coset] = torch.from_numpy(pyref.fr_array([0, 1] + [0] * (n - 2)).view(np.int64)).cuda()
    rng = random.Random(6)
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    to_dev = lambda vals: torch.from_numpy(pyref.fr_array(vals).view(np.int64)).cuda()
    prog_all = g_all.compile(lay.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=e, short_columns=lay.short_columns)
    prog_num = g_num.compile(lay_n.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=1)
    prog_one = g_one.compile(lay1.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=1, rot_scale=1)
    try:
        ext = dom.coeff_to_extended(coeffs, internal=True)
        cols = [ext[i] for i in range(n_cols)]
        cols[lay.t_inv] = to_dev([32 * dom.coset_vanishing_inverse(c) % R for c in range(e)])
        h_ext = torch.zeros((1 << ek, 4), dtype=torch.int64, device="cuda")
        prog_all.evaluate(cols, h_ext, beta=beta, gamma=gamma, theta=theta, y=y, columns_internal=True)
        h_a = dom.extended_to_coeff(h_ext.clone()).clone()
        for cosets in (list(range(e)), [e - 1, 0, 2][: min(3, e)]):
            q = len(cosets)
            cc = dom.coeff_to_cosets(coeffs, cosets, internal=True)                       # (n_cols, q, n, 4)
            for i, c in enumerate(cosets):
                assert bool((cc[:, i] == dom.coeff_to_coset(coeffs, c, internal=True)).all()), (name, c)
            v = torch.zeros((q, n, 4), dtype=torch.int64, device="cuda")
            prog_num.evaluate([cc[j].reshape(q * n, 4) for j in range(n_cols)], v.reshape(q * n, 4), beta=beta, gamma=gamma, theta=theta, y=y,
                              columns_internal=True, segments=q)
            for i, c in enumerate(cosets):                                                 # the divided per-coset program = numerator * 1 / (u_c - 1)
                one = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
                prog_one.evaluate([cc[j, i] for j in range(n_cols)], one, challenges=[dom.coset_vanishing_inverse(c)], beta=beta, gamma=gamma,
                                  theta=theta, y=y, columns_internal=True)
                scaled = h.linear_combination([v[i]], np.stack([fr_words(dom.coset_vanishing_inverse(c))]))
                assert bool((scaled == one).all()) and bool((one == h_ext[c::e]).all()), (name, c)
            parts = dom.cosets_to_partials(v, cosets)
            if q == e:
                h_b = dom.combine_cosets([parts[i] for i in range(q)], cosets=cosets, pieces=dom.quotient_poly_degree, divide_by_vanishing=True)
                assert bool((h_b == h_a).all()), name
    finally:
        prog_all.destroy()
        prog_num.destroy()
        prog_one.destroy()


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(circuits.CONSTRAINT_SYSTEMS))
def test_the_one_call_quotient_equals_the_whole_array_route(pyref, name):
    """hm_quotient_by_cosets_bn256_fr_dev: coefficient columns in, h out.  With all E cosets: word for word what coeff_to_extended
    -> the divided program over 2^extended_k rows -> extended_to_coeff returns, for random columns; with a subset: the same as the
    Python composition of the fused calls (coeff_to_cosets, segments, cosets_to_partials, combine_cosets with the division on its
    matrix)."""
    import torch
    from halo2_experiments_amd.domain import EvaluationDomain
    cs = circuits.CONSTRAINT_SYSTEMS[name]()
    k = 7
    dom = EvaluationDomain(cs.degree(), k)
    ek, n, e = dom.extended_k, dom.n, dom.num_cosets()
    delta = pow(7, 1 << 28, R)
    g_all, lay = circuits.evaluate_h_program(cs, k, ek, delta)
    g_num, lay_n = circuits.evaluate_h_program(cs, k, ek, delta, per_coset=True, divide=False)
    n_cols = lay.num_fixed_entries + cs.num_advice + cs.num_instance
    coeffs = h.random_fr(n_cols * n, 123 + len(name), "cuda", shape=(n_cols, n, 4))          # uniform over the whole of [0, r)
    coeffs[lay.x_coset] = torch.from_numpy(pyref.fr_array([0, 1] + [0] * (n - 2)).view(np.int64)).cuda()
    rng = random.Random(8)
    beta, gamma, theta, y = (rng.randrange(R) for _ in range(4))
    to_dev = lambda vals: torch.from_numpy(pyref.fr_array(vals).view(np.int64)).cuda()
    prog_all = g_all.compile(lay.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=e, short_columns=lay.short_columns)
    prog_num = g_num.compile(lay_n.num_fixed_entries, cs.num_advice, cs.num_instance, num_challenges=0, rot_scale=1)
    try:
        ext = dom.coeff_to_extended(coeffs, internal=True)
        cols = [ext[i] for i in range(n_cols)]
        cols[lay.t_inv] = to_dev([32 * dom.coset_vanishing_inverse(c) % R for c in range(e)])
        h_ext = torch.zeros((1 << ek, 4), dtype=torch.int64, device="cuda")
        prog_all.evaluate(cols, h_ext, beta=beta, gamma=gamma, theta=theta, y=y, columns_internal=True)
        dom.extended_to_coeff(h_ext)                                   # in place: all E n coefficients
        columns = [coeffs[i] for i in range(n_cols)]
        got = prog_num.quotient_by_cosets(dom, columns, cosets=list(range(e)), beta=beta, gamma=gamma, theta=theta, y=y)
        assert got.shape == (e * n, 4) and bool((got == h_ext).all()), name
        sub = [e - 1, 1, 0][: min(3, e)]
        got = prog_num.quotient_by_cosets(dom, columns, cosets=sub, beta=beta, gamma=gamma, theta=theta, y=y)
        q = len(sub)
        cc = dom.coeff_to_cosets(coeffs, sub, internal=True)
        v = torch.zeros((q, n, 4), dtype=torch.int64, device="cuda")
        prog_num.evaluate([cc[j].reshape(q * n, 4) for j in range(n_cols)], v.reshape(q * n, 4), beta=beta, gamma=gamma, theta=theta, y=y,
                          columns_internal=True, segments=q)
        parts = dom.cosets_to_partials(v, sub)
        want = dom.combine_cosets([parts[i] for i in range(q)], cosets=sub, divide_by_vanishing=True)
        assert bool((got == want).all()), name
        # the fixed entries of the table transformed ONCE (a proving key keeps them): same result, their coefficients not read
        nf = lay.num_fixed_entries
        kept = dom.coeff_to_cosets(coeffs[:nf], sub, internal=True)
        pre = [kept[i] for i in range(nf)] + [None] * (n_cols - nf)
        got2 = prog_num.quotient_by_cosets(dom, [None] * nf + columns[nf:], cosets=sub, beta=beta, gamma=gamma, theta=theta, y=y, on_cosets=pre)
        assert bool((got2 == want).all()), name
        # ... and in two steps, as two devices would: each its cosets (hm_quotient_partials), then all partials combined on one
        five = list(range(min(5, e)))
        pa = prog_num.quotient_partials(dom, columns, five[:2], beta=beta, gamma=gamma, theta=theta, y=y)
        pb = prog_num.quotient_partials(dom, columns, five[2:], beta=beta, gamma=gamma, theta=theta, y=y)
        both = ev.quotient_combine(dom, [pa[i] for i in range(pa.shape[0])] + [pb[i] for i in range(pb.shape[0])], five)
        assert bool((both == prog_num.quotient_by_cosets(dom, columns, cosets=five, beta=beta, gamma=gamma, theta=theta, y=y)).all()), name
        assert coeffs[lay.x_coset][1].any() and not coeffs[lay.x_coset][2:].any()          # the inputs are untouched
        with pytest.raises(Exception):
            prog_num.quotient_by_cosets(dom, columns[:-1], cosets=sub)
        with pytest.raises(Exception):
            prog_num.quotient_by_cosets(dom, columns, cosets=[0, 0])
    finally:
        prog_all.destroy()
        prog_num.destroy()
