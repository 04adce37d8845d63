"""GPU suite: a whole (toy) PLONKish proof's polynomial side on the device, then the verifier's check -- the analogue of the
reference's only prover test, which proves and then verifies (/root/reference/src/circuits/merkle_sum_tree.rs:345-358;
full_prover, /root/reference/src/circuits/utils.rs:40-63).

Circuit: rows of (a, b, c) with the gate q * (a * b - c), a range-check lookup a in [0, 64), copy constraints between cells
of b.  The device makes everything create_proof makes between the witness and the opening: the lookup's permuted columns,
the grand products of the lookup and of the permutation argument (two chunks), all polynomials in coefficient form and on
the extended coset, h(X) = (gates, permutation, lookup combined by y) / (X^n - 1) through the GraphEvaluator and
extended_to_coeff.  The check is the one verify_proof makes at a random point x:
        sum_i y^(..) * expression_i(x) == h(x) * (x^n - 1)
with every polynomial's value at x * omega^rot computed by the device Horner kernel, plus h's degree bound (the numerator
is divisible by the vanishing polynomial only if every constraint holds on every row)."""
import random

import numpy as np
import pytest

import halo2_experiments_amd as h
from halo2_experiments_amd import evaluation as ev
from halo2_experiments_amd.domain import EvaluationDomain, FR_MODULUS, FR_ZETA, fr_words
from oracle import graph_ref

pytestmark = pytest.mark.gpu
R = FR_MODULUS


def to_gpu(pyref, values):
    import torch
    return torch.from_numpy(pyref.fr_array(values).view(np.int64)).cuda()


def from_gpu(pyref, t):
    return pyref.fr_from_array(t.cpu().numpy().view(np.uint64))


def run(exprs, fixed, advice, n, **scalars):
    import torch
    g = ev.GraphEvaluator()
    g.add_custom_gates(exprs)
    prog = g.compile(len(fixed), len(advice), 0)
    out = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    prog.evaluate(list(fixed) + list(advice), out, **scalars)
    prog.destroy()
    return out


@pytest.mark.parametrize("tamper", [False, True])
def test_mini_proof_quotient_identity(pyref, tamper):
    import torch
    k, blinding, max_degree, chunk = 8, 5, 4, 2
    n, usable = 1 << k, (1 << k) - blinding - 1
    dom = EvaluationDomain(max_degree, k)
    ek, rot_scale = dom.extended_k, 1 << (dom.extended_k - k)
    omega, delta = dom.omega, pow(7, 1 << 28, R)
    rng = random.Random(2024)
    beta, gamma, theta, y, x = (rng.randrange(2, R) for _ in range(5))
    rnd = lambda cnt: [rng.randrange(R) for _ in range(cnt)]

    # ---- witness and fixed columns (Lagrange basis) ---------------------------------------------------------------
    a = [rng.randrange(64) for _ in range(usable)] + rnd(n - usable)
    b = []
    for _ in range(usable // 2):
        v = rng.randrange(R)
        b += [v, v]                                                  # copy constraint b[2i] == b[2i+1]
    b += rnd(n - usable)
    c = [a[i] * b[i] % R for i in range(usable)] + rnd(n - usable)
    if tamper:
        c[17] = (c[17] + 1) % R                                      # one gate row is violated
    q = [1] * usable + [0] * (n - usable)
    t = [i % 64 for i in range(n)]
    ident = lambda j, i: pow(delta, j, R) * pow(omega, i, R) % R
    sig = [[ident(j, i) for i in range(n)] for j in range(3)]
    for i in range(0, usable, 2):
        sig[1][i], sig[1][i + 1] = ident(1, i + 1), ident(1, i)
    l0 = [1] + [0] * (n - 1)
    l_last = [1 if i == usable else 0 for i in range(n)]
    l_active = [1 if i < usable else 0 for i in range(n)]
    d = lambda col: to_gpu(pyref, col)
    d_a, d_b, d_c, d_q, d_t = d(a), d(b), d(c), d(q), d(t)
    d_sig = [d(s) for s in sig]
    x_col = d([pow(omega, i, R) for i in range(n)])

    # ---- lookup argument: permuted columns and z_L on the device ---------------------------------------------------
    a_perm, s_perm = h.permute_expression_pair(d_a, d_t, usable, blinding_seed=11)
    num = run([(ev.Advice(0) + ev.BETA) * (ev.Advice(1) + ev.GAMMA)], [], [d_a, d_t], n, beta=beta, gamma=gamma)
    den = run([(ev.Advice(0) + ev.BETA) * (ev.Advice(1) + ev.GAMMA)], [], [a_perm, s_perm], n, beta=beta, gamma=gamma)
    h.batch_invert(den)
    z_l = h.grand_product(run([ev.Advice(0) * ev.Advice(1)], [], [num, den], n), fr_words(1))
    assert from_gpu(pyref, z_l[usable:usable + 1]) == [1]
    z_l[usable + 1:] = d(rnd(n - usable - 1))

    # ---- permutation argument: z_0 over (a, b), z_1 over (c) -------------------------------------------------------
    cols_d = [d_a, d_b, d_c]
    zs, start = [], 1
    for s0 in range(0, 3, chunk):
        cc, ss = cols_d[s0:s0 + chunk], d_sig[s0:s0 + chunk]
        w = len(cc)
        den_e = num_e = None
        for j in range(w):
            de = ev.Advice(j) + ev.BETA * ev.Advice(w + j) + ev.GAMMA
            ne = ev.Advice(j) + ev.BETA * ev.Advice(2 * w) * pow(delta, s0 + j, R) + ev.GAMMA
            den_e = de if den_e is None else den_e * de
            num_e = ne if num_e is None else num_e * ne
        den = run([den_e], [], cc + ss + [x_col], n, beta=beta, gamma=gamma)
        num = run([num_e], [], cc + ss + [x_col], n, beta=beta, gamma=gamma)
        h.batch_invert(den)
        z = h.grand_product(run([ev.Advice(0) * ev.Advice(1)], [], [num, den], n), fr_words(start))
        start = from_gpu(pyref, z[usable:usable + 1])[0]
        z[usable + 1:] = d(rnd(n - usable - 1))
        zs.append(z)
    assert start == 1

    # ---- every polynomial: coefficients, then the extended coset ----------------------------------------------------
    fixed_l = [d_q, d_t] + d_sig + [d(l0), d(l_last), d(l_active)]                     # fixed 0 .. 7
    advice_l = [d_a, d_b, d_c, a_perm, s_perm, z_l] + zs                               # advice 0 .. 7
    x_poly = d([0, 1] + [0] * (n - 2))                                                 # the polynomial X, already in coefficient form
    coeffs = dom.lagrange_to_coeff(torch.stack(fixed_l + advice_l))                    # (16, n, 4), one batched call
    ext = dom.coeff_to_extended(torch.cat([coeffs, x_poly.reshape(1, n, 4)]))          # (17, 4n, 4)
    t_inv = d([pow((pow(FR_ZETA * pow(dom.extended_omega, i, R) % R, n, R) - 1) % R, -1, R) for i in range(rot_scale)])
    F, A = ev.Fixed, ev.Advice
    exprs = [F(0) * (A(0) * A(1) - A(2))]
    exprs += ev.permutation_expressions([A(0), A(1), A(2)], [F(2), F(3), F(4)],
                                        [lambda rot, i=i: A(6 + i, rot) for i in range(2)], F(5), F(6), F(7), F(8), chunk, delta,
                                        -(blinding + 1))
    exprs += ev.lookup_expressions([A(0)], [F(1)], lambda rot: A(5, rot), lambda rot: A(3, rot), lambda rot: A(4, rot), F(5), F(6), F(7))
    g = ev.GraphEvaluator()
    g.add_custom_gates(exprs)
    g.add_vanishing_division(F(9))
    prog = g.compile(10, 8, 0, rot_scale=rot_scale, short_columns={9: ek - k})
    h_ext = torch.zeros((dom.extended_len(), 4), dtype=torch.int64, device="cuda")
    ext_cols = [ext[i] for i in range(8)] + [ext[16], t_inv] + [ext[8 + i] for i in range(8)]
    prog.evaluate(ext_cols, h_ext, beta=beta, gamma=gamma, theta=theta, y=y)
    prog.destroy()
    h_coeff = dom.extended_to_coeff(h_ext)                                             # (3n, 4): h(X); in place on h_ext
    torch.cuda.synchronize()
    degree_ok = not h_ext[3 * n:].any()                                                # deg h < 3n iff Z_H divides the numerator
    # ---- the same quotient from j - 1 = 3 of the 4 cosets of the extended domain only (EvaluationDomain.combine_cosets) ----
    # every column onto the coset from its coefficients, the per-coset program (rotations unscaled, 1 / (X^n - 1) a constant),
    # the inverse transform: for a SATISFIED circuit word for word the h above; for the tampered one a different polynomial
    g1 = ev.GraphEvaluator()
    g1.add_custom_gates(exprs)
    g1.add_vanishing_division(ev.Challenge(0))
    prog1 = g1.compile(10, 8, 0, num_challenges=1, rot_scale=1)
    all_coeffs = torch.cat([coeffs, x_poly.reshape(1, n, 4)])
    assert dom.min_cosets() == 3 and dom.num_cosets() == 4
    parts, use = [], [3, 0, 2]
    for c in use:
        cc = dom.coeff_to_coset(all_coeffs, c)
        cols_c = [cc[i] for i in range(8)] + [cc[16], cc[16]] + [cc[8 + i] for i in range(8)]      # (fixed 9, the pattern column, is unread here)
        v = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
        prog1.evaluate(cols_c, v, challenges=[dom.coset_vanishing_inverse(c)], beta=beta, gamma=gamma, theta=theta, y=y)
        parts.append(dom.coset_to_partial(v, c))
    prog1.destroy()
    h_min = dom.combine_cosets(parts, cosets=use)
    # ... and in ONE call from the coefficient arrays (hm_quotient_by_cosets_bn256_fr_dev: the undivided numerator, the division on
    # the recombination matrix)
    g2 = ev.GraphEvaluator()
    g2.add_custom_gates(exprs)
    prog2 = g2.compile(10, 8, 0, rot_scale=1)
    coeff_cols = [all_coeffs[i] for i in range(8)] + [all_coeffs[16], all_coeffs[16]] + [all_coeffs[8 + i] for i in range(8)]
    h_one = prog2.quotient_by_cosets(dom, coeff_cols, cosets=use, beta=beta, gamma=gamma, theta=theta, y=y)
    prog2.destroy()
    assert bool((h_one == h_min).all())
    if not tamper:
        assert h_min.shape == h_coeff.shape and bool((h_min == h_coeff).all())
    else:
        assert not bool((h_min == h_coeff).all())

    # ---- the verifier's check at x ----------------------------------------------------------------------------------
    rots = [0, 1, -1, -(blinding + 1)]
    M = 16
    pts = np.stack([fr_words(x * pow(omega, r, R) % R) for r in rots for _ in range(16)])
    idx = np.array([p for _ in rots for p in range(16)], dtype=np.uint32)
    vals = pyref.fr_from_array(h.eval_polynomial(coeffs, pts, poly_index=idx))
    at = [[None] * M for _ in range(16)]
    for ri, r in enumerate(rots):
        for p in range(16):
            at[p][r % M] = vals[ri * 16 + p]
    fixed_at = at[:8] + [[x] + [None] * (M - 1), None]                                 # fixed 8 = the polynomial X at x
    advice_at = at[8:]
    acc = 0
    for e in exprs:
        acc = (acc * y + graph_ref.evaluate_expression(e, fixed_at, advice_at, [], {"beta": beta, "gamma": gamma, "theta": theta}, 0, 1, M)) % R
    hx = pyref.fr_from_array(h.eval_polynomial(h_coeff.reshape(1, 3 * n, 4).contiguous(), np.stack([fr_words(x)])))[0]
    identity_ok = acc == hx * (pow(x, n, R) - 1) % R
    if tamper:
        assert not (degree_ok and identity_ok)                                         # a violated gate is caught
    else:
        assert degree_ok and identity_ok
