"""CPU suite: oracle/poly_ref.py (kate_division, batch_invert, grand_product, linear_combination) against the
definitions they restate -- division identities, inverses, direct products."""
import random

import pytest

from oracle import poly_ref as pr

R = pr.R


def horner(a, x):
    acc = 0
    for c in reversed(a):
        acc = (acc * x + c) % R
    return acc


@pytest.mark.parametrize("n", [1, 2, 3, 17, 256, 1000])
def test_kate_division_is_the_quotient_by_x_minus_z(n):
    rng = random.Random(n)
    a = [rng.randrange(R) for _ in range(n)]
    for z in (0, 1, R - 1, rng.randrange(R)):
        q = pr.kate_division(a, z)
        assert len(q) == n - 1
        # a(X) = q(X) (X - z) + a(z), coefficient by coefficient
        rem = horner(a, z)
        back = [0] * n
        for i, c in enumerate(q):
            back[i + 1] = (back[i + 1] + c) % R
            back[i] = (back[i] - z * c) % R
        back[0] = (back[0] + rem) % R
        assert back == a
    with pytest.raises(ValueError):
        pr.kate_division([], 5)


def test_batch_invert_skips_zeros():
    rng = random.Random(7)
    v = [rng.randrange(R) for _ in range(100)]
    for i in (0, 13, 14, 99):
        v[i] = 0
    inv = pr.batch_invert(v)
    for a, b in zip(v, inv):
        assert (b == 0) if a == 0 else (a * b % R == 1)
    assert pr.batch_invert([]) == [] and pr.batch_invert([0, 0]) == [0, 0] and pr.batch_invert([1]) == [1]


def test_grand_product_and_linear_combination():
    rng = random.Random(9)
    m = [rng.randrange(R) for _ in range(50)]
    start = rng.randrange(R)
    z = pr.grand_product(m, start)
    assert len(z) == 50 and z[0] == start
    acc = start
    for i in range(50):
        assert z[i] == acc
        acc = acc * m[i] % R
    polys = [[rng.randrange(R) for _ in range(20)] for _ in range(5)]
    cs = [rng.randrange(R) for _ in range(5)]
    out = pr.linear_combination(polys, cs, 20)
    for i in range(20):
        assert out[i] == sum(c * p[i] for c, p in zip(cs, polys)) % R


def test_permutation_factors_telescope():
    """For the identity permutation sigma_j(omega^i) = delta^j omega^i every factor is 1, so z stays at 1: the
    property the permutation argument rests on (upstream permutation/prover.rs)."""
    rng = random.Random(11)
    k, n = 4, 16
    omega = pow(0x03DDB9F5166D18B798865EA93DD31F743215CF6DD39329C8D34F1ED960C37C9C, 1 << (28 - k), R)
    delta = pow(7, 1 << 28, R)
    cols = [[rng.randrange(R) for _ in range(n)] for _ in range(3)]
    sig = [[pow(delta, j, R) * pow(omega, i, R) % R for i in range(n)] for j in range(3)]
    mv = pr.permutation_factors(cols, sig, omega, delta, rng.randrange(R), rng.randrange(R))
    assert mv == [1] * n
    assert pr.grand_product(mv, 1) == [1] * n


def test_permute_expression_pair_properties():
    """The defining properties of the lookup argument's permuted columns (upstream lookup/prover.rs): A' is the sorted
    input, S' a permutation of the table, and on every row either A'[i] == S'[i] or A'[i] == A'[i-1]."""
    rng = random.Random(21)
    for rows, span in ((1, 1), (7, 3), (100, 16), (1000, 64), (513, 513)):
        table = [rng.randrange(span) for _ in range(rows)]
        table[: min(span, rows)] = list(range(min(span, rows)))          # every small value occurs at least once
        present = sorted(set(table))
        inp = [rng.choice(present) for _ in range(rows)]
        if len(set(inp)) > rows:
            continue
        a, s = pr.permute_expression_pair(inp, table, rows)
        assert a == sorted(inp) and sorted(s) == sorted(table)
        for i in range(rows):
            assert a[i] == s[i] or (i > 0 and a[i] == a[i - 1])
    with pytest.raises(KeyError):
        pr.permute_expression_pair([1, 2, 5], [1, 2, 3], 3)
    # the leftovers go to the repeated rows from the back: smallest leftover on the last repeated row
    a, s = pr.permute_expression_pair([4, 4, 4, 4], [4, 9, 7, 8], 4)
    assert a == [4, 4, 4, 4] and s == [4, 9, 8, 7]
