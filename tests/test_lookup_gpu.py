"""GPU suite: the lookup argument's permuted columns (permute_expression_pair) through the C ABI, bit-exact against
oracle/poly_ref.py, plus the defining properties at 2^20 rows."""
import random

import numpy as np
import pytest

import halo2_experiments_amd as h
from halo2_experiments_amd import _lib
from oracle import poly_ref as pr

pytestmark = pytest.mark.gpu
R = pr.R


def to_gpu(pyref, values):
    import torch
    return torch.from_numpy(pyref.fr_array(values).view(np.int64)).cuda()


def from_gpu(pyref, t):
    return pyref.fr_from_array(t.cpu().numpy().view(np.uint64))


def make_pair(rng, n, rows, kind):
    if kind == "range":                     # a range-check lookup: small values, many repeats
        span = max(1, min(rows, 1 << 8))
        table = [i % span for i in range(n)]
        inp = [rng.randrange(span) for _ in range(n)]
    elif kind == "compressed":              # theta-compressed multi-column lookup: random field elements, few repeats
        table = [rng.randrange(R) for _ in range(n)]
        inp = [table[rng.randrange(rows)] for _ in range(n)]
    elif kind == "permutation":             # every value exactly once on both sides
        table = list(range(100, 100 + n))
        inp = table[:rows]
        rng.shuffle(inp)
        inp += [0] * (n - rows)
    else:                                   # edge values around the modulus and word boundaries
        vals = [0, 1, R - 1, R - 2, (1 << 64) - 1, 1 << 64, (1 << 128) + 5, (1 << 192), (1 << 253) + 9, 2 ** 32, 2 ** 32 - 1]
        table = [vals[i % len(vals)] for i in range(n)]
        inp = [vals[rng.randrange(len(vals))] for _ in range(n)]
    return inp, table


@pytest.mark.parametrize("k,blinding,kind", [(1, 0, "range"), (3, 2, "range"), (6, 5, "edge"), (10, 6, "range"), (10, 6, "compressed"),
                                             (11, 6, "permutation"), (12, 6, "edge"), (12, 0, "compressed"), (13, 100, "range"),
                                             (14, 6, "compressed"), (16, 6, "range")])
def test_permute_expression_pair_matches_oracle(pyref, k, blinding, kind):
    """One tile, exactly one tile (2^11), several tiles (global stages), usable rows that are not a power of two."""
    import torch
    n = 1 << k
    rows = n - blinding - 1 if n > blinding + 1 else n
    rng = random.Random(1000 * k + blinding)
    inp, table = make_pair(rng, n, rows, kind)
    a, s = h.permute_expression_pair(to_gpu(pyref, inp), to_gpu(pyref, table), rows, blinding_seed=7)
    want_a, want_s = pr.permute_expression_pair(inp, table, rows)
    assert from_gpu(pyref, a[:rows]) == want_a
    assert from_gpu(pyref, s[:rows]) == want_s
    if rows < n:                            # blinding rows: canonical field elements, not all equal
        tail = from_gpu(pyref, torch.cat([a[rows:], s[rows:]]))
        assert all(0 <= v < R for v in tail) and len(set(tail)) > 1


@pytest.mark.parametrize("kind", ["bytes", "20-bit", "21-bit", "constant", "mostly-zero", "mixed-batch"])
def test_small_keys_take_the_counting_sort_and_agree_with_the_oracle(pyref, kind):
    """Round 4: when every live key of a chain is below 2^20 (range checks, byte tables: the reference's lookups) the 256-bit
    bitonic network is replaced by a counting sort -- histogram, scan, expansion.  Bit-exact against the oracle at 2^14 rows for
    byte values, values that need all 20 bits, values one bit above the limit (the network again), one constant (every wave on
    one counter), a column of mostly zeros, and a batch in which ONE lookup holds large values (the whole chain then takes the
    network)."""
    k, blinding = 14, 6
    n, rows = 1 << k, (1 << k) - blinding - 1
    rng = random.Random(hash(kind) & 0xFFFF)
    def pair(span, top=0):
        table = [(i % span) | top for i in range(n)]
        inp = [table[rng.randrange(rows)] for _ in range(n)]
        return inp, table
    if kind == "bytes":
        pairs = [pair(256)]
    elif kind == "20-bit":
        pairs = [pair(5000, top=1 << 19)]
    elif kind == "21-bit":
        pairs = [pair(5000, top=1 << 20)]
    elif kind == "constant":
        pairs = [([7] * n, [7] * n)]
    elif kind == "mostly-zero":
        table = [0] * n
        for i in range(300):
            table[i] = i
        inp = [0 if rng.random() < 0.95 else rng.randrange(300) for _ in range(n)]
        pairs = [(inp, table)]
    else:
        pairs = [pair(256), make_pair(rng, n, rows, "compressed"), pair(65536), pair(3)]
    outs = h.permute_expression_pairs([to_gpu(pyref, p[0]) for p in pairs], [to_gpu(pyref, p[1]) for p in pairs], rows, blinding_seed=11)
    for (inp, table), (a, s_) in zip(pairs, outs):
        want_a, want_s = pr.permute_expression_pair(inp, table, rows)
        assert from_gpu(pyref, a[:rows]) == want_a and from_gpu(pyref, s_[:rows]) == want_s
    if kind == "bytes":                     # and a missing value is still an error on this path
        bad = list(pairs[0][0])
        bad[123] = 256
        with pytest.raises(_lib.Halo2Mi355xError):
            h.permute_expression_pair(to_gpu(pyref, bad), to_gpu(pyref, pairs[0][1]), rows)


def test_missing_input_value_is_an_error(pyref):
    """Upstream returns Error::ConstraintSystemFailure when an input value does not occur in the table."""
    n = 1 << 10
    table = [i % 50 for i in range(n)]
    inp = [i % 50 for i in range(n)]
    inp[777] = 50
    with pytest.raises(_lib.Halo2Mi355xError) as e:
        h.permute_expression_pair(to_gpu(pyref, inp), to_gpu(pyref, table), n - 7)
    assert "missing from the table" in str(e.value)
    inp[777] = 49
    h.permute_expression_pair(to_gpu(pyref, inp), to_gpu(pyref, table), n - 7)      # and the library is usable afterwards


def test_large_lookup_properties(pyref):
    """2^20 rows (nine phases of global stages): A' sorted, S' a permutation of the table, A'[i] in {S'[i], A'[i-1]}."""
    import torch
    n = 1 << 20
    rows = n - 7
    g = torch.Generator(device="cuda")
    g.manual_seed(5)
    small = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    small[:, 0] = torch.arange(n, device="cuda") % 4096                       # table: 0 .. 4095 repeated
    table = h.linear_combination([small], np.stack([pyref.fr_array([pow(2, 256, R)])[0]]))     # raw small integers -> Montgomery
    inp_small = torch.zeros((n, 4), dtype=torch.int64, device="cuda")
    inp_small[:, 0] = torch.randint(0, 4096, (n,), device="cuda", generator=g)
    inp = h.linear_combination([inp_small], np.stack([pyref.fr_array([pow(2, 256, R)])[0]]))
    a, s = h.permute_expression_pair(inp, table, rows)
    av = inp_small[:rows, 0].sort().values
    # back to small integers: multiply by 2^-256
    back = lambda t: h.linear_combination([t], np.stack([pyref.fr_array([pow(2, -256, R)])[0]]))
    a_int, s_int = back(a[:rows].contiguous()), back(s[:rows].contiguous())
    assert not a_int[:, 1:].any() and not s_int[:, 1:].any()
    assert torch.equal(a_int[:, 0], av)
    assert torch.equal(s_int[:, 0].sort().values, small[:rows, 0].sort().values)
    same_as_table = a_int[:, 0] == s_int[:, 0]
    same_as_prev = torch.zeros(rows, dtype=torch.bool, device="cuda")
    same_as_prev[1:] = a_int[1:, 0] == a_int[:-1, 0]
    assert bool((same_as_table | same_as_prev).all())


@pytest.mark.parametrize("count", [1, 3, 8, 11])
def test_batch_of_lookups_equals_single_calls(pyref, count):
    """All lookup arguments of a circuit in one call (one chain carries eight; eleven = two chains), different kinds and
    one pair repeated; and a failing lookup among them is reported by index."""
    k, blinding = 12, 6
    n, rows = 1 << k, (1 << k) - blinding - 1
    rng = random.Random(900 + count)
    kinds = ["range", "compressed", "edge", "permutation"]
    pairs = [make_pair(rng, n, rows, kinds[i % 4]) for i in range(count)]
    d_in, d_tab = [to_gpu(pyref, p[0]) for p in pairs], [to_gpu(pyref, p[1]) for p in pairs]
    outs = h.permute_expression_pairs(d_in, d_tab, rows, blinding_seed=3)
    assert len(outs) == count
    for (inp, table), (a, s) in zip(pairs, outs):
        want_a, want_s = pr.permute_expression_pair(inp, table, rows)
        assert from_gpu(pyref, a[:rows]) == want_a and from_gpu(pyref, s[:rows]) == want_s
    if count >= 3:
        bad = list(pairs[count - 2][0])
        bad[5] = (max(pairs[count - 2][1]) + 12345) % R
        d_in[count - 2] = to_gpu(pyref, bad)
        with pytest.raises(_lib.Halo2Mi355xError) as e:
            h.permute_expression_pairs(d_in, d_tab, rows)
        assert e.value.missing == [count - 2]
