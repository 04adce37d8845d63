"""ORACLE (test infrastructure only -- never imported by the product): CPU restatement on Python integers of the
Fr vector steps create_proof makes between its NTTs and commitments (upstream halo2_proofs at the tag pinned by
/root/reference/Cargo.toml:10 and its dependency `ff` [UPSTREAM-RECALLED]; reached from the reference through
create_proof, /root/reference/src/circuits/utils.rs:40-48; parity unpinned: the reference holds no vectors for them).

    kate_division        halo2_proofs/src/arithmetic.rs  kate_division
    batch_invert         ff::BatchInvert for slices (zero elements are skipped and stay zero)
    grand_product        halo2_proofs/src/plonk/permutation/prover.rs (and lookup/prover.rs): the loop that pushes
                         z[row] = z[row - 1] * modified_values[row - 1] after z[0] = last_z
    linear_combination   halo2_proofs/src/poly.rs  `Polynomial * F` and `Polynomial + &Polynomial`
    permutation_factors  the numerator / denominator sweep of permutation/prover.rs that feeds the two above
    permute_expression_pair  halo2_proofs/src/plonk/lookup/prover.rs  permute_expression_pair (without the blinding rows)

Values are canonical integers in [0, r); the tests convert to and from the reference's Montgomery words.
"""
from typing import List, Sequence

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def kate_division(a: Sequence[int], b: int) -> List[int]:
    """Follows upstream line by line: b = -b; walk the coefficients from the top, q = lead - tmp, tmp = q * b."""
    if len(a) == 0:
        raise ValueError("kate_division: empty polynomial")        # upstream: usize underflow panic
    b = (-b) % R
    q = [0] * (len(a) - 1)
    tmp = 0
    for i, r in zip(range(len(q) - 1, -1, -1), reversed(a)):
        lead = (r - tmp) % R
        q[i] = lead
        tmp = lead * b % R
    return q


def batch_invert(values: Sequence[int]) -> List[int]:
    """ff::BatchInvert: running products over the non-zero elements, one inversion, walk back."""
    acc = 1
    tmp = []
    for v in values:
        tmp.append(acc)
        if v % R != 0:
            acc = acc * v % R
    acc = pow(acc, R - 2, R)
    out = list(values)
    for i in range(len(values) - 1, -1, -1):
        v = values[i] % R
        if v == 0:
            out[i] = 0
            continue
        out[i] = acc * tmp[i] % R
        acc = acc * v % R
    return out


def grand_product(factors: Sequence[int], start: int) -> List[int]:
    """z = [start]; for row in 1..n: z.push(z[row - 1] * factors[row - 1])  (n = len(factors) entries)."""
    if not factors:
        return []
    z = [start % R]
    for row in range(1, len(factors)):
        z.append(z[row - 1] * factors[row - 1] % R)
    return z


def linear_combination(polys: Sequence[Sequence[int]], coeffs: Sequence[int], n: int) -> List[int]:
    out = [0] * n
    for p, c in zip(polys, coeffs):
        for i in range(n):
            out[i] = (out[i] + c * p[i]) % R
    return out


def permutation_factors(values: Sequence[Sequence[int]], sigmas: Sequence[Sequence[int]], omega: int, delta: int, beta: int, gamma: int,
                        first_column: int = 0) -> List[int]:
    """modified_values of permutation/prover.rs for one chunk of columns: start from ones, multiply the denominators
    (beta * sigma + gamma + value), batch-invert, then multiply the numerators (delta^column * beta * omega^row + gamma
    + value)."""
    n = len(values[0])
    mv = [1] * n
    for col, sig in zip(values, sigmas):
        for i in range(n):
            mv[i] = mv[i] * ((beta * sig[i] + gamma + col[i]) % R) % R
    mv = batch_invert(mv)
    deltaomega = pow(delta, first_column, R)
    for col in values:
        cur = deltaomega
        for i in range(n):
            mv[i] = mv[i] * ((cur * beta + gamma + col[i]) % R) % R
            cur = cur * omega % R
        deltaomega = deltaomega * delta % R
    return mv


def permute_expression_pair(input_values: Sequence[int], table_values: Sequence[int], usable_rows: int):
    """Follows upstream: sort the input; count the table's values in an ordered map; the first occurrence of every input
    value takes that value into the permuted table at its row and one instance out of the map (absent: the error
    ConstraintSystemFailure, here KeyError); repeated rows are collected; the leftover table values, in ascending
    order, are written to repeated rows popped from the BACK of that list."""
    permuted_input = sorted(v % R for v in input_values[:usable_rows])
    leftover = {}
    for v in table_values[:usable_rows]:
        leftover[v % R] = leftover.get(v % R, 0) + 1
    permuted_table = [0] * usable_rows
    repeated_rows = []
    for row, v in enumerate(permuted_input):
        if row == 0 or v != permuted_input[row - 1]:
            permuted_table[row] = v
            if leftover.get(v, 0) == 0:
                raise KeyError("ConstraintSystemFailure: input value not in the table")
            leftover[v] -= 1
        else:
            repeated_rows.append(row)
    for coeff in sorted(leftover):
        for _ in range(leftover[coeff]):
            permuted_table[repeated_rows.pop()] = coeff
    assert not repeated_rows
    return permuted_input, permuted_table
