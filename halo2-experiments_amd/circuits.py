"""The constraint systems of the reference's three `create_proof` configurations (BASELINE.json configs 2-4), as expression
lists for the ``GraphEvaluator`` mirror of evaluation.py -- what ``evaluate_h`` runs per row of the extended domain.

What is transcribed from files of the reference (read as text; cited per gate) and what is recalled:

    MerkleSumTree  /root/reference/src/chips/merkle_sum_tree.rs:32-138    bool / swap / sum gates, the `check == is_lt` gate, the
                   column set (5 chip columns + 5 hash inputs), equality columns: TRANSCRIBED
    MerkleTreeV3   /root/reference/src/chips/merkle_v3.rs:30-82           bool / swap gates, column set: TRANSCRIBED
    Poseidon       /root/reference/src/chips/poseidon/hash.rs:45-72       which columns PoseidonChip::configure allocates
                   (partial_sbox, rc_a, rc_b, equality on the hash inputs, constants in rc_b[0]): TRANSCRIBED
    Pow5Chip       halo2_gadgets::poseidon::Pow5Chip::configure ("full round", "partial rounds", "pad-and-add" gates): RECALLED --
                   the crate is a git dependency that is not in this image (/root/reference/Cargo.toml:11)
    LtChip         zkevm-circuits gadgets::less_than::LtChip::configure ("lt gate": lhs - rhs - sum diff_i 256^i + lt * 2^(8 N);
                   bool check of lt; one u8-range lookup per diff byte): RECALLED (/root/reference/Cargo.toml:14)

The MDS matrix, its inverse and the round constants of the Spec are generic field elements here (halo2_gadgets derives
them with a Grain LFSR): the program's shape -- calculations, rotations, columns -- does not depend on their values.
Selectors are fixed columns, one each (what keygen gives with selector compression off; compression only merges columns).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Sequence, Tuple

from .evaluation import (Advice, Challenge, Constant, Expression, Fixed, GraphEvaluator, Instance, Negated, Product, Scaled, Sum,
                         lookup_expressions, permutation_expressions)
from .domain import FR_MODULUS

R = FR_MODULUS


def degree(e: Expression) -> int:
    """Expression::degree (plonk/circuit.rs): columns 1, constants 0, products add, sums take the maximum."""
    if isinstance(e, (Fixed, Advice, Instance)):
        return 1
    if isinstance(e, Sum):
        return max(degree(e.a), degree(e.b))
    if isinstance(e, Product):
        return degree(e.a) + degree(e.b)
    if isinstance(e, Negated):
        return degree(e.a)
    if isinstance(e, Scaled):
        return degree(e.a)
    return 0


class _Columns:
    """meta.advice_column() / fixed_column() / selector() / instance_column() in allocation order."""

    def __init__(self):
        self.num_advice = self.num_fixed = self.num_instance = 0
        self.equality: List[Tuple[str, int]] = []          # columns in the permutation argument, in enabling order

    def advice(self) -> int:
        self.num_advice += 1
        return self.num_advice - 1

    def fixed(self) -> int:
        self.num_fixed += 1
        return self.num_fixed - 1

    selector = fixed

    def instance(self) -> int:
        self.num_instance += 1
        return self.num_instance - 1

    def enable_equality(self, kind: str, index: int) -> None:
        if (kind, index) not in self.equality:
            self.equality.append((kind, index))


@dataclass
class ConstraintSystem:
    name: str
    source: str
    num_fixed: int
    num_advice: int
    num_instance: int
    gates: List[Tuple[str, List[Expression]]]
    lookups: List[Tuple[List[Expression], List[Expression]]]          # (input expressions, table expressions)
    equality: List[Tuple[str, int]]
    blinding_factors: int = 5

    def polynomials(self) -> List[Expression]:
        return [p for _, polys in self.gates for p in polys]

    def degree(self) -> int:
        """ConstraintSystem::degree: max over the permutation argument (3), the lookup arguments (max(4, 2 + input + table
        degree)) and the gate polynomials."""
        d = 3 if self.equality else 1
        for ins, tabs in self.lookups:
            d = max(d, 4, 2 + max(degree(e) for e in ins) + max(degree(e) for e in tabs))
        for p in self.polynomials():
            d = max(d, degree(p))
        return d

    def permutation_chunk_len(self) -> int:
        return self.degree() - 2

    def permutation_sets(self) -> int:
        c = self.permutation_chunk_len()
        return (len(self.equality) + c - 1) // c


def _rand_constants(count: int, seed: int) -> List[int]:
    x, out = seed, []
    for _ in range(count):
        x = pow(x, 5, R) * 7 % R + 1                      # any generic non-zero elements (see the module docstring)
        out.append(x)
    return out


def pow5_gates(state: Sequence[int], partial_sbox: int, rc_a: Sequence[int], rc_b: Sequence[int], s_full: int, s_partial: int,
               s_pad_and_add: int, rate: int) -> List[Tuple[str, List[Expression]]]:
    """halo2_gadgets Pow5Chip::configure (recalled): the three gates of the Poseidon permutation chip over `state` (WIDTH
    advice columns), `partial_sbox`, the round-constant columns rc_a / rc_b and three selectors."""
    width = len(state)
    m_reg = [_rand_constants(width, 1000 + i) for i in range(width)]
    m_inv = [_rand_constants(width, 2000 + i) for i in range(width)]

    def pow_5(v: Expression) -> Expression:
        v2 = v * v
        return v2 * v2 * v

    full = []
    for nxt in range(width):
        expr = None
        for idx in range(width):
            term = pow_5(Advice(state[idx]) + Fixed(rc_a[idx])) * m_reg[nxt][idx]
            expr = term if expr is None else expr + term
        full.append(Fixed(s_full) * (expr - Advice(state[nxt], 1)))

    cur_0, mid_0 = Advice(state[0]), Advice(partial_sbox)

    def mid(idx: int) -> Expression:
        acc = mid_0 * m_reg[idx][0]
        for j in range(1, width):
            acc = acc + (Advice(state[j]) + Fixed(rc_a[j])) * m_reg[idx][j]
        return acc

    def nxt_lin(idx: int) -> Expression:
        acc = None
        for j in range(width):
            t = Advice(state[j], 1) * m_inv[idx][j]
            acc = t if acc is None else acc + t
        return acc

    partial = [Fixed(s_partial) * (pow_5(cur_0 + Fixed(rc_a[0])) - mid_0),
               Fixed(s_partial) * (pow_5(mid(0) + Fixed(rc_b[0])) - nxt_lin(0))]
    for idx in range(1, width):
        partial.append(Fixed(s_partial) * (mid(idx) + Fixed(rc_b[idx]) - nxt_lin(idx)))

    pad = [Fixed(s_pad_and_add) * (Advice(state[i], -1) + Advice(state[i]) - Advice(state[i], 1)) for i in range(rate)]
    pad.append(Fixed(s_pad_and_add) * (Advice(state[rate], -1) - Advice(state[rate], 1)))
    return [("full round", full), ("partial rounds", partial), ("pad-and-add", pad)]


def _poseidon_chip(cols: _Columns, hash_inputs: Sequence[int], rate: int):
    """PoseidonChip::configure, /root/reference/src/chips/poseidon/hash.rs:45-72: partial_sbox, rc_a, rc_b, equality on the
    hash inputs, constants in rc_b[0]; then Pow5Chip::configure (which allocates its three selectors)."""
    width = len(hash_inputs)
    partial_sbox = cols.advice()
    rc_a = [cols.fixed() for _ in range(width)]
    rc_b = [cols.fixed() for _ in range(width)]
    for c in hash_inputs:
        cols.enable_equality("advice", c)
    cols.enable_equality("fixed", rc_b[0])                 # meta.enable_constant(rc_b[0])
    s_full, s_partial, s_pad = cols.selector(), cols.selector(), cols.selector()
    return pow5_gates(hash_inputs, partial_sbox, rc_a, rc_b, s_full, s_partial, s_pad, rate)


def lt_chip(cols: _Columns, q_enable: Expression, lhs: Expression, rhs: Expression, n_bytes: int = 8):
    """gadgets::less_than::LtChip::configure (recalled): -> (gates, lookups, the `lt` column)."""
    lt = cols.advice()
    diff = [cols.advice() for _ in range(n_bytes)]
    u8 = cols.fixed()
    from_bytes = None
    for i, c in enumerate(diff):
        t = Advice(c) * pow(256, i, R)
        from_bytes = t if from_bytes is None else from_bytes + t
    check_a = lhs - rhs - from_bytes + Advice(lt) * pow(2, 8 * n_bytes, R)
    check_b = Advice(lt) * (Constant(1) - Advice(lt))
    gates = [("lt gate", [q_enable * check_a, q_enable * check_b])]
    lookups = [([Advice(c)], [Fixed(u8)]) for c in diff]
    return gates, lookups, lt


def merkle_sum_tree() -> ConstraintSystem:
    """MerkleSumTreeChip::configure, /root/reference/src/chips/merkle_sum_tree.rs:32-138 (WIDTH 5, RATE 4, LtChip over 8 bytes)."""
    cols = _Columns()
    a, b, c, d, e = (cols.advice() for _ in range(5))      # circuits/merkle_sum_tree.rs:27-31
    inst = cols.instance()
    bool_s, swap_s, sum_s, lt_s = (cols.selector() for _ in range(4))          # :44-47
    for col in (a, b, c, d, e):                                                # :53-57
        cols.enable_equality("advice", col)
    cols.enable_equality("instance", inst)                                     # :58
    A = Advice
    gates = [
        ("bool constraint", [Fixed(bool_s) * A(e) * (Constant(1) - A(e))]),                                          # :62-66
        ("swap constraint", [Fixed(swap_s) * (A(e) * Constant(2) * (A(c) - A(a)) - (A(a, 1) - A(a)) - (A(c) - A(c, 1))),    # :70-92
                             Fixed(swap_s) * (A(e) * Constant(2) * (A(d) - A(b)) - (A(b, 1) - A(b)) - (A(d) - A(d, 1)))]),
        ("sum constraint", [Fixed(sum_s) * (A(b) + A(d) - A(e))]),                                                   # :95-101
    ]
    hash_inputs = [cols.advice() for _ in range(5)]                                                                  # :103
    gates += _poseidon_chip(cols, hash_inputs, rate=4)                                                               # :105-106
    lt_gates, lookups, lt = lt_chip(cols, Fixed(lt_s), A(a), A(b))                                                   # :109-114
    gates += lt_gates
    gates.append(("check == is_lt", [Fixed(lt_s) * (A(lt) - A(c))]))                                                 # :127-138
    return ConstraintSystem("MerkleSumTree depth 20 (config 4)", "chips/merkle_sum_tree.rs:32-138 (+ Pow5Chip, LtChip recalled)",
                            cols.num_fixed, cols.num_advice, cols.num_instance, gates, lookups, cols.equality)


def merkle_v3() -> ConstraintSystem:
    """MerkleTreeV3Chip::configure, /root/reference/src/chips/merkle_v3.rs:30-82 (WIDTH 3, RATE 2)."""
    cols = _Columns()
    a, b, c = (cols.advice() for _ in range(3))
    inst = cols.instance()
    bool_s, swap_s = cols.selector(), cols.selector()                          # :38-39
    for col in (a, b, c):                                                      # :41-43
        cols.enable_equality("advice", col)
    cols.enable_equality("instance", inst)                                     # :44
    A = Advice
    gates = [
        ("bool constraint", [Fixed(bool_s) * A(c) * (Constant(1) - A(c))]),                                          # :48-52
        ("swap constraint", [Fixed(swap_s) * (A(c) * Constant(2) * (A(b) - A(a)) - (A(a, 1) - A(a)) - (A(b) - A(b, 1)))]),  # :57-68
    ]
    hash_inputs = [cols.advice() for _ in range(3)]                                                                  # :70
    gates += _poseidon_chip(cols, hash_inputs, rate=2)                                                               # :72-73
    return ConstraintSystem("MerkleTreeV3 depth 20 (config 3)", "chips/merkle_v3.rs:30-82 (+ Pow5Chip recalled)",
                            cols.num_fixed, cols.num_advice, cols.num_instance, gates, [], cols.equality)


def poseidon() -> ConstraintSystem:
    """PoseidonCircuit::configure, /root/reference/src/circuits/poseidon.rs:36-41 with WIDTH 5, RATE 4 (:79-81): the hash
    inputs, an instance column for the digest, PoseidonChip (hash_with_instance.rs: the same allocations + equality on the
    instance column)."""
    cols = _Columns()
    inst = cols.instance()
    hash_inputs = [cols.advice() for _ in range(5)]
    gates = _poseidon_chip(cols, hash_inputs, rate=4)
    cols.enable_equality("instance", inst)
    return ConstraintSystem("Poseidon (config 2)", "circuits/poseidon.rs:36-41, chips/poseidon/hash_with_instance.rs (+ Pow5Chip recalled)",
                            cols.num_fixed, cols.num_advice, cols.num_instance, gates, [], cols.equality)


CONSTRAINT_SYSTEMS: Dict[str, Callable[[], ConstraintSystem]] = {
    "poseidon_k11": poseidon, "merkle_v3_k17": merkle_v3, "merkle_sum_tree_k18": merkle_sum_tree,
    # the ONE configuration the reference proves for real: test_full_prover, k = 9, a depth-5 path
    # (/root/reference/src/circuits/merkle_sum_tree.rs:345-358, path :172-212) -- the same constraint system as k = 18
    "merkle_sum_tree_k9": merkle_sum_tree,
}


@dataclass
class EvaluateHLayout:
    """Column table of the whole evaluate_h program: the circuit's fixed columns, then (as further fixed-kind entries) the
    permutation sigmas, the permutation z polynomials, l0 / l_last / l_active, the coset column zeta * omega_ext^idx, per
    lookup (z, permuted input, permuted table), and the short inverse-vanishing pattern; then advice; then instance."""
    num_fixed_entries: int
    sigma0: int
    z0: int
    l0: int
    l_last: int
    l_active: int
    x_coset: int
    lookup0: int
    t_inv: int
    short_columns: Dict[int, int] = field(default_factory=dict)


def evaluate_h_program(cs: ConstraintSystem, k: int, extended_k: int, delta: int, with_arguments: bool = True, per_coset: bool = False,
                       divide: bool = True):
    """GraphEvaluator for evaluate_h of `cs`: the custom gates, then (with_arguments) the permutation argument over the
    equality columns, every lookup argument, and divide_by_vanishing_poly as the last multiplication.
    per_coset: the program for ONE coset of the n-th roots inside the extended domain (EvaluationDomain.coeff_to_coset;
    compile with rot_scale = 1, evaluate over 2^k rows): 1 / (X^n - 1) is a single constant there and enters as Challenge(0)
    (compile with num_challenges = 1; the t_inv entry of the column table stays, unread).
    divide=False: the numerator only -- the division rides on the recombination (EvaluationDomain.combine_cosets(...,
    divide_by_vanishing=True)), so one launch can take several cosets as segments (CompiledGraph.evaluate(segments=...)).
    -> (GraphEvaluator, EvaluateHLayout)."""
    nf = cs.num_fixed
    P, nsets, L = len(cs.equality), cs.permutation_sets(), len(cs.lookups)
    lay = EvaluateHLayout(num_fixed_entries=nf, sigma0=nf, z0=nf + P, l0=nf + P + nsets, l_last=nf + P + nsets + 1,
                          l_active=nf + P + nsets + 2, x_coset=nf + P + nsets + 3, lookup0=nf + P + nsets + 4,
                          t_inv=nf + P + nsets + 4 + 3 * L)
    lay.num_fixed_entries = lay.t_inv + 1
    lay.short_columns = {lay.t_inv: extended_k - k}
    polys = list(cs.polynomials())
    if with_arguments:
        kind = {"advice": Advice, "fixed": Fixed, "instance": Instance}
        perm_cols = [kind[kd](i) for kd, i in cs.equality]
        polys += permutation_expressions(perm_cols, [Fixed(lay.sigma0 + j) for j in range(P)],
                                         [lambda r, i=i: Fixed(lay.z0 + i, r) for i in range(nsets)], Fixed(lay.l0), Fixed(lay.l_last),
                                         Fixed(lay.l_active), Fixed(lay.x_coset), cs.permutation_chunk_len(), delta,
                                         -(cs.blinding_factors + 1))
        for j, (ins, tabs) in enumerate(cs.lookups):
            b = lay.lookup0 + 3 * j
            polys += lookup_expressions(ins, tabs, lambda r, b=b: Fixed(b, r), lambda r, b=b: Fixed(b + 1, r),
                                        lambda r, b=b: Fixed(b + 2, r), Fixed(lay.l0), Fixed(lay.l_last), Fixed(lay.l_active))
    g = GraphEvaluator()
    g.add_custom_gates(polys)
    if with_arguments and divide:
        g.add_vanishing_division(Challenge(0) if per_coset else Fixed(lay.t_inv))
    if per_coset:
        lay.short_columns = {}
    return g, lay
