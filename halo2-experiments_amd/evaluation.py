"""Host-side mirror of ``halo2_proofs::plonk::evaluation::GraphEvaluator`` (the gate arithmetic of ``evaluate_h``,
SURVEY.md §8f-4; upstream ``halo2_proofs/src/plonk/evaluation.rs`` and ``plonk/circuit.rs::Expression`` at the tag
pinned by /root/reference/Cargo.toml:10; reached from ``create_proof``, /root/reference/src/circuits/utils.rs:40-48).

Upstream flattens every gate polynomial of the constraint system into a straight-line program with common
sub-expressions shared, then evaluates it once per row of the extended domain:

    Expression     Constant | Fixed(query) | Advice(query) | Instance(query) | Challenge | Negated | Sum | Product | Scaled
    ValueSource    Constant(i) | Intermediate(i) | Fixed(col, rot_idx) | Advice(..) | Instance(..) | Challenge(i)
                   | Beta | Gamma | Theta | Y | PreviousValue
    Calculation    Add | Sub | Mul | Square | Double | Negate | Horner(start, parts, factor) | Store
    add_expression the same peephole rules as upstream: a + (-b) -> Sub, x * 2 -> Double, x * x -> Square, constants
                   0 / 1 folded, commutative operands ordered, every Calculation deduplicated
    custom gates   value = Horner(PreviousValue, [gate polynomials...], Y)

``GraphEvaluator.compile`` lowers that program to the five-word device calculations of ``hm_graph_create`` (one column
table: fixed, then advice, then instance; Challenge / Beta / Gamma / Theta / Y as the per-call constants; Horner as a
MulAdd chain), and ``evaluate`` runs it on extended-domain columns resident in HBM.  Field elements here are Python
integers in [0, r).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import Dict, List, Sequence, Tuple

import numpy as np

from . import _lib
from .arithmetic import _ptr, _stream_ptr, _tensor_rows
from .domain import FR_MODULUS, fr_words

R = FR_MODULUS


# ---- Expression (plonk/circuit.rs) -------------------------------------------------------------------
class Expression:
    def __add__(self, o):
        return Sum(self, _lift(o))

    def __radd__(self, o):
        return Sum(_lift(o), self)

    def __sub__(self, o):
        return Sum(self, Negated(_lift(o)))

    def __rsub__(self, o):
        return Sum(_lift(o), Negated(self))

    def __mul__(self, o):
        if isinstance(o, int):
            return Scaled(self, o % R)
        return Product(self, o)

    def __rmul__(self, o):
        return Scaled(self, o % R)

    def __neg__(self):
        return Negated(self)


def _lift(o) -> "Expression":
    return o if isinstance(o, Expression) else Constant(o % R)


@dataclass(frozen=True)
class Constant(Expression):
    value: int


@dataclass(frozen=True)
class Fixed(Expression):
    column: int
    rotation: int = 0


@dataclass(frozen=True)
class Advice(Expression):
    column: int
    rotation: int = 0


@dataclass(frozen=True)
class Instance(Expression):
    column: int
    rotation: int = 0


@dataclass(frozen=True)
class Challenge(Expression):
    index: int


@dataclass(frozen=True)
class ProofScalar(Expression):
    """beta / gamma / theta of the proof as a leaf: upstream has them as ValueSource kinds only (its permutation and
    lookup terms are hand-written loops); here those terms are expression trees too, so they need a leaf."""
    name: str        # "Beta" | "Gamma" | "Theta"


BETA, GAMMA, THETA = ProofScalar("Beta"), ProofScalar("Gamma"), ProofScalar("Theta")


@dataclass(frozen=True)
class Negated(Expression):
    a: Expression


@dataclass(frozen=True)
class Sum(Expression):
    a: Expression
    b: Expression


@dataclass(frozen=True)
class Product(Expression):
    a: Expression
    b: Expression


@dataclass(frozen=True)
class Scaled(Expression):
    a: Expression
    factor: int


# ---- ValueSource / Calculation (plonk/evaluation.rs) -----------------------------------------------------
# a value source is a tuple whose first element is its kind; tuples order like upstream's derived Ord (by variant,
# then fields), which decides the operand order of commutative calculations
_KINDS = ("Constant", "Intermediate", "Fixed", "Advice", "Instance", "Challenge", "Beta", "Gamma", "Theta", "Y", "PreviousValue")
_RANK = {k: i for i, k in enumerate(_KINDS)}


def _key(vs):
    return (_RANK[vs[0]],) + tuple(vs[1:])


class GraphEvaluator:
    def __init__(self):
        self.constants: List[int] = [0, 1, 2]
        self.rotations: List[int] = []
        self.calculations: List[Tuple] = []           # (calculation tuple, target)
        self._calc_index: Dict[Tuple, int] = {}
        self.num_intermediates = 0

    # -- upstream's builders ------------------------------------------------------------------------------
    def add_rotation(self, rotation: int) -> int:
        if rotation in self.rotations:
            return self.rotations.index(rotation)
        self.rotations.append(rotation)
        return len(self.rotations) - 1

    def add_constant(self, c: int):
        c %= R
        if c in self.constants:
            return ("Constant", self.constants.index(c))
        self.constants.append(c)
        return ("Constant", len(self.constants) - 1)

    def add_calculation(self, calc: Tuple):
        if calc in self._calc_index:
            return ("Intermediate", self._calc_index[calc])
        target = self.num_intermediates
        self.calculations.append((calc, target))
        self._calc_index[calc] = target
        self.num_intermediates += 1
        return ("Intermediate", target)

    def add_expression(self, e: Expression):
        zero, one, two = ("Constant", 0), ("Constant", 1), ("Constant", 2)
        if isinstance(e, Constant):
            return self.add_constant(e.value)
        if isinstance(e, (Fixed, Advice, Instance)):
            rot = self.add_rotation(e.rotation)
            return self.add_calculation(("Store", (type(e).__name__, e.column, rot)))
        if isinstance(e, Challenge):
            return self.add_calculation(("Store", ("Challenge", e.index)))
        if isinstance(e, ProofScalar):
            return self.add_calculation(("Store", (e.name,)))
        if isinstance(e, Negated):
            if isinstance(e.a, Constant):
                return self.add_constant(-e.a.value)
            r = self.add_expression(e.a)
            return r if r == zero else self.add_calculation(("Negate", r))
        if isinstance(e, Sum):
            if isinstance(e.b, Negated):                      # undo subtraction stored as a + (-b)
                ra, rb = self.add_expression(e.a), self.add_expression(e.b.a)
                if ra == zero:
                    return self.add_calculation(("Negate", rb))
                if rb == zero:
                    return ra
                return self.add_calculation(("Sub", ra, rb))
            ra, rb = self.add_expression(e.a), self.add_expression(e.b)
            if ra == zero:
                return rb
            if rb == zero:
                return ra
            return self.add_calculation(("Add", ra, rb) if _key(ra) <= _key(rb) else ("Add", rb, ra))
        if isinstance(e, Product):
            ra, rb = self.add_expression(e.a), self.add_expression(e.b)
            if ra == zero or rb == zero:
                return zero
            if ra == one:
                return rb
            if rb == one:
                return ra
            if ra == two:
                return self.add_calculation(("Double", rb))
            if rb == two:
                return self.add_calculation(("Double", ra))
            if ra == rb:
                return self.add_calculation(("Square", ra))
            return self.add_calculation(("Mul", ra, rb) if _key(ra) <= _key(rb) else ("Mul", rb, ra))
        if isinstance(e, Scaled):
            if e.factor % R == 0:
                return zero
            if e.factor % R == 1:
                return self.add_expression(e.a)
            cst = self.add_constant(e.factor)
            ra = self.add_expression(e.a)
            return self.add_calculation(("Mul", ra, cst))
        raise TypeError(f"not an Expression: {e!r}")

    def add_custom_gates(self, polynomials: Sequence[Expression]):
        """Evaluator::new's custom-gate part: Horner(PreviousValue, parts, Y) over every gate polynomial."""
        parts = tuple(self.add_expression(p) for p in polynomials)
        return self.add_calculation(("Horner", ("PreviousValue",), parts, ("Y",)))

    def add_vanishing_division(self, t_inverse_column: Expression):
        """EvaluationDomain::divide_by_vanishing_poly as the program's last step: value *= t_inv[idx mod 2^(extended_k - k)],
        with the inverse pattern given as a (short, periodically read) column."""
        if not self.calculations:
            raise ValueError("add_vanishing_division: the program is empty")
        last = ("Intermediate", self.calculations[-1][1])
        return self.add_calculation(("Mul", last, self.add_expression(t_inverse_column)))

    # -- lowering to the device program ----------------------------------------------------------------------
    def compile(self, num_fixed: int, num_advice: int, num_instance: int, num_challenges: int = 0, rot_scale: int = 1,
                short_columns: Dict[int, int] = None) -> "CompiledGraph":
        """Column table = fixed | advice | instance.  Per-call constants = challenges..., beta, gamma, theta, y.
        short_columns: {column table index: log2(rows)} for columns shorter than the domain, read periodically (the inverse
        vanishing-polynomial pattern)."""
        short_columns = short_columns or {}
        n_static = len(self.constants)
        dyn_index = {("Challenge", i): n_static + i for i in range(num_challenges)}
        for j, name in enumerate(("Beta", "Gamma", "Theta", "Y")):
            dyn_index[(name,)] = n_static + num_challenges + j
        col_base = {"Fixed": 0, "Advice": num_fixed, "Instance": num_fixed + num_advice}
        n_cols = num_fixed + num_advice + num_instance
        words: List[int] = []
        next_inter = [self.num_intermediates]

        def src(vs) -> int:
            kind = vs[0]
            if kind == "Constant":
                return (0 << 30) | vs[1]
            if kind == "Intermediate":
                return (1 << 30) | vs[1]
            if kind in col_base:
                col = col_base[kind] + vs[1]
                if not 0 <= vs[1] < {"Fixed": num_fixed, "Advice": num_advice, "Instance": num_instance}[kind]:
                    raise ValueError(f"{kind} column {vs[1]} out of range")
                return (2 << 30) | (vs[2] << 20) | (short_columns.get(col, 0) << 14) | col
            if kind == "PreviousValue":
                return 3 << 30
            if vs in dyn_index:
                return (0 << 30) | dyn_index[vs]
            raise ValueError(f"value source {vs!r} has no slot in this compilation")

        ops = {"Add": 0, "Sub": 1, "Mul": 2, "Square": 3, "Double": 4, "Negate": 5, "Store": 6}
        for calc, target in self.calculations:
            name = calc[0]
            if name == "Horner":
                start, parts, factor = calc[1], calc[2], calc[3]
                cur = src(start)
                if not parts:
                    words += [6, cur, 0, 0, target]
                for i, part in enumerate(parts):          # value = value * factor + part
                    t = target if i == len(parts) - 1 else next_inter[0]
                    if t != target:
                        next_inter[0] += 1
                    words += [7, cur, src(factor), src(part), t]
                    cur = (1 << 30) | t
            elif name in ("Add", "Sub", "Mul"):
                words += [ops[name], src(calc[1]), src(calc[2]), 0, target]
            else:
                words += [ops[name], src(calc[1]), 0, 0, target]
        return CompiledGraph(np.array(words, dtype=np.uint32).reshape(-1, 5), list(self.constants), num_challenges + 4,
                             [r * rot_scale for r in self.rotations], n_cols, next_inter[0], num_challenges, dict(short_columns))


class CompiledGraph:
    """A program uploaded to the device (``hm_graph_create``); ``evaluate`` = upstream's per-row loop of evaluate_h."""

    def __init__(self, calcs: np.ndarray, constants: List[int], n_dynamic: int, rotations: List[int], n_columns: int,
                 n_intermediates: int, num_challenges: int, short_columns: Dict[int, int] = None):
        self.short_columns = short_columns or {}
        self.calcs, self.constants, self.n_dynamic = np.ascontiguousarray(calcs), constants, n_dynamic
        self.rotations, self.n_columns, self.n_intermediates = rotations, n_columns, n_intermediates
        self.num_challenges = num_challenges
        lib = _lib.load()
        consts = np.stack([fr_words(c) for c in constants]) if constants else np.zeros((0, 4), dtype=np.uint64)
        rots = np.array(rotations, dtype=np.int32)
        h = ctypes.c_uint64(0)
        _lib.check(lib.hm_graph_create(self.calcs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), self.calcs.shape[0],
                                       _ptr(consts) if len(constants) else None, len(constants), n_dynamic,
                                       rots.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)) if len(rotations) else None,
                                       len(rotations), n_columns, n_intermediates, ctypes.byref(h)))
        self.handle = h.value

    def evaluate(self, columns: Sequence, values, challenges: Sequence[int] = (), beta: int = 0, gamma: int = 0, theta: int = 0,
                 y: int = 0, columns_internal: bool = False, segments: int = 1) -> None:
        """``columns``: GPU tensors (size, 4), fixed then advice then instance; ``values``: (size, 4) GPU tensor holding
        PreviousValue on entry and the program's value on return.  ``columns_internal``: every column holds 32 * value
        (HM_GRAPH_COLUMNS_INTERNAL: what ``EvaluationDomain.coeff_to_extended(..., internal=True)`` writes).
        ``segments`` > 1: the rows are that many back-to-back blocks of size / segments rows (a power of two) and a rotation
        wraps inside its block -- several cosets of the extended domain in one launch (``EvaluationDomain.coeff_to_cosets``)."""
        size = _tensor_rows(values, 4, "values")
        if segments < 1 or size % segments:
            raise ValueError("evaluate: the rows must be a whole number of segments")
        seg = size // segments
        if seg & (seg - 1) or seg == 0:
            raise ValueError("evaluate: the (segment of the) domain must be a power of two")
        if len(columns) != self.n_columns or len(challenges) != self.num_challenges:
            raise ValueError("evaluate: column / challenge count differs from the compiled program's")
        for i, c in enumerate(columns):
            want = (1 << self.short_columns[i]) if i in self.short_columns else size
            if _tensor_rows(c, 4, "column") != want:
                raise ValueError(f"evaluate: column {i} must hold {want} rows")
        ptrs = (ctypes.c_void_p * max(len(columns), 1))(*[c.data_ptr() for c in columns])
        dyn = np.stack([fr_words(v) for v in list(challenges) + [beta, gamma, theta, y]])
        _lib.check(_lib.load().hm_graph_evaluate_segments_dev(ctypes.c_uint64(self.handle), ptrs, len(columns), _ptr(dyn), dyn.shape[0],
                                                              seg.bit_length() - 1, segments, ctypes.c_void_p(values.data_ptr()),
                                                              1 if columns_internal else 0, ctypes.c_void_p(_stream_ptr(values))))

    def _quotient_args(self, domain, coeff_columns, cosets, challenges, beta, gamma, theta, y, on_cosets, who):
        cosets = list(range(domain.min_cosets())) if cosets is None else [int(c) for c in cosets]
        if len(coeff_columns) != self.n_columns or len(challenges) != self.num_challenges:
            raise ValueError(f"{who}: column / challenge count differs from the compiled program's")
        if self.short_columns:
            raise ValueError(f"{who}: the program must not read short columns (compile the undivided numerator)")
        q = len(cosets)
        pre = list(on_cosets) if on_cosets is not None else [None] * len(coeff_columns)
        if len(pre) != len(coeff_columns):
            raise ValueError(f"{who}: on_cosets must have one entry per column")
        cols = [None if pre[i] is not None else c.contiguous() for i, c in enumerate(coeff_columns)]
        for c in cols:
            if c is not None and _tensor_rows(c, 4, "column") != domain.n:
                raise ValueError(f"{who}: every column must hold n coefficients")
        pre = [None if p is None else p.contiguous() for p in pre]
        for p in pre:
            if p is not None and _tensor_rows(p, 4, "on_cosets") != q * domain.n:
                raise ValueError(f"{who}: a column on the cosets must hold len(cosets) * n values")
        ref = next(t for t in cols + pre if t is not None)
        ptrs = (ctypes.c_void_p * len(cols))(*[(c.data_ptr() if c is not None else None) for c in cols])
        pre_ptrs = (ctypes.c_void_p * len(cols))(*[(p.data_ptr() if p is not None else None) for p in pre]) if on_cosets is not None else None
        dyn = np.stack([fr_words(v) for v in list(challenges) + [beta, gamma, theta, y]])
        shifts = np.stack([fr_words(domain.coset_shift(c)) for c in cosets])
        return cosets, q, ref, (cols, pre), ptrs, pre_ptrs, dyn, shifts

    def quotient_by_cosets(self, domain, coeff_columns: Sequence, cosets=None, challenges: Sequence[int] = (), beta: int = 0, gamma: int = 0,
                           theta: int = 0, y: int = 0, on_cosets: Sequence = None):
        """h(X) in ONE call (``hm_quotient_by_cosets_bn256_fr_dev``): this program must be the UNDIVIDED numerator compiled with
        ``rot_scale=1`` (``circuits.evaluate_h_program(per_coset=True, divide=False)``); ``coeff_columns``: (n, 4) coefficient
        tensors, one per entry of the column table; ``cosets``: indices into the extended domain's cosets (default: the
        ``domain.min_cosets()`` that determine the quotient of a satisfied circuit).  -> (len(cosets) * n, 4): the first
        ``len(cosets)`` pieces, i.e. all of h when the circuit is satisfied.  ``on_cosets``: per column None or its values on
        those cosets already, (len(cosets), n, 4) as ``domain.coeff_to_cosets(col, cosets, internal=True)`` returns them (the fixed
        columns of a proving key: transformed once); the coefficient entry of such a column may be None."""
        import torch

        cosets, q, ref, keep, ptrs, pre_ptrs, dyn, shifts = self._quotient_args(domain, coeff_columns, cosets, challenges, beta, gamma, theta, y,
                                                                                on_cosets, "quotient_by_cosets")
        out = torch.empty((q * domain.n, 4), dtype=ref.dtype, device=ref.device)
        _lib.check(_lib.load().hm_quotient_by_cosets_bn256_fr_dev(
            ctypes.c_uint64(self.handle), ptrs, pre_ptrs, len(coeff_columns), _ptr(dyn), dyn.shape[0], domain.k,
            _ptr(fr_words(domain.omega)), _ptr(shifts), q, q,
            ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(_stream_ptr(out))))
        return out

    def quotient_partials(self, domain, coeff_columns: Sequence, cosets, challenges: Sequence[int] = (), beta: int = 0, gamma: int = 0,
                          theta: int = 0, y: int = 0, on_cosets: Sequence = None):
        """The first half of ``quotient_by_cosets`` on THIS device's cosets (``hm_quotient_partials_bn256_fr_dev``): ->
        (len(cosets), n, 4) partials.  ``quotient_combine`` turns the partials of all devices into h."""
        import torch

        cosets, q, ref, keep, ptrs, pre_ptrs, dyn, shifts = self._quotient_args(domain, coeff_columns, cosets, challenges, beta, gamma, theta, y,
                                                                                on_cosets, "quotient_partials")
        out = torch.empty((q, domain.n, 4), dtype=ref.dtype, device=ref.device)
        _lib.check(_lib.load().hm_quotient_partials_bn256_fr_dev(
            ctypes.c_uint64(self.handle), ptrs, pre_ptrs, len(coeff_columns), _ptr(dyn), dyn.shape[0], domain.k,
            _ptr(fr_words(domain.omega)), _ptr(shifts), q, ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(_stream_ptr(out))))
        return out

    def destroy(self) -> None:
        if self.handle:
            _lib.check(_lib.load().hm_graph_destroy(ctypes.c_uint64(self.handle)))
            self.handle = 0


# ---- the permutation and lookup arguments of evaluate_h as expression lists ----------------------------------------
# Upstream writes these terms as explicit loops over the rows (plonk/evaluation.rs::evaluate_h); each term enters the
# running value as value = value * y + term.  As expression trees over the same column table they go through the same
# device evaluator as the gates: add_custom_gates(gates + permutation_expressions(...) + lookup_expressions(...)).
def permutation_expressions(columns: Sequence[Expression], sigmas: Sequence[Expression], z, l0: Expression, l_last: Expression,
                            l_active: Expression, x_coset: Expression, chunk_len: int, delta: int, last_rotation: int) -> List[Expression]:
    """columns[j] / sigmas[j]: the j-th permuted column and its sigma polynomial on the extended domain; z[i](rot): the
    i-th grand-product polynomial at a rotation (one per chunk of chunk_len columns); x_coset: the column zeta * omega_ext^idx
    (upstream's running `beta_term` times ZETA); last_rotation = -(blinding_factors + 1)."""
    nsets = (len(columns) + chunk_len - 1) // chunk_len
    if nsets == 0:
        return []
    out = [(1 - z[0](0)) * l0,
           (z[nsets - 1](0) * z[nsets - 1](0) - z[nsets - 1](0)) * l_last]
    for i in range(1, nsets):
        out.append((z[i](0) - z[i - 1](last_rotation)) * l0)
    current_delta = BETA * x_coset                       # delta_start * beta_term = beta * zeta * omega_ext^idx
    for i in range(nsets):
        cols = columns[i * chunk_len:(i + 1) * chunk_len]
        sigs = sigmas[i * chunk_len:(i + 1) * chunk_len]
        left = z[i](1)
        for c, sg in zip(cols, sigs):
            left = left * (c + BETA * sg + GAMMA)
        right = z[i](0)
        for c in cols:
            right = right * (c + current_delta + GAMMA)
            current_delta = current_delta * delta
        out.append((left - right) * l_active)
    return out


def lookup_expressions(inputs: Sequence[Expression], tables: Sequence[Expression], z, permuted_input, permuted_table,
                       l0: Expression, l_last: Expression, l_active: Expression) -> List[Expression]:
    """One lookup argument: inputs / tables are its input and table expressions (compressed by Horner in theta), z(rot) its
    grand product, permuted_input(rot) / permuted_table(rot) the permuted columns A' and S'."""
    def compress(exprs):
        acc = Constant(0)
        for e in exprs:
            acc = acc * THETA + e
        return acc

    table_value = (compress(inputs) + BETA) * (compress(tables) + GAMMA)
    a, sp = permuted_input(0), permuted_table(0)
    a_minus_s = a - sp
    return [(1 - z(0)) * l0,
            (z(0) * z(0) - z(0)) * l_last,
            (z(1) * (a + BETA) * (sp + GAMMA) - z(0) * table_value) * l_active,
            a_minus_s * l0,
            a_minus_s * (a - permuted_input(-1)) * l_active]


def quotient_combine(domain, partials: Sequence, cosets: Sequence[int], pieces: int = None):
    """``hm_quotient_combine_bn256_fr_dev``: the partials of ALL the cosets used (one (n, 4) tensor per coset, in the order of
    ``cosets``, on this device) -> (pieces * n, 4) coefficients of h; the vanishing division rides on the matrix."""
    import torch

    cosets = [int(c) for c in cosets]
    parts = [p.contiguous() for p in partials]
    if len(parts) != len(cosets) or any(_tensor_rows(p, 4, "partial") != domain.n for p in parts):
        raise ValueError("quotient_combine: one (n, 4) partial per coset")
    pieces = len(cosets) if pieces is None else pieces
    out = torch.empty((pieces * domain.n, 4), dtype=parts[0].dtype, device=parts[0].device)
    ptrs = (ctypes.c_void_p * len(parts))(*[p.data_ptr() for p in parts])
    shifts = np.stack([fr_words(domain.coset_shift(c)) for c in cosets])
    _lib.check(_lib.load().hm_quotient_combine_bn256_fr_dev(ptrs, _ptr(shifts), len(parts), domain.k, pieces, ctypes.c_void_p(out.data_ptr()),
                                                           ctypes.c_void_p(_stream_ptr(out))))
    return out
