"""ORACLE (test infrastructure only -- never imported by the product): CPU restatement of
halo2_proofs::plonk::evaluation::{GraphEvaluator::evaluate, Calculation::evaluate, ValueSource::get} on Python
integers (upstream halo2_proofs/src/plonk/evaluation.rs at the tag pinned by /root/reference/Cargo.toml:10
[UPSTREAM-RECALLED]; parity unpinned: the reference holds no vectors for this path).

It interprets the calculations exactly as upstream stores them (Horner as one calculation over a list of parts),
i.e. BEFORE the product's lowering to MulAdd chains, row by row:

    get_rotation_idx(idx, rot, rot_scale, isize) = (idx + rot * rot_scale) mod isize
    Horner(start, parts, factor): value = start; for part in parts: value = value * factor + part
    result of the graph = the last calculation's intermediate (zero for an empty graph)
"""
from typing import Dict, List, Sequence

R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001


def _evaluate_row(calculations, constants, rotations, cell, challenges, beta, gamma, theta, y, previous_value, idx, rot_scale, isize) -> int:
    """One row of GraphEvaluator::evaluate.  cell(kind, column, row) -> the column's value at that row."""
    inter: Dict[int, int] = {}

    def get(vs):
        kind = vs[0]
        if kind == "Constant":
            return constants[vs[1]]
        if kind == "Intermediate":
            return inter[vs[1]]
        if kind in ("Fixed", "Advice", "Instance"):
            return cell(kind, vs[1], (idx + rotations[vs[2]] * rot_scale) % isize)
        if kind == "Challenge":
            return challenges[vs[1]]
        return {"Beta": beta, "Gamma": gamma, "Theta": theta, "Y": y, "PreviousValue": previous_value}[kind]

    last = 0
    for calc, target in calculations:
        name = calc[0]
        if name == "Add":
            v = (get(calc[1]) + get(calc[2])) % R
        elif name == "Sub":
            v = (get(calc[1]) - get(calc[2])) % R
        elif name == "Mul":
            v = get(calc[1]) * get(calc[2]) % R
        elif name == "Square":
            v = get(calc[1]) ** 2 % R
        elif name == "Double":
            v = 2 * get(calc[1]) % R
        elif name == "Negate":
            v = -get(calc[1]) % R
        elif name == "Store":
            v = get(calc[1])
        elif name == "Horner":
            v = get(calc[1])
            f = get(calc[3])
            for part in calc[2]:
                v = (v * f + get(part)) % R
        else:
            raise ValueError(name)
        inter[target] = v
        last = v
    return last


def evaluate_graph(calculations, constants: Sequence[int], rotations: Sequence[int], fixed: List[List[int]], advice: List[List[int]],
                   instance: List[List[int]], challenges: Sequence[int], beta: int, gamma: int, theta: int, y: int,
                   previous: Sequence[int], rot_scale: int, isize: int) -> List[int]:
    """calculations: [(calc tuple, target)] as built by the GraphEvaluator mirror; columns are lists of ints of length isize."""
    table = {"Fixed": fixed, "Advice": advice, "Instance": instance}
    cell = lambda kind, col, row: table[kind][col][row]
    return [_evaluate_row(calculations, constants, rotations, cell, challenges, beta, gamma, theta, y, previous[idx], idx, rot_scale, isize)
            for idx in range(isize)]


def evaluate_graph_rows(calculations, constants: Sequence[int], rotations: Sequence[int], cell, challenges: Sequence[int], beta: int,
                        gamma: int, theta: int, y: int, previous, rows: Sequence[int], rot_scale: int, isize: int) -> List[int]:
    """The same interpreter on the listed rows only (domains too large to walk in Python: 2^21 rows at k = 18): columns are read
    through cell(kind, column, row) -- a sparse window gathered around the rows, see cells_needed() -- and previous[idx] must
    exist for every listed row."""
    return [_evaluate_row(calculations, constants, rotations, cell, challenges, beta, gamma, theta, y, previous[idx], idx, rot_scale, isize)
            for idx in rows]


def cells_needed(rotations: Sequence[int], rows: Sequence[int], rot_scale: int, isize: int) -> List[int]:
    """Every row index the program can read for the listed rows: (idx + rotation * rot_scale) mod isize over all its rotations."""
    return sorted({(idx + r * rot_scale) % isize for idx in rows for r in list(rotations) + [0]})


def evaluate_expression(e, fixed, advice, instance, challenges, idx: int, rot_scale: int, isize: int) -> int:
    """Direct evaluation of an Expression tree (plonk/circuit.rs::Expression::evaluate) -- independent of the graph."""
    name = type(e).__name__
    if name == "Constant":
        return e.value % R
    if name in ("Fixed", "Advice", "Instance"):
        col = {"Fixed": fixed, "Advice": advice, "Instance": instance}[name][e.column]
        return col[(idx + e.rotation * rot_scale) % isize]
    if name == "Challenge":
        return challenges[e.index]
    if name == "ProofScalar":
        return challenges[{"Beta": "beta", "Gamma": "gamma", "Theta": "theta"}[e.name]]     # challenges may be a dict here
    if name == "Negated":
        return -evaluate_expression(e.a, fixed, advice, instance, challenges, idx, rot_scale, isize) % R
    if name == "Sum":
        return (evaluate_expression(e.a, fixed, advice, instance, challenges, idx, rot_scale, isize) +
                evaluate_expression(e.b, fixed, advice, instance, challenges, idx, rot_scale, isize)) % R
    if name == "Product":
        return (evaluate_expression(e.a, fixed, advice, instance, challenges, idx, rot_scale, isize) *
                evaluate_expression(e.b, fixed, advice, instance, challenges, idx, rot_scale, isize)) % R
    if name == "Scaled":
        return evaluate_expression(e.a, fixed, advice, instance, challenges, idx, rot_scale, isize) * e.factor % R
    raise TypeError(name)


def evaluate_h_permutation_and_lookups(values, y, beta, gamma, theta, isize, rot_scale, extended_omega, zeta, delta,
                                        perm_columns, perm_sigmas, perm_z, chunk_len, last_rotation, l0, l_last, l_active,
                                        lookups, t_inverse=None):
    """Restatement of the permutation and lookup loops of evaluate_h (upstream plonk/evaluation.rs [UPSTREAM-RECALLED]) on
    Python integers; columns are lists of isize integers.  lookups: [(input_values, table_values, z, a_perm, s_perm)] where
    input_values / table_values are lists of already-evaluated input / table expression columns (compressed here by theta).
    Ends with divide_by_vanishing_poly when t_inverse (the 2^(extended_k - k) inverse evaluations) is given."""
    values = list(values)
    rot = lambda idx, r: (idx + r * rot_scale) % isize
    nsets = len(perm_z)
    if nsets:
        for idx in range(isize):
            beta_term = pow(extended_omega, idx, R)
            v = values[idx]
            r_next, r_last = rot(idx, 1), rot(idx, last_rotation)
            v = (v * y + (1 - perm_z[0][idx]) * l0[idx]) % R
            zl = perm_z[-1][idx]
            v = (v * y + (zl * zl - zl) * l_last[idx]) % R
            for i in range(1, nsets):
                v = (v * y + (perm_z[i][idx] - perm_z[i - 1][r_last]) * l0[idx]) % R
            current_delta = beta * zeta % R * beta_term % R
            for i in range(nsets):
                cols = perm_columns[i * chunk_len:(i + 1) * chunk_len]
                sigs = perm_sigmas[i * chunk_len:(i + 1) * chunk_len]
                left = perm_z[i][r_next]
                for c, sg in zip(cols, sigs):
                    left = left * (c[idx] + beta * sg[idx] + gamma) % R
                right = perm_z[i][idx]
                for c in cols:
                    right = right * (c[idx] + current_delta + gamma) % R
                    current_delta = current_delta * delta % R
                v = (v * y + (left - right) * l_active[idx]) % R
            values[idx] = v
    for input_values, table_values, z, a_perm, s_perm in lookups:
        for idx in range(isize):
            ci = 0
            for col in input_values:
                ci = (ci * theta + col[idx]) % R
            ct = 0
            for col in table_values:
                ct = (ct * theta + col[idx]) % R
            table_value = (ci + beta) * (ct + gamma) % R
            r_next, r_prev = rot(idx, 1), rot(idx, -1)
            a_minus_s = (a_perm[idx] - s_perm[idx]) % R
            v = values[idx]
            v = (v * y + (1 - z[idx]) * l0[idx]) % R
            v = (v * y + (z[idx] * z[idx] - z[idx]) * l_last[idx]) % R
            v = (v * y + (z[r_next] * (a_perm[idx] + beta) % R * (s_perm[idx] + gamma) - z[idx] * table_value) * l_active[idx]) % R
            v = (v * y + a_minus_s * l0[idx]) % R
            v = (v * y + a_minus_s * (a_perm[idx] - a_perm[r_prev]) % R * l_active[idx]) % R
            values[idx] = v
    if t_inverse is not None:
        values = [v * t_inverse[i % len(t_inverse)] % R for i, v in enumerate(values)]
    return values
