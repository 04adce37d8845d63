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


def evaluate_graph(calculations, constants: Sequence[int], rotations: Sequence[int], fixed: List[List[int]], advice: List[List[int]],
                   instance: List[List[int]], challenges: Sequence[int], beta: int, gamma: int, theta: int, y: int,
                   previous: Sequence[int], rot_scale: int, isize: int) -> List[int]:
    """calculations: [(calc tuple, target)] as built by the GraphEvaluator mirror; columns are lists of ints of length isize."""
    out = []
    for idx in range(isize):
        inter: Dict[int, int] = {}

        def get(vs):
            kind = vs[0]
            if kind == "Constant":
                return constants[vs[1]]
            if kind == "Intermediate":
                return inter[vs[1]]
            if kind in ("Fixed", "Advice", "Instance"):
                col = {"Fixed": fixed, "Advice": advice, "Instance": instance}[kind][vs[1]]
                return col[(idx + rotations[vs[2]] * rot_scale) % isize]
            if kind == "Challenge":
                return challenges[vs[1]]
            return {"Beta": beta, "Gamma": gamma, "Theta": theta, "Y": y, "PreviousValue": previous[idx]}[kind]

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
        out.append(last)
    return out


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
