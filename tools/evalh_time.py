#!/usr/bin/env python3
"""Time the device GraphEvaluator on the WHOLE evaluate_h program of one of the reference's circuits (circuits.py), at the
extended-domain size of its configuration, every column in a buffer of its own: development aid.
    python tools/evalh_time.py [merkle_sum_tree_k18] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from halo2_experiments_amd import circuits
from halo2_experiments_amd.domain import FR_MODULUS

name = sys.argv[1] if len(sys.argv) > 1 else "merkle_sum_tree_k18"
k = int(sys.argv[2]) if len(sys.argv) > 2 else int(name.rsplit("k", 1)[1])
cs = circuits.CONSTRAINT_SYSTEMS[name]()
ek = k + 3
g, lay = circuits.evaluate_h_program(cs, k, ek, pow(7, 1 << 28, FR_MODULUS))
prog = g.compile(lay.num_fixed_entries, cs.num_advice, cs.num_instance, rot_scale=1 << (ek - k), short_columns=lay.short_columns)
n = 1 << ek
from halo2_experiments_amd.arithmetic import random_fr
_seeds = iter(range(1, 10 ** 6))
def rand_col(rows):
    return random_fr(rows, next(_seeds), "cuda")
ncols = lay.num_fixed_entries + cs.num_advice + cs.num_instance
cols = [rand_col(1 << lay.short_columns[i]) if i in lay.short_columns else rand_col(n) for i in range(ncols)]
values = rand_col(n)
ops = np.bincount(prog.calcs[:, 0], minlength=8)
print(f"{name}: k={k} rows 2^{ek}, {ncols} columns ({ncols * n * 32 / 2**30:.1f} GiB), {prog.calcs.shape[0]} calculations "
      f"(add {ops[0]} sub {ops[1]} mul {ops[2]} sq {ops[3]} dbl {ops[4]} neg {ops[5]} store {ops[6]} muladd {ops[7]})")
for internal in (False, True):
    for _ in range(2):
        prog.evaluate(cols, values, beta=3, gamma=5, theta=7, y=11, columns_internal=internal)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3):
        prog.evaluate(cols, values, beta=3, gamma=5, theta=7, y=11, columns_internal=internal)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    print(f"  columns {'internal' if internal else 'external'}: {dt * 1e3:.3f} ms = {n * prog.calcs.shape[0] / dt / 1e9:.1f} G calculation-rows/s")
prog.destroy()
