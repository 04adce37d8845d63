#!/usr/bin/env python3
"""Time the device GraphEvaluator on Poseidon-like gate programs at prover sizes: development aid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from halo2_experiments_amd import evaluation as ev
from test_evaluation import poseidon_like_gates

def rand_col(n, gen):
    x = torch.randint(-(2 ** 63), 2 ** 63 - 1, (n, 4), dtype=torch.int64, device="cuda", generator=gen)
    x[:, 3] &= 0x0FFFFFFFFFFFFFFF
    return x

for width, ek in ((3, 14), (3, 20), (3, 21), (5, 21), (8, 21)):
    polys, nf, na, ni = poseidon_like_gates(width)
    g = ev.GraphEvaluator(); g.add_custom_gates(polys)
    prog = g.compile(nf, na, ni, num_challenges=1, rot_scale=8)
    n = 1 << ek
    gen = torch.Generator(device="cuda"); gen.manual_seed(1)
    cols = [rand_col(n, gen) for _ in range(nf + na + ni)]
    values = rand_col(n, gen)
    prog.evaluate(cols, values, challenges=[5], y=7); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(3): prog.evaluate(cols, values, challenges=[5], y=7)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 3
    ninst = prog.calcs.shape[0]
    nmul = int(((prog.calcs[:, 0] == 2) | (prog.calcs[:, 0] == 3) | (prog.calcs[:, 0] == 7)).sum())
    print(f"width {width} rows 2^{ek}: {ninst} instructions ({nmul} products), {len(cols)} columns: {dt*1e3:.3f} ms = {n*ninst/dt/1e9:.2f} G instr-rows/s")
    prog.destroy()
