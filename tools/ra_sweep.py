"""Wide differential sweep of Decomposer::RecursiveAssembly (device arm vs oracle) on random sketches."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fiksi_amd as F
from fiksi_amd._lib import FiksiError
from oracle import oracle
from helpers import random_sketch

seed0, count = int(sys.argv[1]), int(sys.argv[2])
ctx = F.Context(0)
stats = dict(planned=0, panic=0, exhausted=0, same_path=0, same_verdict=0, worst=0.0, errors=0)
t0 = time.time()
for seed in range(seed0, seed0 + count):
    s = random_sketch(seed)
    words, flags = s.recursive_plan(20000)
    g = s.graph()
    v_o, plan, steps, fl = oracle.solve_recursive(g, trial_cap=4096, budget=20000)
    assert fl == flags and np.array_equal(words, plan), seed
    if flags:
        stats["panic" if flags & 1 else "exhausted"] += 1
        continue
    stats["planned"] += 1
    try:
        s.solve(F.SolvingOptions(decomposer=F.Decomposer.RecursiveAssembly), ctx, solver=2)
    except FiksiError as e:
        stats["errors"] += 1
        print("ERROR seed", seed, e, flush=True)
        continue
    v, res = s.flatten()["vars"], s.last_result
    sq = float(np.sum(oracle.residuals_batch(dict(g, vars=v)) ** 2))
    sq_o = float(np.sum(oracle.residuals_batch(dict(g, vars=v_o)) ** 2))
    if res["ncomp"] != len(steps):
        print("STEP COUNT seed", seed, res["ncomp"], len(steps), flush=True)
    if res["accepted"] == int(steps["accepted"].sum()) and res["trials"] == int(steps["trials"].sum()):
        stats["same_path"] += 1
        if np.all(np.isfinite(v_o)):
            d = abs(sq - sq_o) / (1e-10 + 1e-6 * sq_o)
            stats["worst"] = max(stats["worst"], d)
            if d > 1: print("SSE seed", seed, sq, sq_o, flush=True)
    if (sq < 1e-4) == (sq_o < 1e-4):
        stats["same_verdict"] += 1
    else:
        print("VERDICT seed", seed, sq, sq_o, res["accepted"], int(steps["accepted"].sum()), res["trials"], int(steps["trials"].sum()), flush=True)
print(stats, f"{time.time() - t0:.0f} s")
