"""Wide differential sweep: FX_STEP_QR (None and SinglePass) against the oracle on many random small sketches — every
variable and counter must be the oracle's bits (correctly rounded atan2 on both sides). Prints mismatching seeds."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
from oracle import oracle
from helpers import random_sketch, mixed_sketch

seed0, count, chunk = int(sys.argv[1]), int(sys.argv[2]), 1000
ctx = fiksi_amd.Context(0)
bad = 0
t0 = time.time()
with oracle.atan2_mode("correctly_rounded"):
    for lo in range(seed0, seed0 + count, chunk):
        flats = [random_sketch(s).flatten() for s in range(lo, lo + chunk)]
        flats += [mixed_sketch(s, fix_some=bool(s & 1)).flatten() for s in range(lo, lo + chunk // 10)]
        b = workloads.concat(flats)
        for dec in (0, 1):
            v, r = ctx.system_solve_batch(b, abi.solving_opts(solver=2, decomposer=dec))
            if dec == 0:
                v_o, r_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=16)
            else:
                v_o, r_o = oracle.solve_single_pass_batch(b, trial_cap=4096, nthreads=16)
            for s in range(len(flats)):
                a0, a1 = int(b["var_off"][s]), int(b["var_off"][s + 1])
                # a component / block whose step turns non-finite: the reference would double lambda for ever; the oracle stops
                # it at its trial cap (4096 trials), the device at the first non-finite trial — both leave it at its start point
                capped = r_o["exit"][s] == 4 or r["exit"][s] >= 4 or int(r_o["trials"][s]) - int(r["trials"][s]) >= 4000
                same = np.array_equal(v[a0:a1].view(np.uint64), v_o[a0:a1].view(np.uint64)) and r["accepted"][s] == r_o["accepted"][s] \
                    and (capped or r["trials"][s] == r_o["trials"][s])
                if not same:
                    bad += 1
                    if bad <= 20:
                        print("MISMATCH seed-index", lo, s, "decomposer", dec, "gpu", r[s], "oracle", r_o[s], flush=True)
        print(f"seeds {lo}..{lo + chunk}: done, mismatches so far {bad}, {time.time() - t0:.0f} s", flush=True)
print("TOTAL mismatches", bad)
