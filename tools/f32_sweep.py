"""f32 compute (fx_lm_opts_default_f32) against the f64 oracle on many sketches: verdict agreement and the distribution of
the final sum of squared residuals (unscaled), cfg5-style (inconsistent ring16) and arbitrary random sketches."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import fiksi_amd
from fiksi_amd import abi, workloads
from oracle import oracle
from helpers import random_sketch

ctx = fiksi_amd.Context(0)
for name, b in (("ring16 inconsistent x 20000", workloads.ring16(20000, inconsistent=True)), ("ring16 x 20000", workloads.ring16(20000)),
                ("random sketches x 6000", workloads.concat([random_sketch(s).flatten() for s in range(30000, 36000)]))):
    v, r = ctx.system_solve_batch(b, abi.solving_opts(f32=True))
    v_o, r_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=16)
    res, res_o = oracle.residuals_batch(b, v), oracle.residuals_batch(b, v_o)
    n = len(r)
    sq = np.array([float((res[b["expr_off"][s]:b["expr_off"][s + 1]] ** 2).sum()) for s in range(n)])
    sq_o = np.array([float((res_o[b["expr_off"][s]:b["expr_off"][s + 1]] ** 2).sum()) for s in range(n)])
    ok = np.isfinite(sq) & np.isfinite(sq_o) & (r_o["trials"] < 4000) & (r["exit"] < 4)
    verdict = ((sq < 1e-4) == (sq_o < 1e-4))[ok].mean()
    rel = np.abs(sq - sq_o)[ok] / (1e-6 + sq_o[ok])
    print(f"{name}: {int(ok.sum())} Systems, same verdict {verdict:.4f}; |dSSE| / (1e-6 + SSE_f64): median {np.median(rel):.2e}, p90 {np.percentile(rel, 90):.2e}, "
          f"p99 {np.percentile(rel, 99):.2e}; f32 trials per System {r['trials'].mean():.2f} (f64 oracle {r_o['trials'].mean():.2f})", flush=True)
