"""Diagnostic: the cfg3 Systems whose normal-equation path leaves the oracle's (different accepted / trial counts):
how many, do they reach the same verdict, how far apart are their final SSEs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import fiksi_amd
from fiksi_amd import workloads
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
ctx = fiksi_amd.Context(0)
N = 100_000
b = workloads.ring16(N)
v, res = ctx.system_solve_batch(b)
pick = np.sort(np.random.default_rng(7).choice(N, size=n, replace=False))
sample = workloads.concat([workloads.shard(b, int(s), N) for s in pick])
v_o, res_o = O.solve_batch(sample, mode=3, nthreads=16)
got = res[pick]
same = (got["accepted"] == res_o["accepted"]) & (got["trials"] == res_o["trials"]) & (got["exit"] == res_o["exit"])
print("same path", same.mean(), "divergent", int((~same).sum()))
d = np.abs(got["sse"] - res_o["sse"])
print("same path: max d / tol", np.max(d[same] / (1e-10 + 1e-6 * np.abs(res_o["sse"][same]))))
vv = v.reshape(N, 32)[pick].ravel()
r = O.residuals_batch(sample, vv).reshape(n, -1)
r_o = O.residuals_batch(sample, v_o).reshape(n, -1)
sq, sq_o = (r * r).sum(1), (r_o * r_o).sum(1)
div = ~same
print("divergent: same verdict", np.mean((sq[div] < 1e-4) == (sq_o[div] < 1e-4)), "of", int(div.sum()))
rel = d[div] / np.maximum(np.abs(res_o["sse"][div]), 1e-300)
conv = res_o["sse"][div] < 1e-8
print("divergent converged (oracle sse < 1e-8):", int(conv.sum()), "gpu sse max among them", got["sse"][div][conv].max() if conv.any() else None)
print("divergent not converged:", int((~conv).sum()), "rel dSSE quantiles 50/90/99/max", np.quantile(rel[~conv], [0.5, 0.9, 0.99, 1.0]) if (~conv).any() else None)
print("accepted diffs", np.bincount(np.abs(got["accepted"][div].astype(int) - res_o["accepted"][div].astype(int)))[:12])
with O.atan2_mode("correctly_rounded"):
    v_q, res_q = O.solve_batch(sample, mode=3, nthreads=16)
from fiksi_amd import abi
sv, sr = ctx.system_solve_batch(sample, abi.solving_opts(solver=2))
print("QR step vs oracle (correctly rounded atan2): vars bit-identical", np.array_equal(sv.view(np.uint64), v_q.view(np.uint64)),
      "accepted", np.array_equal(sr["accepted"], res_q["accepted"]), "trials", np.array_equal(sr["trials"], res_q["trials"]),
      "sse", np.array_equal(sr["sse"], res_q["sse"]), "exit", np.array_equal(sr["exit"], res_q["exit"]))
