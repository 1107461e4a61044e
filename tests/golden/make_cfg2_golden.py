#!/usr/bin/env python3
"""Solve BASELINE cfg2 (one sketch of 5 000 points / 10 000 constraints, fiksi_amd.workloads.large_sketch)
with the CPU ORACLE (~85 s: the reference algorithm clears an (m+n)-vector per column,
solvi qr.rs:287) and store the outcome as a fixture for the GPU parity test. Oracle output, not
reference output (the reference is Rust and cannot be built in this image)."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from fiksi_amd import workloads  # noqa: E402
from oracle import oracle as O  # noqa: E402

b = workloads.large_sketch(5000)
t = time.time()
v, res = O.solve_batch(b, mode=3)
dt = time.time() - t
r = O.residuals_batch(b, v)
out = {
    "workload": "fiksi_amd.workloads.large_sketch(5000, seed=7, noise=0.01)",
    "oracle_seconds": dt,
    "accepted": int(res["accepted"][0]), "trials": int(res["trials"][0]), "exit": int(res["exit"][0]),
    "scale": float(res["scale"][0]), "sse0": float(res["sse0"][0]), "sse": float(res["sse"][0]),
    "sse_unscaled": float((r * r).sum()),
    "vars_every_97th": [float(x) for x in v[::97]],
}
json.dump(out, open(os.path.join(os.path.dirname(__file__), "cfg2_oracle.json"), "w"), indent=1)
print(out["accepted"], out["trials"], out["exit"], out["sse"], dt)
