"""Synthetic batches for the BASELINE.json configurations, as flat fx_batch arrays.

All randomness comes from the reference's 32-bit LCG (fiksi/src/rand.rs:24-39), vectorised over
systems: seed = ``seed0 + system index``, so every generator is reproducible bit for bit.

* ``ring16``            cfg3/cfg4: 16 points on a jittered circle, 16 ring + 8 chord distances +
                        8 three-point angles (32 variables, 32 expressions, 144 Jacobian non-zeros);
                        ``inconsistent=True`` is cfg5's over-constrained variant.
* ``hinged_triangles``  the reference's own bench generator (fiksi/benches/fiksi_bench.rs:15-40).
* ``quadrilateral``     cfg1: the four points / six distances of fiksi/src/tests/basic.rs:95-105.
"""
from __future__ import annotations

from typing import Dict

import numpy as np

from . import abi


class LcgVec:
    """``Rng`` of fiksi/src/rand.rs, one independent stream per system."""

    def __init__(self, seeds):
        self.state = np.asarray(seeds, dtype=np.uint64) & np.uint64(0xFFFFFFFF)

    def next_u32(self) -> np.ndarray:
        self.state = (self.state * np.uint64(1664525) + np.uint64(1013904223)) & np.uint64(0xFFFFFFFF)
        return self.state.astype(np.uint32)

    def next_f64(self) -> np.ndarray:
        return (1.0 / 4294967295.0) * self.next_u32().astype(np.float64)


def _wrap(a):
    a = np.where(a > np.pi, a - 2.0 * np.pi, a)
    return np.where(a < -np.pi, a + 2.0 * np.pi, a)


def ring16(n_systems: int, seed0: int = 1000, inconsistent: bool = False, fix_gauge: bool = False) -> Dict[str, np.ndarray]:
    """cfg3: n independent 32-constraint sketches. Draw order per system: R, cx, cy, then
    (angle jitter, radius jitter) per point, then one noise draw per coordinate, then (only if
    ``inconsistent``) one draw per distance target.

    ``fix_gauge`` fixes points 0 and 1 (fiksi fixes whole elements) at their ground-truth
    positions, which removes the rigid-motion null space (solved positions become directly
    comparable) while keeping the targets consistent.
    """
    n = int(n_systems)
    P = 16
    rng = LcgVec(seed0 + np.arange(n, dtype=np.uint64))
    R = 5.0 + 10.0 * rng.next_f64()
    cx = 20.0 * (rng.next_f64() - 0.5)
    cy = 20.0 * (rng.next_f64() - 0.5)
    truth = np.zeros((n, P, 2))
    for i in range(P):
        th = 2.0 * np.pi * (i + 0.3 * (rng.next_f64() - 0.5)) / P
        ri = R * (1.0 + 0.1 * (rng.next_f64() - 0.5))
        truth[:, i, 0] = cx + ri * np.cos(th)
        truth[:, i, 1] = cy + ri * np.sin(th)
    start = truth.copy()
    for i in range(P):
        for c in range(2):
            start[:, i, c] += 0.05 * R * (2.0 * rng.next_f64() - 1.0)

    m = 32
    tag = np.zeros((n, m), dtype=np.uint8)
    idx = np.zeros((n, m, 4), dtype=np.uint32)
    par = np.zeros((n, m))

    def dist(a, b):
        d = truth[:, a, :] - truth[:, b, :]
        return np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])

    row = 0
    for i in range(P):  # ring distances (i, i+1 mod 16)
        j = (i + 1) % P
        tag[:, row] = abi.POINT_POINT_DISTANCE
        idx[:, row, 0], idx[:, row, 1] = 2 * i, 2 * j
        par[:, row] = dist(i, j)
        row += 1
    for k in range(8):  # chords (2k, 2k+2)
        i, j = 2 * k, (2 * k + 2) % P
        tag[:, row] = abi.POINT_POINT_DISTANCE
        idx[:, row, 0], idx[:, row, 1] = 2 * i, 2 * j
        par[:, row] = dist(i, j)
        row += 1
    for k in range(8):  # angles at 2k+1 between (2k) and (2k+2)
        a, b, c = 2 * k, 2 * k + 1, (2 * k + 2) % P
        tag[:, row] = abi.POINT_POINT_POINT_ANGLE
        idx[:, row, 0], idx[:, row, 1], idx[:, row, 2] = 2 * a, 2 * b, 2 * c
        u = truth[:, a, :] - truth[:, b, :]
        v = truth[:, c, :] - truth[:, b, :]
        par[:, row] = _wrap(np.arctan2(v[:, 1], v[:, 0]) - np.arctan2(u[:, 1], u[:, 0]))
        row += 1
    if inconsistent:  # cfg5: every distance target off by up to 2 %
        for r in range(24):
            par[:, r] *= 1.0 + 0.02 * (2.0 * rng.next_f64() - 1.0)

    var_fixed = np.zeros((n, 2 * P), dtype=np.uint8)
    if fix_gauge:
        var_fixed[:, 0:4] = 1
        start[:, 0:2, :] = truth[:, 0:2, :]
    return {
        "var_off": (np.arange(n + 1, dtype=np.uint64) * (2 * P)).astype(np.uint32),
        "expr_off": (np.arange(n + 1, dtype=np.uint64) * m).astype(np.uint32),
        "vars": start.reshape(-1).copy(),
        "var_fixed": var_fixed.reshape(-1),
        "expr_tag": tag.reshape(-1),
        "expr_idx": idx.reshape(-1),
        "expr_param": par.reshape(-1),
        "var_comp": np.zeros(n * 2 * P, dtype=np.uint16),
        "expr_comp": np.zeros(n * m, dtype=np.uint16),
    }


def ring16_overconstrained(n_systems: int, seed0: int = 1000) -> Dict[str, np.ndarray]:
    """cfg5's theme taken literally — least squares over MORE constraints than unknowns: the inconsistent ring16 sketch plus 16
    distances (i, i + 3) whose targets are the start configuration's own distances: 48 expressions on 32 variables."""
    b = ring16(n_systems, seed0=seed0, inconsistent=True)
    n, P, m0, m = int(n_systems), 16, 32, 48
    pos = b["vars"].reshape(n, P, 2)
    tag = np.zeros((n, m), dtype=np.uint8)
    idx = np.zeros((n, m, 4), dtype=np.uint32)
    par = np.zeros((n, m))
    tag[:, :m0] = b["expr_tag"].reshape(n, m0)
    idx[:, :m0] = b["expr_idx"].reshape(n, m0, 4)
    par[:, :m0] = b["expr_param"].reshape(n, m0)
    for i in range(P):
        j = (i + 3) % P
        tag[:, m0 + i] = abi.POINT_POINT_DISTANCE
        idx[:, m0 + i, 0], idx[:, m0 + i, 1] = 2 * i, 2 * j
        d = pos[:, i, :] - pos[:, j, :]
        par[:, m0 + i] = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
    out = dict(b)
    out["expr_off"] = (np.arange(n + 1, dtype=np.uint64) * m).astype(np.uint32)
    out["expr_tag"] = tag.reshape(-1)
    out["expr_idx"] = idx.reshape(-1)
    out["expr_param"] = par.reshape(-1)
    out["expr_comp"] = np.zeros(n * m, dtype=np.uint16)
    return out


def ring_chords(n_systems: int, n_points: int = 20, step: int = 7, seed0: int = 5000) -> Dict[str, np.ndarray]:
    """n independent sketches of `n_points` points on a jittered circle with the distances (i, i + 1) and (i, i + step): a
    structure whose normal matrix fills in when factored (the long chords couple everything), unlike the banded ring16 or the
    arrow-shaped hinged chains. Consistent targets. 2 n_points variables, 2 n_points expressions."""
    n, P = int(n_systems), int(n_points)
    rng = LcgVec(seed0 + np.arange(n, dtype=np.uint64))
    R = 5.0 + 10.0 * rng.next_f64()
    truth = np.zeros((n, P, 2))
    for i in range(P):
        th = 2.0 * np.pi * (i + 0.3 * (rng.next_f64() - 0.5)) / P
        ri = R * (1.0 + 0.1 * (rng.next_f64() - 0.5))
        truth[:, i, 0] = ri * np.cos(th)
        truth[:, i, 1] = ri * np.sin(th)
    start = truth.copy()
    for i in range(P):
        for c in range(2):
            start[:, i, c] += 0.03 * R * (2.0 * rng.next_f64() - 1.0)
    m = 2 * P
    tag = np.full((n, m), abi.POINT_POINT_DISTANCE, dtype=np.uint8)
    idx = np.zeros((n, m, 4), dtype=np.uint32)
    par = np.zeros((n, m))
    row = 0
    for hop in (1, step):
        for i in range(P):
            j = (i + hop) % P
            idx[:, row, 0], idx[:, row, 1] = 2 * i, 2 * j
            d = truth[:, i, :] - truth[:, j, :]
            par[:, row] = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
            row += 1
    return {
        "var_off": (np.arange(n + 1, dtype=np.uint64) * (2 * P)).astype(np.uint32),
        "expr_off": (np.arange(n + 1, dtype=np.uint64) * m).astype(np.uint32),
        "vars": start.reshape(-1).copy(),
        "var_fixed": np.zeros(n * 2 * P, dtype=np.uint8),
        "expr_tag": tag.reshape(-1),
        "expr_idx": idx.reshape(-1),
        "expr_param": par.reshape(-1),
        "var_comp": np.zeros(n * 2 * P, dtype=np.uint16),
        "expr_comp": np.zeros(n * m, dtype=np.uint16),
    }


RING16_NNZ = 24 * 4 + 8 * 6  # 144
# algorithmic bytes of one Jacobian-assembly evaluation of one ring16 system:
# 8*nv (x) + 28*m (tag 4 + idx 16 + param 8) + 8*m (r) + 8*nnz (J values)
RING16_K1_BYTES = 8 * 32 + 28 * 32 + 8 * 32 + 8 * RING16_NNZ  # 2560


def k1_algorithmic_bytes(batch: Dict[str, np.ndarray], nnz: int, one_structure: bool = False) -> int:
    """Bytes one Jacobian-assembly pass (K1) has to move. SURVEY.md §8d's formula for any batch:
    8*n_vars + 28*n_exprs (tag 4 + fields 16 + parameter 8) + 8*n_exprs + 8*nnz — 2560 B per ring16 System.
    ``one_structure``: a batch whose Systems all share one structure (one sketch, many parameter sets) reads kinds and
    fields from its first System only (fx_eval.hip), so per row only the 8-byte parameter is left of the 28:
    8*n_vars + 8*n_exprs + 8*n_exprs + 8*nnz — 1920 B per ring16 System. That is the figure the counters confirm
    (1936 B per System, profiles/round3_pmc_traffic_500k.json) and the one a roofline fraction may be quoted on."""
    nv = int(batch["var_off"][-1])
    ne = int(batch["expr_off"][-1])
    per_row = 8 if one_structure else 28
    return 8 * nv + per_row * ne + 8 * ne + 8 * int(nnz)


def ring16_two_structures(n_systems: int, seed0: int = 1000) -> Dict[str, np.ndarray]:
    """cfg3 sketches of two structures, interleaved: every odd System lists its first ring distance and its last angle in
    each other's place (rows 0 and 31 swapped, parameters with them). Same geometry, same solution, same Jacobian
    non-zeros — but no longer ONE structure: the batch takes the paths that stream every System's structure."""
    b = ring16(n_systems, seed0=seed0)
    n, m = int(n_systems), 32
    tag = b["expr_tag"].reshape(n, m)
    idx = b["expr_idx"].reshape(n, m, 4)
    par = b["expr_param"].reshape(n, m)
    odd = np.arange(n) % 2 == 1
    for arr in (tag, idx, par):
        tmp = arr[odd, 0].copy()
        arr[odd, 0] = arr[odd, 31]
        arr[odd, 31] = tmp
    return b


def ring16_all_different(n_systems: int, seed0: int = 1000) -> Dict[str, np.ndarray]:
    """The headline's sketches with every System's STRUCTURE its own: the 16 ring distances stay, the eight chords join point 2k to a
    point drawn per System (never itself or a ring neighbour), and the eight angles sit at points drawn per System. 32 constraints on
    32 variables as in ``ring16``, consistent targets — but no two Systems share their index arrays, so no one-structure program, no
    structure class: what the general build of the grouped kernel makes of a batch of unrelated sketches."""
    b = ring16(n_systems, seed0)
    n, P, m = int(n_systems), 16, 32
    rng = LcgVec(seed0 + 77777 + np.arange(n, dtype=np.uint64))
    truth_rng = LcgVec(seed0 + np.arange(n, dtype=np.uint64))  # the same draws as ring16: the ground truth the targets come from
    R = 5.0 + 10.0 * truth_rng.next_f64()
    cx = 20.0 * (truth_rng.next_f64() - 0.5)
    cy = 20.0 * (truth_rng.next_f64() - 0.5)
    truth = np.zeros((n, P, 2))
    for i in range(P):
        th = 2.0 * np.pi * (i + 0.3 * (truth_rng.next_f64() - 0.5)) / P
        ri = R * (1.0 + 0.1 * (truth_rng.next_f64() - 0.5))
        truth[:, i, 0] = cx + ri * np.cos(th)
        truth[:, i, 1] = cy + ri * np.sin(th)
    idx = b["expr_idx"].reshape(n, m, 4).copy()
    par = b["expr_param"].reshape(n, m).copy()
    rows = np.arange(n)
    for k in range(8):  # chord k: point 2k to 2k + 2 + (0 ... 11): anything but itself and its two neighbours
        i = 2 * k
        j = (i + 2 + np.minimum((rng.next_f64() * 12.0).astype(np.int64), 11)) % P
        d = truth[rows, i, :] - truth[rows, j, :]
        idx[:, 16 + k, 0], idx[:, 16 + k, 1] = 2 * i, (2 * j).astype(np.uint32)
        par[:, 16 + k] = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])
    for k in range(8):  # angle k: at a drawn point between its ring neighbours
        bb = np.minimum((rng.next_f64() * 16.0).astype(np.int64), 15)
        a, c = (bb + P - 1) % P, (bb + 1) % P
        idx[:, 24 + k, 0], idx[:, 24 + k, 1], idx[:, 24 + k, 2] = (2 * a).astype(np.uint32), (2 * bb).astype(np.uint32), (2 * c).astype(np.uint32)
        u = truth[rows, a, :] - truth[rows, bb, :]
        v = truth[rows, c, :] - truth[rows, bb, :]
        par[:, 24 + k] = _wrap(np.arctan2(v[:, 1], v[:, 0]) - np.arctan2(u[:, 1], u[:, 0]))
    b["expr_idx"] = idx.reshape(-1)
    b["expr_param"] = par.reshape(-1)
    return b


def hinged_triangles(n_systems: int, n_triangles: int = 11) -> Dict[str, np.ndarray]:
    """fiksi/benches/fiksi_bench.rs:15-40: hinge (0,0); triangle t adds p1=(-1,t), p2=(1,t) and
    distances hinge-p1 = 2, hinge-p2 = 2, p1-p2 = 3. Every system of the batch is identical (the
    reference generator has no randomness)."""
    n, T = int(n_systems), int(n_triangles)
    nv, m = 2 + 4 * T, 3 * T
    v = np.zeros(nv)
    tag = np.full(m, abi.POINT_POINT_DISTANCE, dtype=np.uint8)
    idx = np.zeros((m, 4), dtype=np.uint32)
    par = np.zeros(m)
    for t in range(T):
        p1, p2 = 2 + 4 * t, 4 + 4 * t
        v[p1:p1 + 2] = (-1.0, float(t))
        v[p2:p2 + 2] = (1.0, float(t))
        idx[3 * t + 0, :2] = (0, p1)
        idx[3 * t + 1, :2] = (0, p2)
        idx[3 * t + 2, :2] = (p1, p2)
        par[3 * t: 3 * t + 3] = (2.0, 2.0, 3.0)
    return {
        "var_off": (np.arange(n + 1, dtype=np.uint64) * nv).astype(np.uint32),
        "expr_off": (np.arange(n + 1, dtype=np.uint64) * m).astype(np.uint32),
        "vars": np.tile(v, n),
        "var_fixed": np.zeros(n * nv, dtype=np.uint8),
        "expr_tag": np.tile(tag, n),
        "expr_idx": np.tile(idx.reshape(-1), n),
        "expr_param": np.tile(par, n),
        "var_comp": np.zeros(n * nv, dtype=np.uint16),
        "expr_comp": np.zeros(n * m, dtype=np.uint16),
    }


def quadrilateral(consistent: bool = True) -> Dict[str, np.ndarray]:
    """cfg1: four points, six pairwise distances (fiksi/src/tests/basic.rs:95-105). ``consistent``
    uses the unit-square targets 1,1,1,1,sqrt2,sqrt2 on pairs (01,02,13,23,12,03); otherwise the
    test's geometrically impossible targets 1,1.5,1.7,1.2,2,5."""
    v = np.array([0.123, 0.1, 1.2, 0.0, -0.5, 1.1, 1.599, 1.2])
    pairs = [(0, 1), (0, 2), (1, 3), (2, 3), (1, 2), (0, 3)]
    targets = [1.0, 1.0, 1.0, 1.0, 2.0 ** 0.5, 2.0 ** 0.5] if consistent else [1.0, 1.5, 1.7, 1.2, 2.0, 5.0]
    idx = np.zeros((6, 4), dtype=np.uint32)
    for r, (a, b) in enumerate(pairs):
        idx[r, :2] = (2 * a, 2 * b)
    return {
        "var_off": np.array([0, 8], dtype=np.uint32),
        "expr_off": np.array([0, 6], dtype=np.uint32),
        "vars": v,
        "var_fixed": np.zeros(8, dtype=np.uint8),
        "expr_tag": np.full(6, abi.POINT_POINT_DISTANCE, dtype=np.uint8),
        "expr_idx": idx.reshape(-1),
        "expr_param": np.array(targets),
        "var_comp": np.zeros(8, dtype=np.uint16),
        "expr_comp": np.zeros(6, dtype=np.uint16),
    }


def concat(batches) -> Dict[str, np.ndarray]:
    """Concatenate flat batches into one."""
    out = {k: [] for k in ("vars", "var_fixed", "expr_tag", "expr_idx", "expr_param", "var_comp", "expr_comp")}
    var_off, expr_off = [np.zeros(1, dtype=np.uint64)], [np.zeros(1, dtype=np.uint64)]
    v0 = e0 = 0
    for b in batches:
        for k in out:
            out[k].append(np.asarray(b[k]))
        var_off.append(np.asarray(b["var_off"][1:], dtype=np.uint64) + v0)
        expr_off.append(np.asarray(b["expr_off"][1:], dtype=np.uint64) + e0)
        v0 += int(b["var_off"][-1])
        e0 += int(b["expr_off"][-1])
    res = {k: np.concatenate(v) for k, v in out.items()}
    res["var_off"] = np.concatenate(var_off).astype(np.uint32)
    res["expr_off"] = np.concatenate(expr_off).astype(np.uint32)
    return res


def shard(batch: Dict[str, np.ndarray], rank: int, world: int) -> Dict[str, np.ndarray]:
    """Contiguous shard ``[rank*N/world, (rank+1)*N/world)`` of the systems (cfg4)."""
    n = len(batch["var_off"]) - 1
    lo, hi = (n * rank) // world, (n * (rank + 1)) // world
    v0, v1 = int(batch["var_off"][lo]), int(batch["var_off"][hi])
    e0, e1 = int(batch["expr_off"][lo]), int(batch["expr_off"][hi])
    out = {
        "var_off": (batch["var_off"][lo:hi + 1].astype(np.int64) - v0).astype(np.uint32),
        "expr_off": (batch["expr_off"][lo:hi + 1].astype(np.int64) - e0).astype(np.uint32),
        "expr_idx": batch["expr_idx"][4 * e0:4 * e1].copy(),
    }
    for k in ("vars", "var_fixed", "var_comp"):
        out[k] = batch[k][v0:v1].copy()
    for k in ("expr_tag", "expr_param", "expr_comp"):
        out[k] = batch[k][e0:e1].copy()
    return out


def large_sketch(n_points: int = 5000, seed: int = 7, noise: float = 0.01) -> Dict[str, np.ndarray]:
    """cfg2: ONE large sketch. ``n_points`` ground-truth points uniform in [0,100]^2;
    n-1 chain distances (i,i+1); skip distances (i,i+2) for the first 0.4*n+1 even i; 0.6*n
    three-point angles (i-1,i,i+1), odd i first, then even i; targets from the ground truth; start =
    truth + noise*100*U(-1,1). For n = 5000: 10 000 variables, 10 000 expressions (4999 + 2001
    distances, 3000 angles), 46 000 Jacobian non-zeros."""
    n = int(n_points)
    rng = LcgVec(np.array([seed], dtype=np.uint64))

    def draws(k):
        return np.array([rng.next_f64()[0] for _ in range(k)])

    truth = 100.0 * draws(2 * n).reshape(n, 2)
    start = truth + noise * 100.0 * (2.0 * draws(2 * n).reshape(n, 2) - 1.0)
    n_skip = (2 * n) // 5 + 1
    n_ang = (3 * n) // 5
    tags, idx, par = [], [], []

    def dist(a, b):
        d = truth[a] - truth[b]
        return float(np.sqrt(d[0] * d[0] + d[1] * d[1]))

    for i in range(n - 1):
        tags.append(abi.POINT_POINT_DISTANCE); idx.append((2 * i, 2 * (i + 1), 0, 0)); par.append(dist(i, i + 1))
    for i in list(range(0, n - 2, 2))[:n_skip]:
        tags.append(abi.POINT_POINT_DISTANCE); idx.append((2 * i, 2 * (i + 2), 0, 0)); par.append(dist(i, i + 2))
    for i in (list(range(1, n - 1, 2)) + list(range(2, n - 1, 2)))[:n_ang]:
        u, v = truth[i - 1] - truth[i], truth[i + 1] - truth[i]
        ang = float(_wrap(np.arctan2(v[1], v[0]) - np.arctan2(u[1], u[0])))
        tags.append(abi.POINT_POINT_POINT_ANGLE); idx.append((2 * (i - 1), 2 * i, 2 * (i + 1), 0)); par.append(ang)
    m = len(tags)
    return {
        "var_off": np.array([0, 2 * n], dtype=np.uint32),
        "expr_off": np.array([0, m], dtype=np.uint32),
        "vars": start.reshape(-1).copy(),
        "var_fixed": np.zeros(2 * n, dtype=np.uint8),
        "expr_tag": np.array(tags, dtype=np.uint8),
        "expr_idx": np.array(idx, dtype=np.uint32).reshape(-1),
        "expr_param": np.array(par, dtype=np.float64),
        "var_comp": np.zeros(2 * n, dtype=np.uint16),
        "expr_comp": np.zeros(m, dtype=np.uint16),
    }
