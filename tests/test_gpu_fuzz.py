"""Differential testing on random sketches of every size class (fused kernel, wide kernel, sparse path,
block walker) against the oracle: structure-level quantities exactly, outcomes within the tolerances the
LM tests use for arbitrary (ill-conditioned, partly infeasible) sketches."""
import numpy as np
import pytest

from helpers import random_big_sketch, random_sketch

pytestmark = pytest.mark.gpu


def _compare(b, res, res_o, v, v_o):
    assert np.array_equal(res["ncomp"], res_o["ncomp"])
    assert np.array_equal(res["scale"], res_o["scale"])
    fx = b["var_fixed"] == 1
    assert np.array_equal(v[fx], b["vars"][fx])
    ok = ~(np.isnan(res_o["sse"]) | np.isnan(res["sse"]))
    assert np.array_equal(np.isnan(res_o["sse0"]), np.isnan(res["sse0"]))
    same = (res["accepted"] == res_o["accepted"]) & (res["trials"] == res_o["trials"]) & ok
    d = np.abs(res["sse"] - res_o["sse"])
    assert np.all(d[same] <= 1e-9 + 0.25 * np.abs(res_o["sse"][same])), (d[same].max(), np.where(same)[0][np.argmax(d[same])])
    return float(same.mean()), float(np.mean((res["exit"] == 0) == (res_o["exit"] == 0)))


@pytest.mark.parametrize("n_points", [20, 30, 45, 62, 90])
def test_random_sketches_of_all_size_classes_none(fiksi, oracle, ctx, n_points):
    from fiksi_amd import workloads

    flats = [random_big_sketch(1000 * n_points + s, n_points).flatten() for s in range(12)]
    flats += [random_sketch(7 * n_points + s).flatten() for s in range(6)]  # small ones in the same batch
    b = workloads.concat(flats)
    v, res = ctx.system_solve_batch(b)
    v_o, res_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
    same, verdict = _compare(b, res, res_o, v, v_o)
    assert same > 0.6 and verdict > 0.85, (same, verdict)
    # the first starting SSE of every System is computed from bit-identical inputs
    first = np.isfinite(res_o["sse0"]) & (res_o["ncomp"] == 1)
    assert np.allclose(res["sse0"][first], res_o["sse0"][first], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("n_points", [25, 50, 80])
def test_random_sketches_of_all_size_classes_single_pass(fiksi, oracle, ctx, n_points):
    from fiksi_amd import abi, workloads

    flats = [random_big_sketch(500 * n_points + s, n_points).flatten() for s in range(10)]
    b = workloads.concat(flats)
    for s in range(len(flats)):
        mine = abi.single_pass_blocks(b, s)
        ncomp = int(max([c for c in flats[s]["var_comp"] if c != 0xFFFF], default=-1)) + 1
        ref = [(c, r, vv) for c in range(ncomp) for r, vv in oracle.single_pass_units(b, s, c)]
        assert mine == ref, s
    v, res = ctx.system_solve_batch(b, abi.solving_opts(decomposer=1))
    v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
    same, verdict = _compare(b, res, res_o, v, v_o)
    assert same > 0.5 and verdict > 0.8, (same, verdict)
