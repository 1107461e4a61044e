"""Differential testing on random sketches of every size class (fused kernel, wide kernel, sparse path,
block walker) against the oracle. No System is dropped for taking a different path than the oracle's
(tests/helpers.py: compare_outcomes checks every one):
  * FX_STEP_QR (reference numerics; Systems beyond one wavefront run the refined normal-equation step): SURVEY
    8c's tolerance — same path on >= 95 %, |SSE_gpu - SSE_ref| <= 1e-10 + 1e-6 SSE on the same path, the same
    verdict and a bounded SSE difference on the others;
  * the default normal-equation step (cond^2): structure-level quantities exactly, same-path SSE within 25 %, and
    the fraction of equal verdicts stated per test."""
import numpy as np
import pytest

from helpers import compare_outcomes, random_big_sketch, random_sketch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n_points", [20, 30, 45, 62, 90])
@pytest.mark.parametrize("solver", [0, 2])
def test_random_sketches_of_all_size_classes_none(fiksi, oracle, ctx, n_points, solver):
    from fiksi_amd import abi, workloads

    flats = [random_big_sketch(1000 * n_points + s, n_points).flatten() for s in range(12)]
    flats += [random_sketch(7 * n_points + s).flatten() for s in range(6)]  # small ones in the same batch
    b = workloads.concat(flats)
    one_wave = max(int(f["var_off"][-1]) for f in flats) <= 64 and solver == 2
    with oracle.atan2_mode("correctly_rounded" if solver == 2 else "libm"):
        v, res = ctx.system_solve_batch(b, abi.solving_opts(solver=solver))
        v_o, res_o = oracle.solve_batch(b, mode=3, trial_cap=4096, nthreads=8)
        same, verdict = compare_outcomes(b, v, res, v_o, res_o, oracle, tight=one_wave)
    if one_wave:  # every System on the QR kernel: the oracle's bits
        assert same == 1.0 and verdict == 1.0, (same, verdict)
        assert np.array_equal(v.view(np.uint64), v_o.view(np.uint64))
    elif solver == 2:  # the large Systems of the batch take the refined step
        assert same > 0.8 and verdict > 0.9, (same, verdict)
    else:
        assert same > 0.6 and verdict > 0.85, (same, verdict)
    # the first starting SSE of every System is computed from bit-identical inputs
    first = np.isfinite(res_o["sse0"]) & (res_o["ncomp"] == 1)
    assert np.allclose(res["sse0"][first], res_o["sse0"][first], rtol=1e-9, atol=1e-300)


@pytest.mark.parametrize("n_points", [25, 50, 80])
@pytest.mark.parametrize("solver", [0, 2])
def test_random_sketches_of_all_size_classes_single_pass(fiksi, oracle, ctx, n_points, solver):
    from fiksi_amd import abi, workloads

    flats = [random_big_sketch(500 * n_points + s, n_points).flatten() for s in range(10)]
    b = workloads.concat(flats)
    for s in range(len(flats)):
        mine = abi.single_pass_blocks(b, s)
        ncomp = int(max([c for c in flats[s]["var_comp"] if c != 0xFFFF], default=-1)) + 1
        ref = [(c, r, vv) for c in range(ncomp) for r, vv in oracle.single_pass_units(b, s, c)]
        assert mine == ref, s
    with oracle.atan2_mode("correctly_rounded" if solver == 2 else "libm"):
        v, res = ctx.system_solve_batch(b, abi.solving_opts(decomposer=1, solver=solver))
        v_o, res_o = oracle.solve_single_pass_batch(b, trial_cap=4096, nthreads=8)
        same, verdict = compare_outcomes(b, v, res, v_o, res_o, oracle, tight=False)
    if solver == 2 and n_points == 25:  # blocks of Systems within one wavefront: the oracle's bits
        assert np.array_equal(v.view(np.uint64), v_o.view(np.uint64))
    assert same > 0.5 and verdict > 0.8, (same, verdict)
