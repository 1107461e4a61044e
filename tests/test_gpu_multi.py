"""fx_system_solve_batch_multi (SURVEY 8e as a library entry point): one batch sharded over several contexts, a host
thread each. On the 1-GPU box the contexts share device 0 — the code path (threads, shard views, in-place write-back,
host-side counter sum) is the one an 8-GPU host runs with one context per device."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_sharded_solve_equals_the_single_context_solve(fiksi, ctx):
    from fiksi_amd import abi, workloads
    from fiksi_amd._lib import FiksiError

    b = workloads.concat([workloads.ring16(3001), workloads.hinged_triangles(7, 11), workloads.large_sketch(120, seed=3),
                          workloads.ring16(500, inconsistent=True)])
    v1, r1 = ctx.system_solve_batch(b)
    others = [fiksi.Context(0) for _ in range(3)]
    try:
        for n_ctx in (1, 2, 3):
            v, r, total = abi.Context.system_solve_batch_multi(others[:n_ctx], b)
            assert np.array_equal(v.view(np.uint64), v1.view(np.uint64)), n_ctx
            assert np.array_equal(r, r1), n_ctx
            assert total == {"systems": len(r1), "converged": int((r1["sse_unscaled"] < 1e-4).sum()),
                             "accepted": int(r1["accepted"].sum()), "trials": int(r1["trials"].sum())}
        # more contexts than Systems: empty shards are skipped
        tiny = workloads.ring16(2)
        v, r, total = abi.Context.system_solve_batch_multi(others, tiny)
        vt, rt = ctx.system_solve_batch(tiny)
        assert np.array_equal(v.view(np.uint64), vt.view(np.uint64)) and np.array_equal(r, rt) and total["systems"] == 2
        with pytest.raises(FiksiError):
            abi.Context.system_solve_batch_multi([others[0], others[0]], tiny)  # a context is bound to one host thread
    finally:
        for c in others:
            c.close()


def test_shards_large_enough_to_be_solved_in_chunks(fiksi, ctx):
    """Each context's shard of 70 000 one-wavefront Systems takes the chunked host path (copy stream + two solve streams
    of its own context); two contexts, two host threads: the bits of the resident solve."""
    from fiksi_amd import abi, workloads

    b = workloads.concat([workloads.ring16(69000, seed0=31), workloads.hinged_triangles(1000, 5), workloads.ring16(70000, seed0=900000)])
    assert len(b["var_off"]) - 1 == 140000
    db = ctx.upload(b)
    db.system_solve()
    v1, r1 = db.get_vars(), db.get_results()
    db.free()
    others = [fiksi.Context(0) for _ in range(2)]
    try:
        v, r, total = abi.Context.system_solve_batch_multi(others, b)
        assert np.array_equal(v.view(np.uint64), v1.view(np.uint64))
        assert np.array_equal(r, r1) and total["systems"] == 140000
    finally:
        for c in others:
            c.close()


def test_medium_components_take_one_route_for_every_shard(fiksi, ctx):
    """Components of 65 ... 128 columns go to the wide kernel or to the team kernels by the cost of the batch at hand, and the
    two add in different orders: 2 000 of the reference's 16-triangle sketches (66 variables) go wide as a whole and would go
    to the team kernels in three shards of 667. fx_system_solve_batch_multi decides once, on the whole batch — the bits do not
    depend on the number of contexts."""
    from fiksi_amd import abi, workloads

    b = workloads.hinged_triangles(2000, 16)
    assert int(b["var_off"][1]) == 66
    ctx.set_one_structure_builds(False)  # (this is about the wide / team choice: the sparse one-structure build would take the batch)
    v1, r1 = ctx.system_solve_batch(b)
    others = [fiksi.Context(0) for _ in range(3)]
    for c in others:
        c.set_one_structure_builds(False)
    try:
        for n_ctx in (1, 2, 3):
            v, r, _ = abi.Context.system_solve_batch_multi(others[:n_ctx], b)
            assert np.array_equal(v.view(np.uint64), v1.view(np.uint64)), n_ctx
            assert np.array_equal(r, r1), n_ctx
        # a pinned route is honoured (and must be the same in every context)
        for c in others:
            c.set_wide_routing(0)
        ctx.set_wide_routing(0)
        v0, r0 = ctx.system_solve_batch(b)
        v, r, _ = abi.Context.system_solve_batch_multi(others, b)
        assert np.array_equal(v.view(np.uint64), v0.view(np.uint64)) and np.array_equal(r, r0)
        others[1].set_wide_routing(1)
        with pytest.raises(Exception):
            abi.Context.system_solve_batch_multi(others, b)
    finally:
        ctx.set_wide_routing(-1)
        ctx.set_one_structure_builds(True)
        for c in others:
            c.close()
