"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes loader for ``oracle/libfiksi_oracle.so`` (the C++ CPU restatement of the reference's hot
path, see ``oracle/fo_common.hpp``). Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; nothing under ``fiksi_amd/`` does.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libfiksi_oracle.so")


def build(force: bool = False) -> str:
    """Compile the oracle with g++ (seconds)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _LIB_PATH


class FoResult(C.Structure):
    _fields_ = [
        ("accepted", C.c_uint32),
        ("trials", C.c_uint32),
        ("exit", C.c_uint32),
        ("ncomp", C.c_uint32),
        ("scale", C.c_double),
        ("sse0", C.c_double),
        ("sse", C.c_double),
    ]


RESULT_DTYPE = np.dtype(
    [("accepted", "<u4"), ("trials", "<u4"), ("exit", "<u4"), ("ncomp", "<u4"),
     ("scale", "<f8"), ("sse0", "<f8"), ("sse", "<f8")]
)

EXIT_SSE, EXIT_STEP, EXIT_FTOL, EXIT_MAX_OUTER, EXIT_TRIAL_CAP = range(5)

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.fo_eval_batch.restype = C.c_int64
        _lib.fo_from_triplets.restype = C.c_int64
    return _lib


def _p(a, t=None):
    if a is None:
        return None
    return a.ctypes.data_as(C.c_void_p)


def set_atan2_mode(mode: str) -> None:
    """'libm' (default): the platform's atan2, as the reference computes it here; 'correctly_rounded': binary128
    atan2q rounded to double — the canonical value the bit-for-bit FX_STEP_QR tests run both sides with."""
    lib().fo_set_atan2_mode(C.c_int({"libm": 0, "correctly_rounded": 1}[mode]))


class atan2_mode:
    """with oracle.atan2_mode('correctly_rounded'): ..."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.old = "correctly_rounded" if lib().fo_get_atan2_mode() else "libm"
        set_atan2_mode(self.mode)

    def __exit__(self, *exc):
        set_atan2_mode(self.old)


def atan2(y, x):
    """Element-wise atan2 in the current mode."""
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(y)
    lib().fo_atan2_batch(C.c_uint64(y.size), _p(y), _p(x), _p(out))
    return out


def rng_u32(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.uint32)
    lib().fo_rng_u32(C.c_uint32(seed), C.c_uint32(n), _p(out))
    return out


def rng_f64(seed: int, n: int) -> np.ndarray:
    out = np.zeros(n, dtype=np.float64)
    lib().fo_rng_f64(C.c_uint32(seed), C.c_uint32(n), _p(out))
    return out


def expr_eval(tag: int, vars8, param: float):
    """Residual and gradient of one expression on gathered values (expressions.rs)."""
    v = np.zeros(8, dtype=np.float64)
    vv = np.asarray(vars8, dtype=np.float64)
    v[: len(vv)] = vv
    g = np.zeros(8, dtype=np.float64)
    r = C.c_double(0.0)
    k = lib().fo_expr_eval(C.c_uint8(tag), _p(v), C.c_double(param), C.byref(r), _p(g))
    return r.value, g[:k].copy()


def colamd(nrows: int, ncols: int, rowidx, colptr, aggressive: bool = True) -> np.ndarray:
    r = np.ascontiguousarray(rowidx, dtype=np.int32)
    p = np.ascontiguousarray(colptr, dtype=np.int32).copy()
    rc = lib().fo_colamd(C.c_int(nrows), C.c_int(ncols), _p(r), _p(p), C.c_int(1 if aggressive else 0))
    if rc != 0:
        raise ValueError("colamd rejected the matrix")
    return p


def symbolic(nrows: int, ncols: int, colptr, rowidx):
    cp = np.ascontiguousarray(colptr, dtype=np.int64)
    ri = np.ascontiguousarray(rowidx, dtype=np.int64)
    parents = np.zeros(ncols, dtype=np.int64)
    post = np.zeros(ncols, dtype=np.int64)
    rc = np.zeros(ncols, dtype=np.int64)
    cc = np.zeros(ncols, dtype=np.int64)
    lcp = np.zeros(ncols + 1, dtype=np.int64)
    lri = np.zeros(ncols * (ncols + 1) // 2 + 1, dtype=np.int64)
    lib().fo_symbolic(C.c_int64(nrows), C.c_int64(ncols), _p(cp), _p(ri), _p(parents), _p(post), _p(rc), _p(cc),
                      _p(lcp), _p(lri))
    return dict(parents=parents, post=post, row_counts=rc, col_counts=cc, l_colptr=lcp, l_rowidx=lri[: lcp[-1]].copy())


def symbolic_qr(nrows: int, ncols: int, colptr, rowidx, ordering: str = "colamd"):
    """SymbolicQr::build: permutations and the structures of H and R."""
    cp = np.ascontiguousarray(colptr, dtype=np.int64)
    ri = np.ascontiguousarray(rowidx, dtype=np.int64)
    col_perm = np.zeros(ncols, dtype=np.int64)
    row_perm = np.zeros(nrows, dtype=np.int64)
    h_ptr = np.zeros(ncols + 1, dtype=np.int64)
    r_ptr = np.zeros(ncols + 1, dtype=np.int64)
    h_rows = np.zeros(nrows * ncols + 1, dtype=np.int64)
    r_rows = np.zeros(ncols * (ncols + 1) // 2 + 1, dtype=np.int64)
    lib().fo_symbolic_qr(C.c_int64(nrows), C.c_int64(ncols), _p(cp), _p(ri), C.c_int(1 if ordering == "colamd" else 0),
                         _p(col_perm), _p(row_perm), _p(h_ptr), _p(h_rows), _p(r_ptr), _p(r_rows))
    return dict(col_perm=col_perm, row_perm=row_perm, h_ptr=h_ptr, h_rows=h_rows[: h_ptr[-1]].copy(), r_ptr=r_ptr,
                r_rows=r_rows[: r_ptr[-1]].copy())


def qr_factor_solve(nrows: int, ncols: int, colptr, rowidx, values, b=None, ordering: str = "natural"):
    cp = np.ascontiguousarray(colptr, dtype=np.int64)
    ri = np.ascontiguousarray(rowidx, dtype=np.int64)
    va = np.ascontiguousarray(values, dtype=np.float64)
    cap = ncols * (ncols + 1) // 2 + 1
    rcp = np.zeros(ncols + 1, dtype=np.int64)
    rri = np.zeros(cap, dtype=np.int64)
    rva = np.zeros(cap, dtype=np.float64)
    solved = C.c_int(0)
    bb = None
    if b is not None:
        bb = np.zeros(nrows, dtype=np.float64)
        bb[: len(b)] = np.asarray(b, dtype=np.float64)
    rc = lib().fo_qr_factor_solve(C.c_int64(nrows), C.c_int64(ncols), _p(cp), _p(ri), _p(va),
                                  C.c_int(1 if ordering == "colamd" else 0), _p(bb), _p(rcp), _p(rri), _p(rva),
                                  C.c_int64(cap), C.byref(solved))
    if rc != 0:
        raise RuntimeError("R capacity too small")
    nnz = int(rcp[-1])
    return dict(r_colptr=rcp, r_rowidx=rri[:nnz].copy(), r_values=rva[:nnz].copy(),
                x=None if bb is None else bb[:ncols].copy(), solved=bool(solved.value))


def from_triplets(nrows: int, ncols: int, rows, cols, vals):
    r = np.ascontiguousarray(rows, dtype=np.int64)
    c = np.ascontiguousarray(cols, dtype=np.int64)
    v = np.ascontiguousarray(vals, dtype=np.float64)
    cp = np.zeros(ncols + 1, dtype=np.int64)
    ri = np.zeros(len(v), dtype=np.int64)
    va = np.zeros(len(v), dtype=np.float64)
    nnz = lib().fo_from_triplets(C.c_int64(nrows), C.c_int64(ncols), C.c_int64(len(v)), _p(r), _p(c), _p(v), _p(cp),
                                 _p(ri), _p(va))
    return cp, ri[:nnz].copy(), va[:nnz].copy()


def solve_upper(n: int, colptr, rowidx, values, b):
    cp = np.ascontiguousarray(colptr, dtype=np.int64)
    ri = np.ascontiguousarray(rowidx, dtype=np.int64)
    va = np.ascontiguousarray(values, dtype=np.float64)
    bb = np.ascontiguousarray(b, dtype=np.float64).copy()
    ok = lib().fo_solve_upper(C.c_int64(n), _p(cp), _p(ri), _p(va), _p(bb))
    return bool(ok), bb


# ---- flat batches (dict of numpy arrays with the fx_batch field names) ----------------------

def _batch_args(b):
    n = int(len(b["var_off"]) - 1)
    return n, [
        C.c_uint32(n), _p(b["var_off"]), _p(b["expr_off"]), _p(b["vars"]), _p(b["var_fixed"]), _p(b["expr_tag"]),
        _p(b["expr_idx"]), _p(b["expr_param"]),
    ]


def eval_batch(b):
    """r, (jrow_ptr, jcol, jval): reference sparse Jacobian assembly, row-major."""
    n, args = _batch_args(b)
    total_e = int(b["expr_off"][-1])
    r = np.zeros(total_e, dtype=np.float64)
    cap = 8 * total_e + 8
    jrow = np.zeros(total_e + 1, dtype=np.int64)
    jcol = np.zeros(cap, dtype=np.int32)
    jval = np.zeros(cap, dtype=np.float64)
    nnz = lib().fo_eval_batch(*args, _p(b.get("var_comp")), _p(r), _p(jrow), _p(jcol), _p(jval), C.c_int64(cap))
    assert nnz >= 0
    return r, (jrow, jcol[:nnz].copy(), jval[:nnz].copy())


def solve_batch(b, mode: int = 3, ordering: str = "colamd", trial_cap: int = 0, nthreads: int = 1):
    """assemble::solve (mode 3 = scale+perturb, the reference default; +4 = Optimizer::LBfgs) or the bare LM (mode 0).

    Returns (solved variables, per-system results); the input batch is not modified.
    """
    n, args = _batch_args(b)
    vars_out = b["vars"].copy()
    args[3] = _p(vars_out)
    res = np.zeros(n, dtype=RESULT_DTYPE)
    lib().fo_solve_batch(*args, _p(b.get("var_comp")), _p(b.get("expr_comp")), C.c_uint32(mode),
                         C.c_int(1 if ordering == "colamd" else 0), C.c_uint32(trial_cap), C.c_uint32(nthreads),
                         _p(res))
    return vars_out, res


def first_step_batch(b, ordering: str = "colamd"):
    n, args = _batch_args(b)
    delta = np.zeros_like(b["vars"])
    lib().fo_first_step_batch(*args, _p(b.get("var_comp")), _p(b.get("expr_comp")),
                              C.c_int(1 if ordering == "colamd" else 0), _p(delta))
    return delta


def system_scale_batch(b):
    n = int(len(b["var_off"]) - 1)
    scale = np.zeros(n, dtype=np.float64)
    lib().fo_system_scale_batch(C.c_uint32(n), _p(b["var_off"]), _p(b["expr_off"]), _p(b["vars"]), _p(b["expr_tag"]),
                                _p(b["expr_param"]), _p(scale))
    return scale


def residuals_batch(b, vars_=None):
    n = int(len(b["var_off"]) - 1)
    v = b["vars"] if vars_ is None else np.ascontiguousarray(vars_, dtype=np.float64)
    r = np.zeros(int(b["expr_off"][-1]), dtype=np.float64)
    lib().fo_residuals_batch(C.c_uint32(n), _p(b["var_off"]), _p(b["expr_off"]), _p(v), _p(b["expr_tag"]),
                             _p(b["expr_idx"]), _p(b["expr_param"]), _p(r))
    return r


def analyze_batch(b):
    """System::analyze: 1 per expression that does not increase the rank (over-constraining)."""
    n = int(len(b["var_off"]) - 1)
    dep = np.zeros(int(b["expr_off"][-1]), dtype=np.uint8)
    lib().fo_analyze_batch(C.c_uint32(n), _p(b["var_off"]), _p(b["expr_off"]), _p(b["vars"]), _p(b["expr_tag"]),
                           _p(b["expr_idx"]), _p(b["expr_param"]), _p(dep))
    return dep


def solve_single_pass_batch(b, perturb: bool = True, ordering: str = "colamd", trial_cap: int = 0, nthreads: int = 1,
                            lbfgs: bool = False):
    """assemble::solve with Decomposer::SinglePass. Returns (solved variables, per-system results)."""
    n, args = _batch_args(b)
    vars_out = b["vars"].copy()
    args[3] = _p(vars_out)
    res = np.zeros(n, dtype=RESULT_DTYPE)
    lib().fo_solve_single_pass_batch(*args, _p(b.get("var_comp")), _p(b.get("expr_comp")),
                                     C.c_uint32((1 if perturb else 0) | (4 if lbfgs else 0)),
                                     C.c_int(1 if ordering == "colamd" else 0), C.c_uint32(trial_cap),
                                     C.c_uint32(nthreads), _p(res))
    return vars_out, res


def single_pass_units(b, system: int = 0, comp: int = 0):
    """The SinglePass blocks of one component: list of (expression ids in block order, free variables)."""
    v0, v1 = int(b["var_off"][system]), int(b["var_off"][system + 1])
    e0, e1 = int(b["expr_off"][system]), int(b["expr_off"][system + 1])
    nv, ne = v1 - v0, e1 - e0
    cap = 8 * ne + nv + 8
    uoe = np.zeros(max(ne, 1), dtype=np.int32)
    ro = np.zeros(cap + 1, dtype=np.uint32); rows = np.zeros(cap, dtype=np.uint32)
    vo = np.zeros(cap + 1, dtype=np.uint32); vs = np.zeros(cap, dtype=np.uint32)
    fixed = np.ascontiguousarray(b["var_fixed"][v0:v1])
    tags = np.ascontiguousarray(b["expr_tag"][e0:e1])
    idx = np.ascontiguousarray(b["expr_idx"][4 * e0:4 * e1])
    vc = None if b.get("var_comp") is None else np.ascontiguousarray(b["var_comp"][v0:v1])
    nu = lib().fo_single_pass_units(C.c_uint32(nv), C.c_uint32(ne), _p(fixed), _p(tags), _p(idx), _p(vc), C.c_uint16(comp),
                                    _p(uoe), _p(ro), _p(rows), _p(vo), _p(vs), C.c_uint32(cap))
    assert nu >= 0
    return [(rows[ro[u]:ro[u + 1]].tolist(), vs[vo[u]:vo[u + 1]].tolist()) for u in range(nu)]


def permute(permutation, values):
    """PermutationSequence::build_for_gather_permutation + permute_slice. Returns (permuted values, swaps)."""
    perm = np.ascontiguousarray(permutation, dtype=np.uint32)
    vals = np.ascontiguousarray(values, dtype=np.float64).copy()
    n = lib().fo_permute(C.c_uint32(len(perm)), _p(perm), _p(vals))
    return vals, n


def maximum_matching(nvars: int, expressions, free=None):
    """find_maximum_matching on a raw bipartite graph: `expressions` = list of variable lists.
    Returns (cardinality, variable -> expression or -1)."""
    eptr = np.zeros(len(expressions) + 1, dtype=np.uint32)
    eptr[1:] = np.cumsum([len(e) for e in expressions])
    evars = np.asarray([v for e in expressions for v in e], dtype=np.uint32)
    fr = np.asarray(sorted(range(nvars) if free is None else free), dtype=np.uint32)
    out = np.zeros(max(nvars, 1), dtype=np.uint32)
    card = lib().fo_maximum_matching(C.c_uint32(nvars), C.c_uint32(len(expressions)), _p(eptr), _p(evars),
                                     C.c_uint32(len(fr)), _p(fr), _p(out))
    return card, [int(x) if x != 0xFFFFFFFF else -1 for x in out[:nvars]]


def eval_dense_batch(b):
    """Problem::calculate_residuals_and_jacobian for every System: (residuals, [dense row-major J])."""
    n, args = _batch_args(b)
    off = np.zeros(n + 1, dtype=np.uint64)
    lib().fo_eval_dense_batch(*args, _p(b.get("var_comp")), None, None, _p(off))
    r = np.zeros(int(b["expr_off"][-1]) if n else 0, dtype=np.float64)
    jac = np.zeros(max(int(off[-1]), 1), dtype=np.float64)
    lib().fo_eval_dense_batch(*args, _p(b.get("var_comp")), _p(r), _p(jac), _p(off))
    out = []
    for s in range(n):
        m = int(b["expr_off"][s + 1] - b["expr_off"][s])
        blk = jac[int(off[s]):int(off[s + 1])]
        out.append(blk.reshape(m, -1) if m and len(blk) else blk.reshape(m, 0))
    return r, out


def solve_recursive(g, perturb: bool = True, ordering: str = "colamd", trial_cap: int = 0, budget: int = 0, sizes=None):
    """assemble::solve with Decomposer::RecursiveAssembly on ONE System (fo_recursive.hpp).

    ``g``: the System with its geometric graph — ``vars, var_fixed, expr_tag, expr_idx, expr_param`` (flat batch of one
    System), ``el_kind, el_idx, el_comp`` per element and ``con_valency, con_expr, con_ninc, con_inc, con_comp`` per
    constraint (``fiksi_amd.System.graph()`` produces it). Returns (solved variables, serialised plan words,
    per-step results, flags: bit0 the reference would panic, bit1 search budget exhausted). ``sizes`` (optional list)
    receives (unknowns, rows) of every step's cluster problem."""
    vars_out = np.ascontiguousarray(g["vars"], dtype=np.float64).copy()
    nv, ne = len(vars_out), len(g["expr_tag"])
    nel, ncon = len(g["el_kind"]), len(g["con_valency"])
    cap = 1 << 20
    plan = np.zeros(cap, dtype=np.uint32)
    plan_len = C.c_uint32(0)
    steps = np.zeros(4096, dtype=RESULT_DTYPE)
    step_sizes = np.zeros(2 * 4096, dtype=np.uint32)
    n_steps = C.c_uint32(0)
    flags = C.c_uint32(0)
    a = lambda k, dt: np.ascontiguousarray(g[k], dtype=dt)
    keep = [a("var_fixed", np.uint8), a("expr_tag", np.uint8), a("expr_idx", np.uint32), a("expr_param", np.float64),
            a("el_kind", np.uint8), a("el_idx", np.uint32), a("el_comp", np.uint16), a("con_valency", np.uint8),
            a("con_expr", np.uint32), a("con_ninc", np.uint8), a("con_inc", np.uint32), a("con_comp", np.uint16)]
    lib().fo_solve_recursive(C.c_uint32(nv), _p(vars_out), _p(keep[0]), C.c_uint32(ne), _p(keep[1]), _p(keep[2]), _p(keep[3]),
                             C.c_uint32(nel), _p(keep[4]), _p(keep[5]), _p(keep[6]), C.c_uint32(ncon), _p(keep[7]), _p(keep[8]),
                             _p(keep[9]), _p(keep[10]), _p(keep[11]), C.c_int(1 if perturb else 0),
                             C.c_int(1 if ordering == "colamd" else 0), C.c_uint32(trial_cap), C.c_uint64(budget), _p(plan),
                             C.c_uint32(cap), C.byref(plan_len), _p(steps), C.c_uint32(len(steps)), C.byref(n_steps), C.byref(flags),
                             _p(step_sizes))
    assert plan_len.value <= cap and n_steps.value <= len(steps)
    if sizes is not None:
        sizes.extend((int(step_sizes[2 * i]), int(step_sizes[2 * i + 1])) for i in range(n_steps.value))
    return vars_out, plan[: plan_len.value].copy(), steps[: n_steps.value].copy(), int(flags.value)


def pose_rows(pose, point):
    """Pose2D::transform_point and the two gradient_chain_rule_point rows ([1,0] and [0,1]) at ``pose`` =
    (rotation, tx, ty) for ``point`` = (u, v). Returns (x, y, gradient of x[3], gradient of y[3])."""
    pose = np.ascontiguousarray(pose, dtype=np.float64)
    out = np.zeros(8, dtype=np.float64)
    lib().fo_pose_rows(_p(pose), C.c_double(point[0]), C.c_double(point[1]), _p(out))
    return out[0], out[1], out[2:5].copy(), out[5:8].copy()
