"""Python mirror of the reference builder API (``fiksi::System`` and handles) over the C builder
(``include/fiksi_amd_builder.h``). Names, argument order and error behaviour follow the reference:

* ``System::new/solve/get_element_handles/get_constraint_handles``  fiksi/src/lib.rs:307-466
* ``ElementHandle::{fix,unfix,get_value,update_value,as_any_element}`` fiksi/src/elements/mod.rs:60-112,560-579
* ``ConstraintHandle::{calculate_residual,update_parameter}``        fiksi/src/constraints/mod.rs:88-110,992-1046
* ``SolvingOptions``, ``Decomposer``, ``solve::Optimizer``            fiksi/src/lib.rs:154-237, solve/mod.rs:17-27

Where the reference panics (a handle used with a foreign System, lib.rs `assert_eq!`) this mirror
raises ``AssertionError``; where Rust's type system rejects a call (wrong element kind) it raises
``TypeError``.
"""
from __future__ import annotations

import ctypes as C
import enum
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import abi
from ._lib import FxResult, check, lib


class Optimizer(enum.Enum):  # solve/mod.rs:17-27
    LevenbergMarquardt = 0
    LBfgs = 1


class Decomposer(enum.Enum):  # lib.rs:154-201
    NONE = 0
    SinglePass = 1
    RecursiveAssembly = 2


@dataclass
class SolvingOptions:  # lib.rs:205-237
    optimizer: Optimizer = Optimizer.LevenbergMarquardt
    decomposer: Decomposer = Decomposer.NONE
    perturb: bool = True
    plan_budget: int = 0  # not in the reference: thousands of subgraphs RecursiveAssembly's plan search may grow (0 = default)

    def _to_abi(self):
        o = abi.solving_opts(perturb=self.perturb)
        o.optimizer = self.optimizer.value
        o.decomposer = self.decomposer.value
        o.plan_budget = self.plan_budget
        return o


SolvingOptions.DEFAULT = SolvingOptions()

_default_ctx: Optional[abi.Context] = None


def default_context() -> abi.Context:
    """The device context ``System.solve`` uses when none is given (device 0, created lazily)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = abi.Context(0)
    return _default_ctx


class ElementHandle:
    """``(system_id, id)`` handle, typed by ``tag`` (elements/mod.rs:26-33)."""

    __slots__ = ("system_id", "id", "tag")

    def __init__(self, system_id: int, id: int, tag: int):
        self.system_id, self.id, self.tag = system_id, id, tag

    def _own(self, system: "System"):
        assert self.system_id == system.id, "Tried to get an element that is not part of this `System`"

    def fix(self, system: "System"):  # elements/mod.rs:60-65
        check(lib.fxs_element_fix(system._h, self.id), "fix")

    def unfix(self, system: "System"):  # elements/mod.rs:80-85
        check(lib.fxs_element_unfix(system._h, self.id), "unfix")

    def get_value(self, system: "System"):  # elements/mod.rs:88-100
        self._own(system)
        out = (C.c_double * 4)()
        n = check(lib.fxs_element_get_value(system._h, self.id, out), "get_value")
        vals = tuple(out[i] for i in range(n))
        return vals[0] if n == 1 else vals

    def update_value(self, system: "System", *values: float):  # elements/mod.rs:560-579
        if self.tag == 1 and len(values) == 2:
            check(lib.fxs_point_update_value(system._h, self.id, values[0], values[1]), "update_value")
        elif self.tag == 0 and len(values) == 1:
            check(lib.fxs_length_update_value(system._h, self.id, values[0]), "update_value")
        else:
            raise TypeError("update_value exists for Point (x, y) and Length (length) handles only")

    def as_any_element(self) -> "ElementHandle":
        return self

    def __eq__(self, other):
        return isinstance(other, ElementHandle) and (self.system_id, self.id) == (other.system_id, other.id)

    def __hash__(self):
        return hash((self.system_id, self.id))


class ConstraintHandle:
    __slots__ = ("system_id", "id", "tag")

    def __init__(self, system_id: int, id: int, tag: int):
        self.system_id, self.id, self.tag = system_id, id, tag

    def calculate_residual(self, system: "System") -> float:  # constraints/mod.rs:88-110
        assert self.system_id == system.id, "Tried to evaluate a constraint that is not part of this `System`"
        return float(system.constraint_residuals()[self.id])

    def update_parameter(self, system: "System", value: float):  # constraints/mod.rs:992-1046
        rc = lib.fxs_constraint_update_parameter(system._h, self.id, value)
        if rc < 0:
            raise TypeError("this constraint kind has no parameter")

    def as_any_constraint(self) -> "ConstraintHandle":
        return self

    def __eq__(self, other):
        return isinstance(other, ConstraintHandle) and (self.system_id, self.id) == (other.system_id, other.id)

    def __hash__(self):
        return hash((self.system_id, self.id))


@dataclass
class Analysis:  # lib.rs:245-249
    overconstrained: List[ConstraintHandle]


class System:
    """``fiksi::System``: build with ``elements.*.create`` / ``constraints.*.create``, then ``solve``."""

    def __init__(self):
        h = C.c_void_p()
        check(lib.fxs_system_new(C.byref(h)), "System::new")
        self._h = h
        self.id = int(lib.fxs_system_id(h))
        self.last_result = None

    def __del__(self):
        try:
            if self._h:
                lib.fxs_system_free(self._h)
                self._h = None
        except Exception:
            pass

    # -- lib.rs:329-361
    def get_element_handles(self) -> List[ElementHandle]:
        n = lib.fxs_num_elements(self._h)
        return [ElementHandle(self.id, i, lib.fxs_element_tag_of(self._h, i)) for i in range(n)]

    def get_constraint_handles(self) -> List[ConstraintHandle]:
        n = lib.fxs_num_constraints(self._h)
        return [ConstraintHandle(self.id, i, lib.fxs_constraint_tag_of(self._h, i)) for i in range(n)]

    # -- lib.rs:464-466
    def solve(self, opts: SolvingOptions = SolvingOptions.DEFAULT, ctx: Optional[abi.Context] = None, **lm_kw):
        """``lm_kw``: fields of ``fx_lm_opts`` to set beside the reference's three options (e.g. ``solver=2`` for
        the reference-numerics step FX_STEP_QR)."""
        ctx = ctx or default_context()
        res = FxResult()
        o = opts._to_abi()
        for k, v in lm_kw.items():
            setattr(o.lm, k, v)
        check(lib.fxs_system_solve(self._h, ctx.handle, C.byref(o), C.byref(res)), "System::solve")
        self.last_result = {f[0]: getattr(res, f[0]) for f in FxResult._fields_}

    # -- lib.rs:448-459 (doc-hidden in the reference): constraints that over-constrain the System
    def analyze(self, ctx: Optional[abi.Context] = None) -> "Analysis":
        ctx = ctx or default_context()
        ne = max(int(lib.fxs_num_expressions(self._h)), 1)
        ids = np.zeros(ne, dtype=np.uint32)
        n = C.c_uint32(0)
        check(lib.fxs_system_analyze(self._h, ctx.handle, ids.ctypes.data, C.byref(n)), "System::analyze")
        return Analysis([ConstraintHandle(self.id, int(i), lib.fxs_constraint_tag_of(self._h, int(i))) for i in ids[: n.value]])

    def constraint_residuals(self, ctx: Optional[abi.Context] = None) -> np.ndarray:
        ctx = ctx or default_context()
        n = lib.fxs_num_constraints(self._h)
        out = np.zeros(max(n, 1), dtype=np.float64)
        check(lib.fxs_system_constraint_residuals(self._h, ctx.handle, out.ctypes.data), "calculate_residual")
        return out[:n]

    def components(self):
        """(n_components, component of each element, component of each constraint)."""
        ne, nc = lib.fxs_num_elements(self._h), lib.fxs_num_constraints(self._h)
        n = C.c_uint32(0)
        ec = np.zeros(max(ne, 1), dtype=np.uint16)
        cc = np.zeros(max(nc, 1), dtype=np.uint16)
        check(lib.fxs_components(self._h, C.byref(n), ec.ctypes.data, cc.ctypes.data), "components")
        return n.value, ec[:ne], cc[:nc]

    def flatten(self):
        return flatten([self])

    def graph(self):
        """The System with its geometric graph (graph.rs:98-147): the flat batch arrays of ``flatten()`` plus, per
        element, ``el_kind`` (0 Length, 1 Point, 2 Line, 3 Circle), ``el_idx`` (first variable) and ``el_comp``, and
        per constraint ``con_valency``, ``con_expr`` (first expression), ``con_ninc`` / ``con_inc`` (incident
        primitive elements, 6 slots each) and ``con_comp``."""
        g = self.flatten()
        ne, nc = lib.fxs_num_elements(self._h), lib.fxs_num_constraints(self._h)
        ek = np.zeros(max(ne, 1), dtype=np.uint8); ei = np.zeros(max(ne, 1), dtype=np.uint32)
        cv = np.zeros(max(nc, 1), dtype=np.uint8); ce = np.zeros(max(nc, 1), dtype=np.uint32)
        cn = np.zeros(max(nc, 1), dtype=np.uint8); ci = np.zeros(6 * max(nc, 1), dtype=np.uint32)
        check(lib.fxs_export_graph(self._h, ek.ctypes.data, ei.ctypes.data, cv.ctypes.data, ce.ctypes.data, cn.ctypes.data,
                                   ci.ctypes.data), "export_graph")
        _, ec, cc = self.components()
        g.update(el_kind=ek[:ne], el_idx=ei[:ne], el_comp=ec, con_valency=cv[:nc], con_expr=ce[:nc], con_ninc=cn[:nc],
                 con_inc=ci[: 6 * nc], con_comp=cc)
        return g

    def recursive_plan(self, budget: int = 0):
        """The recombination plan ``Decomposer.RecursiveAssembly`` solves this System by (host only), as the word
        list of ``fxs_recursive_plan``, and its flags (bit0: the reference would panic, bit1: search budget spent)."""
        n, fl = C.c_uint32(0), C.c_uint32(0)
        check(lib.fxs_recursive_plan(self._h, budget, None, 0, C.byref(n), C.byref(fl)), "recursive_plan")
        out = np.zeros(max(n.value, 1), dtype=np.uint32)
        check(lib.fxs_recursive_plan(self._h, budget, out.ctypes.data, n.value, C.byref(n), C.byref(fl)), "recursive_plan")
        return out[: n.value], fl.value


def flatten(systems: Sequence[System]):
    """The fx_batch arrays (numpy copies) of a list of Systems."""
    n = len(systems)
    arr = (C.c_void_p * max(n, 1))(*[s._h for s in systems])
    f = C.c_void_p()
    check(lib.fxs_flatten(arr, n, C.byref(f)), "flatten")
    try:
        b = lib.fxs_flat_batch(f).contents
        var_off = np.ctypeslib.as_array(C.cast(b.var_off, C.POINTER(C.c_uint32)), (n + 1,)).copy()
        expr_off = np.ctypeslib.as_array(C.cast(b.expr_off, C.POINTER(C.c_uint32)), (n + 1,)).copy()
        nv, ne = int(var_off[-1]), int(expr_off[-1])

        def grab(ptr, ctype, count):
            if count == 0:
                return np.zeros(0, dtype=np.dtype(ctype))
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(ctype)), (count,)).copy()

        return {
            "var_off": var_off, "expr_off": expr_off,
            "vars": grab(b.vars, C.c_double, nv), "var_fixed": grab(b.var_fixed, C.c_uint8, nv),
            "expr_tag": grab(b.expr_tag, C.c_uint8, ne), "expr_idx": grab(b.expr_idx, C.c_uint32, 4 * ne),
            "expr_param": grab(b.expr_param, C.c_double, ne),
            "var_comp": grab(b.var_comp, C.c_uint16, nv), "expr_comp": grab(b.expr_comp, C.c_uint16, ne),
        }
    finally:
        lib.fxs_flat_free(f)


def solve_systems(systems: Sequence[System], opts: SolvingOptions = SolvingOptions.DEFAULT,
                  ctx: Optional[abi.Context] = None) -> np.ndarray:
    """Solve many independent Systems in one device batch (one wavefront each)."""
    ctx = ctx or default_context()
    n = len(systems)
    arr = (C.c_void_p * max(n, 1))(*[s._h for s in systems])
    res = np.zeros(n, dtype=abi.RESULT_DTYPE)
    o = opts._to_abi()
    check(lib.fxs_systems_solve(arr, n, ctx.handle, C.byref(o), res.ctypes.data), "solve_systems")
    return res
