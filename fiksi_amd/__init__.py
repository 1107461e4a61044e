"""fiksi_amd — MI355X-native batched geometric-constraint solver behind fiksi's builder API.

The package is a thin host layer over ``libfiksi_amd.so`` (hand-written HIP kernels for gfx950 +
a plain C ABI, ``include/fiksi_amd.h``). It mirrors the reference crate's public surface:

    from fiksi_amd import System, SolvingOptions, elements, constraints
    s = System()
    p0 = elements.Point.create(s, 0., 0.)
    p1 = elements.Point.create(s, 1., 0.5)
    constraints.PointPointDistance.create(s, p0, p1, 2.)
    s.solve(SolvingOptions.DEFAULT)
"""
from . import abi, constraints, elements  # noqa: F401
from .abi import Context, DeviceBatch  # noqa: F401
from .system import (Analysis, ConstraintHandle, Decomposer, ElementHandle, Optimizer, SolvingOptions, System,  # noqa: F401
                     default_context, flatten, solve_systems)
