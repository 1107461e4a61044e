"""``fiksi::elements`` mirror (fiksi/src/elements/mod.rs:280,321,365,437)."""
from __future__ import annotations

from ._lib import lib
from .system import ElementHandle, System

LENGTH, POINT, LINE, CIRCLE = range(4)


def _expect(handle: ElementHandle, tag: int, what: str):
    if not isinstance(handle, ElementHandle) or handle.tag != tag:
        raise TypeError(f"expected an ElementHandle<{what}>")


def _made(system: System, rc: int, tag: int, what: str) -> ElementHandle:
    if rc < 0:
        raise TypeError(f"{what}::create rejected its arguments (fx_status {rc})")
    return ElementHandle(system.id, int(rc), tag)


class Length:
    @staticmethod
    def create(system: System, length: float) -> ElementHandle:
        return _made(system, lib.fxs_length_create(system._h, float(length)), LENGTH, "Length")


class Point:
    @staticmethod
    def create(system: System, x: float, y: float) -> ElementHandle:
        return _made(system, lib.fxs_point_create(system._h, float(x), float(y)), POINT, "Point")


class Line:
    @staticmethod
    def create(system: System, point1: ElementHandle, point2: ElementHandle) -> ElementHandle:
        _expect(point1, POINT, "Point")
        _expect(point2, POINT, "Point")
        return _made(system, lib.fxs_line_create(system._h, point1.id, point2.id), LINE, "Line")


class Circle:
    @staticmethod
    def create(system: System, center: ElementHandle, radius: ElementHandle) -> ElementHandle:
        _expect(center, POINT, "Point")
        _expect(radius, LENGTH, "Length")
        return _made(system, lib.fxs_circle_create(system._h, center.id, radius.id), CIRCLE, "Circle")
