"""``fiksi::constraints`` mirror: the eleven ``*::create`` builders (fiksi/src/constraints/mod.rs:317-891),
same argument order as the reference."""
from __future__ import annotations

import ctypes as C

from ._lib import lib
from .system import ConstraintHandle, ElementHandle, System

(POINT_POINT_COINCIDENCE, POINT_POINT_DISTANCE, POINT_POINT_POINT_ANGLE, POINT_LINE_INCIDENCE,
 POINT_LINE_DISTANCE, POINT_CIRCLE_INCIDENCE, SEGMENT_SEGMENT_LENGTH_EQUALITY, LINE_LINE_ANGLE,
 LINE_LINE_PARALLELISM, LINE_LINE_PERPENDICULARITY, LINE_CIRCLE_TANGENCY) = range(11)


def _create(system: System, tag: int, handles, param: float = 0.0) -> ConstraintHandle:
    for h in handles:
        if not isinstance(h, ElementHandle):
            raise TypeError("constraint arguments must be element handles")
    ids = (C.c_uint32 * len(handles))(*[h.id for h in handles])
    rc = lib.fxs_constraint_create(system._h, tag, ids, len(handles), float(param))
    if rc < 0:
        raise TypeError(f"constraint create rejected its arguments (wrong element kind?) fx_status {rc}")
    return ConstraintHandle(system.id, int(rc), tag)


class PointPointCoincidence:
    VALENCY = 2

    @staticmethod
    def create(system, point1, point2):
        return _create(system, POINT_POINT_COINCIDENCE, [point1, point2])


class PointPointDistance:
    VALENCY = 1

    @staticmethod
    def create(system, point1, point2, distance):
        return _create(system, POINT_POINT_DISTANCE, [point1, point2], distance)


class PointPointPointAngle:
    VALENCY = 1

    @staticmethod
    def create(system, point1, point2, point3, angle):
        return _create(system, POINT_POINT_POINT_ANGLE, [point1, point2, point3], angle)


class PointLineIncidence:
    VALENCY = 1

    @staticmethod
    def create(system, point, line):
        return _create(system, POINT_LINE_INCIDENCE, [point, line])


class PointLineDistance:
    VALENCY = 1

    @staticmethod
    def create(system, point, line, distance):
        return _create(system, POINT_LINE_DISTANCE, [point, line], distance)


class PointCircleIncidence:
    VALENCY = 1

    @staticmethod
    def create(system, point, circle):
        return _create(system, POINT_CIRCLE_INCIDENCE, [point, circle])


class SegmentSegmentLengthEquality:
    VALENCY = 1

    @staticmethod
    def create(system, segment1_point1, segment1_point2, segment2_point1, segment2_point2):
        return _create(system, SEGMENT_SEGMENT_LENGTH_EQUALITY,
                       [segment1_point1, segment1_point2, segment2_point1, segment2_point2])


class LineLineAngle:
    VALENCY = 1

    @staticmethod
    def create(system, line1, line2, angle):
        return _create(system, LINE_LINE_ANGLE, [line1, line2], angle)


class LineLineParallelism:
    VALENCY = 1

    @staticmethod
    def create(system, line1, line2):
        return _create(system, LINE_LINE_PARALLELISM, [line1, line2])


class LineLinePerpendicularity:
    VALENCY = 1

    @staticmethod
    def create(system, line1, line2):
        return _create(system, LINE_LINE_PERPENDICULARITY, [line1, line2])


class LineCircleTangency:
    VALENCY = 1

    @staticmethod
    def create(system, line, circle):
        return _create(system, LINE_CIRCLE_TANGENCY, [line, circle])
