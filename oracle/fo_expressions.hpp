// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates fiksi/src/constraints/expressions.rs:28-961 (the 11 expression variants: variable
// indices, length-scale transform, residual + gradient). kurbo 0.13.0 (not vendored in the
// reference tree) supplies only Point/Vec2 arithmetic: `a - b` component-wise, `Vec2::atan2 =
// y.atan2(x)`, `cross(a,b) = a.x*b.y - a.y*b.x`, `dot`, `hypot2 = length_squared = x*x + y*y`,
// `distance_squared = (a-b).hypot2()`. Operation order below mirrors the Rust source so results
// are bit-identical under IEEE-754 without FMA contraction (compile with -ffp-contract=off).
#pragma once
#include <quadmath.h>

#include <cmath>
#include <cstdint>

namespace fo {

// Variant order of `enum Expression`, expressions.rs:28-40.
enum Tag : uint8_t {
    VariableVariableEquality = 0,
    PointPointDistance = 1,
    PointPointPointAngle = 2,
    PointLineIncidence = 3,
    PointLineDistance = 4,
    PointCircleIncidence = 5,
    SegmentSegmentLengthEquality = 6,
    LineLineAngle = 7,
    LineLineParallelism = 8,
    LineLinePerpendicularity = 9,
    LineCircleTangency = 10,
    NUM_TAGS = 11,
};

// Flat form of one `Expression`: up to four element fields (variable index of a point's x, or of
// a length) in declaration order of the Rust struct, plus the f64 parameter (distance / angle).
struct Expression {
    uint8_t tag;
    uint32_t idx[4];
    double param;
};

// expressions.rs:48-182. Returns the number of variables written to `out`.
inline int variable_indices(const Expression& e, uint32_t out[8]) {
    const uint32_t* f = e.idx;
    switch (e.tag) {
        case VariableVariableEquality:  // :50-54
            out[0] = f[0];
            out[1] = f[1];
            return 2;
        case PointPointDistance:  // :55-64
            out[0] = f[0]; out[1] = f[0] + 1; out[2] = f[1]; out[3] = f[1] + 1;
            return 4;
        case PointPointPointAngle:  // :65-76
        case PointLineIncidence:    // :77-88
        case PointLineDistance:     // :89-100
            out[0] = f[0]; out[1] = f[0] + 1; out[2] = f[1]; out[3] = f[1] + 1;
            out[4] = f[2]; out[5] = f[2] + 1;
            return 6;
        case PointCircleIncidence:  // :101-111 (point, center, radius)
            out[0] = f[0]; out[1] = f[0] + 1; out[2] = f[1]; out[3] = f[1] + 1;
            out[4] = f[2];
            return 5;
        case SegmentSegmentLengthEquality:  // :112-125
        case LineLineAngle:                 // :126-139
        case LineLineParallelism:           // :140-153
        case LineLinePerpendicularity:      // :154-167
            out[0] = f[0]; out[1] = f[0] + 1; out[2] = f[1]; out[3] = f[1] + 1;
            out[4] = f[2]; out[5] = f[2] + 1; out[6] = f[3]; out[7] = f[3] + 1;
            return 8;
        case LineCircleTangency:  // :168-180 (line p1, line p2, center, radius)
            out[0] = f[0]; out[1] = f[0] + 1; out[2] = f[1]; out[3] = f[1] + 1;
            out[4] = f[2]; out[5] = f[2] + 1; out[6] = f[3];
            return 7;
        default:
            return 0;
    }
}

// expressions.rs:195-211: only the two distance-parameterised variants scale.
inline Expression transform(const Expression& e, double length_scale_recip) {
    Expression t = e;
    if (e.tag == PointPointDistance || e.tag == PointLineDistance) {
        t.param = length_scale_recip * e.param;
    }
    return t;
}

// Whether the parameter counts as a length in `calculate_system_scale`
// (assemble/mod.rs:32-44).
inline bool has_distance_param(uint8_t tag) {
    return tag == PointPointDistance || tag == PointLineDistance;
}

namespace detail {

struct V2 { double x, y; };
inline V2 sub(double ax, double ay, double bx, double by) { return V2{ax - bx, ay - by}; }
inline double cross(V2 a, V2 b) { return a.x * b.y - a.y * b.x; }
inline double dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
inline double hypot2(V2 a) { return a.x * a.x + a.y * a.y; }
// Rust's f64::atan2 is the platform libm's (std) or the `libm` crate's (fiksi/src/floatfuncs.rs:48-59): within an
// ulp of the exact value, different bits on different platforms (glibc 2.35 misrounds ~0.07 % of arguments by an
// ulp — tests/test_atan2.py). Mode 0 (default) is what the reference computes on THIS platform: glibc's atan2.
// Mode 1 is the correctly rounded value every such libm approximates, computed independently of the product's
// routine: libquadmath's atan2q in binary128, rounded to double. The bit-for-bit tests of FX_STEP_QR run both
// sides in that canonical mode.
inline int& atan2_mode() {
    static int mode = 0;
    return mode;
}
inline double vatan2(V2 a) {
    if (atan2_mode() == 1) return static_cast<double>(atan2q(static_cast<__float128>(a.y), static_cast<__float128>(a.x)));
    return std::atan2(a.y, a.x);
}
constexpr double PI = 3.14159265358979323846264338327950288;

// expressions.rs:327-352
inline double ppd(const double v[4], double param_distance, double g[4]) {
    double p1x = v[0], p1y = v[1], p2x = v[2], p2y = v[3];
    double distance = std::sqrt((p1x - p2x) * (p1x - p2x) + (p1y - p2y) * (p1y - p2y));
    double residual = distance - param_distance;
    double distance_recip = 1. / distance;
    g[0] = (p1x - p2x) * distance_recip;
    g[1] = (p1y - p2y) * distance_recip;
    g[2] = -(p1x - p2x) * distance_recip;
    g[3] = -(p1y - p2y) * distance_recip;
    return residual;
}

// expressions.rs:393-399 / :665-671 (single-step wrap)
inline double wrap_angle(double angle) {
    if (angle > PI) return angle - 2.0 * PI;
    if (angle < -PI) return angle + 2.0 * PI;
    return angle;
}

}  // namespace detail

// expressions.rs:214-276 dispatch + the per-variant `compute_residual_and_gradient_` bodies.
// `v` holds the gathered variable values in `variable_indices` order; `g` receives the partials
// in the same order. Returns the residual.
inline double compute_residual_and_gradient(const Expression& e, const double v[8], double g[8]) {
    using namespace detail;
    switch (e.tag) {
        case VariableVariableEquality: {  // :294-300
            g[0] = -1.;
            g[1] = 1.;
            return v[1] - v[0];
        }
        case PointPointDistance:  // :327-352
            return ppd(v, e.param, g);
        case PointPointPointAngle: {  // :375-424
            V2 u = sub(v[0], v[1], v[2], v[3]);
            V2 w = sub(v[4], v[5], v[2], v[3]);
            double angle = wrap_angle(vatan2(w) - vatan2(u));
            double residual = angle - e.param;
            double u_squared_recip = 1. / hypot2(u);
            double v_squared_recip = 1. / hypot2(w);
            double d1x = u.y * u_squared_recip;
            double d1y = -u.x * u_squared_recip;
            double d3x = -w.y * v_squared_recip;
            double d3y = w.x * v_squared_recip;
            double d2x = -d1x - d3x;
            double d2y = -d1y - d3y;
            g[0] = d1x; g[1] = d1y; g[2] = d2x; g[3] = d2y; g[4] = d3x; g[5] = d3y;
            return residual;
        }
        case PointLineIncidence: {  // :448-476
            double px = v[0], py = v[1], l1x = v[2], l1y = v[3], l2x = v[4], l2y = v[5];
            V2 u = sub(l2x, l2y, l1x, l1y);
            V2 w = sub(px, py, l1x, l1y);
            double residual = cross(u, w);
            g[0] = -u.y;
            g[1] = u.x;
            g[2] = -py + l2y;
            g[3] = px - l2x;
            g[4] = w.y;
            g[5] = -w.x;
            return residual;
        }
        case PointLineDistance: {  // :503-543
            double px = v[0], py = v[1], l1x = v[2], l1y = v[3], l2x = v[4], l2y = v[5];
            V2 u = sub(l2x, l2y, l1x, l1y);
            V2 w = sub(px, py, l1x, l1y);
            double cr = cross(u, w);
            double line_length_squared = hypot2(u);
            double line_length = std::sqrt(line_length_squared);
            double line_length_recip = 1. / line_length;
            double a = cr / line_length_squared;
            double b = -a * u.x;
            double c = px + a * u.y;
            double residual = line_length_recip * cr - e.param;
            g[0] = -line_length_recip * u.y;
            g[1] = line_length_recip * u.x;
            g[2] = -line_length_recip * (b - l2y + py);
            g[3] = -line_length_recip * (l2x - c);
            g[4] = line_length_recip * (b + w.y);
            g[5] = -line_length_recip * (c - l1x);
            return residual;
        }
        case PointCircleIncidence: {  // :563-575: PPD(point, center, D = radius variable)
            double residual = ppd(v, v[4], g);
            g[4] = -1.;
            return residual;
        }
        case SegmentSegmentLengthEquality: {  // :596-619
            double g1[4], g2[4];
            double r1 = ppd(v, 0., g1);
            double r2 = ppd(v + 4, 0., g2);
            g[0] = -g1[0]; g[1] = -g1[1]; g[2] = -g1[2]; g[3] = -g1[3];
            g[4] = g2[0]; g[5] = g2[1]; g[6] = g2[2]; g[7] = g2[3];
            return r2 - r1;
        }
        case LineLineAngle: {  // :643-695
            V2 u = sub(v[2], v[3], v[0], v[1]);
            V2 w = sub(v[6], v[7], v[4], v[5]);
            double angle = wrap_angle(vatan2(w) - vatan2(u));
            double residual = angle - e.param;
            double u_squared_recip = 1. / hypot2(u);
            double v_squared_recip = 1. / hypot2(w);
            double a1x = -u.y * u_squared_recip;
            double a1y = u.x * u_squared_recip;
            double a2x = w.y * v_squared_recip;
            double a2y = -w.x * v_squared_recip;
            g[0] = a1x; g[1] = a1y; g[2] = -a1x; g[3] = -a1y;
            g[4] = a2x; g[5] = a2y; g[6] = -a2x; g[7] = -a2y;
            return residual;
        }
        case LineLineParallelism: {  // :716-751
            V2 u = sub(v[2], v[3], v[0], v[1]);
            V2 w = sub(v[6], v[7], v[4], v[5]);
            double residual = cross(w, u);
            g[0] = w.y; g[1] = -w.x; g[2] = -w.y; g[3] = w.x;
            g[4] = -u.y; g[5] = u.x; g[6] = u.y; g[7] = -u.x;
            return residual;
        }
        case LineLinePerpendicularity: {  // :772-798
            V2 u = sub(v[2], v[3], v[0], v[1]);
            V2 w = sub(v[6], v[7], v[4], v[5]);
            double residual = dot(w, u);
            g[0] = -w.x; g[1] = -w.y; g[2] = w.x; g[3] = w.y;
            g[4] = -u.x; g[5] = -u.y; g[6] = u.x; g[7] = u.y;
            return residual;
        }
        case LineCircleTangency: {  // :819-873
            double l1x = v[0], l1y = v[1], l2x = v[2], l2y = v[3], cx = v[4], cy = v[5];
            double circle_radius = v[6];
            double length2 = hypot2(sub(l1x, l1y, l2x, l2y));
            double length = std::sqrt(length2);
            if (length == 0.) {  // :838-840
                for (int i = 0; i < 7; ++i) g[i] = 0.;
                return 0.;
            }
            double length_recip = 1. / length;
            double signed_area = l1x * (l2y - cy) + l2x * (cy - l1y) + cx * (l1y - l2y);
            double residual = length_recip * std::fabs(signed_area) - circle_radius;
            // f64::signum: +1 for +0.0 and positives, -1 for -0.0 and negatives, NaN for NaN.
            double sign = std::isnan(signed_area) ? signed_area : (std::signbit(signed_area) ? -1. : 1.);
            double length3_recip = 1. / (length2 * length);
            g[0] = sign * length3_recip * (length2 * (l2y - cy) + signed_area * (l2x - l1x));
            g[1] = sign * length3_recip * (length2 * (-l2x + cx) + signed_area * (l2y - l1y));
            g[2] = sign * length3_recip * (length2 * (cy - l1y) - signed_area * (l2x - l1x));
            g[3] = sign * length3_recip * (length2 * (l1x - cx) - signed_area * (l2y - l1y));
            g[4] = sign * length_recip * (l1y - l2y);
            g[5] = sign * length_recip * (-l1x + l2x);
            g[6] = -1.;
            return residual;
        }
        default:
            return 0.;
    }
}

}  // namespace fo
