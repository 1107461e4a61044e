// Per-expression residual + Jacobian-row math for the HIP kernels (device code, gfx950).
//
// Implements the eleven expression variants of fiksi (reference:
// fiksi/src/constraints/expressions.rs:291-874; variable order :48-182). The floating-point
// operation order follows the reference formula by formula, and this file is compiled with
// -ffp-contract=off, so that for f64 every residual and partial is bit-identical to the reference
// arithmetic (sqrt and division are correctly rounded on gfx950); the only ulp-level difference
// is atan2 in the two angle variants' residuals: the device libm's (<= 1 ulp) in K1 and the normal-equation
// kernels, the correctly rounded atan2_cr (fx_atan2.h, CR = true) in the FX_STEP_QR kernels.
#pragma once
#ifdef FX_HOST_ONLY
#include "fx_hip_shim.h"
#else
#include <hip/hip_runtime.h>
#endif
#include <stdint.h>

#include "fx_atan2.h"

#define FX_TAG_VVE 0
#define FX_TAG_PPD 1
#define FX_TAG_PPPA 2
#define FX_TAG_PLI 3
#define FX_TAG_PLD 4
#define FX_TAG_PCI 5
#define FX_TAG_SSLE 6
#define FX_TAG_LLA 7
#define FX_TAG_LLP 8
#define FX_TAG_LLPERP 9
#define FX_TAG_LCT 10
#define FX_NTAGS 11
// Rows of a RecursiveAssembly cluster problem (fiksi/src/assemble/mod.rs:547-588): a point of a solved
// cluster, moved by the cluster's pose, must land on the point's new position. Internal to the library:
// only the RecursiveAssembly arm (fx_recursive.h) builds them, only the POSE instantiations evaluate them.
// Fields: [0] the pose (rotation, tx, ty: three consecutive variables), [1] the point as solved so far
// (two consecutive fixed variables), [2] the coordinate of the point's new position the row is about.
#define FX_TAG_POSE_X 11
#define FX_TAG_POSE_Y 12
#define FX_NTAGS_POSE 13

namespace fx {

// Number of scalar variables an expression reads (expressions.rs:48-182).
template <bool POSE = false>
__host__ __device__ inline int tag_nvars(int tag) {
    if (POSE && tag >= FX_TAG_POSE_X) return 6;
    switch (tag) {
        case FX_TAG_VVE: return 2;
        case FX_TAG_PPD: return 4;
        case FX_TAG_PPPA:
        case FX_TAG_PLI:
        case FX_TAG_PLD: return 6;
        case FX_TAG_PCI: return 5;
        case FX_TAG_LCT: return 7;
        default: return 8;
    }
}

// Expands the (up to four) element fields of an expression into the system-local indices of
// the scalar variables it reads, in gradient order (expressions.rs:48-182). Points contribute
// (idx, idx+1); entries beyond the variable count are 0. Written without arrays-in-switch so the
// result stays in registers.
template <bool POSE = false, typename I>
__host__ __device__ inline int expand_vars(int tag, const I f[4], uint32_t out[8]) {
    const uint32_t a = f[0], b = f[1], c = f[2], d = f[3];
    if (POSE && tag >= FX_TAG_POSE_X) {
        out[0] = a; out[1] = a + 1; out[2] = a + 2;
        out[3] = b; out[4] = b + 1;
        out[5] = c;
        out[6] = 0; out[7] = 0;
        return 6;
    }
    const int k = tag_nvars(tag);
    out[0] = a;
    out[1] = (tag == FX_TAG_VVE) ? b : a + 1;  // VariableVariableEquality holds two scalars
    out[2] = (k > 2) ? b : 0;
    out[3] = (k > 3) ? b + 1 : 0;
    out[4] = (k > 4) ? c : 0;      // PointCircleIncidence: c is the radius (k == 5)
    out[5] = (k > 5) ? c + 1 : 0;
    out[6] = (k > 6) ? d : 0;      // LineCircleTangency: d is the radius (k == 7)
    out[7] = (k > 7) ? d + 1 : 0;
    return k;
}

template <typename T> struct Math;
template <> struct Math<double> {
    static __device__ __forceinline__ double sqrt_(double x) { return ::sqrt(x); }
    static __device__ __forceinline__ double atan2_(double y, double x) { return ::atan2(y, x); }
    template <bool CR>
    static __device__ __forceinline__ double atan2_sel(double y, double x) {
        if constexpr (CR) return atan2_cr(y, x);
        else return ::atan2(y, x);
    }
    static __device__ __forceinline__ double abs_(double x) { return ::fabs(x); }
    static __device__ __forceinline__ double pi() { return 3.14159265358979323846264338327950288; }
};
template <> struct Math<float> {
    static __device__ __forceinline__ float sqrt_(float x) { return ::sqrtf(x); }
    static __device__ __forceinline__ float atan2_(float y, float x) { return ::atan2f(y, x); }
    template <bool CR>
    static __device__ __forceinline__ float atan2_sel(float y, float x) { return ::atan2f(y, x); }
    static __device__ __forceinline__ float abs_(float x) { return ::fabsf(x); }
    static __device__ __forceinline__ float pi() { return 3.14159265358979323846f; }
};

// distance(p1,p2) - target; partials w.r.t. (p1x, p1y, p2x, p2y)   [expressions.rs:327-352]
template <typename T>
__device__ __forceinline__ T point_point_distance(T p1x, T p1y, T p2x, T p2y, T target, T& g0, T& g1, T& g2, T& g3) {
    T dx = p1x - p2x, dy = p1y - p2y;
    T dist = Math<T>::sqrt_(dx * dx + dy * dy);
    T inv = T(1) / dist;
    g0 = dx * inv;
    g1 = dy * inv;
    g2 = -dx * inv;
    g3 = -dy * inv;
    return dist - target;
}

// single-step wrap into (-pi, pi]   [expressions.rs:393-399, 665-671]
template <typename T>
__device__ __forceinline__ T wrap_pi(T a) {
    const T pi = Math<T>::pi();
    if (a > pi) return a - T(2) * pi;
    if (a < -pi) return a + T(2) * pi;
    return a;
}

// Residual + gradient of one expression. v[] = gathered values in expand_vars order, g[] = partials
// in the same order (entries >= tag_nvars(tag) are left untouched). WANT_G=false drops the
// gradient arithmetic (residual-only evaluation, expressions.rs:883-961).
// CR = true: the two angle residuals use the correctly rounded atan2 of fx_atan2.h instead of the device libm's
// (FX_STEP_QR: every bit of the solve then follows from IEEE arithmetic alone).
// POSE = true adds the two pose rows (Pose2D::transform_point / gradient_chain_rule_point,
// expressions.rs:1120-1157, as assemble/mod.rs:547-588 uses them); they cost a sincos, so only the
// cluster-problem instantiations carry them.
template <typename T, bool WANT_G, bool CR = false, bool POSE = false>
__device__ __forceinline__ T eval_expression(int tag, const T v[8], T param, T g[8]) {
    if constexpr (POSE) {
        if (tag >= FX_TAG_POSE_X) {
            T sn, cs;
            if constexpr (sizeof(T) == 8) ::sincos(v[0], &sn, &cs);
            else ::sincosf(v[0], &sn, &cs);
            const T pu = v[3], pv = v[4];
            const T uc = pu * cs, us = pu * sn, vc = pv * cs, vs = pv * sn;
            const bool is_x = tag == FX_TAG_POSE_X;
            const T gx = is_x ? T(1) : T(0), gy = is_x ? T(0) : T(1);
            if (WANT_G) {
                g[0] = (-us - vc) * gx + (uc - vs) * gy;
                g[1] = gx;
                g[2] = gy;
                g[3] = T(0);
                g[4] = T(0);
                g[5] = T(-1);
            }
            const T moved = is_x ? (v[1] + uc - vs) : (v[2] + us + vc);
            return moved - v[5];
        }
    }
    switch (tag) {
        case FX_TAG_VVE: {  // expressions.rs:294-300
            if (WANT_G) { g[0] = T(-1); g[1] = T(1); }
            return v[1] - v[0];
        }
        case FX_TAG_PPD: {
            T a, b, c, d;
            T r = point_point_distance<T>(v[0], v[1], v[2], v[3], param, a, b, c, d);
            if (WANT_G) { g[0] = a; g[1] = b; g[2] = c; g[3] = d; }
            return r;
        }
        case FX_TAG_PPPA: {  // expressions.rs:375-424: angle at p2 from (p1-p2) to (p3-p2)
            T ux = v[0] - v[2], uy = v[1] - v[3];
            T wx = v[4] - v[2], wy = v[5] - v[3];
            T ang = wrap_pi<T>(Math<T>::template atan2_sel<CR>(wy, wx) - Math<T>::template atan2_sel<CR>(uy, ux));
            if (WANT_G) {
                T ur = T(1) / (ux * ux + uy * uy);
                T wr = T(1) / (wx * wx + wy * wy);
                T d1x = uy * ur, d1y = -ux * ur;
                T d3x = -wy * wr, d3y = wx * wr;
                g[0] = d1x; g[1] = d1y;
                g[2] = -d1x - d3x; g[3] = -d1y - d3y;
                g[4] = d3x; g[5] = d3y;
            }
            return ang - param;
        }
        case FX_TAG_PLI: {  // expressions.rs:448-476: cross(l2-l1, p-l1)
            T px = v[0], py = v[1], l1x = v[2], l1y = v[3], l2x = v[4], l2y = v[5];
            T ux = l2x - l1x, uy = l2y - l1y;
            T wx = px - l1x, wy = py - l1y;
            if (WANT_G) {
                g[0] = -uy; g[1] = ux;
                g[2] = -py + l2y; g[3] = px - l2x;
                g[4] = wy; g[5] = -wx;
            }
            return ux * wy - uy * wx;
        }
        case FX_TAG_PLD: {  // expressions.rs:503-543: signed distance of p to line (l1,l2)
            T px = v[0], py = v[1], l1x = v[2], l1y = v[3], l2x = v[4], l2y = v[5];
            T ux = l2x - l1x, uy = l2y - l1y;
            T wx = px - l1x, wy = py - l1y;
            T cr = ux * wy - uy * wx;
            T len2 = ux * ux + uy * uy;
            T len = Math<T>::sqrt_(len2);
            T linv = T(1) / len;
            if (WANT_G) {
                T a = cr / len2;
                T b = -a * ux;
                T c = px + a * uy;
                g[0] = -linv * uy;
                g[1] = linv * ux;
                g[2] = -linv * (b - l2y + py);
                g[3] = -linv * (l2x - c);
                g[4] = linv * (b + wy);
                g[5] = -linv * (c - l1x);
            }
            return linv * cr - param;
        }
        case FX_TAG_PCI: {  // expressions.rs:563-575: distance(p, center) - radius variable
            T a, b, c, d;
            T r = point_point_distance<T>(v[0], v[1], v[2], v[3], v[4], a, b, c, d);
            if (WANT_G) { g[0] = a; g[1] = b; g[2] = c; g[3] = d; g[4] = T(-1); }
            return r;
        }
        case FX_TAG_SSLE: {  // expressions.rs:596-619: |s2| - |s1|
            T a0, a1, a2, a3, b0, b1, b2, b3;
            T r1 = point_point_distance<T>(v[0], v[1], v[2], v[3], T(0), a0, a1, a2, a3);
            T r2 = point_point_distance<T>(v[4], v[5], v[6], v[7], T(0), b0, b1, b2, b3);
            if (WANT_G) {
                g[0] = -a0; g[1] = -a1; g[2] = -a2; g[3] = -a3;
                g[4] = b0; g[5] = b1; g[6] = b2; g[7] = b3;
            }
            return r2 - r1;
        }
        case FX_TAG_LLA: {  // expressions.rs:643-695: angle from line1 direction to line2 direction
            T ux = v[2] - v[0], uy = v[3] - v[1];
            T wx = v[6] - v[4], wy = v[7] - v[5];
            T ang = wrap_pi<T>(Math<T>::template atan2_sel<CR>(wy, wx) - Math<T>::template atan2_sel<CR>(uy, ux));
            if (WANT_G) {
                T ur = T(1) / (ux * ux + uy * uy);
                T wr = T(1) / (wx * wx + wy * wy);
                T a1x = -uy * ur, a1y = ux * ur;
                T a2x = wy * wr, a2y = -wx * wr;
                g[0] = a1x; g[1] = a1y; g[2] = -a1x; g[3] = -a1y;
                g[4] = a2x; g[5] = a2y; g[6] = -a2x; g[7] = -a2y;
            }
            return ang - param;
        }
        case FX_TAG_LLP: {  // expressions.rs:716-751: cross(v, u)
            T ux = v[2] - v[0], uy = v[3] - v[1];
            T wx = v[6] - v[4], wy = v[7] - v[5];
            if (WANT_G) {
                g[0] = wy; g[1] = -wx; g[2] = -wy; g[3] = wx;
                g[4] = -uy; g[5] = ux; g[6] = uy; g[7] = -ux;
            }
            return wx * uy - wy * ux;
        }
        case FX_TAG_LLPERP: {  // expressions.rs:772-798: dot(v, u)
            T ux = v[2] - v[0], uy = v[3] - v[1];
            T wx = v[6] - v[4], wy = v[7] - v[5];
            if (WANT_G) {
                g[0] = -wx; g[1] = -wy; g[2] = wx; g[3] = wy;
                g[4] = -ux; g[5] = -uy; g[6] = ux; g[7] = uy;
            }
            return wx * ux + wy * uy;
        }
        case FX_TAG_LCT: {  // expressions.rs:819-873: |area(l1,l2,c)|/|l1-l2| - radius
            T l1x = v[0], l1y = v[1], l2x = v[2], l2y = v[3], cx = v[4], cy = v[5], rad = v[6];
            T ex = l1x - l2x, ey = l1y - l2y;
            T len2 = ex * ex + ey * ey;
            T len = Math<T>::sqrt_(len2);
            if (len == T(0)) {  // degenerate line: reference returns (0, 0-vector), :838-840
                if (WANT_G) { g[0] = g[1] = g[2] = g[3] = g[4] = g[5] = g[6] = T(0); }
                return T(0);
            }
            T linv = T(1) / len;
            T area = l1x * (l2y - cy) + l2x * (cy - l1y) + cx * (l1y - l2y);
            if (WANT_G) {
                // Rust f64::signum: +1 for +0.0, -1 for -0.0, NaN stays NaN
                T sgn = (area != area) ? area : (signbit(area) ? T(-1) : T(1));
                T l3inv = T(1) / (len2 * len);
                g[0] = sgn * l3inv * (len2 * (l2y - cy) + area * (l2x - l1x));
                g[1] = sgn * l3inv * (len2 * (-l2x + cx) + area * (l2y - l1y));
                g[2] = sgn * l3inv * (len2 * (cy - l1y) - area * (l2x - l1x));
                g[3] = sgn * l3inv * (len2 * (l1x - cx) - area * (l2y - l1y));
                g[4] = sgn * linv * (l1y - l2y);
                g[5] = sgn * linv * (-l1x + l2x);
                g[6] = T(-1);
            }
            return linv * Math<T>::abs_(area) - rad;
        }
        default:
            return T(0);
    }
}

}  // namespace fx
