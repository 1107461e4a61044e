// atan2 with the correctly rounded result (gfx950 device code and plain host C++ from the same source).
//
// Why it exists: the reference's two angle residuals are atan2 differences (expressions.rs:393-399, :665-671), and
// Rust's f64::atan2 is the platform libm's — glibc's, which returns the correctly rounded value. The device libm's
// atan2 is within an ulp of that, not equal to it, and one ulp in a residual is enough to send an ill-conditioned
// LM path elsewhere. FX_STEP_QR, whose linear algebra replays the reference's operation by operation, evaluates
// its angle residuals with this routine instead: the whole solve then carries the reference's bits.
//
// How: everything in double-double arithmetic (two_sum / two_prod with fma, ~106 bits). With a = min(|x|, |y|),
// b = max(|x|, |y|): q = a / b; c = the nearest multiple of 1/64; t = (q - c) / (1 + q c), |t| <= 2^-7;
// atan(q) = atan(c) [table, tools/gen_atan_table.py] + atan(t) [odd series through t^17, the low-order terms in
// double-double]; then the octant (pi/2 - r, pi - r) and the sign. The result carries an error near 2^-100
// relative, so rounding it to double gives the correctly rounded atan2 unless the exact value lies within ~2^-47
// ulp of a rounding boundary. tests/test_atan2.py compares it with glibc on millions of arguments (CPU build).
// Zeros, infinities, NaN and exponents beyond +-900 take the platform's atan2 (exact special values).
#pragma once
#include <math.h>

#include "fx_atan2_table.h"

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define FX_HD __host__ __device__ inline
#else
#define FX_HD inline
#endif

namespace fx {
namespace atan2cr {

#if defined(__HIP_DEVICE_COMPILE__)
static __device__ __constant__ const double ATAN_TAB_DEV[65][2] = {FX_ATAN_TAB_VALUES};
#define FX_ATAN_TAB ATAN_TAB_DEV
#else
static const double ATAN_TAB_HOST[65][2] = {FX_ATAN_TAB_VALUES};
#define FX_ATAN_TAB ATAN_TAB_HOST
#endif

struct dd {
    double h, l;
};
FX_HD dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return dd{s, (a - (s - bb)) + (b - bb)};
}
FX_HD dd quick_two_sum(double a, double b) {  // |a| >= |b|
    const double s = a + b;
    return dd{s, b - (s - a)};
}
FX_HD dd two_prod(double a, double b) {
    const double p = a * b;
    return dd{p, ::fma(a, b, -p)};
}
FX_HD dd add(dd a, dd b) {
    dd s = two_sum(a.h, b.h);
    const dd t = two_sum(a.l, b.l);
    s.l += t.h;
    s = quick_two_sum(s.h, s.l);
    s.l += t.l;
    return quick_two_sum(s.h, s.l);
}
FX_HD dd neg(dd a) { return dd{-a.h, -a.l}; }
FX_HD dd mul(dd a, dd b) {
    dd p = two_prod(a.h, b.h);
    p.l += a.h * b.l + a.l * b.h;
    return quick_two_sum(p.h, p.l);
}
FX_HD dd mul(dd a, double b) {
    dd p = two_prod(a.h, b);
    p.l += a.l * b;
    return quick_two_sum(p.h, p.l);
}
FX_HD dd div(dd a, dd b) {  // three quotient digits, each corrected against the exact remainder
    const double q1 = a.h / b.h;
    dd r = add(a, neg(mul(b, q1)));
    const double q2 = r.h / b.h;
    r = add(r, neg(mul(b, q2)));
    const double q3 = r.h / b.h;
    dd q = quick_two_sum(q1, q2);
    return add(q, dd{q3, 0.0});
}

// atan of q in [0, 1] (double-double in, double-double out)
FX_HD dd atan_unit(dd q) {
    const double ic = ::rint(q.h * 64.0);
    const int i = (int)ic;
    const double c = ic * (1.0 / 64.0);
    dd t = q;
    if (i != 0) {
        const dd num = add(q, dd{-c, 0.0});
        const dd den = add(mul(q, c), dd{1.0, 0.0});
        t = div(num, den);
    }
    const dd s = mul(t, t);
    // atan(t) = t + t s (-1/3 + s (1/5 + s (-1/7 + s D))), D = 1/9 - s/11 + s^2/13 - s^3/15 + s^4/17 in double
    const double sh = s.h;
    const double D = 1.0 / 9.0 + sh * (-1.0 / 11.0 + sh * (1.0 / 13.0 + sh * (-1.0 / 15.0 + sh * (1.0 / 17.0))));
    dd e = add(dd{-SEVENTH_H, -SEVENTH_L}, mul(s, D));
    e = add(dd{FIFTH_H, FIFTH_L}, mul(s, e));
    e = add(dd{-THIRD_H, -THIRD_L}, mul(s, e));
    const dd r = add(t, mul(mul(t, s), e));
    return add(dd{FX_ATAN_TAB[i][0], FX_ATAN_TAB[i][1]}, r);
}

}  // namespace atan2cr

FX_HD double atan2_cr(double y, double x) {
    using namespace atan2cr;
    const double ax = ::fabs(x), ay = ::fabs(y);
    // special values and extreme exponents: the platform routine (exact results, or far from anything a sketch holds)
    if (!(ax > 1e-270 && ax < 1e270 && ay > 1e-270 && ay < 1e270)) return ::atan2(y, x);
    const bool swap = ay > ax;
    const double lo = swap ? ax : ay, hi = swap ? ay : ax;
    const double q1 = lo / hi;
    if (q1 < 1e-200) return ::atan2(y, x);
    const double q2 = ::fma(-q1, hi, lo) / hi;  // the remainder is exact
    dd r = atan_unit(quick_two_sum(q1, q2));
    if (swap) r = add(dd{PI_2_H, PI_2_L}, neg(r));
    if (x < 0.0) r = add(dd{PI_H, PI_L}, neg(r));
    return (y < 0.0) ? -r.h : r.h;
}

}  // namespace fx
#undef FX_HD
