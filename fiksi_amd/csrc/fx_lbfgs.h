// Hager-Zhang line search of `Optimizer::LBfgs` as a resumable state machine.
//
// Reference: fiksi/src/solve/lbfgs.rs:218-506 (`mod hager_zhang`): strong / approximate Wolfe test
// (:305-320), `update` with its U3 bisection loop (:323-362), `secant2` (:369-406), the fixed [0, 5]
// initial bracket (:410-419), `search` (:423-449) and `run` (:453-463).
//
// The reference evaluates phi(p) = |r(x + p d)|^2 from eight call sites scattered over nested
// functions. On the device one evaluation is a whole residual + Jacobian pass of the wavefront, so the
// search is turned inside out: the caller owns the single evaluation site and this machine says which
// step length to try next —
//
//     double p = hz.start(phi0, dphi0);
//     for (;;) { HzParam r = evaluate(p); if (hz.feed(r, p, result)) break; }
//
// — taking exactly the reference's decisions in the reference's order, so the sequence of trial points
// is identical. When `feed` returns true the last point evaluated is `result.p` (the reference's
// guarantee, :445-447). Scalar code, no memory: every lane of the wavefront runs it redundantly on
// wave-uniform values. Also compiles as plain C++ (a CPU-side unit test drives it against a
// line-by-line restatement of the nested reference code, tests/cpp/hz_harness.cpp).
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define FX_HD __host__ __device__ inline
#else
#define FX_HD inline
#endif

namespace fx {

struct HzParam {  // lbfgs.rs:247-255
    double p, phi, dphi;
};

struct HzMachine {
    // lbfgs.rs:223-245
    static constexpr double DELTA = 1e-4, SIGMA = 0.9, EPSILON = 1e-6, THETA = 0.5, GAMMA = 0.66;
    static constexpr uint32_t MAX_ITERATIONS = 100;
    // The reference's U3 loop (:343-358) has no exit when the objective is discontinuous (an angle
    // residual wrapping) or NaN; it is cut after this many bisections and `capped` is raised.
    static constexpr uint32_t U3_CAP = 200;

    enum State : int { S_FIRST, S_BRACKET, S_SEC1, S_SEC2, S_BISECT, S_U3, S_FINAL };
    enum Cont : int { K_AFTER_SEC1, K_AFTER_SEC2, K_AFTER_BISECT };

    double phi0, dphi0;
    HzParam a, b, c;     // bracket and the point `search` would return at the end (:424, :436)
    HzParam c1;          // secant2's first secant point
    HzParam a_, b_;      // secant2's intermediate bracket
    HzParam aa, bb;      // U3 working bracket
    HzParam oa, ob;      // result of the last `update`
    int state, cont;
    uint32_t it, u3_it;
    bool capped;

    FX_HD bool satisfies_wolfe(HzParam q) const {  // :305-320
        if ((q.phi <= phi0 + q.p * (DELTA * dphi0)) && (q.dphi >= SIGMA * dphi0)) return true;
        if (q.phi <= phi0 + EPSILON && (2. * DELTA - 1.) * dphi0 >= q.dphi && q.dphi >= SIGMA * dphi0) return true;
        return false;
    }
    static FX_HD double secant(HzParam x, HzParam y) { return (x.p * y.dphi - y.p * x.dphi) / (y.dphi - x.dphi); }  // :289-291

    // `run` (:453-463): the first trial is the unit step.
    FX_HD double start(double phi_at_0, double dphi_at_0) {
        phi0 = phi_at_0;
        dphi0 = dphi_at_0;
        state = S_FIRST;
        cont = K_AFTER_SEC1;
        it = 0;
        u3_it = 0;
        capped = false;
        a = b = c = c1 = a_ = b_ = aa = bb = oa = ob = HzParam{0., 0., 0.};
        return 1.;
    }

    // Takes the evaluation of the step length handed out last. Returns true when the search is over
    // (`out` is the accepted point, and it is the last one evaluated), false with the next step length
    // in `next_p`.
    FX_HD bool feed(HzParam r, double& next_p, HzParam& out) {
        switch (state) {
            case S_FIRST:
                if (satisfies_wolfe(r)) {
                    out = r;
                    return true;
                }
                c = r;
                a = HzParam{0., phi0, dphi0};  // `bracket` (:410-419)
                next_p = 5.;
                state = S_BRACKET;
                return false;
            case S_BRACKET:
                b = r;
                it = 0;
                next_p = secant(a, b);
                state = S_SEC1;
                return false;
            case S_SEC1:  // secant2 (:374-381)
                c1 = r;
                if (satisfies_wolfe(r)) {
                    out = r;
                    return true;
                }
                if (!begin_update(a, b, r, K_AFTER_SEC1, next_p)) return false;
                return resume(next_p, out);
            case S_SEC2:  // :385-403
                if (satisfies_wolfe(r)) {
                    out = r;
                    return true;
                }
                if (!begin_update(a_, b_, r, K_AFTER_SEC2, next_p)) return false;
                return resume(next_p, out);
            case S_BISECT:  // :434-441
                c = r;
                if (satisfies_wolfe(r)) {
                    out = r;
                    return true;
                }
                if (!begin_update(a, b, r, K_AFTER_BISECT, next_p)) return false;
                return resume(next_p, out);
            case S_U3:  // :343-358
                if (r.dphi >= 0.) {
                    oa = aa;
                    ob = r;
                    return resume(next_p, out);
                }
                if (r.phi <= phi0 + EPSILON) aa = r; else bb = r;
                u3_it += 1;
                if (u3_it >= U3_CAP) {
                    capped = true;
                    oa = aa;
                    ob = bb;
                    return resume(next_p, out);
                }
                next_p = (1. - THETA) * aa.p + THETA * bb.p;
                return false;
            default:  // S_FINAL (:445-448)
                out = c;
                return true;
        }
    }

  private:
    // `update` (:323-362). True: finished at once, (oa, ob) set. False: U3 needs an evaluation.
    FX_HD bool begin_update(HzParam ua, HzParam ub, HzParam uc, int k, double& next_p) {
        cont = k;
        if (uc.p < ua.p || uc.p > ub.p) {  // U0
            oa = ua;
            ob = ub;
            return true;
        }
        if (uc.dphi >= 0.) {  // U1
            oa = ua;
            ob = uc;
            return true;
        }
        if (uc.phi <= phi0 + EPSILON) {  // U2
            oa = uc;
            ob = ub;
            return true;
        }
        aa = ua;  // U3
        bb = uc;
        u3_it = 0;
        state = S_U3;
        next_p = (1. - THETA) * aa.p + THETA * bb.p;
        return false;
    }

    // what follows an `update` at its three call sites
    FX_HD bool resume(double& next_p, HzParam& out) {
        HzParam na, nb;
        switch (cont) {
            case K_AFTER_SEC1:
                a_ = oa;
                b_ = ob;
                if (c1.p == b_.p) {  // :384
                    next_p = secant(b, b_);
                    state = S_SEC2;
                    return false;
                }
                if (c1.p == a_.p) {  // :394
                    next_p = secant(a, a_);
                    state = S_SEC2;
                    return false;
                }
                na = a_;
                nb = b_;
                break;
            case K_AFTER_SEC2:
                na = oa;
                nb = ob;
                break;
            default:  // K_AFTER_BISECT (:441)
                a = oa;
                b = ob;
                return next_iteration(next_p);
        }
        // back in `search` with secant2's bracket (:433-444)
        if (nb.p - na.p > GAMMA * (b.p - a.p)) {
            next_p = 0.5 * (a.p + b.p);
            state = S_BISECT;
            return false;
        }
        a = na;
        b = nb;
        (void)out;
        return next_iteration(next_p);
    }

    FX_HD bool next_iteration(double& next_p) {
        it += 1;
        if (capped || it >= MAX_ITERATIONS) {
            next_p = c.p;  // :445-447
            state = S_FINAL;
            return false;
        }
        next_p = secant(a, b);
        state = S_SEC1;
        return false;
    }
};

// Exit reasons of the L-BFGS driver, in fx_result.exit terms (FX_EXIT_*):
//   start SSE < 1e-4 or SSE < 1e-6 after a step  -> FX_EXIT_SSE        (lbfgs.rs:54-56, :186-188)
//   |SSE change| < 1e-10                          -> FX_EXIT_FTOL       (:183-185)
//   100 iterations                                -> FX_EXIT_MAX_OUTER  (:32)
//   U3 bisection cut                              -> FX_EXIT_TRIAL_CAP
struct LbfgsConst {
    static constexpr uint32_t MAX_HISTORY = 5, MAX_ITERATIONS = 100;
    static constexpr double START_THRESHOLD = 1e-4, CONVERGENCE_THRESHOLD = 1e-10, RESIDUAL_THRESHOLD = 1e-6;
};

}  // namespace fx
