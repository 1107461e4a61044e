// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates `Optimizer::LBfgs`: fiksi/src/solve/lbfgs.rs:20-193 (L-BFGS, history 5, two-loop recursion),
// :199-216 (gradient J^T r on the dense row-major Jacobian, dot product), :218-506 (Hager-Zhang line
// search: approximate Wolfe conditions, secant2, update with the U3 bisection, fixed [0, 5] initial
// bracket), and the dense Jacobian of fiksi/src/subsystem.rs:106-124 with the scatter of
// fiksi/src/constraints/expressions.rs:993-1008 (a later entry of the same column overwrites).
//
// PARITY UNPINNED for this optimizer: no test, bench or example of the reference selects
// `Optimizer::LBfgs` (it is reachable only through the public `SolvingOptions::optimizer`), so there is
// no golden vector or threshold to hold this restatement against; it follows the source line by line.
// Two quirks of the source are kept as they are: (1) for k < 5 the history ring is read at
// `(k + i) % 5`, which addresses slots that were never written (zeros) for most i, so the first
// iterations use little or none of the curvature history; (2) `update`'s U3 loop has no exit for a
// NaN objective — here it is capped (U3_CAP) so that the oracle terminates, and the cap is reported.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "fo_lm.hpp"

namespace fo {

enum LbfgsExit : uint32_t {
    LBFGS_EXIT_RESIDUAL = 0,   // start SSE < 1e-4 (lbfgs.rs:54-56) or SSE < 1e-6 after a step (:186-188)
    LBFGS_EXIT_STALLED = 2,    // |SSE change| < 1e-10 (:183-185)
    LBFGS_EXIT_MAX_ITER = 3,   // 100 iterations (:32, :83)
    LBFGS_EXIT_CAPPED = 4,     // oracle-only: the uncapped U3 bisection loop was cut
};

struct LbfgsStats {
    uint32_t iterations = 0;  // line searches done
    uint32_t evaluations = 0; // residual + Jacobian evaluations
    uint32_t exit = LBFGS_EXIT_MAX_ITER;
    double sse_initial = 0.;
    double sse_final = 0.;
};

namespace lbfgs_detail {

constexpr uint32_t U3_CAP = 200;

// subsystem.rs:106-124 + expressions.rs:993-1008: row-major, entries of other columns stay zero
inline void residuals_and_dense_jacobian(const Subsystem& p, const double* variables, double* residuals,
                                         double* jacobian) {
    const size_t nv = p.free_variables.size();
    uint32_t idx[8];
    double vals[8] = {0, 0, 0, 0, 0, 0, 0, 0}, grad[8];
    for (size_t row = 0; row < p.expressions.size(); ++row) {
        const Expression& e = p.all_expressions[p.expressions[row]];
        int k = variable_indices(e, idx);
        for (int i = 0; i < k; ++i) vals[i] = p.value_of(idx[i], variables);
        residuals[row] = compute_residual_and_gradient(e, vals, grad);
        for (int i = 0; i < k; ++i) {
            int32_t f = p.free_index[idx[i]];
            if (f >= 0) jacobian[row * nv + static_cast<size_t>(f)] = grad[i];
        }
    }
}

inline double sum_squares(const std::vector<double>& v) {  // utils.rs:11-19
    double s = 0.;
    for (double x : v) s += x * x;
    return s;
}

inline void compute_gradient(const std::vector<double>& jacobian, const std::vector<double>& residuals,
                             std::vector<double>& gradient) {  // lbfgs.rs:199-210
    const size_t nv = gradient.size(), ne = residuals.size();
    for (size_t i = 0; i < nv; ++i) {
        double g = 0.;
        for (size_t c = 0; c < ne; ++c) g += jacobian[c * nv + i] * residuals[c];
        gradient[i] = g;
    }
}

inline double dot_product(const std::vector<double>& a, const std::vector<double>& b) {  // lbfgs.rs:213-216
    double s = 0.;
    for (size_t i = 0; i < a.size(); ++i) s += a[i] * b[i];
    return s;
}

struct Param {  // lbfgs.rs:247-255
    double p, phi, dphi;
};

struct Eval {  // lbfgs.rs:258-285
    const Subsystem& problem;
    const std::vector<double>& variables;
    std::vector<double>& variables_scratch;
    std::vector<double>& jacobian;
    std::vector<double>& residuals;
    std::vector<double>& gradient;
    const std::vector<double>& direction;
    uint32_t evaluations = 0;

    Param calculate_phi(double p) {
        for (size_t i = 0; i < variables.size(); ++i) variables_scratch[i] = variables[i] + p * direction[i];
        residuals_and_dense_jacobian(problem, variables_scratch.data(), residuals.data(), jacobian.data());
        compute_gradient(jacobian, residuals, gradient);
        evaluations += 1;
        return Param{p, sum_squares(residuals), dot_product(gradient, direction)};
    }
};

constexpr double DELTA = 1e-4, SIGMA = 0.9, EPSILON = 1e-6, THETA = 0.5, GAMMA = 0.66;  // lbfgs.rs:223-237
constexpr uint32_t LS_MAX_ITERATIONS = 100;                                              // :245

inline double secant(Param a, Param b) { return (a.p * b.dphi - b.p * a.dphi) / (b.dphi - a.dphi); }  // :289-291

struct HagerZhang {
    double phi0, dphi0;
    bool capped = false;

    bool satisfies_wolfe(Param c) const {  // :305-320
        if ((c.phi <= phi0 + c.p * (DELTA * dphi0)) && (c.dphi >= SIGMA * dphi0)) return true;
        if (c.phi <= phi0 + EPSILON && (2. * DELTA - 1.) * dphi0 >= c.dphi && c.dphi >= SIGMA * dphi0) return true;
        return false;
    }

    void update(Eval& ev, Param a, Param b, Param c, Param& oa, Param& ob) {  // :323-362
        if (c.p < a.p || c.p > b.p) {  // U0
            oa = a;
            ob = b;
            return;
        }
        if (c.dphi >= 0.) {  // U1
            oa = a;
            ob = c;
        } else if (c.phi <= phi0 + EPSILON) {  // U2
            oa = c;
            ob = b;
        } else {  // U3
            Param aa = a, bb = c;
            for (uint32_t it = 0;; ++it) {
                if (it >= U3_CAP) {
                    capped = true;
                    oa = aa;
                    ob = bb;
                    return;
                }
                Param d = ev.calculate_phi((1. - THETA) * aa.p + THETA * bb.p);
                if (d.dphi >= 0.) {
                    oa = aa;
                    ob = d;
                    return;
                } else if (d.phi <= phi0 + EPSILON) {
                    aa = d;
                } else {
                    bb = d;
                }
            }
        }
    }

    // returns true with `out` set when a point satisfying the Wolfe conditions was found; :369-406
    bool secant2(Eval& ev, Param a, Param b, Param& out, Param& oa, Param& ob) {
        Param c = ev.calculate_phi(secant(a, b));
        if (satisfies_wolfe(c)) {
            out = c;
            return true;
        }
        Param a_, b_;
        update(ev, a, b, c, a_, b_);
        if (c.p == b_.p) {
            Param c_ = ev.calculate_phi(secant(b, b_));
            if (satisfies_wolfe(c_)) {
                out = c_;
                return true;
            }
            update(ev, a_, b_, c_, oa, ob);
        } else if (c.p == a_.p) {
            Param c_ = ev.calculate_phi(secant(a, a_));
            if (satisfies_wolfe(c_)) {
                out = c_;
                return true;
            }
            update(ev, a_, b_, c_, oa, ob);
        } else {
            oa = a_;
            ob = b_;
        }
        return false;
    }

    Param run(Eval& ev) {  // :453-463, bracket :410-419, search :423-449
        Param c = ev.calculate_phi(1.);
        if (satisfies_wolfe(c)) return c;
        Param a{0., phi0, dphi0};
        Param b = ev.calculate_phi(5.);
        for (uint32_t it = 0; it < LS_MAX_ITERATIONS; ++it) {
            Param out, a_, b_;
            if (secant2(ev, a, b, out, a_, b_)) return out;
            if (b_.p - a_.p > GAMMA * (b.p - a.p)) {
                c = ev.calculate_phi(0.5 * (a.p + b.p));
                if (satisfies_wolfe(c)) return c;
                Param na, nb;
                update(ev, a, b, c, na, nb);
                a = na;
                b = nb;
            } else {
                a = a_;
                b = b_;
            }
            if (capped) break;
        }
        ev.calculate_phi(c.p);  // :445-447: the buffers must hold the returned point
        return c;
    }
};

}  // namespace lbfgs_detail

// lbfgs.rs:20-193
inline LbfgsStats lbfgs(const Subsystem& problem, double* variables_inout) {
    using namespace lbfgs_detail;
    constexpr uint32_t MAX_HISTORY = 5, MAX_ITERATIONS = 100;
    constexpr double CONVERGENCE_THRESHOLD = 1e-10, RESIDUAL_THRESHOLD = 1e-6;
    LbfgsStats st;
    const size_t nv = problem.num_variables(), ne = problem.num_residuals();
    std::vector<double> variables(variables_inout, variables_inout + nv);
    std::vector<double> residuals(ne, 0.), jacobian(ne * nv, 0.);
    residuals_and_dense_jacobian(problem, variables.data(), residuals.data(), jacobian.data());
    st.evaluations = 1;
    double prev = sum_squares(residuals);
    st.sse_initial = st.sse_final = prev;
    if (prev < 1e-4) {
        st.exit = LBFGS_EXIT_RESIDUAL;
        return st;
    }
    std::vector<double> gradient(nv, 0.);
    compute_gradient(jacobian, residuals, gradient);
    std::vector<double> s_history(nv * MAX_HISTORY, 0.), y_history(nv * MAX_HISTORY, 0.), rho_history(MAX_HISTORY, 0.);
    std::vector<double> alpha(MAX_HISTORY, 0.), direction(nv, 0.), scratch(nv, 0.);

    for (uint32_t k = 0; k < MAX_ITERATIONS; ++k) {
        const uint32_t history_len = k < MAX_HISTORY ? k : MAX_HISTORY;
        direction = gradient;
        for (uint32_t i = history_len; i-- > 0;) {
            const size_t h = (k + i) % MAX_HISTORY;
            const double* s_i = &s_history[h * nv];
            const double* y_i = &y_history[h * nv];
            double dp = 0.;
            for (size_t j = 0; j < nv; ++j) dp += s_i[j] * direction[j];
            alpha[i] = rho_history[h] * dp;
            for (size_t j = 0; j < nv; ++j) direction[j] -= alpha[i] * y_i[j];
        }
        if (k > 0) {
            const size_t h = (k - 1) % MAX_HISTORY;
            double s_dot_y = 0., y_dot_y = 0.;
            for (size_t j = 0; j < nv; ++j) {
                s_dot_y += s_history[h * nv + j] * y_history[h * nv + j];
                y_dot_y += y_history[h * nv + j] * y_history[h * nv + j];
            }
            if (y_dot_y > 0.) {
                double scale = s_dot_y / y_dot_y;
                for (double& d : direction) d *= scale;
            }
        }
        for (uint32_t i = 0; i < history_len; ++i) {
            const size_t h = (k + i) % MAX_HISTORY;
            const double* s_i = &s_history[h * nv];
            const double* y_i = &y_history[h * nv];
            double dp = 0.;
            for (size_t j = 0; j < nv; ++j) dp += y_i[j] * direction[j];
            double beta = rho_history[h] * dp;
            for (size_t j = 0; j < nv; ++j) direction[j] += s_i[j] * (alpha[i] - beta);
        }
        for (double& d : direction) d *= -1.;

        const size_t h = k % MAX_HISTORY;
        for (size_t j = 0; j < nv; ++j) y_history[h * nv + j] = gradient[j];
        scratch = variables;
        HagerZhang hz{prev, dot_product(gradient, direction)};
        Eval ev{problem, variables, scratch, jacobian, residuals, gradient, direction};
        Param c = hz.run(ev);
        st.evaluations += ev.evaluations;
        st.iterations += 1;
        variables = scratch;
        double s_dot_y = 0.;
        for (size_t j = 0; j < nv; ++j) {
            s_history[h * nv + j] = c.p * direction[j];
            y_history[h * nv + j] = gradient[j] - y_history[h * nv + j];
            s_dot_y += s_history[h * nv + j] * y_history[h * nv + j];
        }
        rho_history[h] = 1.0 / s_dot_y;
        st.sse_final = c.phi;
        if (hz.capped) {
            st.exit = LBFGS_EXIT_CAPPED;
            break;
        }
        if (std::fabs(prev - c.phi) < CONVERGENCE_THRESHOLD) {
            st.exit = LBFGS_EXIT_STALLED;
            break;
        }
        if (c.phi < RESIDUAL_THRESHOLD) {
            st.exit = LBFGS_EXIT_RESIDUAL;
            break;
        }
        prev = c.phi;
    }
    for (size_t j = 0; j < nv; ++j) variables_inout[j] = variables[j];
    return st;
}

}  // namespace fo
