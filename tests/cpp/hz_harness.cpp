// Test harness (CPU): Optimizer::LBfgs with the PRODUCT's line-search state machine
// (fiksi_amd/csrc/fx_lbfgs.h) driven by the oracle's evaluation, next to the oracle's line-by-line
// restatement of the nested reference code. Both must visit the same trial points and return
// bit-identical variables. Built by tests/test_lbfgs.py into its own shared library; it pulls in the
// oracle's C API source for the batch plumbing (make_system, parallel_for).
#include "../../oracle/fo_capi.cpp"
#include "../../fiksi_amd/csrc/fx_lbfgs.h"

namespace {

// fo::lbfgs (oracle/fo_lbfgs.hpp) with the line search replaced by fx::HzMachine
fo::LbfgsStats lbfgs_with_machine(const fo::Subsystem& problem, double* variables_inout) {
    using namespace fo::lbfgs_detail;
    constexpr uint32_t MAX_HISTORY = 5, MAX_ITERATIONS = 100;
    fo::LbfgsStats st;
    const size_t nv = problem.num_variables(), ne = problem.num_residuals();
    std::vector<double> variables(variables_inout, variables_inout + nv);
    std::vector<double> residuals(ne, 0.), jacobian(ne * nv, 0.);
    residuals_and_dense_jacobian(problem, variables.data(), residuals.data(), jacobian.data());
    st.evaluations = 1;
    double prev = sum_squares(residuals);
    st.sse_initial = st.sse_final = prev;
    if (prev < 1e-4) {
        st.exit = fo::LBFGS_EXIT_RESIDUAL;
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
            double dp = 0.;
            for (size_t j = 0; j < nv; ++j) dp += s_history[h * nv + j] * direction[j];
            alpha[i] = rho_history[h] * dp;
            for (size_t j = 0; j < nv; ++j) direction[j] -= alpha[i] * y_history[h * nv + j];
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
            double dp = 0.;
            for (size_t j = 0; j < nv; ++j) dp += y_history[h * nv + j] * direction[j];
            double beta = rho_history[h] * dp;
            for (size_t j = 0; j < nv; ++j) direction[j] += s_history[h * nv + j] * (alpha[i] - beta);
        }
        for (double& d : direction) d *= -1.;
        const size_t h = k % MAX_HISTORY;
        for (size_t j = 0; j < nv; ++j) y_history[h * nv + j] = gradient[j];
        scratch = variables;

        // ---- the part under test
        Eval ev{problem, variables, scratch, jacobian, residuals, gradient, direction};
        fx::HzMachine hz;
        double p = hz.start(prev, dot_product(gradient, direction));
        fx::HzParam c{0., 0., 0.};
        for (;;) {
            Param r = ev.calculate_phi(p);
            if (hz.feed(fx::HzParam{r.p, r.phi, r.dphi}, p, c)) break;
        }
        // ----

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
            st.exit = fo::LBFGS_EXIT_CAPPED;
            break;
        }
        if (std::fabs(prev - c.phi) < 1e-10) {
            st.exit = fo::LBFGS_EXIT_STALLED;
            break;
        }
        if (c.phi < 1e-6) {
            st.exit = fo::LBFGS_EXIT_RESIDUAL;
            break;
        }
        prev = c.phi;
    }
    for (size_t j = 0; j < nv; ++j) variables_inout[j] = variables[j];
    return st;
}

}  // namespace

// Bare L-BFGS (no scaling, no perturbation) of every component of every System, with the oracle's
// line search (which = 0) or the product's machine (which = 1). vars are updated in place.
extern "C" int hz_lbfgs_batch(uint32_t n_systems, const uint32_t* var_off, const uint32_t* expr_off, double* vars,
                              const uint8_t* var_fixed, const uint8_t* expr_tag, const uint32_t* expr_idx,
                              const double* expr_param, const uint16_t* var_comp, const uint16_t* expr_comp, int which,
                              uint32_t* iterations, uint32_t* evaluations, uint32_t* exits) {
    for (uint32_t s = 0; s < n_systems; ++s) {
        FlatSystem sys = make_system(s, var_off, expr_off, vars, var_fixed, expr_tag, expr_idx, expr_param, var_comp, expr_comp);
        std::vector<double> snapshot = sys.variables;
        iterations[s] = evaluations[s] = 0;
        for (const Component& comp : sys.components) {
            if (comp.variables.empty()) continue;
            Subsystem sub;
            sub.system_variables = snapshot.data();
            sub.all_expressions = sys.expressions.data();
            sub.expressions = comp.expressions;
            sub.free_index.assign(sys.variables.size(), -1);
            for (uint32_t v : comp.variables)
                if (!sys.fixed[v]) {
                    sub.free_index[v] = static_cast<int32_t>(sub.free_variables.size());
                    sub.free_variables.push_back(v);
                }
            std::vector<double> x(sub.free_variables.size());
            for (size_t k = 0; k < x.size(); ++k) x[k] = snapshot[sub.free_variables[k]];
            LbfgsStats st = which ? lbfgs_with_machine(sub, x.data()) : lbfgs(sub, x.data());
            iterations[s] += st.iterations;
            evaluations[s] += st.evaluations;
            exits[s] = st.exit;
            for (size_t k = 0; k < x.size(); ++k) vars[var_off[s] + sub.free_variables[k]] = x[k];
        }
    }
    return 0;
}
