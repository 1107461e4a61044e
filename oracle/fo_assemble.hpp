// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates fiksi/src/assemble/mod.rs:32-167 (`calculate_system_scale` and `solve` with
// `Decomposer::None`, Levenberg-Marquardt), fiksi/src/utils.rs:11-33 (root mean squares) and the
// post-solve residual check fiksi/src/constraints/mod.rs:144-168.
//
// The System is given in flat form (what fiksi::System holds after building, lib.rs:256-303):
// all variables, all expressions, the fixed-variable mask, and per connected component (in the
// order of `Graph::connected_components()`, graph.rs:256-258, empty components skipped) the
// ascending lists of its elements' variables and of its expressions.
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "fo_expressions.hpp"
#include "fo_lbfgs.hpp"
#include "fo_lm.hpp"
#include "fo_rand.hpp"

namespace fo {

struct Component {
    std::vector<uint32_t> variables;    // variables of the component's elements (fixed included), ascending
    std::vector<uint32_t> expressions;  // expression ids, ascending
};

struct FlatSystem {
    std::vector<double> variables;
    std::vector<uint8_t> fixed;  // 1 = in System::fixed_variables
    std::vector<Expression> expressions;
    std::vector<Component> components;
};

// assemble/mod.rs:32-44 + utils.rs:11-33: sqrt( (sum v^2 + sum d^2) / count ), sequential sums,
// variables first, then the distance parameters in expression order.
inline double calculate_system_scale(const FlatSystem& s) {
    double sum = 0.;
    size_t n = 0;
    for (double v : s.variables) {
        sum += v * v;
        n += 1;
    }
    for (const Expression& e : s.expressions) {
        if (has_distance_param(e.tag)) {
            sum += e.param * e.param;
            n += 1;
        }
    }
    return std::sqrt(sum / static_cast<double>(n));
}

struct SolveStats {
    double scale = 0.;
    std::vector<LmStats> components;
};

// assemble/mod.rs:46-167 (None arm). Mutates `s.variables` like `System::solve`.
// `optimizer`: 0 = LevenbergMarquardt, 1 = LBfgs (solve/mod.rs:17-27; dispatch assemble/mod.rs:148-158).
inline LmStats run_optimizer(int optimizer, const Subsystem& subsystem, double* free_values, QrOrdering ordering,
                             uint32_t trial_cap) {
    if (optimizer == 1) {
        LbfgsStats b = lbfgs(subsystem, free_values);
        LmStats st;
        st.accepted = b.iterations;
        st.trials = b.evaluations;
        st.exit = b.exit;
        st.sse_initial = b.sse_initial;
        st.sse_final = b.sse_final;
        return st;
    }
    return levenberg_marquardt(subsystem, free_values, ordering, trial_cap);
}

inline SolveStats solve(FlatSystem& s, bool perturb, QrOrdering ordering = QrOrdering::Colamd,
                        uint32_t trial_cap = 0, int optimizer = 0) {
    SolveStats out;
    Rng rng = Rng::from_seed(42);  // :47

    double system_scale = calculate_system_scale(s);
    out.scale = system_scale;
    double system_scale_recip = 1. / system_scale;

    std::vector<double> variables_transformed(s.variables.size());
    for (size_t i = 0; i < s.variables.size(); ++i) variables_transformed[i] = s.variables[i] * system_scale_recip;
    std::vector<Expression> expressions_transformed;
    expressions_transformed.reserve(s.expressions.size());
    for (const Expression& e : s.expressions) expressions_transformed.push_back(transform(e, system_scale_recip));

    for (const Component& component : s.components) {
        if (component.variables.empty()) continue;  // `elements.is_empty()`, :87-89

        std::vector<uint32_t> free_variables;  // BTreeSet: ascending, :91-111
        for (uint32_t v : component.variables) {
            if (!s.fixed[v]) free_variables.push_back(v);
        }

        if (perturb) {  // :113-124 (constants 1/8196 and 1/65568 are the reference's, quirk Q3)
            for (uint32_t fv : free_variables) {
                double& variable = variables_transformed[fv];
                double a = rng.next_f64();
                double b = rng.next_f64();
                variable += variable * (1. / 8196.) * a + (1. / 65568.) * b;
            }
        }

        std::vector<double> free_values(free_variables.size());
        for (size_t k = 0; k < free_variables.size(); ++k) free_values[k] = variables_transformed[free_variables[k]];

        Subsystem subsystem;
        subsystem.system_variables = variables_transformed.data();
        subsystem.all_expressions = expressions_transformed.data();
        subsystem.expressions = component.expressions;
        subsystem.free_variables = free_variables;
        subsystem.free_index.assign(s.variables.size(), -1);
        for (size_t k = 0; k < free_variables.size(); ++k) subsystem.free_index[free_variables[k]] = static_cast<int32_t>(k);

        LmStats st = run_optimizer(optimizer, subsystem, free_values.data(), ordering, trial_cap);
        out.components.push_back(st);

        // :161-166 — only `system.variables` is written back (quirk Q2).
        for (size_t k = 0; k < free_variables.size(); ++k) s.variables[free_variables[k]] = system_scale * free_values[k];
    }
    return out;
}

// Expression::calculate_residual with IdentityVariableMap over unscaled variables
// (expressions.rs:883-961, constraints/mod.rs:144-168 for valency 1).
inline double expression_residual(const Expression& e, const double* variables) {
    uint32_t idx[8];
    double vals[8] = {0, 0, 0, 0, 0, 0, 0, 0}, grad[8];
    int k = variable_indices(e, idx);
    for (int i = 0; i < k; ++i) vals[i] = variables[idx[i]];
    return compute_residual_and_gradient(e, vals, grad);
}

}  // namespace fo
