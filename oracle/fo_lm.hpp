// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates fiksi/src/subsystem.rs:26-166 + fiksi/src/variable_map.rs:57-72 (the `Problem` the
// optimizer sees) and fiksi/src/solve/lm.rs:21-197 (Levenberg-Marquardt on the augmented system
// [J; sqrt(lambda) I] delta = [-r; 0] with solvi's sparse Householder QR, COLAMD ordering).
#pragma once
#include <cmath>
#include <cstdint>
#include <vector>

#include "fo_expressions.hpp"
#include "fo_qr.hpp"
#include "fo_sparse.hpp"

namespace fo {

// subsystem.rs:9-38. `free_index[v]` = rank of global variable v among the free variables
// (IndexSet insertion order = ascending, assemble/mod.rs:91-111), or -1 when fixed
// (variable_map.rs:57-72: fixed variables read the scaled system snapshot).
struct Subsystem {
    const double* system_variables;       // all (scaled, perturbed) system variables
    const Expression* all_expressions;    // all (scaled) expressions of the System
    std::vector<uint32_t> expressions;    // chosen expression ids, ascending
    std::vector<uint32_t> free_variables; // global indices, ascending
    std::vector<int32_t> free_index;      // size = number of system variables

    uint32_t num_variables() const { return static_cast<uint32_t>(free_variables.size()); }
    uint32_t num_residuals() const { return static_cast<uint32_t>(expressions.size()); }

    double value_of(uint32_t v, const double* free_values) const {
        int32_t f = free_index[v];
        return f >= 0 ? free_values[f] : system_variables[v];
    }

    // subsystem.rs:93-104
    void calculate_residuals(const double* variables, double* residuals) const {
        uint32_t idx[8];
        double vals[8] = {0, 0, 0, 0, 0, 0, 0, 0}, grad[8];
        for (size_t row = 0; row < expressions.size(); ++row) {
            const Expression& e = all_expressions[expressions[row]];
            int k = variable_indices(e, idx);
            for (int i = 0; i < k; ++i) vals[i] = value_of(idx[i], variables);
            residuals[row] = compute_residual_and_gradient(e, vals, grad);
        }
    }

    // subsystem.rs:126-166: COO triplets (row, free column, partial) for free variables only.
    void calculate_residuals_and_sparse_jacobian(const double* variables, double* residuals,
                                                 TripletMat& jacobian) const {
        uint32_t idx[8];
        double vals[8] = {0, 0, 0, 0, 0, 0, 0, 0}, grad[8];
        for (size_t row = 0; row < expressions.size(); ++row) {
            const Expression& e = all_expressions[expressions[row]];
            int k = variable_indices(e, idx);
            for (int i = 0; i < k; ++i) vals[i] = value_of(idx[i], variables);
            residuals[row] = compute_residual_and_gradient(e, vals, grad);
            for (int i = 0; i < k; ++i) {
                int32_t f = free_index[idx[i]];
                if (f >= 0) jacobian.push_triplet(row, static_cast<size_t>(f), grad[i]);
            }
        }
    }
};

// Why the loop ended (the reference returns nothing; this is bookkeeping for parity tests).
enum LmExit : uint32_t {
    LM_EXIT_SSE = 0,       // sum_squared_residuals < 1e-8 (lm.rs:110-112)
    LM_EXIT_STEP = 1,      // |delta|^2 < 1e-12 (lm.rs:139-142)
    LM_EXIT_FTOL = 2,      // relative decrease <= 1e-6 (lm.rs:164-168)
    LM_EXIT_MAX_OUTER = 3, // 100 outer steps used up (lm.rs:109)
    LM_EXIT_TRIAL_CAP = 4, // oracle-only safety cap on the uncapped inner loop (quirk Q8)
};

struct LmStats {
    uint32_t accepted = 0;  // accepted steps = Gauss-Newton iterations (one J evaluation each)
    uint32_t trials = 0;    // factor + solve attempts
    uint32_t exit = LM_EXIT_MAX_OUTER;
    double sse_initial = 0.;
    double sse_final = 0.;  // SSE of the returned point
    double last_delta_norm2 = 0.;
};

inline double norm_squared(const double* v, size_t n) {  // lm.rs:195-197 (sequential sum)
    double s = 0.;
    for (size_t i = 0; i < n; ++i) s += v[i] * v[i];
    return s;
}

// lm.rs:21-193. `ordering` is Colamd in the reference (lm.rs:103); Natural is offered for tests.
// `trial_cap` bounds the reference's unbounded inner loop (0 = uncapped, as in the reference).
// If `first_delta` is non-null it receives the first successfully solved delta (ncols values).
// `Problem` (solve/mod.rs:29-49): num_variables, num_residuals, calculate_residuals,
// calculate_residuals_and_sparse_jacobian — Subsystem above, or the cluster problem of fo_recursive.hpp.
template <class Problem>
inline LmStats levenberg_marquardt(const Problem& problem, double* variables,
                                   QrOrdering ordering = QrOrdering::Colamd, uint32_t trial_cap = 0,
                                   double* first_delta = nullptr) {
    LmStats stats;
    const size_t nrows = problem.num_residuals();
    const size_t ncols = problem.num_variables();

    std::vector<double> variables_scratch(variables, variables + ncols);
    std::vector<double> residuals(nrows, 0.), residuals_scratch(nrows, 0.), b_augmented(nrows + ncols, 0.);

    TripletMat sparse_jacobian(nrows, ncols);
    problem.calculate_residuals_and_sparse_jacobian(variables_scratch.data(), residuals.data(), sparse_jacobian);
    for (double& r : residuals) r = -r;
    for (size_t idx = 0; idx < ncols; ++idx) sparse_jacobian.push_triplet(nrows + idx, idx, 0.);  // lm.rs:92-96
    SparseColMat csc = SparseColMat::from_triplet_mat(sparse_jacobian);

    SymbolicQr sparse_sqr = SymbolicQr::build(csc.structure, ordering);  // once per call, lm.rs:103
    Qr sparse_qr(sparse_sqr);

    double sum_squared_residuals = norm_squared(residuals.data(), nrows);
    stats.sse_initial = sum_squared_residuals;
    stats.sse_final = sum_squared_residuals;

    double lambda = 0.5;
    bool done = false;
    for (int step = 0; step < 100 && !done; ++step) {
        if (sum_squared_residuals < 1e-8) {
            stats.exit = LM_EXIT_SSE;
            break;
        }
        for (;;) {
            if (trial_cap != 0 && stats.trials >= trial_cap) {
                stats.exit = LM_EXIT_TRIAL_CAP;
                done = true;
                break;
            }
            // lm.rs:119-125: the damping entry is the last stored entry of every column.
            double sqrt_lambda = std::sqrt(lambda);
            for (size_t idx = 0; idx < ncols; ++idx) {
                csc.values[csc.structure.column_pointers[idx + 1] - 1] = sqrt_lambda;
            }
            sparse_qr.factorize(csc);
            stats.trials += 1;

            for (size_t i = 0; i < nrows; ++i) b_augmented[i] = residuals[i];
            for (size_t i = nrows; i < nrows + ncols; ++i) b_augmented[i] = 0.;
            bool solved = sparse_qr.solve_mut(b_augmented.data());
            if (!solved) {
                lambda *= 8.;
                continue;
            }
            const double* delta = b_augmented.data();
            if (first_delta) {
                for (size_t i = 0; i < ncols; ++i) first_delta[i] = delta[i];
                first_delta = nullptr;
            }
            double dn2 = norm_squared(delta, ncols);
            stats.last_delta_norm2 = dn2;
            if (dn2 < 1e-12) {
                stats.exit = LM_EXIT_STEP;
                done = true;
                break;
            }
            for (size_t idx = 0; idx < ncols; ++idx) variables_scratch[idx] = variables[idx] + delta[idx];
            problem.calculate_residuals(variables_scratch.data(), residuals_scratch.data());
            double sum_squared_residuals_scratch = norm_squared(residuals_scratch.data(), nrows);

            if (sum_squared_residuals_scratch < sum_squared_residuals) {
                lambda *= 0.125;
                if (lambda < 1e-50) lambda = 1e-50;
                for (size_t idx = 0; idx < ncols; ++idx) variables[idx] = variables_scratch[idx];
                stats.accepted += 1;
                stats.sse_final = sum_squared_residuals_scratch;
                if ((sum_squared_residuals - sum_squared_residuals_scratch) / sum_squared_residuals <= 1e-6) {
                    stats.exit = LM_EXIT_FTOL;
                    done = true;
                    break;
                }
                sum_squared_residuals = sum_squared_residuals_scratch;
                sparse_jacobian.clear();
                problem.calculate_residuals_and_sparse_jacobian(variables_scratch.data(), residuals.data(),
                                                                sparse_jacobian);
                for (double& r : residuals) r = -r;
                for (size_t idx = 0; idx < ncols; ++idx) sparse_jacobian.push_triplet(nrows + idx, idx, 0.);
                csc = SparseColMat::from_triplet_mat(sparse_jacobian);
                break;
            } else {
                lambda *= 2.;
            }
        }
    }
    return stats;
}

}  // namespace fo
