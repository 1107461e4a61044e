// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates fiksi/src/analyze/numerical/mod.rs:8-163 (`System::analyze`: over-constraint detection by
// row-wise incremental Gauss-Jordan elimination on the dense Jacobian, all variables free) and the
// dense-gradient scatter of fiksi/src/constraints/expressions.rs:962-1090 (later duplicate entries
// OVERWRITE earlier ones, quirk Q4 — unlike the sparse path, which sums).
#pragma once
#include <cmath>
#include <cstdint>
#include <utility>
#include <vector>

#include "fo_expressions.hpp"

namespace fo {

// numerical/mod.rs:33-117. `matrix` row-major nrows x ncols, `column_indices` a permutation of
// 0..ncols. Returns which rows increase the rank.
inline std::vector<bool> incremental_gauss_jordan_elimination(std::vector<double>& matrix, size_t nrows, size_t ncols,
                                                              std::vector<size_t>& column_indices) {
    const double EPSILON = 1e-8;  // :8
    const size_t constraints = nrows, variables = ncols;
    std::vector<bool> constraint_increases_rank(constraints, false);
    size_t current_col = 0;
    for (size_t row = 0; row < std::min(constraints, variables); ++row) {
        size_t rank = 0;
        for (size_t row_idx = 0; row_idx < row; ++row_idx) {
            size_t column_idx = column_indices[rank];
            double factor = matrix[row * variables + column_idx];
            for (size_t col = 0; col < variables; ++col) matrix[row * variables + col] -= factor * matrix[row_idx * variables + col];
            if (constraint_increases_rank[row_idx]) rank += 1;
        }
        bool pivot_found = false;
        for (size_t idx = current_col; idx < variables; ++idx) {
            size_t real_idx = column_indices[idx];
            if (std::fabs(matrix[row * variables + real_idx]) > EPSILON) {
                std::swap(column_indices[current_col], column_indices[idx]);
                pivot_found = true;
                break;
            }
        }
        if (!pivot_found) continue;
        double factor = matrix[row * variables + column_indices[current_col]];
        for (size_t col = 0; col < variables; ++col) matrix[row * variables + col] *= 1. / factor;
        size_t column_idx = column_indices[current_col];
        for (size_t row_idx = 0; row_idx < row; ++row_idx) {
            double f = matrix[row_idx * variables + column_idx];
            for (size_t col = 0; col < variables; ++col) matrix[row_idx * variables + col] -= f * matrix[row * variables + col];
        }
        current_col += 1;
        constraint_increases_rank[row] = true;
    }
    return constraint_increases_rank;
}

// numerical/mod.rs:123-163 on flat data: `dependent[e]` = 1 when expression e does not increase the
// rank (its constraint is reported as over-constraining).
inline void find_overconstraints(const double* variables, size_t nvars, const Expression* exprs, size_t nexprs,
                                 uint8_t* dependent) {
    std::vector<double> jacobian(nexprs * nvars, 0.);
    for (size_t row = 0; row < nexprs; ++row) {
        uint32_t idx[8];
        double vals[8] = {0, 0, 0, 0, 0, 0, 0, 0}, grad[8];
        int k = variable_indices(exprs[row], idx);
        for (int i = 0; i < k; ++i) vals[i] = variables[idx[i]];
        compute_residual_and_gradient(exprs[row], vals, grad);
        for (int i = 0; i < k; ++i) jacobian[row * nvars + idx[i]] = grad[i];  // expressions.rs:1003-1007: overwrite
    }
    std::vector<size_t> column_pivots(nvars);
    for (size_t c = 0; c < nvars; ++c) column_pivots[c] = c;
    std::vector<bool> independent = incremental_gauss_jordan_elimination(jacobian, nexprs, nvars, column_pivots);
    for (size_t e = 0; e < nexprs; ++e) dependent[e] = independent[e] ? 0 : 1;
}

}  // namespace fo
