// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates solvi/src/decomposition/sparse/qr.rs: SymbolicQr::build (:118-206), numeric (:209-223),
// apply_householder (:226-240), calculate_householder (:244-275), Qr::factorize (:281-322),
// q_tr_mul_mut (:328-346), solve_mut (:351-356). Left-looking sparse Householder QR (CSparse
// cs_qr) on the column-permuted matrix.
#pragma once
#include <cmath>
#include <vector>

#include "fo_colamd.hpp"
#include "fo_sparse.hpp"
#include "fo_symbolic.hpp"

namespace fo {

enum class QrOrdering { Natural, Colamd };  // qr.rs:49-61

struct SymbolicQr {
    std::vector<size_t> row_permutation;
    SparseColMatStructure r_structure;
    SparseColMatStructure h_structure;
    std::vector<size_t> col_permutation;
    PermutationSequence inv_col_permutation_sequence;

    // qr.rs:118-206
    static SymbolicQr build(const SparseColMatStructure& a, QrOrdering ordering) {
        SymbolicQr out;
        SparseColMatStructure permuted;
        const SparseColMatStructure* ap = &a;
        if (ordering == QrOrdering::Colamd) {
            std::vector<int> rows(a.row_indices.size());
            for (size_t k = 0; k < rows.size(); ++k) rows[k] = static_cast<int>(a.row_indices[k]);
            std::vector<int> p(a.column_pointers.size());
            for (size_t k = 0; k < p.size(); ++k) p[k] = static_cast<int>(a.column_pointers[k]);
            bool ok = colamd(static_cast<int>(a.nrows), static_cast<int>(a.ncols), rows, p);
            (void)ok;  // `.expect("valid column ordering")` in the reference
            std::vector<size_t> permutation(a.ncols);
            for (size_t k = 0; k < a.ncols; ++k) permutation[k] = static_cast<size_t>(p[k]);
            permuted = a.permute_columns(permutation);
            ap = &permuted;
            std::vector<size_t> inv_permutation(a.ncols, 0);
            for (size_t idx = 0; idx < a.ncols; ++idx) inv_permutation[permutation[idx]] = idx;
            out.inv_col_permutation_sequence = PermutationSequence::build_for_gather_permutation(inv_permutation);
            out.col_permutation = std::move(permutation);
        } else {
            std::vector<size_t> permutation(a.ncols);
            for (size_t k = 0; k < a.ncols; ++k) permutation[k] = k;
            out.inv_col_permutation_sequence = PermutationSequence::build_for_gather_permutation(permutation);
            out.col_permutation = std::move(permutation);
        }
        std::vector<size_t> parents = elimination_tree<false>(*ap);
        std::vector<size_t> post = post_order(parents);
        CholeskyCounts counts = CholeskyCounts::build(*ap, parents, post);
        CholeskyStructure cs = CholeskyStructure::build(*ap, parents, post, counts);
        out.r_structure = std::move(cs.l_structure);
        out.row_permutation = std::move(cs.row_permutation);
        out.h_structure = std::move(cs.h_structure);
        return out;
    }
};

// qr.rs:226-240
inline void apply_householder(double* x, double beta, const size_t* rows, const double* values, size_t len) {
    double tau = 0.;
    for (size_t idx = 0; idx < len; ++idx) tau = tau + values[idx] * x[rows[idx]];
    tau = tau * beta;
    for (size_t idx = 0; idx < len; ++idx) x[rows[idx]] = x[rows[idx]] - values[idx] * tau;
}

// qr.rs:244-275 (= CSparse cs_house). Overwrites `v` with the Householder vector; returns
// (norm, beta).
inline void calculate_householder(double* v, size_t len, double& norm, double& beta) {
    double sigma = 0.;
    for (size_t k = 1; k < len; ++k) sigma = sigma + v[k] * v[k];
    if (sigma == 0.) {
        norm = std::fabs(v[0]);
        beta = (v[0] >= 0.) ? 0. : 2.;
        v[0] = 1.;
    } else {
        norm = std::sqrt(sigma + v[0] * v[0]);
        if (v[0] <= 0.) {
            v[0] = v[0] - norm;
        } else {
            v[0] = -sigma / (v[0] + norm);
        }
        beta = -(1. / (norm * v[0]));
    }
}

struct Qr {
    const SymbolicQr* s;
    SparseColMat r;
    std::vector<double> h_values, h_betas, x;

    // qr.rs:209-223
    explicit Qr(const SymbolicQr& sym) : s(&sym) {
        r.structure = sym.r_structure;
        r.values.assign(sym.r_structure.row_indices.size(), 0.);
        h_values.assign(sym.h_structure.row_indices.size(), 0.);
        h_betas.assign(sym.h_structure.ncols, 0.);
        x.assign(sym.h_structure.nrows, 0.);
    }

    // qr.rs:281-322
    void factorize(const SparseColMat& a) {
        std::fill(r.values.begin(), r.values.end(), 0.);
        std::fill(h_values.begin(), h_values.end(), 0.);
        size_t n = a.ncols();
        const auto& hs = s->h_structure;
        for (size_t j = 0; j < n; ++j) {
            std::fill(x.begin(), x.end(), 0.);
            size_t col = s->col_permutation[j];
            for (size_t p = a.structure.column_pointers[col]; p < a.structure.column_pointers[col + 1]; ++p) {
                x[s->row_permutation[a.structure.row_indices[p]]] = a.values[p];
            }
            size_t rbeg = r.structure.column_pointers[j], rend = r.structure.column_pointers[j + 1];
            for (size_t p = rbeg; p < rend; ++p) {
                size_t r_row = r.structure.row_indices[p];
                if (r_row == j) continue;
                size_t hb = hs.column_pointers[r_row], he = hs.column_pointers[r_row + 1];
                apply_householder(x.data(), h_betas[r_row], hs.row_indices.data() + hb, h_values.data() + hb, he - hb);
                r.values[p] = x[r_row];
                x[r_row] = 0.;
            }
            size_t hb = hs.column_pointers[j], he = hs.column_pointers[j + 1];
            for (size_t p = hb; p < he; ++p) {
                size_t row = hs.row_indices[p];
                h_values[p] = x[row];
                x[row] = 0.;
            }
            double norm, beta;
            calculate_householder(h_values.data() + hb, he - hb, norm, beta);
            h_betas[j] = beta;
            r.values[rend - 1] = norm;
        }
    }

    // qr.rs:328-346
    void q_tr_mul_mut(double* b) const {
        const auto& hs = s->h_structure;
        std::vector<double> y(hs.nrows, 0.);
        for (size_t i = 0; i < hs.nrows; ++i) y[s->row_permutation[i]] = b[i];
        for (size_t j = 0; j < hs.ncols; ++j) {
            size_t hb = hs.column_pointers[j], he = hs.column_pointers[j + 1];
            apply_householder(y.data(), h_betas[j], hs.row_indices.data() + hb, h_values.data() + hb, he - hb);
        }
        for (size_t i = 0; i < hs.nrows; ++i) b[i] = y[i];
    }

    // qr.rs:351-356
    bool solve_mut(double* b) const {
        q_tr_mul_mut(b);
        bool solved = r.solve_upper_triangular_mut(b);
        s->inv_col_permutation_sequence.permute_slice(b);
        return solved;
    }
};

}  // namespace fo
