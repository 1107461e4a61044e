// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates solvi/src/triplet_mat.rs:94-117 (COO builder) and
// solvi/src/sparse_col_mat.rs:190-201 (CSC structure), :452-499 (permute_columns),
// :690-737 (from_triplet_mat), :788-826 (solve_upper_triangular_mut).
#pragma once
#include <algorithm>
#include <numeric>
#include <vector>

#include "fo_common.hpp"

namespace fo {

// triplet_mat.rs: three parallel Vecs; pushing outside the shape grows it (:94-100).
struct TripletMat {
    size_t nrows = 0, ncols = 0;
    std::vector<size_t> row_indices, col_indices;
    std::vector<double> values;

    TripletMat(size_t r, size_t c) : nrows(r), ncols(c) {}

    void push_triplet(size_t row, size_t col, double value) {  // :94-100
        nrows = std::max(nrows, row + 1);
        ncols = std::max(ncols, col + 1);
        row_indices.push_back(row);
        col_indices.push_back(col);
        values.push_back(value);
    }

    void clear() {  // :111-117 (shape resets to 0x0 and is regrown by pushes)
        nrows = 0;
        ncols = 0;
        row_indices.clear();
        col_indices.clear();
        values.clear();
    }
};

// sparse_col_mat.rs:190-201
struct SparseColMatStructure {
    size_t nrows = 0, ncols = 0;
    std::vector<size_t> row_indices;
    std::vector<size_t> column_pointers;

    const size_t* col_begin(size_t j) const { return row_indices.data() + column_pointers[j]; }
    const size_t* col_end(size_t j) const { return row_indices.data() + column_pointers[j + 1]; }
    size_t col_len(size_t j) const { return column_pointers[j + 1] - column_pointers[j]; }

    // sparse_col_mat.rs:452-499 (structure only: the `values` quirk Q7 does not affect it).
    SparseColMatStructure permute_columns(const std::vector<size_t>& column_permutation) const {
        SparseColMatStructure out;
        out.nrows = nrows;
        out.ncols = column_permutation.size();
        out.row_indices.reserve(row_indices.size());
        out.column_pointers.reserve(column_permutation.size() + 1);
        out.column_pointers.push_back(0);
        for (size_t idx = 0; idx < column_permutation.size(); ++idx) {
            size_t j = column_permutation[idx];
            out.column_pointers.push_back(out.column_pointers[idx] + col_len(j));
            out.row_indices.insert(out.row_indices.end(), col_begin(j), col_end(j));
        }
        return out;
    }
};

struct SparseColMat {
    SparseColMatStructure structure;
    std::vector<double> values;

    size_t nrows() const { return structure.nrows; }
    size_t ncols() const { return structure.ncols; }

    // sparse_col_mat.rs:690-737: argsort by (col,row), sum duplicates, fill column pointers for
    // empty columns. The reference sorts with `sort_unstable_by_key`; a stable sort is used here so
    // that duplicates are summed in push order (for two duplicates — the only case fiksi produces,
    // quirk Q4 — the sum is order-independent).
    static SparseColMat from_triplet_mat(const TripletMat& a) {
        size_t nnz = a.values.size();
        SparseColMat out;
        out.structure.nrows = a.nrows;
        out.structure.ncols = a.ncols;
        out.structure.column_pointers.assign(a.ncols + 1, 0);
        out.structure.row_indices.reserve(nnz);
        out.values.reserve(nnz);

        std::vector<size_t> indices(nnz);
        std::iota(indices.begin(), indices.end(), size_t{0});
        std::stable_sort(indices.begin(), indices.end(), [&](size_t x, size_t y) {
            if (a.col_indices[x] != a.col_indices[y]) return a.col_indices[x] < a.col_indices[y];
            return a.row_indices[x] < a.row_indices[y];
        });

        size_t prev_row = NONE, prev_col = NONE;
        for (size_t idx : indices) {
            size_t row = a.row_indices[idx];
            size_t col = a.col_indices[idx];
            if (row == prev_row && col == prev_col) {
                out.values.back() += a.values[idx];
            } else {
                if (col != prev_col) {
                    for (size_t c = prev_col + 1; c <= col; ++c) {  // wrapping_add(1) on NONE -> 0
                        out.structure.column_pointers[c] = out.values.size();
                    }
                }
                out.values.push_back(a.values[idx]);
                out.structure.row_indices.push_back(row);
            }
            prev_row = row;
            prev_col = col;
        }
        for (size_t c = prev_col + 1; c <= a.ncols; ++c) {
            out.structure.column_pointers[c] = out.values.size();
        }
        return out;
    }

    // sparse_col_mat.rs:788-826. Returns false on an exactly-zero (or structurally absent)
    // diagonal.
    bool solve_upper_triangular_mut(double* b) const {
        for (size_t ii = structure.nrows; ii-- > 0;) {
            size_t i = ii;
            size_t begin = structure.column_pointers[i], end = structure.column_pointers[i + 1];
            double diag = 0.;
            if (end > begin && structure.row_indices[end - 1] == i) diag = values[end - 1];
            if (diag == 0.) return false;
            double coeff = b[i] / diag;
            b[i] = coeff;
            for (size_t p = begin; p < end; ++p) {
                size_t row = structure.row_indices[p];
                if (!(row < i)) break;  // take_while(row < i)
                b[row] = b[row] - coeff * values[p];
            }
        }
        return true;
    }
};

}  // namespace fo
