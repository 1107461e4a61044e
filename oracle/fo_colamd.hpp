// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates colamd_rs (a c2rust port of SuiteSparse COLAMD, Davis/Gilbert/Larimore/Ng, "Algorithm
// 836"), as used by solvi's SymbolicQr:
//   colamd_rs/src/colamd.rs:139-158 (colamd_recommended), :354-494 (driver), :502-655
//   (init_rows_cols), :656-809 (init_scoring), :810-1075 (find_ordering), :1087-1137
//   (order_children), :1139-1220 (detect_super_cols), :1222-1305 (garbage_collection),
//   :1306-1326 (clear_mark); colamd_rs/src/options.rs:25-29 (defaults 10, 10, aggressive).
// Only well-formed (sorted, duplicate-free) columns are accepted: that is all solvi ever passes
// (SparseColMat::from_triplet_mat sorts and de-duplicates); "jumbled" input returns false.
#pragma once
#include <climits>
#include <cmath>
#include <cstdint>
#include <vector>

namespace fo {

struct ColamdOptions {  // options.rs:8-29
    double dense_row_control = 10.;
    double dense_column_control = 10.;
    bool aggressive_row_absorption = true;
};

namespace colamd_detail {

constexpr int EMPTY = -1;
constexpr int ALIVE = 0;
constexpr int DEAD = -1;
constexpr int DEAD_PRINCIPAL = -1;
constexpr int DEAD_NON_PRINCIPAL = -2;

// The reference overlays several fields in unions (colamd.rs:31-109); the overlay pairs are
// kept as single ints here with both names noted.
struct Col {
    int start;       // index into A of the column's row list, or DEAD_* when ordered/absorbed
    int length;
    int thickness;   // shared1: thickness | parent
    int score;       // shared2: score | order
    int prev;        // shared3: headhash | hash | prev
    int degree_next; // shared4: degree_next | hash_next
};
struct Row {
    int start;
    int length;
    int degree;  // shared1: degree | p
    int mark;    // shared2: mark | first_column
};

// colamd.rs:1306-1326
inline int clear_mark(int tag_mark, int max_mark, int n_row, std::vector<Row>& rows) {
    if (tag_mark <= 0 || tag_mark >= max_mark) {
        for (int r = 0; r < n_row; ++r) {
            if (rows[r].mark >= ALIVE) rows[r].mark = 0;
        }
        tag_mark = 1;
    }
    return tag_mark;
}

// colamd.rs:1222-1305. Compacts the column lists then the row lists to the front of A and
// returns the new first-free index.
inline int garbage_collection(int n_row, int n_col, std::vector<Row>& rows, std::vector<Col>& cols,
                              std::vector<int>& A, int pfree) {
    int pdest = 0;
    for (int c = 0; c < n_col; ++c) {
        if (cols[c].start >= ALIVE) {
            int psrc = cols[c].start;
            cols[c].start = pdest;
            int length = cols[c].length;
            for (int j = 0; j < length; ++j) {
                int r = A[psrc++];
                if (rows[r].mark >= ALIVE) A[pdest++] = r;
            }
            cols[c].length = pdest - cols[c].start;
        }
    }
    for (int r = 0; r < n_row; ++r) {
        if (rows[r].mark < ALIVE || rows[r].length == 0) {
            rows[r].mark = DEAD;
        } else {
            int psrc = rows[r].start;
            rows[r].mark = A[psrc];  // shared2.first_column
            A[psrc] = -r - 1;
        }
    }
    int psrc = pdest;
    while (psrc < pfree) {
        if (A[psrc++] < 0) {
            psrc--;
            int r = -A[psrc] - 1;
            A[psrc] = rows[r].mark;  // restore first column index
            rows[r].start = pdest;
            int length = rows[r].length;
            for (int j = 0; j < length; ++j) {
                int c = A[psrc++];
                if (cols[c].start >= ALIVE) A[pdest++] = c;
            }
            rows[r].length = pdest - rows[r].start;
        }
    }
    return pdest;
}

// colamd.rs:1139-1220
inline void detect_super_cols(std::vector<Col>& cols, std::vector<int>& A, std::vector<int>& head,
                              int row_start, int row_length) {
    for (int rp = row_start; rp < row_start + row_length; ++rp) {
        int col = A[rp];
        if (cols[col].start < ALIVE) continue;
        int hash = cols[col].prev;  // shared3.hash
        int head_column = head[hash];
        int first_col;
        if (head_column > EMPTY) {
            first_col = cols[head_column].prev;  // shared3.headhash
        } else {
            first_col = -(head_column + 2);
        }
        for (int super_c = first_col; super_c != EMPTY; super_c = cols[super_c].degree_next) {
            int length = cols[super_c].length;
            int prev_c = super_c;
            for (int c = cols[super_c].degree_next; c != EMPTY; c = cols[c].degree_next) {
                if (cols[c].length != length || cols[c].score != cols[super_c].score) {
                    prev_c = c;
                    continue;
                }
                int cp1 = cols[super_c].start, cp2 = cols[c].start;
                int i = 0;
                for (; i < length; ++i) {
                    if (A[cp1++] != A[cp2++]) break;
                }
                if (i != length) {
                    prev_c = c;
                    continue;
                }
                // identical: absorb c into super_c
                cols[super_c].thickness += cols[c].thickness;
                cols[c].thickness = super_c;  // shared1.parent
                cols[c].start = DEAD_NON_PRINCIPAL;
                cols[c].score = EMPTY;  // shared2.order
                cols[prev_c].degree_next = cols[c].degree_next;  // hash_next
            }
        }
        if (head_column > EMPTY) {
            cols[head_column].prev = EMPTY;  // headhash
        } else {
            head[hash] = EMPTY;
        }
    }
}

}  // namespace colamd_detail

// colamd.rs:139-158 — recommended length of the whole i32 workspace (indices + Col + Row
// records; sizeof(Colamd_Col) = 24 B = 6 ints, sizeof(Colamd_Row) = 16 B = 4 ints).
inline size_t colamd_recommended(size_t nnz, size_t n_row, size_t n_col) {
    size_t c = (n_col + 1) * 24 / 4;
    size_t r = (n_row + 1) * 16 / 4;
    return 2 * nnz + c + r + n_col + nnz / 5;
}

// colamd_rs/src/lib.rs:103-140 + colamd.rs:354-494. `row_indices` holds the nnz row indices of
// the CSC matrix (only its first p[n_col] entries are read); `p` (length n_col+1) holds the column
// pointers on entry and the permutation in p[0..n_col] on exit (p[n_col] = -1 is not replicated;
// callers take the first n_col entries). Returns false on invalid input.
inline bool colamd(int n_row, int n_col, const std::vector<int>& row_indices, std::vector<int>& p,
                   const ColamdOptions& options = ColamdOptions()) {
    using namespace colamd_detail;
    if (n_row < 0 || n_col < 0) return false;
    if (static_cast<int>(p.size()) != n_col + 1) return false;
    int nnz = p[n_col];
    if (nnz < 0 || p[0] != 0) return false;

    // Index workspace as in solvi's call (qr.rs:123-134): Alen = recommended - Col - Row records.
    size_t total = colamd_recommended(static_cast<size_t>(nnz), static_cast<size_t>(n_row),
                                      static_cast<size_t>(n_col));
    size_t col_size = (static_cast<size_t>(n_col) + 1) * 6, row_size = (static_cast<size_t>(n_row) + 1) * 4;
    int Alen = static_cast<int>(total - col_size - row_size);
    std::vector<int> A(static_cast<size_t>(Alen), 0);
    for (int k = 0; k < nnz; ++k) A[k] = row_indices[k];
    std::vector<Col> cols(static_cast<size_t>(n_col) + 1);
    std::vector<Row> rows(static_cast<size_t>(n_row) + 1);

    // === init_rows_cols (colamd.rs:502-655) ================================================
    for (int col = 0; col < n_col; ++col) {
        cols[col].start = p[col];
        cols[col].length = p[col + 1] - p[col];
        if (cols[col].length < 0) return false;
        cols[col].thickness = 1;
        cols[col].score = 0;
        cols[col].prev = EMPTY;
        cols[col].degree_next = EMPTY;
    }
    for (int row = 0; row < n_row; ++row) {
        rows[row].length = 0;
        rows[row].mark = -1;
    }
    for (int col = 0; col < n_col; ++col) {
        int last_row = -1;
        for (int cp = p[col]; cp < p[col + 1]; ++cp) {
            int row = A[cp];
            if (row < 0 || row >= n_row) return false;
            if (row <= last_row || rows[row].mark == col) return false;  // jumbled: unsupported here
            rows[row].length += 1;
            rows[row].mark = col;
            last_row = row;
        }
    }
    if (n_row > 0) {
        rows[0].start = p[n_col];
        rows[0].degree = rows[0].start;  // shared1.p
        rows[0].mark = -1;
        for (int row = 1; row < n_row; ++row) {
            rows[row].start = rows[row - 1].start + rows[row - 1].length;
            rows[row].degree = rows[row].start;
            rows[row].mark = -1;
        }
    }
    for (int col = 0; col < n_col; ++col) {
        for (int cp = p[col]; cp < p[col + 1]; ++cp) {
            int row = A[cp];
            A[rows[row].degree] = col;
            rows[row].degree += 1;
        }
    }
    for (int row = 0; row < n_row; ++row) {
        rows[row].mark = 0;
        rows[row].degree = rows[row].length;
    }

    // From here on p[0..n_col] is the `head` array of the degree lists / hash buckets.
    std::vector<int>& head = p;

    // === init_scoring (colamd.rs:656-809) ==================================================
    int dense_row_count, dense_col_count;
    if (options.dense_row_control < 0.) {
        dense_row_count = n_col - 1;
    } else {
        double t = options.dense_row_control * std::sqrt(static_cast<double>(n_col));
        dense_row_count = static_cast<int>(16.0 > t ? 16.0 : t);
    }
    if (options.dense_column_control < 0.) {
        dense_col_count = n_row - 1;
    } else {
        double t = options.dense_column_control * std::sqrt(static_cast<double>(n_row < n_col ? n_row : n_col));
        dense_col_count = static_cast<int>(16.0 > t ? 16.0 : t);
    }
    int max_deg = 0;
    int n_col2 = n_col;
    int n_row2 = n_row;
    // kill empty columns
    for (int c = n_col - 1; c >= 0; --c) {
        if (cols[c].length == 0) {
            cols[c].score = --n_col2;  // shared2.order
            cols[c].start = DEAD_PRINCIPAL;
        }
    }
    // kill dense columns
    for (int c = n_col - 1; c >= 0; --c) {
        if (cols[c].start < ALIVE) continue;
        if (cols[c].length > dense_col_count) {
            cols[c].score = --n_col2;
            for (int cp = cols[c].start; cp < cols[c].start + cols[c].length; ++cp) rows[A[cp]].degree -= 1;
            cols[c].start = DEAD_PRINCIPAL;
        }
    }
    // kill dense and empty rows
    for (int r = 0; r < n_row; ++r) {
        int deg = rows[r].degree;
        if (deg > dense_row_count || deg == 0) {
            rows[r].mark = DEAD;
            --n_row2;
        } else {
            max_deg = max_deg > deg ? max_deg : deg;
        }
    }
    // initial column scores
    for (int c = n_col - 1; c >= 0; --c) {
        if (cols[c].start < ALIVE) continue;
        int score = 0;
        int cp = cols[c].start, new_cp = cp, cp_end = cp + cols[c].length;
        while (cp < cp_end) {
            int row = A[cp++];
            if (rows[row].mark < ALIVE) continue;
            A[new_cp++] = row;
            score += rows[row].degree - 1;
            score = score < n_col ? score : n_col;
        }
        int col_length = new_cp - cols[c].start;
        if (col_length == 0) {
            cols[c].score = --n_col2;
            cols[c].start = DEAD_PRINCIPAL;
        } else {
            cols[c].length = col_length;
            cols[c].score = score;
        }
    }
    // degree lists
    for (int c = 0; c <= n_col; ++c) head[c] = EMPTY;
    int min_score = n_col;
    for (int c = n_col - 1; c >= 0; --c) {
        if (cols[c].start < ALIVE) continue;
        int score = cols[c].score;
        int next_col = head[score];
        cols[c].prev = EMPTY;
        cols[c].degree_next = next_col;
        if (next_col != EMPTY) cols[next_col].prev = c;
        head[score] = c;
        min_score = min_score < score ? min_score : score;
    }

    // === find_ordering (colamd.rs:810-1075) ================================================
    const bool aggressive = options.aggressive_row_absorption;
    int pfree = 2 * nnz;
    int max_mark = INT_MAX - n_col;
    int tag_mark = clear_mark(0, max_mark, n_row, rows);
    min_score = 0;
    for (int k = 0; k < n_col2;) {
        // select pivot column of minimum score
        while (head[min_score] == EMPTY && min_score < n_col) min_score++;
        int pivot_col = head[min_score];
        int next_col = cols[pivot_col].degree_next;
        head[min_score] = next_col;
        if (next_col != EMPTY) cols[next_col].prev = EMPTY;
        int pivot_col_score = cols[pivot_col].score;
        cols[pivot_col].score = k;  // shared2.order
        int pivot_col_thickness = cols[pivot_col].thickness;
        k += pivot_col_thickness;

        int needed_memory = pivot_col_score < n_col - k ? pivot_col_score : n_col - k;
        if (pfree + needed_memory >= Alen) {
            pfree = garbage_collection(n_row, n_col, rows, cols, A, pfree);
            tag_mark = clear_mark(0, max_mark, n_row, rows);
        }

        // construct the pivot row pattern
        int pivot_row_start = pfree;
        int pivot_row_degree = 0;
        cols[pivot_col].thickness = -pivot_col_thickness;
        for (int cp = cols[pivot_col].start, cp_end = cp + cols[pivot_col].length; cp < cp_end; ++cp) {
            int row = A[cp];
            if (rows[row].mark < ALIVE) continue;
            for (int rp = rows[row].start, rp_end = rp + rows[row].length; rp < rp_end; ++rp) {
                int col = A[rp];
                int col_thickness = cols[col].thickness;
                if (col_thickness > 0 && cols[col].start >= ALIVE) {
                    cols[col].thickness = -col_thickness;
                    A[pfree++] = col;
                    pivot_row_degree += col_thickness;
                }
            }
        }
        cols[pivot_col].thickness = pivot_col_thickness;
        max_deg = max_deg > pivot_row_degree ? max_deg : pivot_row_degree;

        // kill all rows used to construct the pivot row
        for (int cp = cols[pivot_col].start, cp_end = cp + cols[pivot_col].length; cp < cp_end; ++cp) {
            rows[A[cp]].mark = DEAD;
        }
        int pivot_row_length = pfree - pivot_row_start;
        int pivot_row = pivot_row_length > 0 ? A[cols[pivot_col].start] : EMPTY;

        // approximate degree: set differences
        for (int rp = pivot_row_start, rp_end = rp + pivot_row_length; rp < rp_end; ++rp) {
            int col = A[rp];
            int col_thickness = -cols[col].thickness;
            cols[col].thickness = col_thickness;
            int cur_score = cols[col].score;
            int prev_col = cols[col].prev;
            int nxt = cols[col].degree_next;
            if (prev_col == EMPTY) head[cur_score] = nxt; else cols[prev_col].degree_next = nxt;
            if (nxt != EMPTY) cols[nxt].prev = prev_col;
            for (int cp = cols[col].start, cp_end = cp + cols[col].length; cp < cp_end; ++cp) {
                int row = A[cp];
                int row_mark = rows[row].mark;
                if (row_mark < ALIVE) continue;
                int set_difference = row_mark - tag_mark;
                if (set_difference < 0) set_difference = rows[row].degree;
                set_difference -= col_thickness;
                if (set_difference == 0 && aggressive) {
                    rows[row].mark = DEAD;
                } else {
                    rows[row].mark = set_difference + tag_mark;
                }
            }
        }

        // add up set differences for each column; hash for supercolumn detection
        for (int rp = pivot_row_start, rp_end = rp + pivot_row_length; rp < rp_end; ++rp) {
            int col = A[rp];
            uint32_t hash = 0;
            int cur_score = 0;
            int cp = cols[col].start, new_cp = cp, cp_end = cp + cols[col].length;
            while (cp < cp_end) {
                int row = A[cp++];
                int row_mark = rows[row].mark;
                if (row_mark < ALIVE) continue;
                A[new_cp++] = row;
                hash += static_cast<uint32_t>(row);
                cur_score += row_mark - tag_mark;
                cur_score = cur_score < n_col ? cur_score : n_col;
            }
            cols[col].length = new_cp - cols[col].start;
            if (cols[col].length == 0) {
                // nothing left but the pivot row: order now
                cols[col].start = DEAD_PRINCIPAL;
                pivot_row_degree -= cols[col].thickness;
                cols[col].score = k;  // order
                k += cols[col].thickness;
            } else {
                cols[col].score = cur_score;
                hash %= static_cast<uint32_t>(n_col + 1);
                int head_column = head[hash];
                int first_col;
                if (head_column > EMPTY) {
                    first_col = cols[head_column].prev;  // headhash
                    cols[head_column].prev = col;
                } else {
                    first_col = -(head_column + 2);
                    head[hash] = -(col + 2);
                }
                cols[col].degree_next = first_col;  // hash_next
                cols[col].prev = static_cast<int>(hash);
            }
        }

        detect_super_cols(cols, A, head, pivot_row_start, pivot_row_length);
        cols[pivot_col].start = DEAD_PRINCIPAL;
        tag_mark = clear_mark(tag_mark + max_deg + 1, max_mark, n_row, rows);

        // finalize the new pivot row and column scores
        int new_rp = pivot_row_start;
        for (int rp = pivot_row_start, rp_end = rp + pivot_row_length; rp < rp_end; ++rp) {
            int col = A[rp];
            if (cols[col].start < ALIVE) continue;
            A[new_rp++] = col;
            A[cols[col].start + cols[col].length] = pivot_row;
            cols[col].length += 1;
            int cur_score = cols[col].score + pivot_row_degree;
            int max_score = n_col - k - cols[col].thickness;
            cur_score -= cols[col].thickness;
            cur_score = cur_score < max_score ? cur_score : max_score;
            cols[col].score = cur_score;
            int nxt = head[cur_score];
            cols[col].degree_next = nxt;
            cols[col].prev = EMPTY;
            if (nxt != EMPTY) cols[nxt].prev = col;
            head[cur_score] = col;
            min_score = min_score < cur_score ? min_score : cur_score;
        }
        if (pivot_row_degree > 0) {
            rows[pivot_row].start = pivot_row_start;
            rows[pivot_row].length = new_rp - pivot_row_start;
            rows[pivot_row].degree = pivot_row_degree;
            rows[pivot_row].mark = 0;
        }
    }

    // === order_children (colamd.rs:1087-1137) ==============================================
    for (int i = 0; i < n_col; ++i) {
        if (cols[i].start != DEAD_PRINCIPAL && cols[i].score == EMPTY) {
            int parent = i;
            do {
                parent = cols[parent].thickness;  // shared1.parent
            } while (cols[parent].start != DEAD_PRINCIPAL);
            int c = i;
            int order = cols[parent].score;
            do {
                cols[c].score = order++;
                cols[c].thickness = parent;  // collapse tree
                c = cols[c].thickness;       // immediate parent (now `parent`)
            } while (cols[c].score == EMPTY);
            cols[parent].score = order;
        }
    }
    for (int c = 0; c < n_col; ++c) p[cols[c].score] = c;
    return true;
}

}  // namespace fo
