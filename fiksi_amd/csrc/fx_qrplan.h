// Host-side symbolic analysis for the reference-numerics LM step (fx_step_solver FX_STEP_QR).
//
// The reference solves every LM trial as the least-squares problem [J; sqrt(lambda) I] delta = [-r; 0] with
// solvi's sparse Householder QR (fiksi/src/solve/lm.rs:98-132, solvi/src/decomposition/sparse/qr.rs:118-356).
// Which floating-point operations that QR performs, and in which order, is fixed by its symbolic phase:
//   - the COLAMD column order                       (colamd_rs::colamd, default knobs; qr.rs:121-170)
//   - the elimination tree of A^T A and its post-order (cholesky.rs:31-84, solvi/src/utils.rs:49-117)
//   - the row permutation of Davis' "Direct Methods" section 5.3 (cholesky.rs:381-442)
//   - the row patterns of the Householder vectors and of R's columns, both ascending (cholesky.rs:444-571).
// The north-star keeps that phase on the host. This header redoes it for one component of one System and
// hands the device kernel (fx_kernels.hip, qr_step) exactly what it needs to replay the numeric phase
// operation by operation: the two permutations, the Householder row lists, and which Householder vectors
// touch which column. Plain C++ (no HIP): tests/test_qr_plan.py checks it on the CPU against an independent restatement of SymbolicQr::build.
#pragma once
#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <vector>

namespace fx {
namespace qr {

struct Csc {  // pattern of a sparse matrix by columns, rows ascending inside a column
    int nrows = 0, ncols = 0;
    std::vector<int> ptr, idx;
};

// ------------------------------------------------------------------------------------------
// COLAMD: approximate-minimum-degree column ordering (Davis, Gilbert, Larimore, Ng; ACM TOMS
// algorithm 836) as colamd_rs runs it for solvi: knobs (10, 10, aggressive), workspace of
// colamd_recommended() entries. The permutation must equal the reference's entry for entry —
// ties are broken by list positions, so the data structures below follow the published
// algorithm closely: one integer pool holding the column lists and the row lists, degree lists
// threaded through the columns, supercolumn detection by hashing, and the same compaction rule.
// ------------------------------------------------------------------------------------------
class Colamd {
  public:
    // perm[k] = column of `a` that comes k-th. False for malformed input (unsorted / duplicate rows).
    static bool order(const Csc& a, std::vector<int>& perm) {
        Colamd w;
        if (!w.load(a)) return false;
        w.score_columns();
        w.eliminate();
        w.number_absorbed_columns();
        perm.assign((size_t)w.ncol, 0);
        for (int c = 0; c < w.ncol; ++c) perm[(size_t)w.col[c].order] = c;
        return true;
    }

  private:
    static constexpr int NONE = -1;
    struct Column {
        int start = 0;       // first entry of the row list in the pool; < 0 once ordered / absorbed
        int len = 0;
        int thickness = 1;   // columns this (super)column stands for; negative while it sits in the pivot row
        int parent = NONE;   // supercolumn that absorbed it
        int score = 0;
        int order = NONE;
        int prev = NONE, next = NONE;          // degree list
        int bucket = 0, bucket_next = NONE;    // supercolumn hash
        int bucket_head = NONE;                // first hashed column, kept on the degree-list head sharing the slot
    };
    struct Row {
        int start = 0, len = 0, degree = 0;
        int mark = 0;  // < 0: dead
        int first = 0; // scratch of compact()
    };
    static constexpr int ORDERED = -1, ABSORBED = -2;

    int nrow = 0, ncol = 0, nnz = 0, pool_len = 0, ncol_live = 0, max_deg = 0;
    std::vector<int> pool, head;
    std::vector<Column> col;
    std::vector<Row> row;

    bool alive(int c) const { return col[c].start >= 0; }

    bool load(const Csc& a) {
        nrow = a.nrows;
        ncol = a.ncols;
        if (nrow < 0 || ncol < 0 || (int)a.ptr.size() != ncol + 1 || a.ptr[0] != 0) return false;
        nnz = a.ptr[ncol];
        if (nnz < 0) return false;
        // colamd_recommended(nnz, n_row, n_col) minus the column / row records (24 and 16 bytes each)
        const size_t rec = ((size_t)ncol + 1) * 6 + ((size_t)nrow + 1) * 4;
        pool_len = (int)(2 * (size_t)nnz + rec + (size_t)ncol + (size_t)nnz / 5 - rec);
        pool.assign((size_t)pool_len, 0);
        col.assign((size_t)ncol + 1, Column());
        row.assign((size_t)nrow + 1, Row());
        std::vector<int> seen((size_t)nrow, -1);
        for (int c = 0; c < ncol; ++c) {
            col[c].start = a.ptr[c];
            col[c].len = a.ptr[c + 1] - a.ptr[c];
            if (col[c].len < 0) return false;
            int last = -1;
            for (int p = a.ptr[c]; p < a.ptr[c + 1]; ++p) {
                const int r = a.idx[p];
                if (r < 0 || r >= nrow || r <= last || seen[r] == c) return false;
                pool[p] = r;
                row[r].len += 1;
                seen[r] = c;
                last = r;
            }
        }
        // the row form goes behind the column form
        int at = nnz;
        for (int r = 0; r < nrow; ++r) {
            row[r].start = at;
            at += row[r].len;
        }
        std::vector<int> fill((size_t)nrow);
        for (int r = 0; r < nrow; ++r) fill[r] = row[r].start;
        for (int c = 0; c < ncol; ++c)
            for (int p = a.ptr[c]; p < a.ptr[c + 1]; ++p) pool[fill[a.idx[p]]++] = c;
        for (int r = 0; r < nrow; ++r) {
            row[r].mark = 0;
            row[r].degree = row[r].len;
        }
        head.assign((size_t)ncol + 1, NONE);
        return true;
    }

    static int knob(double control, int dim, int fallback) {
        if (control < 0.) return fallback;
        const double t = control * std::sqrt((double)dim);
        return (int)(16.0 > t ? 16.0 : t);
    }

    void retire(int c) {  // ordered last (empty / dense columns)
        col[c].order = --ncol_live;
        col[c].start = ORDERED;
    }

    void score_columns() {
        const int dense_row = knob(10., ncol, ncol - 1);
        const int dense_col = knob(10., nrow < ncol ? nrow : ncol, nrow - 1);
        ncol_live = ncol;
        max_deg = 0;
        for (int c = ncol - 1; c >= 0; --c)
            if (col[c].len == 0) retire(c);
        for (int c = ncol - 1; c >= 0; --c) {
            if (!alive(c) || col[c].len <= dense_col) continue;
            for (int p = col[c].start; p < col[c].start + col[c].len; ++p) row[pool[p]].degree -= 1;
            retire(c);
        }
        for (int r = 0; r < nrow; ++r) {
            if (row[r].degree > dense_row || row[r].degree == 0) row[r].mark = -1;
            else max_deg = std::max(max_deg, row[r].degree);
        }
        for (int c = ncol - 1; c >= 0; --c) {
            if (!alive(c)) continue;
            int score = 0, keep = col[c].start;
            for (int p = col[c].start, e = p + col[c].len; p < e; ++p) {
                const int r = pool[p];
                if (row[r].mark < 0) continue;
                pool[keep++] = r;
                score = std::min(score + row[r].degree - 1, ncol);
            }
            if (keep == col[c].start) {
                retire(c);
            } else {
                col[c].len = keep - col[c].start;
                col[c].score = score;
            }
        }
        for (int c = ncol - 1; c >= 0; --c)
            if (alive(c)) push_degree(c, col[c].score);
    }

    void push_degree(int c, int score) {
        const int nx = head[score];
        col[c].prev = NONE;
        col[c].next = nx;
        if (nx != NONE) col[nx].prev = c;
        head[score] = c;
    }

    int reset_marks(int tag, int limit) {
        if (tag <= 0 || tag >= limit) {
            for (int r = 0; r < nrow; ++r)
                if (row[r].mark >= 0) row[r].mark = 0;
            tag = 1;
        }
        return tag;
    }

    // compacts the live column lists, then the live row lists, to the front of the pool
    int compact(int used) {
        int dst = 0;
        for (int c = 0; c < ncol; ++c) {
            if (!alive(c)) continue;
            int src = col[c].start;
            col[c].start = dst;
            for (int k = 0; k < col[c].len; ++k) {
                const int r = pool[src++];
                if (row[r].mark >= 0) pool[dst++] = r;
            }
            col[c].len = dst - col[c].start;
        }
        for (int r = 0; r < nrow; ++r) {
            if (row[r].mark < 0 || row[r].len == 0) {
                row[r].mark = -1;
            } else {  // tag the first entry of the list with the row it belongs to
                row[r].first = pool[row[r].start];
                pool[row[r].start] = -r - 1;
            }
        }
        int src = dst;
        while (src < used) {
            if (pool[src] >= 0) {
                ++src;
                continue;
            }
            const int r = -pool[src] - 1;
            pool[src] = row[r].first;
            row[r].start = dst;
            for (int k = 0; k < row[r].len; ++k) {
                const int c = pool[src++];
                if (alive(c)) pool[dst++] = c;
            }
            row[r].len = dst - row[r].start;
        }
        return dst;
    }

    // columns of the new pivot row with identical row lists become one supercolumn
    void merge_identical(int list, int len) {
        for (int p = list; p < list + len; ++p) {
            const int c0 = pool[p];
            if (!alive(c0)) continue;
            const int slot = col[c0].bucket;
            const int h = head[slot];
            const int first = h > NONE ? col[h].bucket_head : -(h + 2);
            for (int s = first; s != NONE; s = col[s].bucket_next) {
                int before = s;
                for (int c = col[s].bucket_next; c != NONE; c = col[c].bucket_next) {
                    bool same = col[c].len == col[s].len && col[c].score == col[s].score;
                    for (int k = 0; same && k < col[s].len; ++k) same = pool[col[s].start + k] == pool[col[c].start + k];
                    if (!same) {
                        before = c;
                        continue;
                    }
                    col[s].thickness += col[c].thickness;
                    col[c].parent = s;
                    col[c].start = ABSORBED;
                    col[c].order = NONE;
                    col[before].bucket_next = col[c].bucket_next;
                }
            }
            if (h > NONE) col[h].bucket_head = NONE;
            else head[slot] = NONE;
        }
    }

    void eliminate() {
        int used = 2 * nnz;
        const int mark_limit = INT_MAX - ncol;
        int tag = reset_marks(0, mark_limit);
        int min_score = 0;
        for (int k = 0; k < ncol_live;) {
            while (head[min_score] == NONE && min_score < ncol) ++min_score;
            const int piv = head[min_score];
            head[min_score] = col[piv].next;
            if (col[piv].next != NONE) col[col[piv].next].prev = NONE;
            const int piv_score = col[piv].score, piv_thick = col[piv].thickness;
            col[piv].order = k;
            k += piv_thick;

            if (used + std::min(piv_score, ncol - k) >= pool_len) {
                used = compact(used);
                tag = reset_marks(0, mark_limit);
            }

            // pattern of the pivot row: union of the live rows of the pivot column
            const int prow_start = used;
            int prow_degree = 0;
            col[piv].thickness = -piv_thick;
            for (int p = col[piv].start, e = p + col[piv].len; p < e; ++p) {
                const int r = pool[p];
                if (row[r].mark < 0) continue;
                for (int q = row[r].start, qe = q + row[r].len; q < qe; ++q) {
                    const int c = pool[q];
                    if (col[c].thickness > 0 && alive(c)) {
                        prow_degree += col[c].thickness;
                        col[c].thickness = -col[c].thickness;
                        pool[used++] = c;
                    }
                }
            }
            col[piv].thickness = piv_thick;
            max_deg = std::max(max_deg, prow_degree);
            for (int p = col[piv].start, e = p + col[piv].len; p < e; ++p) row[pool[p]].mark = -1;
            const int prow_len = used - prow_start;
            const int prow = prow_len > 0 ? pool[col[piv].start] : NONE;

            // set differences |row \ pivot row| for every row that meets the pivot row
            for (int p = prow_start; p < prow_start + prow_len; ++p) {
                const int c = pool[p];
                const int thick = -col[c].thickness;
                col[c].thickness = thick;
                if (col[c].prev == NONE) head[col[c].score] = col[c].next;
                else col[col[c].prev].next = col[c].next;
                if (col[c].next != NONE) col[col[c].next].prev = col[c].prev;
                for (int q = col[c].start, qe = q + col[c].len; q < qe; ++q) {
                    const int r = pool[q];
                    if (row[r].mark < 0) continue;
                    int diff = row[r].mark - tag;
                    if (diff < 0) diff = row[r].degree;
                    diff -= thick;
                    row[r].mark = diff == 0 ? -1 : diff + tag;  // aggressive absorption: the row is a subset
                }
            }

            // new scores; hash of the surviving row lists
            for (int p = prow_start; p < prow_start + prow_len; ++p) {
                const int c = pool[p];
                uint32_t hash = 0;
                int score = 0, keep = col[c].start;
                for (int q = col[c].start, qe = q + col[c].len; q < qe; ++q) {
                    const int r = pool[q];
                    if (row[r].mark < 0) continue;
                    pool[keep++] = r;
                    hash += (uint32_t)r;
                    score = std::min(score + row[r].mark - tag, ncol);
                }
                col[c].len = keep - col[c].start;
                if (col[c].len == 0) {  // only the pivot row is left of it: ordered now
                    col[c].start = ORDERED;
                    prow_degree -= col[c].thickness;
                    col[c].order = k;
                    k += col[c].thickness;
                    continue;
                }
                col[c].score = score;
                const int slot = (int)(hash % (uint32_t)(ncol + 1));
                const int h = head[slot];
                if (h > NONE) {  // the slot doubles as a degree-list head: chain through that column
                    col[c].bucket_next = col[h].bucket_head;
                    col[h].bucket_head = c;
                } else {
                    col[c].bucket_next = -(h + 2);
                    head[slot] = -(c + 2);
                }
                col[c].bucket = slot;
            }

            merge_identical(prow_start, prow_len);
            col[piv].start = ORDERED;
            tag = reset_marks(tag + max_deg + 1, mark_limit);

            // the pivot row replaces the rows it absorbed; final scores, back into the degree lists
            int keep = prow_start;
            for (int p = prow_start; p < prow_start + prow_len; ++p) {
                const int c = pool[p];
                if (!alive(c)) continue;
                pool[keep++] = c;
                pool[col[c].start + col[c].len] = prow;
                col[c].len += 1;
                int score = col[c].score + prow_degree - col[c].thickness;
                score = std::min(score, ncol - k - col[c].thickness);
                col[c].score = score;
                push_degree(c, score);
                min_score = std::min(min_score, score);
            }
            if (prow_degree > 0) {
                row[prow].start = prow_start;
                row[prow].len = keep - prow_start;
                row[prow].degree = prow_degree;
                row[prow].mark = 0;
            }
        }
    }

    // absorbed columns follow their supercolumn in the order
    void number_absorbed_columns() {
        for (int i = 0; i < ncol; ++i) {
            if (col[i].start == ORDERED || col[i].order != NONE) continue;
            int top = i;
            do top = col[top].parent; while (col[top].start != ORDERED);
            int at = col[top].order;
            for (int c = i; col[c].order == NONE;) {
                col[c].order = at++;
                col[c].parent = top;
                c = top;  // (the published code steps to the freshly collapsed parent link)
            }
            col[top].order = at;
        }
    }
};

// ------------------------------------------------------------------------------------------
// symbolic QR of the column-permuted pattern
// ------------------------------------------------------------------------------------------
struct Symbolic {
    std::vector<int> col_perm;   // position -> original column
    std::vector<int> row_perm;   // original row -> permuted row
    std::vector<int> parent;     // elimination tree over positions
    std::vector<int> hptr, hrows;  // Householder vector k: permuted rows, ascending, first = k
    std::vector<int> rptr, rrows;  // column j of R: positions k < j ascending, then j
};

// elimination tree of A^T A without forming it (cholesky.rs:31-84)
inline void elimination_tree(const Csc& a, std::vector<int>& parent) {
    parent.assign((size_t)a.ncols, -1);
    std::vector<int> ancestor((size_t)a.ncols, -1), last_col((size_t)a.nrows, -1);
    for (int j = 0; j < a.ncols; ++j) {
        for (int p = a.ptr[j]; p < a.ptr[j + 1]; ++p) {
            const int r = a.idx[p];
            for (int k = last_col[r]; k != -1 && k < j;) {
                const int up = ancestor[k];
                ancestor[k] = j;
                if (up == -1) parent[k] = j;
                k = up;
            }
            last_col[r] = j;
        }
    }
}

// depth-first post-order, children of a node visited from the highest-numbered one down, roots ascending
// (solvi/src/utils.rs:49-117)
inline void post_order(const std::vector<int>& parent, std::vector<int>& post) {
    const int n = (int)parent.size();
    std::vector<int> child((size_t)n, -1), sibling((size_t)n, -1), stack;
    for (int v = 0; v < n; ++v) {
        if (parent[v] == -1) continue;
        sibling[v] = child[parent[v]];
        child[parent[v]] = v;
    }
    post.clear();
    post.reserve((size_t)n);
    for (int root = 0; root < n; ++root) {
        if (parent[root] != -1) continue;
        int v = root;
        for (;;) {
            while (child[v] != -1) {
                const int c = child[v];
                child[v] = sibling[c];
                stack.push_back(v);
                v = c;
            }
            post.push_back(v);
            if (stack.empty()) break;
            v = stack.back();
            stack.pop_back();
        }
    }
}

// Returns false when the reference itself could not run on this pattern (malformed input, or a column no
// row can be assigned to: the reference would index past its workspace, cholesky.rs:395-401).
inline bool analyze(const Csc& a, bool use_colamd, Symbolic& out) {
    const int m = a.nrows, n = a.ncols;
    out = Symbolic();
    if (use_colamd) {
        if (!Colamd::order(a, out.col_perm)) return false;
    } else {
        out.col_perm.resize((size_t)n);
        for (int j = 0; j < n; ++j) out.col_perm[j] = j;
    }
    Csc b;  // the permuted pattern
    b.nrows = m;
    b.ncols = n;
    b.ptr.assign(1, 0);
    for (int j = 0; j < n; ++j) {
        const int c = out.col_perm[j];
        b.idx.insert(b.idx.end(), a.idx.begin() + a.ptr[c], a.idx.begin() + a.ptr[c + 1]);
        b.ptr.push_back((int)b.idx.size());
    }
    elimination_tree(b, out.parent);
    std::vector<int> post;
    post_order(out.parent, post);
    // first column of every row, "first" in post-order (cholesky.rs:165-170)
    std::vector<int> first((size_t)m, -1);
    for (int j : post)
        for (int p = b.ptr[j]; p < b.ptr[j + 1]; ++p)
            if (first[b.idx[p]] == -1) first[b.idx[p]] = j;

    // Row permutation (Davis section 5.3, cholesky.rs:381-442): the rows whose first column is k queue up at k;
    // k takes the front row as its pivot row and passes the others on to its parent's queue.
    out.row_perm.assign((size_t)m, -1);
    {
        std::vector<int> next((size_t)m, 0), front((size_t)n, -1), back((size_t)n, -1), queued((size_t)n, 0);
        for (int i = m - 1; i >= 0; --i) {
            const int k = first[i];
            if (k == -1) continue;
            if (queued[k]++ == 0) back[k] = i;
            next[i] = front[k];
            front[k] = i;
        }
        for (int k = 0; k < n; ++k) {
            if (front[k] == -1) return false;  // structurally rank deficient
            const int i = front[k];
            out.row_perm[i] = k;
            if (--queued[k] <= 0) continue;
            const int up = out.parent[k];
            if (up == -1) continue;
            if (queued[up] == 0) back[up] = back[k];
            next[back[k]] = front[up];
            front[up] = next[i];
            queued[up] += queued[k];
        }
        int k = n;
        for (int i = 0; i < m; ++i)
            if (out.row_perm[i] == -1) out.row_perm[i] = k++;
    }

    // Row patterns (cholesky.rs:444-571). R(:, j): the positions reached from the first columns of column j's rows
    // by climbing the tree below j. H(:, j): j, the permuted rows of column j below j, and the patterns of j's
    // children. Both sorted ascending.
    std::vector<std::vector<int>> h((size_t)n);
    std::vector<int> seen((size_t)std::max(m, n) + (size_t)n, 0), reach;
    out.rptr.assign(1, 0);
    for (int j = 0; j < n; ++j) {
        const int stamp = j + 1;
        seen[j] = stamp;
        h[j].push_back(j);
        reach.clear();
        const size_t r0 = out.rrows.size();
        for (int p = b.ptr[j]; p < b.ptr[j + 1]; ++p) {
            const int i = b.idx[p];
            for (int k = first[i]; k != -1 && k < j && seen[k] != stamp; k = out.parent[k]) {
                reach.push_back(k);
                seen[k] = stamp;
            }
            const int ip = out.row_perm[i];
            if (ip > j && seen[ip] < stamp) {
                h[j].push_back(ip);
                seen[ip] = stamp;
            }
        }
        while (!reach.empty()) {
            const int k = reach.back();
            reach.pop_back();
            out.rrows.push_back(k);
            if (out.parent[k] != j) continue;
            for (int r : h[k])
                if (seen[r] < stamp) {
                    seen[r] = stamp;
                    h[j].push_back(r);
                }
        }
        std::sort(out.rrows.begin() + (ptrdiff_t)r0, out.rrows.end());
        out.rrows.push_back(j);
        out.rptr.push_back((int)out.rrows.size());
    }
    out.hptr.assign(1, 0);
    for (int j = 0; j < n; ++j) {
        std::sort(h[j].begin(), h[j].end());
        out.hrows.insert(out.hrows.end(), h[j].begin(), h[j].end());
        out.hptr.push_back((int)out.hrows.size());
    }
    return true;
}

}  // namespace qr
}  // namespace fx
