// ORACLE — test infrastructure only (see fo_common.hpp).
// Restates the symbolic phase of solvi's sparse QR:
//   solvi/src/utils.rs:49-117 (post_order), :153-188 (node_depth_levels)
//   solvi/src/permutation.rs:41-80 (PermutationSequence)
//   solvi/src/decomposition/sparse/cholesky.rs:31-84 (elimination_tree), :149-332
//   (CholeskyCounts::build, Gilbert-Ng-Peyton), :359-594 (CholeskyStructure::build, Davis §5.3).
#pragma once
#include <algorithm>
#include <cassert>
#include <utility>
#include <vector>

#include "fo_common.hpp"
#include "fo_sparse.hpp"

namespace fo {

// utils.rs:49-117. `head[parent]` ends up as
// the highest-numbered child because nodes are linked in ascending order, each pushed on the
// front of its parent's list.
inline std::vector<size_t> post_order(const std::vector<size_t>& parents) {
    size_t n = parents.size();
    std::vector<size_t> head(n, NONE), next(n, NONE);
    for (size_t node = 0; node < n; ++node) {
        size_t parent = parents[node];
        if (parent != NONE) {
            next[node] = head[parent];
            head[parent] = node;
        }
    }
    std::vector<size_t> post;
    post.reserve(n);
    std::vector<size_t> stack;
    for (size_t root = 0; root < n; ++root) {
        if (parents[root] != NONE) continue;
        size_t node = root;
        for (;;) {
            while (head[node] != NONE) {
                size_t child = head[node];
                head[node] = next[child];
                stack.push_back(node);
                node = child;
            }
            post.push_back(node);
            if (stack.empty()) break;
            node = stack.back();
            stack.pop_back();
        }
    }
    return post;
}

// utils.rs:153-188
inline std::pair<std::vector<size_t>, size_t> node_depth_levels(const std::vector<size_t>& parents) {
    size_t n = parents.size();
    size_t max_level = 0;
    std::vector<size_t> levels(n, 0);
    std::vector<size_t> path;
    for (size_t start = 0; start < n; ++start) {
        size_t node = start;
        while (levels[node] == 0 && parents[node] != NONE) {
            path.push_back(node);
            node = parents[node];
        }
        size_t level = levels[node];
        while (!path.empty()) {
            size_t v = path.back();
            path.pop_back();
            level += 1;
            levels[v] = level;
            max_level = std::max(max_level, level);
        }
    }
    return {levels, max_level};
}

// permutation.rs:27-80
struct PermutationSequence {
    std::vector<std::pair<size_t, size_t>> swap_sequence;
    size_t indices_len = 0;

    // permutation.rs:41-65: after applying the swaps, p[i] = a[permutation[i]].
    static PermutationSequence build_for_gather_permutation(const std::vector<size_t>& permutation) {
        PermutationSequence out;
        out.indices_len = permutation.size();
        std::vector<bool> seen(permutation.size(), false);
        std::vector<size_t> stack;
        for (size_t start : permutation) {
            size_t i = start;
            while (!seen[i]) {
                stack.push_back(i);
                seen[i] = true;
                i = permutation[i];
            }
            if (!stack.empty()) {
                size_t pivot = stack[0];
                for (size_t k = stack.size(); k-- > 1;) {  // drain(..).skip(1).rev()
                    out.swap_sequence.emplace_back(pivot, stack[k]);
                }
                stack.clear();
            }
        }
        return out;
    }

    // permutation.rs:70-80
    template <typename T>
    void permute_slice(T* slice) const {
        for (const auto& s : swap_sequence) std::swap(slice[s.first], slice[s.second]);
    }
};

// cholesky.rs:31-84. SYMMETRIC=false: elimination tree of AᵀA without forming it.
template <bool SYMMETRIC>
inline std::vector<size_t> elimination_tree(const SparseColMatStructure& a) {
    size_t m = a.nrows, n = a.ncols;
    std::vector<size_t> parents(n, NONE), ancestors(n, NONE);
    std::vector<size_t> prev_col(SYMMETRIC ? 0 : m, NONE);
    for (size_t col = 0; col < n; ++col) {
        for (const size_t* rp = a.col_begin(col); rp != a.col_end(col); ++rp) {
            size_t row = *rp;
            size_t k = SYMMETRIC ? row : prev_col[row];
            while (k != NONE) {
                if (k >= col) break;
                size_t col_next = ancestors[k];
                ancestors[k] = col;
                if (col_next == NONE) parents[k] = col;
                k = col_next;
            }
            if (!SYMMETRIC) prev_col[row] = col;
        }
    }
    return parents;
}

// cholesky.rs:96-103
struct CholeskyCounts {
    std::vector<size_t> row_counts, col_counts, levels, first_columns;

    // cholesky.rs:149-332
    static CholeskyCounts build(const SparseColMatStructure& a, const std::vector<size_t>& parents,
                                const std::vector<size_t>& postorder) {
        size_t m = a.nrows, n = a.ncols;
        assert(n == parents.size() && n == postorder.size());

        std::vector<size_t> levels = node_depth_levels(parents).first;

        std::vector<size_t> places_in_postorder(n, 0);
        for (size_t place = 0; place < n; ++place) places_in_postorder[postorder[place]] = place;

        std::vector<size_t> subtree_size(n, 1);
        for (size_t j : postorder) {
            size_t parent = parents[j];
            if (parent != NONE) subtree_size[parent] += subtree_size[j];
        }

        std::vector<size_t> first_descendants(n, 0);
        for (size_t place = 0; place < n; ++place) {
            size_t j = postorder[place];
            first_descendants[j] = postorder[place + 1 - subtree_size[j]];
        }

        std::vector<size_t> first_columns(m, NONE);
        for (size_t j : postorder) {
            for (const size_t* rp = a.col_begin(j); rp != a.col_end(j); ++rp) {
                if (first_columns[*rp] == NONE) first_columns[*rp] = j;
            }
        }

        std::vector<std::vector<size_t>> hadj_f(n);
        for (size_t place = 0; place < n; ++place) {
            size_t j = postorder[place];
            for (const size_t* rp = a.col_begin(j); rp != a.col_end(j); ++rp) {
                size_t f = first_columns[*rp];
                if (place > places_in_postorder[f]) hadj_f[f].push_back(j);
            }
        }

        std::vector<ptrdiff_t> vertex_weights(n, 0);
        for (size_t j = 0; j < n; ++j) vertex_weights[j] = subtree_size[j] == 1 ? 1 : 0;

        std::vector<size_t> col_counts(n, 1);
        std::vector<size_t> prev_nbr(n, NONE), prev_f(n, NONE);
        std::vector<size_t> dsu_parent(n);
        for (size_t j = 0; j < n; ++j) dsu_parent[j] = j;
        auto find = [&](size_t x) {
            // recursive path compression in the reference (cholesky.rs:257-262); iterative here.
            size_t root = x;
            while (dsu_parent[root] != root) root = dsu_parent[root];
            while (dsu_parent[x] != root) {
                size_t nx = dsu_parent[x];
                dsu_parent[x] = root;
                x = nx;
            }
            return root;
        };

        for (size_t j_place = 0; j_place < n; ++j_place) {
            size_t j = postorder[j_place];
            if (parents[j] != NONE) vertex_weights[parents[j]] -= 1;
            size_t first_descendant_place = places_in_postorder[first_descendants[j]];
            for (size_t u : hadj_f[j]) {
                // cholesky.rs:282: `+1` on both sides with wrapping for the "unseen" encoding.
                if (first_descendant_place + 1 > prev_nbr[u] + 1) {
                    vertex_weights[j] += 1;
                    size_t p_leaf = prev_f[u];
                    if (p_leaf != NONE) {
                        size_t q = find(p_leaf);
                        col_counts[u] += levels[j] - levels[q];
                        vertex_weights[q] -= 1;
                    } else {
                        col_counts[u] += levels[j] - levels[u];
                    }
                    prev_f[u] = j;
                }
                prev_nbr[u] = j_place;
            }
            size_t parent = parents[j];
            if (parent != NONE) dsu_parent[j] = parent;
        }

        for (size_t j = 0; j < n; ++j) {
            size_t parent = parents[j];
            if (parent != NONE) vertex_weights[parent] += vertex_weights[j];
        }

        CholeskyCounts out;
        out.row_counts.resize(n);
        for (size_t j = 0; j < n; ++j) out.row_counts[j] = static_cast<size_t>(vertex_weights[j]);
        out.col_counts = std::move(col_counts);
        out.levels = std::move(levels);
        out.first_columns = std::move(first_columns);
        return out;
    }
};

// cholesky.rs:119-138
struct CholeskyStructure {
    SparseColMatStructure l_structure;    // structure of R (= Lᵀ of AᵀA)
    std::vector<size_t> row_permutation;  // original row -> permuted row, length m + n
    SparseColMatStructure h_structure;    // Householder vectors, one per column

    // cholesky.rs:359-594
    static CholeskyStructure build(const SparseColMatStructure& a, const std::vector<size_t>& parents,
                                   const std::vector<size_t>& postorder, const CholeskyCounts& cholesky) {
        (void)postorder;  // only used for the (unused) Householder row counts in the reference
        size_t m = a.nrows, n = a.ncols;
        const auto& col_counts = cholesky.col_counts;
        const auto& first_columns = cholesky.first_columns;

        size_t m_fictitious = m;
        std::vector<size_t> row_permutation(m + n, NONE);
        {
            // cholesky.rs:381-442 (Davis, "Direct Methods for Sparse Linear Systems", §5.3)
            std::vector<size_t> next(m, 0), head(n, NONE), tail(n, NONE);
            std::vector<ptrdiff_t> nqueue(n, 0);
            for (size_t ii = m; ii-- > 0;) {
                size_t i = ii;
                size_t k = first_columns[i];
                if (k == NONE) continue;
                if (nqueue[k] == 0) tail[k] = i;
                nqueue[k] += 1;
                next[i] = head[k];
                head[k] = i;
            }
            for (size_t k = 0; k < n; ++k) {
                size_t i;
                if (head[k] == NONE) {
                    i = m_fictitious;
                    m_fictitious += 1;
                } else {
                    i = head[k];
                }
                row_permutation[i] = k;
                nqueue[k] -= 1;
                if (nqueue[k] <= 0) continue;
                size_t parent = parents[k];
                if (parent != NONE) {
                    if (nqueue[parent] == 0) tail[parent] = tail[k];
                    next[tail[k]] = head[parent];
                    head[parent] = next[i];
                    nqueue[parent] += nqueue[k];
                }
            }
            size_t k = n;
            for (size_t i = 0; i < m; ++i) {
                if (row_permutation[i] == NONE) {
                    row_permutation[i] = k;
                    k += 1;
                }
            }
        }

        // cholesky.rs:444-504
        std::vector<std::vector<size_t>> h_row_indices(n);
        size_t num_non_zero = 0;
        for (size_t c : col_counts) num_non_zero += c;
        std::vector<size_t> row_indices(num_non_zero, 0);
        {
            std::vector<size_t> stack;
            stack.reserve(n);
            std::vector<size_t> marker(m + n, 0);
            size_t start = 0;
            for (size_t j = 0; j < n; ++j) {
                marker[j] = j + 1;
                h_row_indices[j].push_back(j);
                for (const size_t* rp = a.col_begin(j); rp != a.col_end(j); ++rp) {
                    size_t i = *rp;
                    size_t k = first_columns[i];
                    while (k != NONE && k < j && marker[k] != j + 1) {
                        stack.push_back(k);
                        marker[k] = j + 1;
                        k = parents[k];
                    }
                    size_t ip = row_permutation[i];
                    if (ip > j && marker[ip] < j + 1) {
                        h_row_indices[j].push_back(ip);
                        marker[ip] = j + 1;
                    }
                }
                size_t idx = start;
                while (!stack.empty()) {
                    size_t k = stack.back();
                    stack.pop_back();
                    row_indices[idx] = k;
                    idx += 1;
                    if (parents[k] == j) {
                        for (size_t row : h_row_indices[k]) {
                            if (marker[row] < j + 1) {
                                marker[row] = j + 1;
                                h_row_indices[j].push_back(row);
                            }
                        }
                    }
                }
                std::sort(row_indices.begin() + start, row_indices.begin() + idx);
                row_indices[idx] = j;
                start += col_counts[j];
            }
        }

        // cholesky.rs:506-567 computes Householder row counts / vertex weights that are never
        // read afterwards; they have no effect on the returned structure and are omitted.

        for (auto& col : h_row_indices) std::sort(col.begin(), col.end());  // :569-571

        CholeskyStructure out;
        out.l_structure.nrows = n;
        out.l_structure.ncols = n;
        out.l_structure.column_pointers.assign(1, 0);
        {
            size_t sum = 0;
            for (size_t c : col_counts) {
                sum += c;
                out.l_structure.column_pointers.push_back(sum);
            }
        }
        out.l_structure.row_indices = std::move(row_indices);
        out.h_structure.nrows = m;
        out.h_structure.ncols = n;
        out.h_structure.column_pointers.assign(1, 0);
        {
            size_t sum = 0;
            for (const auto& col : h_row_indices) {
                sum += col.size();
                out.h_structure.column_pointers.push_back(sum);
                out.h_structure.row_indices.insert(out.h_structure.row_indices.end(), col.begin(), col.end());
            }
        }
        out.row_permutation = std::move(row_permutation);
        return out;
    }
};

}  // namespace fo
