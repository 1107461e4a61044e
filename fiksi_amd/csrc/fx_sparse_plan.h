// Host-side structure of the large-component path (fx_sparse.hip): ordering, symbolic Cholesky, gather lists and the
// elimination-tree schedules. Pure host code, no floating point — the reference keeps COLAMD + symbolic analysis on the
// host as well (solvi/src/decomposition/sparse/qr.rs:118-206).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "../../include/fiksi_amd.h"
#include "fx_expr.h"

namespace fx {
namespace sparse_plan {

// reverse Cuthill-McKee order of the column graph of A (adjacency given as sorted lists)
inline std::vector<uint32_t> rcm_order(const std::vector<std::vector<uint32_t>>& adj) {
    const uint32_t n = (uint32_t)adj.size();
    std::vector<uint32_t> order;
    order.reserve(n);
    std::vector<uint8_t> seen(n, 0);
    std::vector<uint32_t> by_degree(n);
    for (uint32_t i = 0; i < n; ++i) by_degree[i] = i;
    std::stable_sort(by_degree.begin(), by_degree.end(),
                     [&](uint32_t x, uint32_t y) { return adj[x].size() < adj[y].size(); });
    std::vector<uint32_t> nb;
    for (uint32_t start : by_degree) {
        if (seen[start]) continue;
        // pseudo-peripheral start: walk to the last node of a BFS twice
        uint32_t root = start;
        for (int pass = 0; pass < 2; ++pass) {
            std::vector<uint32_t> q{root};
            std::vector<uint8_t> mark(n, 0);
            mark[root] = 1;
            size_t head = 0;
            while (head < q.size()) {
                uint32_t u = q[head++];
                for (uint32_t w : adj[u])
                    if (!mark[w] && !seen[w]) {
                        mark[w] = 1;
                        q.push_back(w);
                    }
            }
            root = q.back();
        }
        size_t head = order.size();
        order.push_back(root);
        seen[root] = 1;
        while (head < order.size()) {
            uint32_t u = order[head++];
            nb.clear();
            for (uint32_t w : adj[u])
                if (!seen[w]) {
                    seen[w] = 1;
                    nb.push_back(w);
                }
            std::stable_sort(nb.begin(), nb.end(), [&](uint32_t x, uint32_t y) { return adj[x].size() < adj[y].size(); });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
    std::reverse(order.begin(), order.end());
    return order;  // order[new] = old
}

// Nested-dissection order of the column graph (George's automatic scheme): split the level structure
// of a breadth-first search from a pseudo-peripheral node at its median level, number the two halves
// recursively and the separator last. Each half holds at most half of the nodes, so the recursion is
// O(log n) deep; the separators become the top of the elimination tree and the halves independent
// subtrees — that independence is what the device schedule runs in parallel. Pieces of up to `leaf`
// nodes (and pieces a median level cannot split) are numbered by reverse Cuthill-McKee.
inline std::vector<uint32_t> nd_order(const std::vector<std::vector<uint32_t>>& adj, uint32_t leaf = 48) {
    const uint32_t n = (uint32_t)adj.size();
    std::vector<uint32_t> order;
    order.reserve(n);
    std::vector<uint32_t> piece(n, 0);   // id of the piece a node currently belongs to
    std::vector<uint32_t> level(n, 0), local(n, 0);
    uint32_t next_piece = 1;

    auto rcm_piece = [&](const std::vector<uint32_t>& nodes) {
        std::vector<std::vector<uint32_t>> sub(nodes.size());
        for (uint32_t k = 0; k < nodes.size(); ++k) local[nodes[k]] = k;
        const uint32_t id = piece[nodes[0]];
        for (uint32_t k = 0; k < nodes.size(); ++k)
            for (uint32_t w : adj[nodes[k]])
                if (piece[w] == id) sub[k].push_back(local[w]);
        for (uint32_t k : rcm_order(sub)) order.push_back(nodes[k]);
    };

    struct Job { std::vector<uint32_t> nodes; bool emit_only; };  // emit_only: a separator, numbered as is
    std::vector<Job> jobs;
    {
        std::vector<uint32_t> all(n);
        for (uint32_t i = 0; i < n; ++i) all[i] = i;
        if (n) jobs.push_back({std::move(all), false});
    }
    std::vector<uint32_t> queue;
    while (!jobs.empty()) {
        Job job = std::move(jobs.back());
        jobs.pop_back();
        if (job.emit_only) {
            order.insert(order.end(), job.nodes.begin(), job.nodes.end());
            continue;
        }
        const uint32_t id = next_piece++;
        for (uint32_t v : job.nodes) piece[v] = id;
        if (job.nodes.size() <= leaf) {
            rcm_piece(job.nodes);
            continue;
        }
        // one connected part at a time: the rest of the piece is pushed back untouched
        auto bfs = [&](uint32_t root) {
            queue.assign(1, root);
            const uint32_t tag = next_piece++;
            piece[root] = tag;
            level[root] = 0;
            for (size_t head = 0; head < queue.size(); ++head) {
                uint32_t u = queue[head];
                for (uint32_t w : adj[u])
                    if (piece[w] == id) {
                        piece[w] = tag;
                        level[w] = level[u] + 1;
                        queue.push_back(w);
                    }
            }
            for (uint32_t v : queue) piece[v] = id;  // restore
        };
        bfs(job.nodes[0]);
        if (queue.size() < job.nodes.size()) {  // disconnected: split off this part
            std::vector<uint32_t> part = queue, rest;
            const uint32_t tag = next_piece++;
            for (uint32_t v : part) piece[v] = tag;
            for (uint32_t v : job.nodes)
                if (piece[v] == id) rest.push_back(v);
            jobs.push_back({std::move(rest), false});
            jobs.push_back({std::move(part), false});
            continue;
        }
        bfs(queue.back());  // twice from the far end: a pseudo-peripheral root
        bfs(queue.back());
        const uint32_t depth = level[queue.back()];
        uint32_t cut = 0;
        {
            std::vector<uint32_t> count(depth + 1, 0);
            for (uint32_t v : queue) count[level[v]]++;
            uint32_t below = 0;
            while (cut < depth && 2 * (below + count[cut]) < queue.size()) below += count[cut++];
        }
        std::vector<uint32_t> lo, hi, sep;
        for (uint32_t v : queue) {
            if (level[v] < cut) lo.push_back(v);
            else if (level[v] > cut) hi.push_back(v);
            else sep.push_back(v);
        }
        if (lo.empty() || hi.empty()) {  // too few levels to cut (clique-like piece)
            rcm_piece(job.nodes);
            continue;
        }
        // numbered in pop order: lo, hi, then the separator
        jobs.push_back({std::move(sep), true});
        jobs.push_back({std::move(hi), false});
        jobs.push_back({std::move(lo), false});
    }
    return order;  // order[new] = old
}


constexpr uint32_t NOPARENT = 0xFFFFFFFFu;
constexpr uint32_t ND_LEAF = 12;                    // nodes of the column graph below which nested dissection stops
constexpr uint32_t TEAM_WAVES = 16;                 // wavefronts of the workgroup a segment is scheduled for
constexpr uint64_t TEAM_SYNC_COST = 3;              // a workgroup barrier, in the units of `work` (round trips of a wavefront)
constexpr uint32_t TEAM_PARTS_MIN_COLUMNS = 1536;   // smaller factors are one segment

// Elimination-tree schedule for workgroups ("teams" of TEAM_WAVES wavefronts). A segment is a set of columns closed
// under "descendant of" inside the columns not yet taken by earlier segments; one workgroup runs a segment: its lists
// (ascending columns, one wavefront walks a list) grouped in levels, a workgroup barrier after each level. Level 0
// holds whole subtrees under a work cap, the columns above them form chains (a column joins the chain of its only
// child above the cap; where several meet a new chain starts, one level above the deepest list feeding it).
// Segments 0 .. nparts-1 are parts — independent forests, each a workgroup of the same launch — and segment nparts,
// the top, is everything above them: one workgroup, after the parts (factorization) or before them (backward sweep).
struct TeamSchedule {
    uint32_t nparts = 0;
    std::vector<uint32_t> seg_lev;              // [nseg + 1] first level of each segment
    std::vector<uint32_t> lev_list;             // [nlev + 1] first list of each level
    std::vector<uint32_t> list_ptr, list_cols;  // [nlists + 1], [nv]
    std::vector<uint32_t> col_seg;              // [nv] segment of a column
    bool empty() const { return seg_lev.empty(); }
    uint32_t nseg() const { return seg_lev.empty() ? 0u : (uint32_t)seg_lev.size() - 1u; }
};

// Lists and levels of ONE segment (`cols` ascending; parents outside the segment do not count), appended to `out`.
// Returns the critical-path estimate: per level max(longest list, level work / TEAM_WAVES) + a barrier.
inline uint64_t schedule_segment(const std::vector<uint32_t>& cols, const std::vector<uint32_t>& parent,
                                 const std::vector<uint64_t>& work, std::vector<uint32_t>& local_of /* [nv] scratch */,
                                 TeamSchedule& out) {
    const uint32_t n = (uint32_t)cols.size();
    if (out.lev_list.empty()) out.lev_list.push_back(0);
    if (out.list_ptr.empty()) out.list_ptr.push_back(0);
    if (!n) return 0;
    for (uint32_t i = 0; i < n; ++i) local_of[cols[i]] = i;
    std::vector<uint32_t> par(n, NOPARENT);
    std::vector<uint64_t> w(n), sub(n);
    uint64_t total = 0;
    for (uint32_t i = 0; i < n; ++i) {
        const uint32_t pa = parent[cols[i]];
        // (a parent outside the segment has a larger number than every column of a part; for the top no parent is outside)
        if (pa != NOPARENT && local_of[pa] < n && cols[local_of[pa]] == pa) par[i] = local_of[pa];
        w[i] = sub[i] = work[cols[i]];
        total += w[i];
    }
    for (uint32_t i = 0; i < n; ++i)
        if (par[i] != NOPARENT) sub[par[i]] += sub[i];  // children come before parents
    std::vector<uint32_t> list_of(n), list_level, upper_children(n), feeder(n), below(n);
    std::vector<uint64_t> list_work;
    auto build = [&](uint64_t cap) -> uint64_t {
        list_level.clear();
        list_work.clear();
        std::fill(upper_children.begin(), upper_children.end(), 0u);
        std::fill(below.begin(), below.end(), 0u);  // deepest level among the lists feeding column i, plus one
        for (uint32_t i = 0; i < n; ++i)
            if (sub[i] > cap && par[i] != NOPARENT) {
                upper_children[par[i]]++;
                feeder[par[i]] = i;
            }
        for (uint32_t i = n; i-- > 0;) {  // level-0 lists: subtrees under the cap, numbered from their roots downwards
            if (sub[i] > cap) continue;
            const uint32_t pa = par[i];
            if (pa == NOPARENT || sub[pa] > cap) {
                list_of[i] = (uint32_t)list_level.size();
                list_level.push_back(0);
                list_work.push_back(sub[i]);
                if (pa != NOPARENT) below[pa] = std::max(below[pa], 1u);
            } else {
                list_of[i] = list_of[pa];
            }
        }
        for (uint32_t i = 0; i < n; ++i) {  // chains above the cap, bottom-up
            if (sub[i] <= cap) continue;
            uint32_t q;
            if (upper_children[i] == 1) {
                q = list_of[feeder[i]];  // extends its only upper child's chain (its other children are level-0 subtrees)
                list_work[q] += w[i];
            } else {
                q = (uint32_t)list_level.size();
                list_level.push_back(std::max(below[i], 1u));
                list_work.push_back(w[i]);
            }
            list_of[i] = q;
            if (par[i] != NOPARENT) below[par[i]] = std::max(below[par[i]], list_level[q] + 1);
        }
        uint32_t nlevels = 0;
        for (uint32_t v : list_level) nlevels = std::max(nlevels, v + 1);
        std::vector<uint64_t> longest(nlevels, 0), sum(nlevels, 0);
        for (size_t q = 0; q < list_level.size(); ++q) {
            longest[list_level[q]] = std::max(longest[list_level[q]], list_work[q]);
            sum[list_level[q]] += list_work[q];
        }
        uint64_t cost = 0;
        for (uint32_t v = 0; v < nlevels; ++v) cost += std::max(longest[v], (sum[v] + TEAM_WAVES - 1) / TEAM_WAVES) + TEAM_SYNC_COST;
        return cost;
    };
    uint64_t best_cap = total, best_cost = ~0ull;
    for (uint64_t cap = total;; cap = cap * 3 / 4) {
        const uint64_t cost = build(cap);
        if (cost < best_cost) {
            best_cost = cost;
            best_cap = cap;
        }
        if (cap < 12) break;
    }
    build(best_cap);
    // lists by level, within a level the heaviest first (the wavefronts take them round-robin); columns ascending in a list
    const uint32_t nlists = (uint32_t)list_level.size();
    std::vector<uint32_t> order(nlists);
    for (uint32_t q = 0; q < nlists; ++q) order[q] = q;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
        if (list_level[x] != list_level[y]) return list_level[x] < list_level[y];
        return list_work[x] > list_work[y];
    });
    std::vector<uint32_t> new_id(nlists);
    for (uint32_t q = 0; q < nlists; ++q) new_id[order[q]] = q;
    const uint32_t list0 = (uint32_t)out.list_ptr.size() - 1, col0 = out.list_ptr.back();
    std::vector<uint32_t> count(nlists, 0);
    for (uint32_t i = 0; i < n; ++i) count[new_id[list_of[i]]]++;
    for (uint32_t q = 0; q < nlists; ++q) out.list_ptr.push_back(out.list_ptr.back() + count[q]);
    out.list_cols.resize((size_t)col0 + n);
    {
        std::vector<uint32_t> fill(nlists);
        for (uint32_t q = 0; q < nlists; ++q) fill[q] = out.list_ptr[list0 + q];
        for (uint32_t i = 0; i < n; ++i) out.list_cols[fill[new_id[list_of[i]]]++] = cols[i];
    }
    uint32_t at = 0;
    while (at < nlists) {  // one lev_list entry per level
        const uint32_t lvl = list_level[order[at]];
        while (at < nlists && list_level[order[at]] == lvl) ++at;
        out.lev_list.push_back(list0 + at);
    }
    return best_cost;
}

// target_parts = 0: one segment (the top) holds everything. Otherwise the maximal subtrees under total / target_parts
// are packed into at most target_parts parts, heaviest first into the lightest part. Returns the critical-path
// estimate: the slowest part plus the top.
inline uint64_t build_team_schedule(const std::vector<uint32_t>& parent, const std::vector<uint64_t>& work, uint32_t target_parts,
                                    TeamSchedule& out) {
    const uint32_t nv = (uint32_t)parent.size();
    out = TeamSchedule();
    out.col_seg.assign(nv, 0);
    std::vector<uint32_t> local_of(nv, 0);
    std::vector<std::vector<uint32_t>> seg_cols;
    if (target_parts && nv) {
        std::vector<uint64_t> sub(work);
        uint64_t total = 0;
        for (uint32_t j = 0; j < nv; ++j) {
            total += work[j];
            if (parent[j] != NOPARENT) sub[parent[j]] += sub[j];
        }
        const uint64_t cap = std::max<uint64_t>(total / target_parts, 1);
        std::vector<uint32_t> roots;  // maximal subtrees under the cap
        for (uint32_t j = 0; j < nv; ++j)
            if (sub[j] <= cap && (parent[j] == NOPARENT || sub[parent[j]] > cap)) roots.push_back(j);
        std::stable_sort(roots.begin(), roots.end(), [&](uint32_t x, uint32_t y) { return sub[x] > sub[y]; });
        const uint32_t np = (uint32_t)std::min<size_t>(roots.size(), target_parts);
        std::vector<uint64_t> load(np, 0);
        std::vector<uint32_t> part_of_root(nv, NOPARENT);
        for (uint32_t r : roots) {
            uint32_t lightest = 0;
            for (uint32_t p = 1; p < np; ++p)
                if (load[p] < load[lightest]) lightest = p;
            load[lightest] += sub[r];
            part_of_root[r] = lightest;
        }
        // a column's part is that of the subtree root above it; columns above every root are the top
        std::vector<uint32_t> seg(nv, np);
        for (uint32_t j = nv; j-- > 0;) {
            if (part_of_root[j] != NOPARENT) seg[j] = part_of_root[j];
            else if (parent[j] != NOPARENT && seg[parent[j]] != np && sub[j] <= cap) seg[j] = seg[parent[j]];
        }
        out.nparts = np;
        seg_cols.assign((size_t)np + 1, {});
        for (uint32_t j = 0; j < nv; ++j) {
            seg_cols[seg[j]].push_back(j);
            out.col_seg[j] = seg[j];
        }
    } else {
        seg_cols.assign(1, {});
        seg_cols[0].resize(nv);
        for (uint32_t j = 0; j < nv; ++j) seg_cols[0][j] = j;
    }
    uint64_t slowest_part = 0, top = 0;
    out.seg_lev.push_back(0);
    out.lev_list.push_back(0);
    out.list_ptr.push_back(0);
    for (size_t sgm = 0; sgm < seg_cols.size(); ++sgm) {
        const uint64_t cost = schedule_segment(seg_cols[sgm], parent, work, local_of, out);
        if (sgm + 1 == seg_cols.size()) top = cost;
        else slowest_part = std::max(slowest_part, cost);
        out.seg_lev.push_back((uint32_t)out.lev_list.size() - 1);
    }
    return slowest_part + top;
}

struct ComponentPlan {
    uint32_t m = 0, nv = 0, nnz_j = 0, nnz_a = 0, nnz_l = 0;
    std::vector<uint32_t> rows, fvar;
    std::vector<uint32_t> jrow_ptr, jslot;
    std::vector<uint32_t> jcol;                        // new column of every entry of J (row-major), for the refined step
    std::vector<uint32_t> perm;                        // new column -> old column
    std::vector<uint32_t> apair_ptr, apairs;           // gather lists of A
    std::vector<uint32_t> cptr, cidx, crow;            // columns of J (permuted order) for the rhs
    std::vector<uint32_t> lcolptr, lrow, lpair_ptr, lpairs, lpair_k;
    std::vector<int32_t> l2a;
    std::vector<uint32_t> rptr, ridx, rcol;            // strictly lower part of L by rows
    TeamSchedule solo;                                 // the whole factor as one segment: one workgroup per System
    TeamSchedule parts;                                // subtrees dealt to workgroups + the top (large Systems; else empty)
};

// Builds every index structure of one component. `colof[v]` = free column of system variable v
// (ascending rank among the component's free variables) or -1.
inline void plan_component(const fx_batch* b, uint32_t s, const std::vector<uint32_t>& rows,
                           const std::vector<uint32_t>& fvar, ComponentPlan& P, uint32_t nd_leaf = ND_LEAF) {
    const uint32_t e0 = b->expr_off[s], nvt = b->var_off[s + 1] - b->var_off[s];
    P.rows = rows;
    P.fvar = fvar;
    P.m = (uint32_t)rows.size();
    P.nv = (uint32_t)fvar.size();
    std::vector<int32_t> colof(nvt, -1);
    for (uint32_t k = 0; k < P.nv; ++k) colof[fvar[k]] = (int32_t)k;

    // --- row patterns (old column numbering), adjacency of the column graph
    std::vector<std::vector<uint32_t>> rowcols(P.m);
    std::vector<std::vector<uint32_t>> adj(P.nv);
    std::vector<uint32_t> entry_cols(8 * (size_t)P.m, 0xFFFFFFFFu);
    for (uint32_t r = 0; r < P.m; ++r) {
        uint32_t e = e0 + rows[r];
        uint32_t vars8[8];
        int k = expand_vars<true>((int)b->expr_tag[e], b->expr_idx + 4 * (size_t)e, vars8);
        auto& rc = rowcols[r];
        for (int q = 0; q < k; ++q) {
            int32_t c = colof[vars8[q]];
            if (c < 0) continue;
            entry_cols[8 * (size_t)r + q] = (uint32_t)c;
            if (std::find(rc.begin(), rc.end(), (uint32_t)c) == rc.end()) rc.push_back((uint32_t)c);
        }
        for (uint32_t x : rc)
            for (uint32_t y : rc)
                if (x != y) adj[x].push_back(y);
    }
    for (auto& a : adj) {
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
    }
    P.perm = nd_order(adj, nd_leaf);
    std::vector<uint32_t> iperm(P.nv);
    for (uint32_t k = 0; k < P.nv; ++k) iperm[P.perm[k]] = k;

    // --- J in CSR with columns in the permuted numbering, slots ascending by new column
    P.jrow_ptr.assign((size_t)P.m + 1, 0);
    P.jslot.assign(P.m, 0xFFFFFFFFu);
    std::vector<uint32_t> jcol;  // new column of every J entry
    for (uint32_t r = 0; r < P.m; ++r) {
        std::vector<uint32_t> nc;
        for (uint32_t c : rowcols[r]) nc.push_back(iperm[c]);
        std::sort(nc.begin(), nc.end());
        uint32_t slots = 0;
        for (int q = 0; q < 8; ++q) {
            uint32_t sl = 0xFu, c = entry_cols[8 * (size_t)r + q];
            if (c != 0xFFFFFFFFu) sl = (uint32_t)(std::find(nc.begin(), nc.end(), iperm[c]) - nc.begin());
            slots |= sl << (4 * q);
        }
        P.jslot[r] = slots;
        jcol.insert(jcol.end(), nc.begin(), nc.end());
        P.jrow_ptr[r + 1] = (uint32_t)jcol.size();
    }
    P.nnz_j = (uint32_t)jcol.size();
    P.jcol = jcol;

    // --- columns of J (for the rhs) and pattern of A (lower triangle, new numbering)
    std::vector<uint32_t> ccount(P.nv + 1, 0);
    for (uint32_t c : jcol) ccount[c + 1]++;
    for (uint32_t c = 0; c < P.nv; ++c) ccount[c + 1] += ccount[c];
    P.cptr = ccount;
    P.cidx.assign(P.nnz_j, 0);
    P.crow.assign(P.nnz_j, 0);
    {
        std::vector<uint32_t> fill(P.cptr.begin(), P.cptr.end() - 1);
        for (uint32_t r = 0; r < P.m; ++r)
            for (uint32_t p = P.jrow_ptr[r]; p < P.jrow_ptr[r + 1]; ++p) {
                uint32_t dst = fill[jcol[p]]++;
                P.cidx[dst] = p;
                P.crow[dst] = r;
            }
    }
    // A[i][j] (i >= j) exists when some row holds both columns; list rows per (i,j) in row order
    std::vector<std::vector<uint32_t>> acol(P.nv);  // rows i of column j (lower, incl. diagonal)
    for (uint32_t r = 0; r < P.m; ++r)
        for (uint32_t p = P.jrow_ptr[r]; p < P.jrow_ptr[r + 1]; ++p)
            for (uint32_t q = P.jrow_ptr[r]; q <= p; ++q) acol[jcol[q]].push_back(jcol[p]);
    std::vector<uint32_t> acolptr(P.nv + 1, 0);
    for (uint32_t j = 0; j < P.nv; ++j) {
        std::sort(acol[j].begin(), acol[j].end());
        acol[j].erase(std::unique(acol[j].begin(), acol[j].end()), acol[j].end());
        acolptr[j + 1] = acolptr[j] + (uint32_t)acol[j].size();
    }
    P.nnz_a = acolptr[P.nv];
    auto a_index = [&](uint32_t i, uint32_t j) {
        return acolptr[j] + (uint32_t)(std::lower_bound(acol[j].begin(), acol[j].end(), i) - acol[j].begin());
    };
    std::vector<uint32_t> acount(P.nnz_a + 1, 0);
    for (uint32_t r = 0; r < P.m; ++r)
        for (uint32_t p = P.jrow_ptr[r]; p < P.jrow_ptr[r + 1]; ++p)
            for (uint32_t q = P.jrow_ptr[r]; q <= p; ++q) acount[a_index(jcol[p], jcol[q]) + 1]++;
    for (uint32_t k = 0; k < P.nnz_a; ++k) acount[k + 1] += acount[k];
    P.apair_ptr = acount;
    P.apairs.assign(2 * (size_t)acount[P.nnz_a], 0);
    {
        std::vector<uint32_t> fill(P.apair_ptr.begin(), P.apair_ptr.end() - 1);
        for (uint32_t r = 0; r < P.m; ++r)
            for (uint32_t p = P.jrow_ptr[r]; p < P.jrow_ptr[r + 1]; ++p)
                for (uint32_t q = P.jrow_ptr[r]; q <= p; ++q) {
                    uint32_t dst = fill[a_index(jcol[p], jcol[q])]++;
                    P.apairs[2 * (size_t)dst] = p;
                    P.apairs[2 * (size_t)dst + 1] = q;
                }
    }

    // --- symbolic Cholesky: pattern(L_j) = pattern(A_j) U (patterns of the etree children \ child)
    std::vector<std::vector<uint32_t>> lcol(P.nv);
    std::vector<std::vector<uint32_t>> children(P.nv);
    for (uint32_t j = 0; j < P.nv; ++j) {
        std::vector<uint32_t> pat = acol[j];  // sorted, starts with j (the diagonal always exists: damping)
        if (pat.empty() || pat[0] != j) pat.insert(pat.begin(), j);
        for (uint32_t ch : children[j]) {
            std::vector<uint32_t> merged;
            merged.reserve(pat.size() + lcol[ch].size());
            std::set_union(pat.begin(), pat.end(), lcol[ch].begin() + 1, lcol[ch].end(), std::back_inserter(merged));
            pat.swap(merged);
        }
        // entries of a child's pattern are > child and >= j by construction; drop anything < j
        pat.erase(pat.begin(), std::lower_bound(pat.begin(), pat.end(), j));
        lcol[j] = pat;
        if (pat.size() > 1) children[pat[1]].push_back(j);  // etree parent = first sub-diagonal row
    }
    P.lcolptr.assign((size_t)P.nv + 1, 0);
    for (uint32_t j = 0; j < P.nv; ++j) P.lcolptr[j + 1] = P.lcolptr[j] + (uint32_t)lcol[j].size();
    P.nnz_l = P.lcolptr[P.nv];
    P.lrow.resize(P.nnz_l);
    P.l2a.assign(P.nnz_l, -1);
    for (uint32_t j = 0; j < P.nv; ++j) {
        std::copy(lcol[j].begin(), lcol[j].end(), P.lrow.begin() + P.lcolptr[j]);
        for (size_t t = 0; t < acol[j].size(); ++t) {
            uint32_t i = acol[j][t];
            uint32_t li = P.lcolptr[j] + (uint32_t)(std::lower_bound(lcol[j].begin(), lcol[j].end(), i) - lcol[j].begin());
            P.l2a[li] = (int32_t)(acolptr[j] + t);
        }
    }
    auto l_index = [&](uint32_t i, uint32_t j) {
        return P.lcolptr[j] + (uint32_t)(std::lower_bound(lcol[j].begin(), lcol[j].end(), i) - lcol[j].begin());
    };
    // gather lists: column k updates L[i][j] for every pair j <= i of its sub-diagonal rows
    std::vector<uint32_t> lcount((size_t)P.nnz_l + 1, 0);
    for (uint32_t k = 0; k < P.nv; ++k)
        for (size_t p = 1; p < lcol[k].size(); ++p)
            for (size_t q = p; q < lcol[k].size(); ++q) lcount[l_index(lcol[k][q], lcol[k][p]) + 1]++;
    for (uint32_t t = 0; t < P.nnz_l; ++t) lcount[t + 1] += lcount[t];
    P.lpair_ptr = lcount;
    P.lpairs.assign(2 * (size_t)lcount[P.nnz_l], 0);
    {
        std::vector<uint32_t> fill(P.lpair_ptr.begin(), P.lpair_ptr.end() - 1);
        for (uint32_t k = 0; k < P.nv; ++k)
            for (size_t p = 1; p < lcol[k].size(); ++p)
                for (size_t q = p; q < lcol[k].size(); ++q) {
                    uint32_t dst = fill[l_index(lcol[k][q], lcol[k][p])]++;
                    P.lpairs[2 * (size_t)dst] = P.lcolptr[k] + (uint32_t)q;      // L[i][k]
                    P.lpairs[2 * (size_t)dst + 1] = P.lcolptr[k] + (uint32_t)p;  // L[j][k]
                }
    }

    P.lpair_k.assign(P.lpairs.size() / 2, 0);
    for (uint32_t t = 0; t < P.nnz_l; ++t)
        for (uint32_t pp = P.lpair_ptr[t]; pp < P.lpair_ptr[t + 1]; ++pp) P.lpair_k[pp] = t;

    // --- L by rows (forward sweep gathers)
    P.rptr.assign((size_t)P.nv + 1, 0);
    for (uint32_t j = 0; j < P.nv; ++j)
        for (size_t t = 1; t < lcol[j].size(); ++t) P.rptr[lcol[j][t] + 1]++;
    for (uint32_t j = 0; j < P.nv; ++j) P.rptr[j + 1] += P.rptr[j];
    P.ridx.assign(P.rptr[P.nv], 0);
    P.rcol.assign(P.rptr[P.nv], 0);
    {
        std::vector<uint32_t> fill(P.rptr.begin(), P.rptr.end() - 1);
        for (uint32_t j = 0; j < P.nv; ++j)
            for (size_t t = 1; t < lcol[j].size(); ++t) {
                uint32_t dst = fill[lcol[j][t]]++;
                P.ridx[dst] = P.lcolptr[j] + (uint32_t)t;
                P.rcol[dst] = j;
            }
    }


#ifdef FX_PLAN_TIMING
    const auto t_sched0 = std::chrono::steady_clock::now();
#endif
    // --- schedules (elimination tree: parent = first sub-diagonal row; a column depends only on its descendants)
    std::vector<uint64_t> work(P.nv, 0);
    for (uint32_t j = 0; j < P.nv; ++j) {
        // critical-path cost of a column, in dependent round trips of its wavefront: a floor, plus the 64-wide passes
        // over its products, over row j of L (forward sweep) and over its own entries
        const uint64_t nprod = P.lpair_ptr[P.lcolptr[j + 1]] - P.lpair_ptr[P.lcolptr[j]];
        const uint64_t nrow = P.rptr[j + 1] - P.rptr[j], len = P.lcolptr[j + 1] - P.lcolptr[j];
        work[j] = 8 + (nprod + 63) / 64 + (nrow + 63) / 64 + (len > 64 ? 2 * ((len + 63) / 64) : 0);
    }
    std::vector<uint32_t> parent(P.nv, NOPARENT);
    for (uint32_t j = 0; j < P.nv; ++j)
        if (lcol[j].size() > 1) parent[j] = lcol[j][1];
    build_team_schedule(parent, work, 0, P.solo);
    if (P.nv >= TEAM_PARTS_MIN_COLUMNS) {
        uint64_t best = ~0ull;
        for (uint32_t target : {32u, 64u, 128u, 240u}) {
            if (P.nv / target < 24u && target != 32u) break;
            TeamSchedule t;
            const uint64_t cost = build_team_schedule(parent, work, target, t);
            if (cost < best) {
                best = cost;
                P.parts = std::move(t);
            }
        }
    }
#ifdef FX_PLAN_TIMING
    fprintf(stderr, "[plan] schedules %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sched0).count());
#endif
}

}  // namespace sparse_plan
}  // namespace fx
