// Host-side structure of the large-component path (fx_sparse.hip): ordering, symbolic Cholesky, gather lists and the
// elimination-tree schedules. Pure host code, no floating point — the reference keeps COLAMD + symbolic analysis on the
// host as well (solvi/src/decomposition/sparse/qr.rs:118-206).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

#include "../../include/fiksi_amd.h"
#include "fx_expr.h"
#include "fx_front_plan.h"

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
// (ascending columns, walked in order) grouped in levels, a workgroup barrier after each level; the lists of a level are
// dealt to the workgroup's wavefronts here (heaviest first, each to the least loaded wavefront) and a wavefront's share of
// a level is ONE run of columns — it sets its pipeline up once per level, not once per list. Level 0
// holds whole subtrees under a work cap, the columns above them form chains (a column joins the chain of its only
// child above the cap; where several meet a new chain starts, one level above the deepest list feeding it).
// Segments 0 .. nparts-1 are parts — independent forests, each a workgroup of the same launch — and segment nparts,
// the top, is everything above them: one workgroup, after the parts (factorization) or before them (backward sweep).
struct TeamSchedule {
    uint32_t nparts = 0;
    std::vector<uint32_t> seg_lev;  // [nseg + 1] first level of each segment
    std::vector<uint32_t> wptr;     // [nlev * TEAM_WAVES + 1]: wavefront w walks cols[wptr[q * TEAM_WAVES + w] .. wptr[q * TEAM_WAVES + w + 1]) in level q
    std::vector<uint32_t> cols;     // [nv] the columns in walking order: segment by segment, level by level, wavefront by wavefront
    std::vector<uint32_t> col_seg;  // [nv] segment of a column
    bool empty() const { return seg_lev.empty(); }
    uint32_t nseg() const { return seg_lev.empty() ? 0u : (uint32_t)seg_lev.size() - 1u; }
    uint32_t nlev() const { return wptr.empty() ? 0u : ((uint32_t)wptr.size() - 1u) / TEAM_WAVES; }
};

// Lists and levels of ONE segment (`cols` ascending; parents outside the segment do not count), appended to `out`.
// Returns the critical-path estimate: per level max(longest list, level work / TEAM_WAVES) + a barrier.
inline uint64_t schedule_segment(const std::vector<uint32_t>& cols, const std::vector<uint32_t>& parent,
                                 const std::vector<uint64_t>& work, std::vector<uint32_t>& local_of /* [nv] scratch */,
                                 TeamSchedule& out) {
    const uint32_t n = (uint32_t)cols.size();
    if (out.wptr.empty()) out.wptr.push_back(0);
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
    // per level: the lists dealt to the wavefronts, heaviest first, each to the least loaded one; a wavefront's columns of a
    // level are contiguous in `cols` (list after list, ascending inside a list)
    const uint32_t nlists = (uint32_t)list_level.size();
    uint32_t nlevels = 0;
    for (uint32_t v : list_level) nlevels = std::max(nlevels, v + 1);
    std::vector<std::vector<uint32_t>> by_level(nlevels);
    for (uint32_t q = 0; q < nlists; ++q) by_level[list_level[q]].push_back(q);
    std::vector<std::vector<uint32_t>> list_cols_local(nlists);
    for (uint32_t i = 0; i < n; ++i) list_cols_local[list_of[i]].push_back(cols[i]);
    uint64_t cost = 0;
    for (uint32_t v = 0; v < nlevels; ++v) {
        std::vector<uint32_t>& ls = by_level[v];
        std::stable_sort(ls.begin(), ls.end(), [&](uint32_t x, uint32_t y) { return list_work[x] > list_work[y]; });
        uint64_t load[TEAM_WAVES] = {0};
        std::vector<uint32_t> share[TEAM_WAVES];
        for (uint32_t q : ls) {
            uint32_t w = 0;
            for (uint32_t k = 1; k < TEAM_WAVES; ++k)
                if (load[k] < load[w]) w = k;
            load[w] += list_work[q];
            share[w].push_back(q);
        }
        uint64_t slowest = 0;
        for (uint32_t w = 0; w < TEAM_WAVES; ++w) {
            slowest = std::max(slowest, load[w]);
            for (uint32_t q : share[w]) out.cols.insert(out.cols.end(), list_cols_local[q].begin(), list_cols_local[q].end());
            out.wptr.push_back((uint32_t)out.cols.size());
        }
        cost += slowest + TEAM_SYNC_COST;
    }
    return cost;
}

// The parts of a factor: the maximal subtrees under total / target_parts, packed into at most target_parts parts
// (heaviest first into the lightest part); seg[j] = part of column j, or the number of parts (returned) for the top —
// the columns above every such subtree.
inline uint32_t partition_tree(const std::vector<uint32_t>& parent, const std::vector<uint64_t>& work, uint32_t target_parts,
                               std::vector<uint32_t>& seg) {
    const uint32_t nv = (uint32_t)parent.size();
    std::vector<uint64_t> sub(work);
    uint64_t total = 0;
    for (uint32_t j = 0; j < nv; ++j) {
        total += work[j];
        if (parent[j] != NOPARENT) sub[parent[j]] += sub[j];
    }
    const uint64_t cap = std::max<uint64_t>(total / std::max(target_parts, 1u), 1);
    std::vector<uint32_t> roots;  // maximal subtrees under the cap
    for (uint32_t j = 0; j < nv; ++j)
        if (sub[j] <= cap && (parent[j] == NOPARENT || sub[parent[j]] > cap)) roots.push_back(j);
    std::stable_sort(roots.begin(), roots.end(), [&](uint32_t x, uint32_t y) { return sub[x] > sub[y]; });
    // (one part per subtree while there are at most 240 of them — a part per CU at most; beyond, the lightest are packed)
    const uint32_t np = (uint32_t)std::min<size_t>(roots.size(), std::max(target_parts, 240u));
    std::vector<uint64_t> load(np, 0);
    std::vector<uint32_t> part_of_root(nv, NOPARENT);
    for (uint32_t r : roots) {
        uint32_t lightest = 0;
        for (uint32_t q = 1; q < np; ++q)
            if (load[q] < load[lightest]) lightest = q;
        load[lightest] += sub[r];
        part_of_root[r] = lightest;
    }
    // a column's part is that of the subtree root above it; columns above every root are the top
    seg.assign(nv, np);
    for (uint32_t j = nv; j-- > 0;) {
        if (part_of_root[j] != NOPARENT) seg[j] = part_of_root[j];
        else if (parent[j] != NOPARENT && seg[parent[j]] != np && sub[j] <= cap) seg[j] = seg[parent[j]];
    }
    return np;
}

// The schedule of a factor whose columns are dealt to segments already (seg[j] in [0, nparts]: nparts = the top;
// nparts = 0: one segment holds everything). Returns the critical-path estimate: the slowest part plus the top.
inline uint64_t build_team_schedule(const std::vector<uint32_t>& parent, const std::vector<uint64_t>& work, const std::vector<uint32_t>& seg,
                                    uint32_t nparts, TeamSchedule& out) {
    const uint32_t nv = (uint32_t)parent.size();
    out = TeamSchedule();
    out.nparts = nparts;
    out.col_seg.assign(nv, 0);
    std::vector<uint32_t> local_of(nv, 0);
    std::vector<std::vector<uint32_t>> seg_cols((size_t)nparts + 1);
    for (uint32_t j = 0; j < nv; ++j) {
        const uint32_t sg = nparts ? seg[j] : 0u;
        seg_cols[sg].push_back(j);
        out.col_seg[j] = sg;
    }
    uint64_t slowest_part = 0, top = 0;
    out.seg_lev.push_back(0);
    out.wptr.push_back(0);
    for (size_t sgm = 0; sgm < seg_cols.size(); ++sgm) {
        const uint64_t cost = schedule_segment(seg_cols[sgm], parent, work, local_of, out);
        if (sgm + 1 == seg_cols.size()) top = cost;
        else slowest_part = std::max(slowest_part, cost);
        out.seg_lev.push_back(out.nlev());
    }
    return slowest_part + top;
}

// What the parts schedule of a large factor carries beyond lists (fx_sparse_team.h: the LDS builds of the spt kernels).
// The columns are numbered so that every segment is one run of columns — and of entries of L — parts first, the top
// last: a workgroup keeps its segment's entries in LDS at [entry - seg_ent[s]]. A product L_ik L_jk belongs to the
// segment of column k; the products of a top entry that belong to a part are summed BY that part, from its LDS, into
// a slot of a contribution buffer (runs: a part's share of one entry's product list, which is sorted by k and therefore
// by segment), the top subtracts its entries' slots before it starts; likewise the forward sweep's row gathers.
struct PartsExtra {
    std::vector<uint32_t> seg_col, seg_ent;          // [nseg + 1] first column / first entry of L of each segment
    std::vector<uint32_t> tpair_ptr, tpairs, tpair_k;  // the top's own products: [n_top_entries + 1], 2 per product, target entry
    std::vector<uint32_t> frun_ptr, frun;            // [nparts + 1]; 3 per run: slot, first product, end (into lpairs)
    std::vector<uint32_t> fslot_ptr;                 // [n_top_entries + 1]: the slots of a top entry are consecutive
    std::vector<uint32_t> brun_ptr, brun;            // the same for the right-hand side: 3 per run: slot, first row entry, end (into ridx / rcol)
    std::vector<uint32_t> bslot_ptr;                 // [n_top_columns + 1]
    std::vector<uint32_t> rmid;                      // [n_top_columns] first entry of row j of L whose column is in the top
    std::vector<uint32_t> cmid;                      // [nv] first entry of column j whose row is in the top (part columns; = end for top columns)
    std::vector<uint32_t> erow_ptr, erows;           // [nseg + 1], [m]: the block's rows by segment (a row's columns lie in one part and the top)
    uint32_t max_part_ent = 0, max_part_cols = 0;
    bool empty() const { return seg_col.empty(); }
};

// A segment's index data as one self-contained block of words ("blob") in SEGMENT-LOCAL numbering — entry of L minus
// the segment's first entry, column minus its first column — so that a workgroup can copy it into LDS once and walk its
// columns without a single index load from HBM (fx_sparse_team.h). Layout (32-bit words):
//   [0] nlev  [1] ncols  [2] nent  [3] nprod  [4] nrowent  [5..11] word offsets of: wptr, cdesc, pair_ptr, lrow, pairs, pair_k,
//   rows  [12] 0  [13] nlev (the two together: seg_lev of a one-segment schedule)  [14..15] 0
//   wptr[nlev * TEAM_WAVES + 1] (positions in cdesc) | cdesc[ncols][8]: column, beg, end, rbeg, rend, pbeg, pend0, mid |
//   pair_ptr[nent + 1] | lrow[nent] as u16 | pairs[nprod]: one word each, x | y << 16 | pair_k[nprod] as u16 |
//   rows[nrowent]: one word each, entry | column << 16        (16-bit local indices: a segment has < 65 536 entries)
// `pairs` / `pair_ptr` / `pair_k`: the gather lists the segment's entries are computed from (for the top of a parts
// schedule: its own products only, PartsExtra); rows: row j of L for the forward sweep (the top: its own columns only).
constexpr uint32_t BLOB_HDR = 16;
constexpr uint32_t BLOB_SOLO_MAX_ENTRIES = 12000;  // a factor beyond this cannot sit in LDS with its index data anyway
struct SegmentBlobs {
    std::vector<uint32_t> words;     // all segments, each 16-byte aligned
    std::vector<uint32_t> seg_off;   // [nseg + 1] first word of each segment's blob
    uint32_t max_words = 0;          // largest part (the top excluded)
    uint32_t top_words = 0;
    bool empty() const { return seg_off.empty(); }
};

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
    PartsExtra px;                                     // ... and what its LDS builds need
    SegmentBlobs solo_blob, parts_blobs;               // the schedules' index data, segment by segment, for LDS
    FrontPlan fronts_solo, fronts_parts;               // the multifrontal build (fx_front_plan.h): one segment / parts + top
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
    // --- large factors: the parts (subtrees of the elimination tree, one workgroup each) and the top. A first symbolic
    // pass (patterns only) finds the tree of this order; the columns are then numbered again, part by part and the top
    // last, ascending inside each — a topological order of the same tree (children stay before parents, parts do not
    // see each other), so the fill is the same and every segment becomes one run of columns and of entries of L.
    std::vector<uint32_t> col_seg;  // (new numbering; empty: one segment)
    uint32_t nparts = 0;
    if (P.nv >= TEAM_PARTS_MIN_COLUMNS) {
        std::vector<std::vector<uint32_t>> pat(P.nv);  // lower pattern of A, then of L, by column
        for (uint32_t r = 0; r < P.m; ++r) {
            std::vector<uint32_t> nc;
            for (uint32_t c : rowcols[r]) nc.push_back(iperm[c]);
            std::sort(nc.begin(), nc.end());
            for (size_t x = 0; x < nc.size(); ++x)
                for (size_t y = x; y < nc.size(); ++y) pat[nc[x]].push_back(nc[y]);
        }
        std::vector<uint32_t> parent0(P.nv, NOPARENT);
        std::vector<uint64_t> work0(P.nv, 0);
        std::vector<std::vector<uint32_t>> kids(P.nv);
        for (uint32_t j = 0; j < P.nv; ++j) {
            std::vector<uint32_t>& q = pat[j];
            q.push_back(j);
            std::sort(q.begin(), q.end());
            q.erase(std::unique(q.begin(), q.end()), q.end());
            for (uint32_t ch : kids[j]) {
                std::vector<uint32_t> merged;
                std::set_union(q.begin(), q.end(), pat[ch].begin() + 1, pat[ch].end(), std::back_inserter(merged));
                q.swap(merged);
                std::vector<uint32_t>().swap(pat[ch]);  // (a child's pattern is used once)
            }
            q.erase(q.begin(), std::lower_bound(q.begin(), q.end(), j));
            if (q.size() > 1) {
                parent0[j] = q[1];
                kids[q[1]].push_back(j);
            }
            work0[j] = 8 + q.size() / 4;  // (weights for the partition only: the schedules use the real gather lists)
        }
        const uint32_t target = std::min(240u, std::max(16u, P.nv / 160u));
        std::vector<uint32_t> seg0;
        nparts = partition_tree(parent0, work0, target, seg0);
        std::vector<uint32_t> order(P.nv);
        for (uint32_t j = 0; j < P.nv; ++j) order[j] = j;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return seg0[x] < seg0[y]; });
        std::vector<uint32_t> perm1(P.nv);
        col_seg.resize(P.nv);
        for (uint32_t k = 0; k < P.nv; ++k) {
            perm1[k] = P.perm[order[k]];
            col_seg[k] = seg0[order[k]];
        }
        P.perm.swap(perm1);
        for (uint32_t k = 0; k < P.nv; ++k) iperm[P.perm[k]] = k;
    }

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
    build_team_schedule(parent, work, std::vector<uint32_t>(), 0, P.solo);
    if (nparts) {
        build_team_schedule(parent, work, col_seg, nparts, P.parts);
        PartsExtra& X = P.px;
        const uint32_t nseg = nparts + 1;
        X.seg_col.assign((size_t)nseg + 1, P.nv);
        for (uint32_t j = P.nv; j-- > 0;) X.seg_col[col_seg[j]] = j;  // (monotone: the first column of each segment)
        for (uint32_t sgm = nseg; sgm-- > 0;)
            if (X.seg_col[sgm] > X.seg_col[sgm + 1]) X.seg_col[sgm] = X.seg_col[sgm + 1];  // (an empty segment)
        X.seg_ent.resize((size_t)nseg + 1);
        for (uint32_t sgm = 0; sgm <= nseg; ++sgm) X.seg_ent[sgm] = P.lcolptr[X.seg_col[sgm]];
        for (uint32_t sgm = 0; sgm < nparts; ++sgm) {
            X.max_part_ent = std::max(X.max_part_ent, X.seg_ent[sgm + 1] - X.seg_ent[sgm]);
            X.max_part_cols = std::max(X.max_part_cols, X.seg_col[sgm + 1] - X.seg_col[sgm]);
        }
        const uint32_t ctop = X.seg_col[nparts], etop = X.seg_ent[nparts];
        std::vector<uint32_t> ecol(P.nnz_l);  // column of an entry of L
        for (uint32_t j = 0; j < P.nv; ++j)
            for (uint32_t k = P.lcolptr[j]; k < P.lcolptr[j + 1]; ++k) ecol[k] = j;
        // the products of the top's entries: runs by part (a product list is sorted by source column), then the top's own
        std::vector<std::vector<uint32_t>> fruns(nparts), bruns(nparts);
        X.tpair_ptr.assign(1, 0);
        X.fslot_ptr.assign(1, 0);
        uint32_t fslots = 0;
        for (uint32_t e = etop; e < P.nnz_l; ++e) {
            uint32_t pp = P.lpair_ptr[e];
            const uint32_t pe = P.lpair_ptr[e + 1];
            while (pp < pe) {
                const uint32_t sgm = col_seg[ecol[P.lpairs[2 * (size_t)pp]]];
                uint32_t q = pp;
                while (q < pe && col_seg[ecol[P.lpairs[2 * (size_t)q]]] == sgm) ++q;
                if (sgm < nparts) {
                    fruns[sgm].insert(fruns[sgm].end(), {fslots++, pp, q});
                } else {
                    for (uint32_t t = pp; t < q; ++t) {
                        X.tpairs.push_back(P.lpairs[2 * (size_t)t]);
                        X.tpairs.push_back(P.lpairs[2 * (size_t)t + 1]);
                        X.tpair_k.push_back(e);
                    }
                }
                pp = q;
            }
            X.tpair_ptr.push_back((uint32_t)X.tpair_k.size());
            X.fslot_ptr.push_back(fslots);
        }
        // the forward sweep's row gathers of the top's columns, likewise (row j of L is sorted by column)
        X.bslot_ptr.assign(1, 0);
        uint32_t bslots = 0;
        for (uint32_t j = ctop; j < P.nv; ++j) {
            uint32_t pp = P.rptr[j];
            const uint32_t pe = P.rptr[j + 1];
            uint32_t mid = pe;
            while (pp < pe) {
                const uint32_t sgm = col_seg[P.rcol[pp]];
                uint32_t q = pp;
                while (q < pe && col_seg[P.rcol[q]] == sgm) ++q;
                if (sgm < nparts) bruns[sgm].insert(bruns[sgm].end(), {bslots++, pp, q});
                else mid = std::min(mid, pp);
                pp = q;
            }
            X.rmid.push_back(mid);
            X.bslot_ptr.push_back(bslots);
        }
        X.frun_ptr.assign(1, 0);
        X.brun_ptr.assign(1, 0);
        for (uint32_t sgm = 0; sgm < nparts; ++sgm) {
            X.frun.insert(X.frun.end(), fruns[sgm].begin(), fruns[sgm].end());
            X.frun_ptr.push_back((uint32_t)X.frun.size() / 3);
            X.brun.insert(X.brun.end(), bruns[sgm].begin(), bruns[sgm].end());
            X.brun_ptr.push_back((uint32_t)X.brun.size() / 3);
        }
        // the rows by segment: the part of a row's columns (they lie in ONE part and the top — two parts never share a row, or
        // their columns would meet in A), the top for rows without a part's column
        {
            std::vector<uint32_t> rseg(P.m, nparts);
            for (uint32_t r2 = 0; r2 < P.m; ++r2)
                for (uint32_t q = P.jrow_ptr[r2]; q < P.jrow_ptr[r2 + 1]; ++q) rseg[r2] = std::min(rseg[r2], col_seg[P.jcol[q]]);
            X.erow_ptr.assign((size_t)nseg + 1, 0);
            for (uint32_t r2 = 0; r2 < P.m; ++r2) X.erow_ptr[rseg[r2] + 1]++;
            for (uint32_t sgm = 0; sgm < nseg; ++sgm) X.erow_ptr[sgm + 1] += X.erow_ptr[sgm];
            X.erows.resize(P.m);
            std::vector<uint32_t> fill(X.erow_ptr.begin(), X.erow_ptr.end() - 1);
            for (uint32_t r2 = 0; r2 < P.m; ++r2) X.erows[fill[rseg[r2]]++] = r2;
        }
        // backward sweep of a part column: its entries whose rows are in the top come last (rows ascend)
        X.cmid.resize(P.nv);
        for (uint32_t j = 0; j < P.nv; ++j) {
            uint32_t k = P.lcolptr[j + 1];
            if (j < ctop)
                while (k > P.lcolptr[j] + 1 && P.lrow[k - 1] >= ctop) --k;
            X.cmid[j] = k;
        }
    }
    // --- the multifrontal build: the same tree cut into fronts that fit a row of 16 lanes (ok = false: no such build)
    build_front_plan(P.nv, P.lcolptr, P.lrow, P.l2a, acolptr, std::vector<uint32_t>(), 0, P.fronts_solo);
    if (nparts) build_front_plan(P.nv, P.lcolptr, P.lrow, P.l2a, acolptr, col_seg, nparts, P.fronts_parts);
    // --- the segments' index data as LDS-ready blobs
    auto build_blobs = [&](const TeamSchedule& T, bool with_top_lists, SegmentBlobs& out) {
        const uint32_t nseg = T.nseg();
        out.seg_off.assign(1, 0);
        for (uint32_t sgm = 0; sgm < nseg; ++sgm) {  // (16-bit local indices)
            const uint32_t c0 = T.nparts ? P.px.seg_col[sgm] : 0u, c1 = T.nparts ? P.px.seg_col[sgm + 1] : P.nv;
            if (c1 - c0 >= 0xFFFFu || P.lcolptr[c1] - P.lcolptr[c0] >= 0xFFFFu) {
                out = SegmentBlobs();
                return;
            }
        }
        for (uint32_t sgm = 0; sgm < nseg; ++sgm) {
            const bool is_top = with_top_lists && sgm == T.nparts;
            const uint32_t q0 = T.seg_lev[sgm], q1 = T.seg_lev[sgm + 1], nlev = q1 - q0;
            const uint32_t t_first = T.wptr[q0 * TEAM_WAVES], t_last = T.wptr[q1 * TEAM_WAVES], nc = t_last - t_first;
            // the segment's run of columns / entries (a one-segment schedule: everything)
            const uint32_t cb = T.nparts ? P.px.seg_col[sgm] : 0u, ce = T.nparts ? P.px.seg_col[sgm + 1] : P.nv;
            const uint32_t eb = P.lcolptr[cb], ee = P.lcolptr[ce], ne = ee - eb;
            const uint32_t etop = is_top ? eb : 0u;
            auto pptr = [&](uint32_t e) { return is_top ? P.px.tpair_ptr[e - etop] : P.lpair_ptr[e]; };
            const uint32_t p0 = pptr(eb), np = pptr(ee) - p0;
            // row entries, segment-local positions: row j's (for the top: from its first own column on)
            std::vector<uint32_t> rpos((size_t)(ce - cb) + 1, 0);
            for (uint32_t j = cb; j < ce; ++j) {
                const uint32_t rb = is_top ? P.px.rmid[j - cb] : P.rptr[j];
                rpos[j - cb + 1] = rpos[j - cb] + (P.rptr[j + 1] - rb);
            }
            const uint32_t nr = rpos[ce - cb];
            std::vector<uint32_t> w(BLOB_HDR, 0);
            auto align2 = [&]() { if (w.size() & 1u) w.push_back(0); };
            w[0] = nlev; w[1] = nc; w[2] = ne; w[3] = np; w[4] = nr; w[12] = 0; w[13] = nlev;
            w[5] = (uint32_t)w.size();
            for (uint32_t q = q0 * TEAM_WAVES; q <= q1 * TEAM_WAVES; ++q) w.push_back(T.wptr[q] - t_first);
            align2();
            w[6] = (uint32_t)w.size();
            for (uint32_t t = t_first; t < t_last; ++t) {
                const uint32_t j = T.cols[t], beg = P.lcolptr[j], end = P.lcolptr[j + 1];
                const uint32_t mid = (T.nparts && !P.px.cmid.empty()) ? P.px.cmid[j] : end;
                w.insert(w.end(), {j - cb, beg - eb, end - eb, rpos[j - cb], rpos[j - cb + 1], pptr(beg) - p0, pptr(std::min(beg + 64u, end)) - p0, mid - eb});
            }
            w[7] = (uint32_t)w.size();
            for (uint32_t e = eb; e <= ee; ++e) w.push_back(pptr(e) - p0);
            auto push16 = [&](const std::vector<uint32_t>& h) {  // two 16-bit values per word, low half first
                for (size_t i = 0; i < h.size(); i += 2) w.push_back((h[i] & 0xFFFFu) | ((i + 1 < h.size() ? h[i + 1] : 0u) << 16));
            };
            w[8] = (uint32_t)w.size();
            {
                std::vector<uint32_t> h;
                for (uint32_t e = eb; e < ee; ++e) h.push_back(P.lrow[e] >= cb && P.lrow[e] < ce ? P.lrow[e] - cb : 0xFFFFu);
                push16(h);
            }
            w[9] = (uint32_t)w.size();
            for (uint32_t t = 0; t < np; ++t) {
                const uint32_t x = is_top ? P.px.tpairs[2 * (size_t)(p0 + t)] : P.lpairs[2 * (size_t)(p0 + t)];
                const uint32_t y = is_top ? P.px.tpairs[2 * (size_t)(p0 + t) + 1] : P.lpairs[2 * (size_t)(p0 + t) + 1];
                w.push_back((x - eb) | ((y - eb) << 16));
            }
            w[10] = (uint32_t)w.size();
            {
                std::vector<uint32_t> h;
                for (uint32_t t = 0; t < np; ++t) h.push_back((is_top ? P.px.tpair_k[p0 + t] : P.lpair_k[p0 + t]) - eb);
                push16(h);
            }
            w[11] = (uint32_t)w.size();
            for (uint32_t j = cb; j < ce; ++j)
                for (uint32_t r = is_top ? P.px.rmid[j - cb] : P.rptr[j]; r < P.rptr[j + 1]; ++r) w.push_back((P.ridx[r] - eb) | ((P.rcol[r] - cb) << 16));
            while (w.size() & 3u) w.push_back(0);
            if (sgm == T.nparts) out.top_words = (uint32_t)w.size();
            else out.max_words = std::max(out.max_words, (uint32_t)w.size());
            out.words.insert(out.words.end(), w.begin(), w.end());
            out.seg_off.push_back((uint32_t)out.words.size());
        }
    };
    if (P.nnz_l <= BLOB_SOLO_MAX_ENTRIES) build_blobs(P.solo, false, P.solo_blob);
    if (nparts) build_blobs(P.parts, true, P.parts_blobs);
#ifdef FX_PLAN_TIMING
    fprintf(stderr, "[plan] schedules %.2f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_sched0).count());
#endif
}

}  // namespace sparse_plan
}  // namespace fx
