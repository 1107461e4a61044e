// Host-side structure of the MULTIFRONTAL build of the large-component path (fx_front.h is its device side): the
// elimination tree of fx_sparse_plan.h's symbolic Cholesky cut into FRONTS — a front is a connected piece of the tree (its
// pivot columns, in elimination order) together with the rows their columns of L reach (its boundary) —, small enough that a
// front's dense matrix, the right-hand side riding along as one more column, fits ONE ROW OF 16 LANES: pivots + boundary
// <= 15. The walkers of fx_sparse_team.h eliminate one column at a time, a wavefront per column, 0.6 - 1.1 us each; a
// front's columns are eliminated inside a lane row's registers (fx_grouped_rows.h's DPP-row Cholesky, stopped after the
// pivots: what is left in the boundary's lanes is the Schur complement, the contribution to the parent front), four fronts
// per wavefront, some 70 ns per pivot. Chain-like sketches — cfg2, the reference's hinged triangles — have separators of a
// few columns, so their fronts are small; a structure with a front beyond 15 columns keeps the walkers (ok = false).
// Pure host code, no floating point: as the reference keeps COLAMD + the symbolic analysis on the host
// (solvi/src/decomposition/sparse/qr.rs:118-206; what this replaces is the numeric phase, qr.rs:281-356).
#pragma once
#include <stdint.h>

#include <algorithm>
#include <vector>

namespace fx {
namespace sparse_plan {

constexpr uint32_t MF_N = 16;              // lanes of a front's row: pivots, boundary, the right-hand side
constexpr uint32_t MF_FMAX = MF_N - 1;     // columns of a front
constexpr uint32_t MF_TS = MF_N + 1;       // a staging tile in LDS: 17 rows of 17 doubles (rows of 17: no bank conflicts; row 16
constexpr uint32_t MF_TILE = MF_TS * MF_TS;  // takes what a child's padding adds — it is never read)
constexpr uint32_t MF_LS = MF_N + 1;       // doubles between the columns of an L / contribution block: 17, not 16 — sixteen lanes that store or
                                           // load their columns side by side then fall into sixteen different LDS banks (a stride of 128 bytes: into one)
constexpr uint32_t MF_HDR = 20;            // words of a segment blob's header
constexpr uint32_t MF_FRONT_WORDS = 8;     // words of a front's record
constexpr uint32_t MF_CHILD_WORDS = 6;     // words of a child's entry
constexpr uint32_t MF_U_GLOBAL = 1u;       // front flag: the contribution block goes to global memory (a part's root: its parent is in the top)
constexpr uint32_t MF_KIDS_W8 = 2u;        // front flags: every child's block has at most 7 (3) rows — two (four) children are added at a time, a group of
constexpr uint32_t MF_KIDS_W4 = 4u;        // 8 (4) lanes each (a hinge point shared by 60 triangles is one front with 60 children)

// Storage of a front, laid out so that a lane stores its 16 registers with ONE address and sixteen immediate offsets, no
// test per register (what the test would keep out lands in padding):
//   L block, (npiv + 1) x 17 doubles: column c of the front's pivots at [17 c, 17 c + 16) — rows 0 .. 15 of the lane's registers, the
//     diagonal slot overwritten with 1 / d —, then the right-hand side's lane at [17 npiv, ...): its rows below npiv are y of the pivots;
//   contribution block, 17 + (nbnd + 1) x 17 doubles, u_off pointing behind the first 17: column k of the boundary (k = nbnd: the
//     right-hand side) at [17 k - npiv, 17 k - npiv + 16) — row i of the lane's registers at 17 k + (i - npiv), so the Schur
//     complement's rows are [17 k, 17 k + nbnd) and what lies around them is padding (npiv + nbnd <= 15: a column's padding takes
//     the next column's rows above npiv, the 17 doubles in front those of the first column).
// One segment's fronts as a self-contained block of words (copied into LDS once per launch), everything the device walks:
//   [0] fronts [1] levels [2] entries of A in the segment [3] columns [4] first entry of A [5] first column
//   [6 .. 10] word offsets of: lev_ptr, fronts, recs, cols, children [11] doubles of L storage [12] doubles of local
//   contribution storage (slots reused: a block lives from its front's level to its parent's) [13] words in all [14] most fronts
//   in a level [15] doubles of global contribution storage it writes [16] the segment's largest front + 2 (diagnostics)
//   lev_ptr[levels + 1] | fronts[.][8]: npiv | nbnd << 8 | nchild << 16 | flags << 24, rec_off, nrec, cols_off, child_off,
//   l_off, u_off, id (of the whole plan) | recs: li | lj << 4 | (entry of A - first) << 8 | cols: the front's columns
//   (pivots in elimination order, then the boundary ascending; numbers of the whole factor) | children, 6 words per child:
//   u_off | global << 31, nb, 16 bytes: the ROW of this front that row r of the child's block adds into (r < nb; 16 — the
//   tile's spare row — beyond). Column c of the child's block goes to the same place as a column for c < nb, and its right-hand
//   side (c = nb) to this front's right-hand side (column pivots + boundary).
struct FrontPlan {
    bool ok = false;
    uint32_t nfronts = 0, nseg = 0;
    uint32_t max_levels = 0, max_level_fronts = 0;
    uint32_t max_l_doubles = 0, max_u_doubles = 0;   // per segment (LDS)
    uint32_t global_u_doubles = 0;                   // the parts' roots (global memory)
    uint32_t max_blob_words = 0, top_blob_words = 0;
    uint32_t max_seg_a = 0, max_seg_cols = 0, top_a = 0, top_cols = 0;
    uint32_t max_ts = 0;                             // the largest staging tile's side
    std::vector<uint32_t> words;    // all segments' blobs, each 16-byte aligned
    std::vector<uint32_t> seg_off;  // [nseg + 1]
    std::vector<uint32_t> seg_a;    // [nseg + 1] first entry of A of each segment (A entries are by column: contiguous)
    std::vector<uint32_t> seg_col;  // [nseg + 1]
};

// lcolptr / lrow: the factor's pattern by columns (first entry of a column: its diagonal); l2a[k] >= 0: entry k of L starts
// as entry l2a[k] of A; acolptr: entries of A by column; col_seg: segment of a column (empty: one segment); nparts: the
// parts (segments 0 .. nparts - 1; segment nparts is the top). Columns are numbered so that children come before parents
// and every segment is one run of columns.
inline void build_front_plan(uint32_t nv, const std::vector<uint32_t>& lcolptr, const std::vector<uint32_t>& lrow, const std::vector<int32_t>& l2a,
                             const std::vector<uint32_t>& acolptr, const std::vector<uint32_t>& col_seg, uint32_t nparts, FrontPlan& out) {
    out = FrontPlan();
    if (nv == 0) return;
    const uint32_t nseg = nparts ? nparts + 1u : 1u;
    auto seg_of = [&](uint32_t j) { return nparts ? col_seg[j] : 0u; };
    constexpr uint32_t NONE = 0xFFFFFFFFu;
    // ---- fronts by amalgamation, bottom-up: a column starts as its own front (boundary = its column of L below the
    // diagonal); a front is merged into the front of its root's parent when the two together still fit a row of lanes
    // (the child's boundary lies inside the parent's pivots + boundary, so the merged boundary is the parent's)
    struct Sn {
        std::vector<uint32_t> piv;  // ascending
        uint32_t root = 0;          // last pivot
        bool alive = true;
    };
    std::vector<Sn> sn(nv);
    std::vector<uint32_t> sn_of(nv);
    auto parent_col = [&](uint32_t j) { return lcolptr[j + 1] - lcolptr[j] > 1u ? lrow[lcolptr[j] + 1] : NONE; };
    auto bnd_size = [&](uint32_t root) { return lcolptr[root + 1] - lcolptr[root] - 1u; };
    for (uint32_t j = 0; j < nv; ++j) {
        sn[j].piv.assign(1, j);
        sn[j].root = j;
        sn_of[j] = j;
    }
    for (uint32_t j = 0; j < nv; ++j) {  // (ascending roots: a front has taken in its children before it looks at its parent)
        const uint32_t s = sn_of[j];
        if (sn[s].root != j) continue;
        const uint32_t pc = parent_col(j);
        if (pc == NONE || seg_of(pc) != seg_of(j)) continue;
        const uint32_t p = sn_of[pc];
        if (sn[s].piv.size() + sn[p].piv.size() + bnd_size(sn[p].root) > MF_FMAX) continue;
        std::vector<uint32_t> merged(sn[s].piv.size() + sn[p].piv.size());
        std::merge(sn[s].piv.begin(), sn[s].piv.end(), sn[p].piv.begin(), sn[p].piv.end(), merged.begin());
        sn[p].piv.swap(merged);
        for (uint32_t c : sn[s].piv) sn_of[c] = p;
        sn[s].alive = false;
        std::vector<uint32_t>().swap(sn[s].piv);
    }
    // ---- the fronts, numbered by their roots (children before parents)
    struct Front {
        std::vector<uint32_t> cols;  // pivots, then boundary
        uint32_t npiv = 0, nbnd = 0, seg = 0, level = 0, parent = NONE;
        std::vector<uint32_t> children;
        uint32_t l_off = 0, u_off = 0, flags = 0;
        std::vector<uint32_t> recs;
    };
    std::vector<Front> fr;
    std::vector<uint32_t> front_of(nv, NONE);
    {
        std::vector<uint32_t> id_of_sn(nv, NONE);
        for (uint32_t j = 0; j < nv; ++j) {
            const uint32_t s = sn_of[j];
            if (sn[s].root != j) continue;
            id_of_sn[s] = (uint32_t)fr.size();
            Front f;
            f.cols = sn[s].piv;
            f.npiv = (uint32_t)f.cols.size();
            for (uint32_t k = lcolptr[j] + 1; k < lcolptr[j + 1]; ++k) f.cols.push_back(lrow[k]);
            f.nbnd = (uint32_t)f.cols.size() - f.npiv;
            f.seg = seg_of(j);
            if (f.cols.size() > MF_FMAX) return;  // (a single column with more than 14 rows below it: no multifrontal build)
            fr.push_back(std::move(f));
        }
        for (uint32_t j = 0; j < nv; ++j) front_of[j] = id_of_sn[sn_of[j]];
    }
    const uint32_t nf = (uint32_t)fr.size();
    for (uint32_t f = 0; f < nf; ++f) {
        Front& F = fr[f];
        if (F.nbnd) {
            F.parent = front_of[F.cols[F.npiv]];  // the front of the first boundary column = of the root's parent
            fr[F.parent].children.push_back(f);
        }
    }
    // levels inside a segment; a part's root (parent in the top) hands its contribution over through global memory
    uint32_t gu = 0;
    for (uint32_t f = 0; f < nf; ++f) {
        Front& F = fr[f];
        for (uint32_t c : F.children)
            if (fr[c].seg == F.seg) F.level = std::max(F.level, fr[c].level + 1u);
        if (F.parent != NONE && fr[F.parent].seg != F.seg) {
            F.flags |= MF_U_GLOBAL;
            F.u_off = gu + MF_LS;
            gu += MF_LS + (F.nbnd + 1u) * MF_LS;
        }
    }
    out.global_u_doubles = gu;
    // ---- the records of a front: where the entries of A of its pivot columns go
    std::vector<uint32_t> local(nv, NONE);
    for (uint32_t f = 0; f < nf; ++f) {
        Front& F = fr[f];
        for (uint32_t t = 0; t < F.cols.size(); ++t) local[F.cols[t]] = t;
        for (uint32_t lj = 0; lj < F.npiv; ++lj) {
            const uint32_t j = F.cols[lj];
            for (uint32_t k = lcolptr[j]; k < lcolptr[j + 1]; ++k) {
                if (l2a[k] < 0) continue;
                const uint32_t li = local[lrow[k]];
                if (li == NONE) return;  // (cannot happen: a column's rows lie inside its front)
                F.recs.push_back(li | (lj << 4) | ((uint32_t)l2a[k] << 8));  // (the entry of A: made segment-local below)
            }
        }
        for (uint32_t t = 0; t < F.cols.size(); ++t) local[F.cols[t]] = NONE;
    }
    // ---- segments: first column / first entry of A, the fronts of each by level
    out.nseg = nseg;
    out.nfronts = nf;
    out.seg_col.assign((size_t)nseg + 1, nv);
    for (uint32_t j = nv; j-- > 0;) out.seg_col[seg_of(j)] = j;
    for (uint32_t s = nseg; s-- > 0;)
        if (out.seg_col[s] > out.seg_col[s + 1]) out.seg_col[s] = out.seg_col[s + 1];
    out.seg_a.resize((size_t)nseg + 1);
    for (uint32_t s = 0; s <= nseg; ++s) out.seg_a[s] = acolptr[out.seg_col[s]];
    std::vector<std::vector<uint32_t>> seg_fronts(nseg);
    for (uint32_t f = 0; f < nf; ++f) seg_fronts[fr[f].seg].push_back(f);
    out.seg_off.assign(1, 0);
    for (uint32_t s = 0; s < nseg; ++s) {
        std::vector<uint32_t>& ids = seg_fronts[s];
        std::stable_sort(ids.begin(), ids.end(), [&](uint32_t x, uint32_t y) { return fr[x].level < fr[y].level; });
        uint32_t nlev = 0;
        for (uint32_t f : ids) nlev = std::max(nlev, fr[f].level + 1u);
        // storage inside the segment: L blocks one after the other; the local contribution blocks in slots of one size that are
        // taken when a front's level starts and given back when its parent's level is over (LDS is what bounds the top of a
        // large sketch: a balanced tree keeps about half of its fronts' blocks alive at a time)
        uint32_t l_at = 0, u_at = 0, gu_seg = 0;
        {
            uint32_t slot = 0;  // doubles of a slot: the segment's largest local block
            for (uint32_t f : ids)
                if (!(fr[f].flags & MF_U_GLOBAL) && fr[f].nbnd) slot = std::max(slot, MF_LS + (fr[f].nbnd + 1u) * MF_LS);
            std::vector<uint32_t> free_slots;
            std::vector<std::vector<uint32_t>> release(nlev + 1u);  // slots to give back once level q is over
            uint32_t nslots = 0;
            size_t at = 0;
            for (uint32_t q = 0; q < nlev; ++q) {
                for (; at < ids.size() && fr[ids[at]].level == q; ++at) {
                    Front& F = fr[ids[at]];
                    F.l_off = l_at;
                    l_at += (F.npiv + 1u) * MF_LS;
                    if (F.flags & MF_U_GLOBAL) {
                        gu_seg += MF_LS + (F.nbnd + 1u) * MF_LS;
                    } else if (F.nbnd) {  // (a root hands nothing on)
                        uint32_t s2;
                        if (free_slots.empty()) s2 = nslots++;
                        else {
                            s2 = free_slots.back();
                            free_slots.pop_back();
                        }
                        F.u_off = s2 * slot + MF_LS;
                        release[fr[F.parent].level].push_back(s2);
                    }
                }
                for (uint32_t s2 : release[q]) free_slots.push_back(s2);
            }
            u_at = nslots * slot;
        }
        l_at = (l_at + 1u) & ~1u;
        u_at = (u_at + 1u) & ~1u;
        const uint32_t a0 = out.seg_a[s], na = out.seg_a[s + 1] - a0, c0 = out.seg_col[s], nc = out.seg_col[s + 1] - c0;
        if (na >= (1u << 24)) return;
        uint32_t ts = 3;  // the staging tile's side: the largest front, its right-hand side's column, a spare row; odd (LDS banks)
        for (uint32_t f : ids) ts = std::max(ts, fr[f].npiv + fr[f].nbnd + 2u);
        ts |= 1u;
        std::vector<uint32_t> w(MF_HDR, 0);
        w[16] = ts;
        w[0] = (uint32_t)ids.size();
        w[1] = nlev;
        w[2] = na;
        w[3] = nc;
        w[4] = a0;
        w[5] = c0;
        w[11] = l_at;
        w[12] = u_at;
        w[15] = gu_seg;
        w[6] = (uint32_t)w.size();
        {
            std::vector<uint32_t> lev_ptr(nlev + 1u, 0);
            for (uint32_t f : ids) lev_ptr[fr[f].level + 1]++;
            uint32_t widest = 0;
            for (uint32_t q = 0; q < nlev; ++q) {
                widest = std::max(widest, lev_ptr[q + 1]);
                lev_ptr[q + 1] += lev_ptr[q];
            }
            w[14] = widest;
            out.max_level_fronts = std::max(out.max_level_fronts, widest);
            w.insert(w.end(), lev_ptr.begin(), lev_ptr.end());
        }
        w[7] = (uint32_t)w.size();
        const size_t fr_at = w.size();
        w.resize(w.size() + (size_t)ids.size() * MF_FRONT_WORDS, 0);
        std::vector<uint32_t> recs, cols, kids;
        for (size_t q = 0; q < ids.size(); ++q) {
            const Front& F = fr[ids[q]];
            uint32_t* d = &w[fr_at + q * MF_FRONT_WORDS];
            if (F.children.size() > 255u) return;
            uint32_t kflags = 0, widest = 0;
            for (uint32_t c : F.children) widest = std::max(widest, fr[c].nbnd);
            if (!F.children.empty()) kflags = widest <= 3u ? MF_KIDS_W4 : widest <= 7u ? MF_KIDS_W8 : 0u;
            d[0] = F.npiv | (F.nbnd << 8) | ((uint32_t)F.children.size() << 16) | ((F.flags | kflags) << 24);
            d[1] = (uint32_t)recs.size();
            d[2] = (uint32_t)F.recs.size();
            d[3] = (uint32_t)cols.size();
            d[4] = (uint32_t)kids.size();
            d[5] = F.l_off;
            d[6] = F.u_off;
            d[7] = ids[q];
            for (uint32_t r : F.recs) {
                const uint32_t a = r >> 8;
                if (a < a0 || a - a0 >= na) return;  // (cannot happen: a pivot column's entries of A belong to its segment)
                recs.push_back((r & 0xFFu) | ((a - a0) << 8));
            }
            cols.insert(cols.end(), F.cols.begin(), F.cols.end());
            for (uint32_t c : F.children) {
                const Front& C = fr[c];
                kids.push_back(C.u_off | ((C.flags & MF_U_GLOBAL) ? 0x80000000u : 0u));
                kids.push_back(C.nbnd);
                uint8_t map[MF_N];  // row of this front for row r of the child's block; the tile's spare row past the block
                for (uint32_t r = 0; r < MF_N; ++r) map[r] = (uint8_t)MF_N;
                for (uint32_t r = 0; r < C.nbnd; ++r) {
                    const uint32_t col = C.cols[C.npiv + r];
                    const auto it = std::find(F.cols.begin(), F.cols.end(), col);
                    if (it == F.cols.end()) return;  // (cannot happen)
                    map[r] = (uint8_t)(it - F.cols.begin());
                }
                for (uint32_t r = 0; r < MF_N; r += 4) kids.push_back(map[r] | (map[r + 1] << 8) | (map[r + 2] << 16) | ((uint32_t)map[r + 3] << 24));
            }
        }
        w[8] = (uint32_t)w.size();
        w.insert(w.end(), recs.begin(), recs.end());
        w[9] = (uint32_t)w.size();
        w.insert(w.end(), cols.begin(), cols.end());
        w[10] = (uint32_t)w.size();
        w.insert(w.end(), kids.begin(), kids.end());
        while (w.size() & 3u) w.push_back(0);
        w[13] = (uint32_t)w.size();
        const bool is_top = nparts && s == nparts;
        if (is_top) {
            out.top_blob_words = (uint32_t)w.size();
            out.top_a = na;
            out.top_cols = nc;
        } else {
            out.max_blob_words = std::max(out.max_blob_words, (uint32_t)w.size());
            out.max_seg_a = std::max(out.max_seg_a, na);
            out.max_seg_cols = std::max(out.max_seg_cols, nc);
        }
        out.max_levels = std::max(out.max_levels, nlev);
        out.max_ts = std::max(out.max_ts, ts);
        out.max_l_doubles = std::max(out.max_l_doubles, l_at);
        out.max_u_doubles = std::max(out.max_u_doubles, u_at);
        out.words.insert(out.words.end(), w.begin(), w.end());
        out.seg_off.push_back((uint32_t)out.words.size());
    }
    out.ok = true;
}

}  // namespace sparse_plan
}  // namespace fx
