// The table "programs" the host compiles ONCE per structure for the kernels that run batches of one structure
// (fx_grouped_c.hip, fx_grouped_s.hip, the grouped / wide FX_STEP_QR builds) and the plans of the reference's sparse QR
// (fx_qrplan.h; qr.rs:118-206 stays on the host, as COLAMD does). Structure only: no value enters a program.
#include "fx_host.h"

namespace fxh {

// ---- FX_STEP_QR: plans of the reference's sparse QR (fx_qrplan.h) ---------------------------------------
// One component (or SinglePass block) of one System: rows = its expressions in row order, free = its free
// variables in column order (both system-local). The pattern of the augmented matrix [J; sqrt(lambda) I] as
// lm.rs:81-98 builds it: column c holds the rows that read free[c] (ascending, an expression reading it twice
// once) and, last, the damping row m + c.

bool build_qr_plan(const uint8_t* expr_tag, const uint16_t* expr_idx16, const uint32_t* rows, uint32_t m, const uint32_t* free_,
                   uint32_t n, uint32_t nvt, QrHostPlan& out) {
    out = QrHostPlan();
    out.n = n;
    out.m = m;
    if (n > 64u) return false;
    if (n == 0) {  // every variable of the component is fixed: the reference's LM takes one trial with an empty step
        for (uint32_t i = 0; i < m; ++i) out.u16.push_back((uint16_t)i);
        out.u16.push_back(0);
        out.ok = true;
        return true;
    }
    std::vector<int32_t> colof(nvt, -1);
    for (uint32_t c = 0; c < n; ++c) colof[free_[c]] = (int32_t)c;
    std::vector<std::vector<int>> cols(n);
    for (uint32_t r = 0; r < m; ++r) {
        uint32_t vars8[8];
        const int k = fx::expand_vars<true>((int)(expr_tag[rows[r]] & 0x7F), expr_idx16 + 4 * (size_t)rows[r], vars8);
        for (int q = 0; q < k; ++q) {
            const int32_t c = vars8[q] < nvt ? colof[vars8[q]] : -1;
            if (c >= 0 && (cols[c].empty() || cols[c].back() != (int)r)) cols[c].push_back((int)r);
        }
    }
    fx::qr::Csc a;
    a.nrows = (int)(m + n);
    a.ncols = (int)n;
    a.ptr.assign(1, 0);
    for (uint32_t c = 0; c < n; ++c) {
        a.idx.insert(a.idx.end(), cols[c].begin(), cols[c].end());
        a.idx.push_back((int)(m + c));
        a.ptr.push_back((int)a.idx.size());
    }
    fx::qr::Symbolic sy;
    if (!fx::qr::analyze(a, true, sy)) return false;
    if (sy.hrows.size() > 0xFFFFu) return false;
    // The rows of every Householder vector below its diagonal entry are padded to a multiple of eight with row m + n — the
    // row of zeros the kernel keeps under the matrix: its register window then loads, multiplies and stores whole blocks of
    // eight with no per-entry select (a padded slot adds 0 * x = +0.0 to a sum that is never -0.0, and writes back the
    // zero it read). hptr counts the padded entries.
    std::vector<uint16_t> hptr_p(n + 1, 0), hrows_p;
    for (uint32_t j = 0; j < n; ++j) {
        hptr_p[j] = (uint16_t)hrows_p.size();
        const int b = sy.hptr[j], e = sy.hptr[j + 1];
        for (int q = b; q < e; ++q) hrows_p.push_back((uint16_t)sy.hrows[q]);
        const int below = e - b - 1;
        for (int q = below; q < ((below + 7) & ~7); ++q) hrows_p.push_back((uint16_t)(m + n));
        if (hrows_p.size() > 0xFFFFu) return false;
    }
    hptr_p[n] = (uint16_t)hrows_p.size();
    out.nnzh = (uint32_t)hrows_p.size();
    out.u16.reserve(n + (m + n) + (n + 1) + hrows_p.size());
    for (uint32_t j = 0; j < n; ++j) out.u16.push_back((uint16_t)sy.col_perm[j]);
    for (uint32_t i = 0; i < m + n; ++i) out.u16.push_back((uint16_t)sy.row_perm[i]);
    out.u16.insert(out.u16.end(), hptr_p.begin(), hptr_p.end());
    out.u16.insert(out.u16.end(), hrows_p.begin(), hrows_p.end());
    out.u64.assign(2 * (size_t)n, 0);
    for (uint32_t j = 0; j < n; ++j)
        for (int p = sy.rptr[j]; p < sy.rptr[j + 1] - 1; ++p) {
            const uint32_t k = (uint32_t)sy.rrows[p];
            out.u64[j] |= 1ull << k;
            out.u64[n + k] |= 1ull << j;
        }
    out.ok = true;
    return true;
}

// The program of the grouped kernel's one-structure build (fx_grouped_c.hip): everything about a System's STRUCTURE that kernel
// needs, written once for a batch whose Systems all share it — one component, at most NV = 32 (48) variables and expressions
// (every expression a row of the component), 17 ... 32 (33 ... 48) free variables: two (three) matrix columns per lane. Jt J is kept by its pattern: a slot per structural non-zero of
// the lower triangle (all NV diagonal entries included: the columns past the free variables are identity padding), one slot of
// zero behind them. Words:
// [0] version [1] variables [2] expressions [3] free variables [4] products (padded to 64) [5] right-hand-side entries (padded
// to 64) [6] slots (even, the zero slot included) [7] compact Jacobian entries (even) [8] the zero slot [17] words in all [18] words
// of the part the f64 builds copy [9 ... 12] byte offsets of the f32 build's gather tables behind it;
// then, at the byte offsets of fx_device.h's GcTable: vcol (i8 [NV]: variable -> free column, -1 = fixed), fidx (u8 [NV]: free
// column -> variable), rtag (u8 [NV]), gbase (u16 [NV]: first compact entry of a row), gvar (u8 [NV][8]: the variables a row
// reads, gradient order), the load table (u8 [16][NC NV]: lane l's element i of column l + 16 q at [l][NV q + i] — the slot of
// (max, min), or the zero slot), the right-hand side (u32: entry | row << 8 | column << 16) and behind it the products (u32:
// entry a | entry b << 8 | slot << 16, 0xFFFFFFFF = padding; rows ascending, a ascending, b from a upward, a pair of entries on
// one column twice — the order fx_grouped.hip builds its lists in, and so the order of the additions).
template <int NC, int RC>
static bool build_gc_program_t(const uint16_t* var_info, const uint8_t* expr_tag, const uint16_t* expr_comp, const uint16_t* expr_idx16,
                               uint32_t nvt, uint32_t net, GcHostProgram& out) {  // (free variables: 1 ... NV; the caller picks the smallest build)
    using TK = fx::GcTable<NC, RC>;
    constexpr uint32_t NV = TK::NV, NR = TK::NR;
    out = GcHostProgram();
    out.nc = NC;
    out.rc = RC;
    if (nvt == 0 || nvt > NV || net == 0 || net > NR) return false;
    int8_t vcol[NV];
    uint8_t fidx[NV] = {0}, rtag[NR] = {0}, gvar[NR][8] = {{0}};
    uint16_t gbase[NR] = {0};
    uint32_t nfree = 0;
    for (uint32_t i = 0; i < NV; ++i) vcol[i] = -1;
    for (uint32_t i = 0; i < nvt; ++i) {
        if ((var_info[i] & fx::VAR_COMP_MASK) != 0) return false;  // (a variable of no component carries another number)
        if (!(var_info[i] & fx::VAR_FIXED_BIT)) {
            vcol[i] = (int8_t)nfree;
            fidx[nfree++] = (uint8_t)i;
        }
    }
    if (nfree == 0 || nfree > NV) return false;
    int gcol[NR][8];
    uint32_t ng = 0;
    for (uint32_t r = 0; r < net; ++r) {
        if (expr_comp[r] != 0) return false;
        const int tag = (int)(expr_tag[r] & 0x7F);
        if (tag >= FX_TAG_POSE_X) return false;
        uint32_t vars8[8];
        const int k = fx::expand_vars(tag, expr_idx16 + 4 * (size_t)r, vars8);
        rtag[r] = (uint8_t)tag;
        gbase[r] = (uint16_t)ng;
        for (int e = 0; e < 8; ++e) {
            if (vars8[e] >= nvt) return false;
            gvar[r][e] = (uint8_t)vars8[e];
            gcol[r][e] = e < k ? (int)vcol[vars8[e]] : -1;
        }
        ng += (uint32_t)k;
    }
    if (ng > 256u) return false;  // (a byte per compact Jacobian entry in the lists)
    // the pattern of the lower triangle, slots in packed-triangle order
    std::vector<int32_t> slot_of(NV * (NV + 1u) / 2u, -1);
    auto tri = [](uint32_t hi, uint32_t lo) { return hi * (hi + 1u) / 2u + lo; };
    for (uint32_t j = 0; j < NV; ++j) slot_of[tri(j, j)] = 0;
    for (uint32_t r = 0; r < net; ++r)
        for (int a = 0; a < 8; ++a)
            for (int bb = a; bb < 8; ++bb)
                if (gcol[r][a] >= 0 && gcol[r][bb] >= 0) {
                    const uint32_t ca = (uint32_t)gcol[r][a], cb = (uint32_t)gcol[r][bb];
                    slot_of[tri(std::max(ca, cb), std::min(ca, cb))] = 0;
                }
    uint32_t nslots = 0;
    for (int32_t& sl : slot_of)
        if (sl == 0) sl = (int32_t)nslots++;
    const uint32_t zero = nslots++;
    nslots = (nslots + 3u) & ~3u;  // (whole 16-byte vectors in f32 too)
    if (nslots > 256u) return false;
    std::vector<uint32_t> pw, pe;
    for (uint32_t r = 0; r < net; ++r)
        for (int a = 0; a < 8; ++a) {
            if (gcol[r][a] < 0) continue;
            const uint32_t ca = (uint32_t)gcol[r][a];
            pe.push_back((gbase[r] + (uint32_t)a) | (r << 8) | (ca << 16));
            for (int bb = a; bb < 8; ++bb) {
                if (gcol[r][bb] < 0) continue;
                const uint32_t cb = (uint32_t)gcol[r][bb];
                const uint32_t w = (gbase[r] + (uint32_t)a) | ((gbase[r] + (uint32_t)bb) << 8) |
                                   ((uint32_t)slot_of[tri(std::max(ca, cb), std::min(ca, cb))] << 16);
                pw.push_back(w);
                if (a != bb && ca == cb) pw.push_back(w);
            }
        }
    while (pw.size() % 64u) pw.push_back(0xFFFFFFFFu);
    while (pe.size() % 64u) pe.push_back(0xFFFFFFFFu);
    std::vector<uint8_t> lt((size_t)16 * NC * NV);
    for (uint32_t l = 0; l < 16u; ++l)
        for (uint32_t q = 0; q < (uint32_t)NC; ++q)
            for (uint32_t i = 0; i < NV; ++i) {
                const uint32_t j = l + 16u * q;
                const int32_t sl = slot_of[tri(std::max(i, j), std::min(i, j))];
                lt[(size_t)l * NC * NV + NV * q + i] = (uint8_t)(sl >= 0 ? (uint32_t)sl : zero);
            }
    std::vector<uint32_t>& w = out.words;
    w.assign(20, 0);
    auto put = [&](const void* src, size_t bytes) -> uint32_t {
        const uint32_t at = (uint32_t)w.size() * 4u;
        w.resize(w.size() + (bytes + 15u) / 16u * 4u, 0u);
        memcpy(reinterpret_cast<unsigned char*>(w.data()) + at, src, bytes);
        return at;
    };
    // (the tables are a multiple of 16 bytes each, so `put` places them back to back where GcTable says)
    bool placed = put(vcol, sizeof(vcol)) == TK::VCOL;
    placed = put(fidx, sizeof(fidx)) == TK::FIDX && placed;
    placed = put(rtag, sizeof(rtag)) == TK::RTAG && placed;
    placed = put(gbase, sizeof(gbase)) == TK::GBASE && placed;
    placed = put(gvar, sizeof(gvar)) == TK::GVAR && placed;
    placed = put(lt.data(), lt.size()) == TK::LT && placed;
    placed = put(pe.data(), pe.size() * 4u) == TK::PE && placed;
    (void)put(pw.data(), pw.size() * 4u);
    if (!placed) return false;
    w[18] = (uint32_t)w.size();  // what the f64 builds copy; behind it, for the f32 build:
    // the same products and right-hand-side entries by TARGET — slot by slot, column by column, each target's in list order — for
    // an assembly without LDS float atomics (ds_add_f32 costs 192 cycles an instruction on gfx950, ds_add_f64 14:
    // tools/probes/lds_atomic_f32_probe.hip). u16 each: entry a | entry b << 8; entry | row << 8.
    {
        std::vector<std::vector<uint16_t>> by_slot(nslots), by_col(NV);
        for (uint32_t x : pw)
            if (x != 0xFFFFFFFFu) by_slot[x >> 16].push_back((uint16_t)(x & 0xFFFFu));
        for (uint32_t x : pe)
            if (x != 0xFFFFFFFFu) by_col[x >> 16].push_back((uint16_t)(x & 0xFFFFu));
        std::vector<uint16_t> sptr(1, 0), spw, cptr(1, 0), cpe;
        for (auto& v : by_slot) {
            spw.insert(spw.end(), v.begin(), v.end());
            sptr.push_back((uint16_t)spw.size());
        }
        for (auto& v : by_col) {
            cpe.insert(cpe.end(), v.begin(), v.end());
            cptr.push_back((uint16_t)cpe.size());
        }
        w[9] = put(sptr.data(), sptr.size() * 2u);
        w[10] = put(spw.data(), spw.size() * 2u);
        w[11] = put(cptr.data(), cptr.size() * 2u);
        w[12] = put(cpe.data(), cpe.size() * 2u);
    }
    w[0] = 1u;
    w[1] = nvt;
    w[2] = net;
    w[3] = nfree;
    w[4] = (uint32_t)pw.size();
    w[5] = (uint32_t)pe.size();
    w[6] = nslots;
    w[7] = (ng + 3u) & ~3u;
    w[8] = zero;
    w[17] = (uint32_t)w.size();
    out.nslots = nslots;
    out.ng = (ng + 3u) & ~3u;
    out.words_f64 = w[18];
    return true;
}
bool build_gc_program(const uint16_t* var_info, const uint8_t* expr_tag, const uint16_t* expr_comp, const uint16_t* expr_idx16,
                             uint32_t nvt, uint32_t net, uint32_t max_free, GcHostProgram& out) {
    // the smallest build that holds the structure: its free variables decide, unless its variables (fixed ones included) or its
    // expressions need the next one's tables — the columns past the free variables are identity padding either way
    // (... and an over-constrained structure the instantiation with twice the rows)
    if (max_free <= 16u && build_gc_program_t<1, 1>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out)) return true;
    if (max_free <= 16u && build_gc_program_t<1, 2>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out)) return true;
    if (max_free <= 32u && build_gc_program_t<2, 2>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out)) return true;
    if (max_free <= 32u && build_gc_program_t<2, 4>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out)) return true;
    return build_gc_program_t<3, 3>(var_info, expr_tag, expr_comp, expr_idx16, nvt, net, out);
}

// The program of the grouped kernel's SPARSE build (fx_grouped_s.hip): batches of one structure whose single component is too
// wide for a register-resident factor (49 free variables and more) but whose Cholesky factor is small — the reference's bench
// sketch of 16 hinged triangles has 66 variables and a factor of 291 entries. Everything the kernel does is a walk over tables:
// the row lists and product lists of fx_grouped_c.hip, and the factorisation as a level schedule — a minimum-degree order of the
// columns, the factor's pattern by columns (column k: its diagonal slot, then its rows ascending), the levels of its
// elimination tree (the columns of a level are independent), per level the update triples L(i, j) -= L(i, k) L(j, k) and the
// (slot, column, row) entries of the triangular solves. Words: [0] version [1] variables [2] expressions [3] free variables
// [4] products (padded to 64) [5] right-hand-side entries (padded to 64) [6] factor slots (even) [7] compact Jacobian entries
// (even) [8] levels [9] update triples [10] below-diagonal entries [11 ...] byte offsets of the tables, in the order of the
// `put` calls below [31] words in all.
bool build_gs_program(const uint16_t* var_info, const uint8_t* expr_tag, const uint16_t* expr_comp, const uint16_t* expr_idx16,
                             uint32_t nvt, uint32_t net, GsHostProgram& out) {
    out = GsHostProgram();
    if (nvt == 0 || nvt > 255u || net == 0 || net > 255u) return false;  // (a byte per variable / row / column id in the tables)
    std::vector<int16_t> vcol(nvt, -1);
    std::vector<uint16_t> fidx;
    for (uint32_t i = 0; i < nvt; ++i) {
        if ((var_info[i] & fx::VAR_COMP_MASK) != 0) return false;
        if (!(var_info[i] & fx::VAR_FIXED_BIT)) {
            vcol[i] = (int16_t)fidx.size();
            fidx.push_back((uint16_t)i);
        }
    }
    const uint32_t n = (uint32_t)fidx.size();
    if (n <= 32u || n > 255u) return false;
    std::vector<uint8_t> rtag(net), gvar((size_t)net * 8, 0);
    std::vector<uint16_t> gbase(net);
    std::vector<int> gcol((size_t)net * 8, -1);
    uint32_t ng = 0;
    std::vector<uint8_t> adj((size_t)n * n, 0);  // pattern of Jt J
    for (uint32_t r = 0; r < net; ++r) {
        if (expr_comp[r] != 0) return false;
        const int tag = (int)(expr_tag[r] & 0x7F);
        if (tag >= FX_TAG_POSE_X) return false;
        uint32_t vars8[8];
        const int k = fx::expand_vars(tag, expr_idx16 + 4 * (size_t)r, vars8);
        rtag[r] = (uint8_t)tag;
        gbase[r] = (uint16_t)ng;
        for (int e = 0; e < 8; ++e) {
            if (vars8[e] >= nvt) return false;
            gvar[(size_t)r * 8 + e] = (uint8_t)vars8[e];
            gcol[(size_t)r * 8 + e] = e < k ? (int)vcol[vars8[e]] : -1;
        }
        // (a distance row keeps TWO of its four entries: the other two are their exact negatives — expressions.rs:291-317 —,
        // and the lists below carry the sign)
        ng += tag == FX_TAG_PPD ? 2u : (uint32_t)k;
        for (int a = 0; a < k; ++a)
            for (int bb = 0; bb < k; ++bb)
                if (gcol[(size_t)r * 8 + a] >= 0 && gcol[(size_t)r * 8 + bb] >= 0)
                    adj[(size_t)gcol[(size_t)r * 8 + a] * n + (size_t)gcol[(size_t)r * 8 + bb]] = 1;
    }
    if (ng > 1023u) return false;
    // ---- minimum-degree order on the graph of Jt J (ties: the lower column), eliminating on a copy
    std::vector<uint32_t> order, pos(n, 0);
    {
        std::vector<uint8_t> g = adj;
        std::vector<uint8_t> gone(n, 0);
        std::vector<uint32_t> deg(n, 0), nbr;
        auto degree = [&](uint32_t c) {
            uint32_t dg = 0;
            for (uint32_t e = 0; e < n; ++e) dg += (!gone[e] && e != c && g[(size_t)c * n + e]) ? 1u : 0u;
            return dg;
        };
        for (uint32_t c = 0; c < n; ++c) deg[c] = degree(c);
        for (uint32_t step = 0; step < n; ++step) {
            uint32_t best = n, bdeg = 0xFFFFFFFFu;
            for (uint32_t c = 0; c < n; ++c)
                if (!gone[c] && deg[c] < bdeg) {
                    bdeg = deg[c];
                    best = c;
                }
            gone[best] = 1;
            pos[best] = step;
            order.push_back(best);
            nbr.clear();
            for (uint32_t a = 0; a < n; ++a)
                if (!gone[a] && g[(size_t)best * n + a]) nbr.push_back(a);
            for (uint32_t a : nbr)
                for (uint32_t bb : nbr) g[(size_t)a * n + bb] = 1;
            for (uint32_t a : nbr) deg[a] = degree(a);  // (only the eliminated column's neighbours change their degree)
        }
    }
    // ---- the factor's pattern in elimination order: lp[i][k] (positions), with fill
    std::vector<uint8_t> lp((size_t)n * n, 0);
    for (uint32_t a = 0; a < n; ++a)
        for (uint32_t bb = 0; bb < n; ++bb)
            if (a == bb || adj[(size_t)a * n + bb]) {
                const uint32_t pi = std::max(pos[a], pos[bb]), pk = std::min(pos[a], pos[bb]);
                lp[(size_t)pi * n + pk] = 1;
            }
    for (uint32_t k = 0; k < n; ++k)
        for (uint32_t i = k + 1; i < n; ++i)
            if (lp[(size_t)i * n + k])
                for (uint32_t j = k + 1; j <= i; ++j)
                    if (lp[(size_t)j * n + k]) lp[(size_t)i * n + j] = 1;
    // slots: column position k holds its diagonal, then its rows (positions ascending)
    std::vector<uint16_t> cbase(n + 1, 0);
    std::vector<int32_t> slot((size_t)n * n, -1);
    std::vector<uint8_t> rowof;  // free COLUMN id of a slot's row
    uint32_t nl = 0;
    for (uint32_t k = 0; k < n; ++k) {
        cbase[k] = (uint16_t)nl;
        for (uint32_t i = k; i < n; ++i)
            if (lp[(size_t)i * n + k]) {
                slot[(size_t)i * n + k] = (int32_t)nl++;
                rowof.push_back((uint8_t)order[i]);
            }
        if (nl > 1023u) return false;
    }
    cbase[n] = (uint16_t)nl;
    // levels of the elimination tree: a column waits for every column that updates it
    std::vector<uint32_t> level(n, 0);
    uint32_t nlev = 0;
    for (uint32_t k = 0; k < n; ++k) {
        for (uint32_t j = 0; j < k; ++j)
            if (lp[(size_t)k * n + j]) level[k] = std::max(level[k], level[j] + 1u);
        nlev = std::max(nlev, level[k] + 1u);
    }
    std::vector<uint8_t> lcol;      // column POSITIONS in level order
    std::vector<uint16_t> lptr(1, 0);
    std::vector<uint32_t> uptr(1, 0), eptr(1, 0), upd, ent;
    for (uint32_t lv = 0; lv < nlev; ++lv) {
        for (uint32_t k = 0; k < n; ++k) {
            if (level[k] != lv) continue;
            lcol.push_back((uint8_t)k);
            for (uint32_t i = k + 1; i < n; ++i) {
                if (!lp[(size_t)i * n + k]) continue;
                // (slot | column id of k << 10 | column id of the row << 18): the triangular solves' entries
                ent.push_back((uint32_t)slot[(size_t)i * n + k] | (order[k] << 10) | (order[i] << 18));
                for (uint32_t j = k + 1; j <= i; ++j)
                    if (lp[(size_t)j * n + k])
                        upd.push_back((uint32_t)slot[(size_t)i * n + j] | ((uint32_t)slot[(size_t)i * n + k] << 10) | ((uint32_t)slot[(size_t)j * n + k] << 20));
            }
        }
        lptr.push_back((uint16_t)lcol.size());
        uptr.push_back((uint32_t)upd.size());
        eptr.push_back((uint32_t)ent.size());
    }
    // per column position: its column id; per column id: the slot of its diagonal
    std::vector<uint8_t> colid(n);
    std::vector<uint16_t> dslot(n);
    for (uint32_t k = 0; k < n; ++k) {
        colid[k] = (uint8_t)order[k];
        dslot[order[k]] = cbase[k];
    }
    // products of Jt J into the factor's slots, right-hand side entries (the order of fx_grouped_c.hip's lists)
    std::vector<uint32_t> pw, pe;
    for (uint32_t r = 0; r < net; ++r) {
        const bool ppd = rtag[r] == FX_TAG_PPD;
        auto gent = [&](int e) -> uint32_t { return gbase[r] + (uint32_t)(ppd && e >= 2 ? e - 2 : e); };  // where entry e's value (or its negative) is kept
        auto gneg = [&](int e) -> uint32_t { return ppd && e >= 2 ? 1u : 0u; };
        for (int a = 0; a < 8; ++a) {
            const int ca = gcol[(size_t)r * 8 + a];
            if (ca < 0) continue;
            pe.push_back(gent(a) | (r << 10) | ((uint32_t)ca << 20) | (gneg(a) << 31));
            for (int bb = a; bb < 8; ++bb) {
                const int cb = gcol[(size_t)r * 8 + bb];
                if (cb < 0) continue;
                const uint32_t pi = std::max(pos[(uint32_t)ca], pos[(uint32_t)cb]), pk = std::min(pos[(uint32_t)ca], pos[(uint32_t)cb]);
                const uint32_t w = gent(a) | (gent(bb) << 10) | ((uint32_t)slot[(size_t)pi * n + pk] << 20) | ((gneg(a) ^ gneg(bb)) << 31);
                pw.push_back(w);
                if (a != bb && ca == cb) pw.push_back(w);
            }
        }
    }
    while (pw.size() % 64u) pw.push_back(0xFFFFFFFFu);
    while (pe.size() % 64u) pe.push_back(0xFFFFFFFFu);
    std::vector<uint32_t>& w = out.words;
    w.assign(32, 0);
    auto put = [&](const void* src, size_t bytes) -> uint32_t {
        const uint32_t at = (uint32_t)w.size() * 4u;
        w.resize(w.size() + (bytes + 15u) / 16u * 4u, 0u);
        if (bytes) memcpy(reinterpret_cast<unsigned char*>(w.data()) + at, src, bytes);
        return at;
    };
    w[11] = put(vcol.data(), vcol.size() * 2);
    w[12] = put(fidx.data(), fidx.size() * 2);
    w[13] = put(rtag.data(), rtag.size());
    w[14] = put(gbase.data(), gbase.size() * 2);
    w[15] = put(gvar.data(), gvar.size());
    w[16] = put(dslot.data(), dslot.size() * 2);
    w[17] = put(cbase.data(), cbase.size() * 2);
    w[18] = put(rowof.data(), rowof.size());
    w[19] = put(lcol.data(), lcol.size());
    w[20] = put(lptr.data(), lptr.size() * 2);
    w[21] = put(uptr.data(), uptr.size() * 4);
    w[22] = put(eptr.data(), eptr.size() * 4);
    w[23] = put(upd.data(), upd.size() * 4);
    w[24] = put(ent.data(), ent.size() * 4);
    w[25] = put(pw.data(), pw.size() * 4);
    w[26] = put(pe.data(), pe.size() * 4);
    w[27] = put(colid.data(), colid.size());
    w[0] = 1u;
    w[1] = nvt;
    w[2] = net;
    w[3] = n;
    w[4] = (uint32_t)pw.size();
    w[5] = (uint32_t)pe.size();
    w[6] = (nl + 1u) & ~1u;
    w[7] = (ng + 1u) & ~1u;
    w[8] = nlev;
    w[9] = (uint32_t)upd.size();
    w[10] = (uint32_t)ent.size();
    w[31] = (uint32_t)w.size();
    out.nl = (nl + 1u) & ~1u;
    out.ng = (ng + 1u) & ~1u;
    out.nvt = nvt;
    out.net = net;
    out.nfree = n;
    return true;
}

// The same analysis compiled into a table-driven program for the grouped FX_STEP_QR build (fx_grouped.hip: four Systems per
// wavefront, one per row of 16 lanes; batches of ONE structure, so one program serves every System). The permuted augmented
// matrix [J | -r; sqrt(lambda) I | 0] is stored by its symbolic patterns — per column position j the rows of R(:, j) above the
// diagonal and of the Householder vector H(:, j) from it down — plus the dense right-hand side and one slot of zero for the
// padding. Every access of the factorisation is then an offset from a table: per Householder step k its active columns
// (those with k in R's pattern, and the right-hand side), one lane each, and per (active column, vector entry) one word
// holding both offsets of the multiply-add. The arithmetic and its order are the one-wavefront QR kernel's (fx_kernels.hip).
// Words: [0] n [1] m [2] nx (doubles per System, even) [3] zero slot [4] scat (u16 [m][8], 0xFFFF = dropped) [5] rhs_off (u16 [m])
// [6] damp_off (u16 [n]) [7] cpos (u16 [n]: free column -> position) [8] steps ([n][3]: diag | len << 16, entries' first word,
// active columns) [9] bptr (u16 [n + 1]) [10] bent (row << 16 | offset of R(row, i)) [11] words in all [12] first right-hand
// side entry [13] longest vector (entries below the diagonal, padded to fours).
// `wide`: the program of the one-wavefront QR build of the wide kernel (fx_wide.hip: components of up to 128 columns and
// 256 rows, Householder vectors of any length) — the same tables with offsets in ELEMENTS (the matrix may pass 64 KB).
bool build_qrg_program(const uint8_t* expr_tag, const uint16_t* expr_idx16, const uint32_t* rows, uint32_t m, const uint32_t* free_,
                       uint32_t n, uint32_t nvt, QrgHostProgram& out, bool wide) {
    out = QrgHostProgram();
    out.n = n;
    out.m = m;
    if (n == 0 || n > (wide ? 128u : 32u) || m == 0 || m > (wide ? 256u : 64u)) return false;
    std::vector<int32_t> colof(nvt, -1);
    for (uint32_t c = 0; c < n; ++c) colof[free_[c]] = (int32_t)c;
    std::vector<std::vector<int>> cols(n);
    std::vector<int32_t> gcol((size_t)m * 8, -1);
    for (uint32_t r = 0; r < m; ++r) {
        uint32_t vars8[8];
        const int k = fx::expand_vars<true>((int)(expr_tag[rows[r]] & 0x7F), expr_idx16 + 4 * (size_t)rows[r], vars8);
        for (int q = 0; q < k; ++q) {
            const int32_t c = vars8[q] < nvt ? colof[vars8[q]] : -1;
            gcol[(size_t)r * 8 + q] = c;
            if (c >= 0 && (cols[c].empty() || cols[c].back() != (int)r)) cols[c].push_back((int)r);
        }
    }
    fx::qr::Csc a;
    a.nrows = (int)(m + n);
    a.ncols = (int)n;
    a.ptr.assign(1, 0);
    for (uint32_t c = 0; c < n; ++c) {
        a.idx.insert(a.idx.end(), cols[c].begin(), cols[c].end());
        a.idx.push_back((int)(m + c));
        a.ptr.push_back((int)a.idx.size());
    }
    fx::qr::Symbolic sy;
    if (!fx::qr::analyze(a, true, sy)) return false;
    const uint32_t Mq = m + n;
    std::vector<uint32_t> cpos(n, 0);
    for (uint32_t j = 0; j < n; ++j) cpos[(uint32_t)sy.col_perm[j]] = j;
    // storage: column position j holds the rows of R(:, j) above the diagonal, then those of H(:, j) (j first)
    std::vector<std::vector<int>> prow(n);
    std::vector<uint32_t> cbase(n + 1, 0);
    for (uint32_t j = 0; j < n; ++j) {
        for (int p = sy.rptr[j]; p < sy.rptr[j + 1] - 1; ++p) prow[j].push_back(sy.rrows[p]);
        for (int p = sy.hptr[j]; p < sy.hptr[j + 1]; ++p) prow[j].push_back(sy.hrows[p]);
        if (!std::is_sorted(prow[j].begin(), prow[j].end()) || std::adjacent_find(prow[j].begin(), prow[j].end()) != prow[j].end()) return false;
        cbase[j + 1] = cbase[j] + (uint32_t)prow[j].size();
    }
    const uint32_t rhsbase = cbase[n];
    uint32_t nx = rhsbase + Mq + 1u;
    const uint32_t zero = nx - 1u;
    nx = (nx + 1u) & ~1u;
    const uint32_t osc = wide ? 1u : 8u;  // offsets in elements / in bytes
    if (osc * nx > 0xFFF0u) return false;
    bool bad = false;
    auto at = [&](int r, uint32_t j) -> uint32_t {  // offset of entry (permuted row r, column position j; j == n: right-hand side)
        if (j == n) return rhsbase + (uint32_t)r;
        auto it = std::lower_bound(prow[j].begin(), prow[j].end(), r);
        if (it == prow[j].end() || *it != r) {
            bad = true;
            return zero;
        }
        return cbase[j] + (uint32_t)(it - prow[j].begin());
    };
    auto atb = [&](int r, uint32_t j) -> uint32_t { return osc * at(r, j); };  // ... as the kernel takes them
    std::vector<uint16_t> scat((size_t)m * 8, 0xFFFFu), rhs_off(m), damp(n), cpos16(n), bptr(n + 1, 0);
    for (uint32_t r = 0; r < m; ++r) {
        for (int q = 0; q < 8; ++q)
            if (gcol[(size_t)r * 8 + q] >= 0) scat[(size_t)r * 8 + q] = (uint16_t)atb(sy.row_perm[r], cpos[(uint32_t)gcol[(size_t)r * 8 + q]]);
        rhs_off[r] = (uint16_t)atb(sy.row_perm[r], n);
    }
    for (uint32_t c = 0; c < n; ++c) {
        damp[c] = (uint16_t)atb(sy.row_perm[m + c], cpos[c]);
        cpos16[c] = (uint16_t)cpos[c];
    }
    std::vector<uint32_t> steps(3 * (size_t)n, 0), ent, bent;
    uint32_t max_len = 0;
    for (uint32_t k = 0; k < n; ++k) {
        const int hb = sy.hptr[k], he = sy.hptr[k + 1];
        if (he <= hb || sy.hrows[hb] != (int)k) return false;
        const uint32_t below = (uint32_t)(he - hb - 1), len = (below + 3u) & ~3u;
        max_len = std::max(max_len, len);
        std::vector<uint32_t> active;  // column positions the vector is applied to, ascending, then the right-hand side
        for (uint32_t j = k + 1; j < n; ++j)
            if (std::binary_search(sy.rrows.begin() + sy.rptr[j], sy.rrows.begin() + sy.rptr[j + 1] - 1, (int)k)) active.push_back(j);
        active.push_back(n);
        const uint32_t na = (uint32_t)active.size();
        steps[3 * k] = atb((int)k, k) | (len << 16);
        steps[3 * k + 1] = (uint32_t)ent.size();
        steps[3 * k + 2] = na;
        const size_t e0 = ent.size();
        ent.resize(e0 + (size_t)(len + 1u) * na, (osc * zero) | ((osc * zero) << 16));
        for (uint32_t i = 0; i < na; ++i) {
            ent[e0 + i] = atb((int)k, active[i]);
            for (uint32_t u = 0; u < below; ++u) {
                const int r = sy.hrows[hb + 1 + (int)u];
                ent[e0 + (size_t)(1u + u) * na + i] = atb(r, active[i]) | (atb(r, k) << 16);
            }
        }
    }
    for (uint32_t i = 0; i < n; ++i) {
        bptr[i] = (uint16_t)bent.size();
        for (int p = sy.rptr[i]; p < sy.rptr[i + 1] - 1; ++p) bent.push_back(((osc * (uint32_t)sy.rrows[p]) << 16) | atb(sy.rrows[p], i));
    }
    bptr[n] = (uint16_t)bent.size();
    if (bad || (!wide && max_len > 32u) || max_len > 0xFFFFu) return false;
    std::vector<uint32_t>& w = out.words;
    w.assign(16, 0);
    auto put16 = [&](const std::vector<uint16_t>& v) -> uint32_t {
        const uint32_t o = (uint32_t)w.size();
        w.resize(o + (v.size() + 1) / 2, 0);
        memcpy(w.data() + o, v.data(), v.size() * 2);
        return o;
    };
    auto put32 = [&](const std::vector<uint32_t>& v) -> uint32_t {
        const uint32_t o = (uint32_t)w.size();
        w.insert(w.end(), v.begin(), v.end());
        return o;
    };
    w[0] = n; w[1] = m; w[2] = nx; w[3] = zero;
    w[4] = put16(scat); w[5] = put16(rhs_off); w[6] = put16(damp); w[7] = put16(cpos16);
    const uint32_t o_ent_rel = 0;
    (void)o_ent_rel;
    w[8] = put32(steps);
    w[9] = put16(bptr);
    w[10] = put32(bent);
    {  // the Jacobian rows are kept compact: row r's entries start at gbase[r], one per variable of its expression kind
        std::vector<uint16_t> gbase(m);
        uint32_t ng = 0;
        for (uint32_t r = 0; r < m; ++r) {
            gbase[r] = (uint16_t)ng;
            ng += (uint32_t)fx::tag_nvars<true>((int)(expr_tag[rows[r]] & 0x7F));
        }
        w[15] = put16(gbase);
        out.ng = (ng + 3u) & ~3u;
    }
    w.resize((w.size() + 3u) & ~size_t(3), 0);
    w[14] = (uint32_t)w.size();  // the small tables end here (the kernel keeps them in LDS); the per-entry words stay in global memory
    const uint32_t o_ent = put32(ent);
    for (uint32_t k = 0; k < n; ++k) w[w[8] + 3 * k + 1] += o_ent;  // entries' first word, from the start of the program
    w.resize((w.size() + 3u) & ~size_t(3), 0);
    w[11] = (uint32_t)w.size();
    w[12] = osc * rhsbase;
    w[13] = max_len;
    out.nx = nx;
    out.ok = true;
    return true;
}

}  // namespace fxh
