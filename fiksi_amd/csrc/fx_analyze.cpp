// Host analysis of a batch (include/fiksi_amd.h: fx_batch): validation — the data invariants the reference enforces by
// construction (handles of one System, indices < variables.len(): fiksi/src/lib.rs:363-445) — and everything the device needs
// besides the raw arrays: compact per-variable / per-expression words, the limits that size kernels and LDS, structure
// classes, the CSR pattern of the Jacobian (sparse_col_mat.rs:690-737 after the row / column transposition) and the
// tag-sorted order of the row-parallel kernels. Structure only: no value is computed here.
#include "fx_host.h"

namespace fxh {

// distinct free columns of a row (ascending) and the slot of each gradient entry; returns the count
inline int row_columns(const uint32_t vars8[8], int k, const int32_t* free_rank, int32_t cols[8], uint32_t* slots_out) {
    int ncols = 0;
    for (int q = 0; q < k; ++q) {
        int32_t col = free_rank[vars8[q]];
        if (col < 0) continue;
        bool seen = false;
        for (int t = 0; t < ncols; ++t) seen = seen || cols[t] == col;
        if (!seen) cols[ncols++] = col;
    }
    std::sort(cols, cols + ncols);
    if (slots_out) {
        uint32_t slots = 0;
        for (int q = 0; q < 8; ++q) {
            uint32_t sl = 0xFu;
            if (q < k) {
                int32_t col = free_rank[vars8[q]];
                if (col >= 0) sl = (uint32_t)(std::find(cols, cols + ncols, col) - cols);
            }
            slots |= sl << (4 * q);
        }
        *slots_out = slots;
    }
    return ncols;
}

// CSR pattern from the compact per-variable / per-expression arrays (host copies of what the device
// holds): columns = system-local rank among the free variables, ascending inside a row, duplicates
// merged (sparse_col_mat.rs:690-737 after the row/column transposition).
void build_csr(uint32_t n, const uint32_t* var_off, const uint32_t* expr_off, const uint16_t* var_info,
               const uint8_t* expr_tag, const uint16_t* expr_idx16, CsrPlan& out) {
    const uint32_t ne = n ? expr_off[n] : 0;
    out.jrow_ptr.assign((size_t)ne + 1, 0);
    out.jslot.assign(ne, 0xFFFFFFFFu);
    std::vector<uint32_t> local_cols[MAX_RANGES];
    uint32_t range_lo[MAX_RANGES + 1] = {0};
    uint32_t nr = 1;
    parallel_ranges(n, ne, [&](uint32_t t, uint32_t s_lo, uint32_t s_hi) {
        range_lo[t] = s_lo;
        std::vector<int32_t> free_rank;
        auto& lc = local_cols[t];
        lc.reserve((size_t)(expr_off[s_hi] - expr_off[s_lo]) * 5);
        for (uint32_t s = s_lo; s < s_hi; ++s) {
            const uint32_t v0 = var_off[s], nvt = var_off[s + 1] - v0;
            const uint32_t e0 = expr_off[s], net = expr_off[s + 1] - e0;
            free_rank.assign(nvt, -1);
            int32_t rank = 0;
            for (uint32_t i = 0; i < nvt; ++i) {
                uint16_t info = var_info[v0 + i];
                if ((info & fx::VAR_COMP_MASK) != fx::VAR_COMP_NONE && !(info & fx::VAR_FIXED_BIT)) free_rank[i] = rank++;
            }
            for (uint32_t i = 0; i < net; ++i) {
                const uint32_t e = e0 + i;
                uint32_t vars8[8];
                int k = fx::expand_vars<true>((int)(expr_tag[e] & 0x7F), expr_idx16 + 4 * (size_t)e, vars8);
                int32_t cols[8];
                uint32_t slots;
                int ncols = row_columns(vars8, k, free_rank.data(), cols, &slots);
                out.jslot[e] = slots;
                for (int q = 0; q < ncols; ++q) lc.push_back((uint32_t)cols[q]);
                out.jrow_ptr[(size_t)e + 1] = (uint32_t)ncols;  // count; prefix-summed below
            }
        }
    }, &nr);
    for (uint32_t e = 0; e < ne; ++e) out.jrow_ptr[e + 1] += out.jrow_ptr[e];
    out.jcol.resize(out.jrow_ptr[ne]);
    for (uint32_t t = 0; t < nr; ++t)
        if (!local_cols[t].empty())
            std::copy(local_cols[t].begin(), local_cols[t].end(), out.jcol.begin() + out.jrow_ptr[expr_off[range_lo[t]]]);
}

void build_eval_plan(uint32_t n, const uint32_t* var_off, const uint32_t* expr_off, const uint16_t* var_info,
                     const uint8_t* expr_tagx, const uint16_t* expr_idx16, EvalPlan& out) {
    const uint32_t ne = n ? expr_off[n] : 0;
    out.expr_var0.assign(ne, 0);
    out.row_perm.assign(ne, 0);
    out.row_sysoff.assign(ne, 0);
    std::vector<uint32_t> expr_sys(ne, 0);
    std::vector<uint8_t> row_simple(ne, 0);
    parallel_ranges(n, ne, [&](uint32_t, uint32_t s_lo, uint32_t s_hi) {
        for (uint32_t s = s_lo; s < s_hi; ++s) {
            const uint32_t v0 = var_off[s];
            for (uint32_t e = expr_off[s]; e < expr_off[s + 1]; ++e) {
                out.expr_var0[e] = v0;
                expr_sys[e] = s;
                uint32_t vars8[8];
                const int k = fx::expand_vars<true>((int)(expr_tagx[e] & 0x7F), expr_idx16 + 4 * (size_t)e, vars8);
                bool all_free = true, distinct = true;  // "simple": every variable free, none read twice
                for (int q = 0; q < k; ++q) {
                    const uint16_t info = var_info[v0 + vars8[q]];
                    all_free = all_free && (info & fx::VAR_COMP_MASK) != fx::VAR_COMP_NONE && !(info & fx::VAR_FIXED_BIT);
                    for (int u = q + 1; u < k; ++u) distinct = distinct && vars8[q] != vars8[u];
                }
                row_simple[e] = (all_free && distinct) ? 1 : 0;
            }
        }
    });
    // tag-sorted thread -> row assignment inside every block of 256 rows (stable counting sort)
    const uint32_t nblk = (ne + 255u) / 256u;
    out.blk_info.assign(nblk, fx::BlockInfo{});
    parallel_ranges(nblk, ne, [&](uint32_t, uint32_t b_lo, uint32_t b_hi) {
        for (uint32_t blk = b_lo; blk < b_hi; ++blk) {
            const uint32_t r0 = blk * 256u;
            uint32_t nrw = std::min<uint32_t>(256, ne - r0), t = 0;
            uint8_t sorted[256];
            for (int tag = 0; tag < FX_NTAGS_POSE; ++tag)
                for (uint32_t i = 0; i < nrw; ++i)
                    if ((expr_tagx[r0 + i] & 0x7F) == tag) sorted[t++] = (uint8_t)i;
            // The sorted order, rotated by a wavefront per block: wavefront w of every workgroup lands on the same SIMD of its
            // CU, so without the rotation the expensive kinds (the angle rows: two atan2, three times a distance row's
            // instructions) of all eight resident blocks pile up on one SIMD while the other three idle.
            const uint32_t shift = nrw == 256u ? 64u * (blk & 3u) : 0u;
            for (uint32_t i = 0; i < nrw; ++i) out.row_perm[r0 + i] = sorted[(i + shift) % nrw];
            fx::BlockInfo bi{};
            bi.sys0 = expr_sys[r0];
            bool simple = true;
            for (uint32_t i = 0; i < nrw; ++i) {
                simple = simple && row_simple[r0 + i];
                // one byte per row: fits while the block's rows belong to at most 256 consecutive Systems. Systems
                // without expressions take an index without taking a row, so a block can span more — such a block
                // is not "simple": its rows then read their System's first variable from expr_var0 instead
                const uint32_t off = expr_sys[r0 + i] - bi.sys0;
                simple = simple && off <= 255u;
                out.row_sysoff[r0 + i] = (uint8_t)(off & 0xFFu);
            }
            bi.flags = simple ? 1u : 0u;  // jbase / jcount are filled when the CSR structure is built
            out.blk_info[blk] = bi;
        }
    });
}

// Checks the batch and builds the plan. Mirrors the data invariants the reference enforces by
// construction (handles of the same System, indices < variables.len()).
int analyze(const fx_batch* b, HostPlan* plan) {
    if (!b) return fail(FX_ERR_INVALID, "batch is NULL");
    const uint32_t n = b->n_systems;
    if (n > 0 && (!b->var_off || !b->expr_off)) return fail(FX_ERR_INVALID, "var_off/expr_off is NULL");
    if (n == 0) {
        if (plan) *plan = HostPlan();
        return FX_OK;
    }
    if (b->var_off[0] != 0 || b->expr_off[0] != 0) return fail(FX_ERR_INVALID, "offset arrays must start at 0");
    for (uint32_t s = 0; s < n; ++s) {
        if (b->var_off[s + 1] < b->var_off[s] || b->expr_off[s + 1] < b->expr_off[s])
            return fail(FX_ERR_INVALID, "offsets of system %u decrease", s);
    }
    const uint32_t nv = b->var_off[n], ne = b->expr_off[n];
    if (nv > 0 && (!b->vars || !b->var_fixed)) return fail(FX_ERR_INVALID, "vars/var_fixed is NULL");
    if (ne > 0 && (!b->expr_tag || !b->expr_idx || !b->expr_param)) return fail(FX_ERR_INVALID, "expr_* is NULL");

    // The caller's word that every System has the first one's structure (fx_ctx_set_batch_hints): the first System alone is analysed,
    // its outcome stands for the others — regular offsets are checked here, everything else by verify_one_structure, which the caller
    // of this function runs beside the device's work and before any result reaches the user.
    if (plan && g_hint_one_structure && n >= 2) {
        const uint32_t nv0 = b->var_off[1], ne0 = b->expr_off[1];
        bool regular = nv0 <= FX_MAX_SYSTEM_VARS && (uint64_t)nv0 * n == nv && (uint64_t)ne0 * n == ne;
        for (uint32_t s = 1; regular && s <= n; ++s) regular = b->var_off[s] == s * nv0 && b->expr_off[s] == s * ne0;
        if (regular) {
            fx_batch first = *b;
            first.n_systems = 1;
            HostPlan one;
            g_hint_one_structure = false;
            const int rc1 = analyze(&first, &one);
            g_hint_one_structure = true;
            if (rc1) return rc1;
            if (one.n_large == 0) {  // (Systems beyond one wavefront keep a host copy and their own analysis: the ordinary way)
                HostPlan& p = *plan;
                p = HostPlan();
                p.n_systems = n;
                p.n_vars = nv;
                p.n_exprs = ne;
                p.nnz = one.nnz * n;
                p.max_free = one.max_free; p.max_rows = one.max_rows; p.max_vars = one.max_vars; p.max_exprs = one.max_exprs;
                p.max_vars_all = one.max_vars_all; p.max_exprs_all = one.max_exprs_all; p.max_pairs = one.max_pairs; p.max_ents = one.max_ents;
                p.max_pairs_tri = one.max_pairs_tri;
                p.uniform = 1u;
                p.hinted = true;
                p.sys_ncomp.assign(n, one.sys_ncomp[0]);
                p.sys_large.assign(n, 0);
                p.same_as_prev.assign(n, 1);
                p.same_as_prev[0] = 0;
                // the structure arrays: System 0's only (an upload of a batch of one structure sends one period and the device fills
                // in the rest; upload_planned reads them at offset 0 for such a batch)
                p.var_info = one.var_info;
                p.expr_comp = one.expr_comp;
                p.expr_idx16 = one.expr_idx16;
                p.expr_tagx = one.expr_tagx;
                return FX_OK;
            }
        }
    }
    HostPlan local;
    HostPlan& p = plan ? *plan : local;
    p = HostPlan();
    p.n_systems = n;
    p.n_vars = nv;
    p.n_exprs = ne;
    p.sys_ncomp.assign(n, 0);
    p.sys_large.assign(n, 0);
    p.var_info.resize(nv);
    p.expr_comp.resize(ne);
    p.expr_idx16.resize(4 * (size_t)ne);
    p.expr_tagx.resize(ne);
    p.same_as_prev.assign(n, 0);

    struct Partial {
        uint64_t nnz = 0;
        uint32_t max_free = 0, max_rows = 0, max_vars = 0, max_exprs = 0, max_vars_all = 0, max_exprs_all = 0, n_large = 0;
        uint32_t w_max_free = 0, w_max_rows = 0, w_max_vars = 0;
        uint32_t max_pairs = 0, max_ents = 0, max_pairs_large = 0, max_ents_large = 0, max_pairs_tri = 0;
        int err = FX_OK;
        uint32_t err_system = 0;
        uint32_t first = 0;      // first System of the range
        bool chain_same = true;  // every System of the range came out as the one before it (sizes and analysed arrays)
        char msg[192] = {0};
    } part[MAX_RANGES];
    uint32_t nr = 1;
    const int n_tags = g_allow_pose ? FX_NTAGS_POSE : FX_NTAGS;  // read here: the ranges below may run on other threads
    // One sketch, many parameter sets (the last System has the first one's sizes and kinds): nearly every System will take
    // the analysis of the one before it, a few nanoseconds per item — threads then pay from some millions of items on
    uint64_t min_items = 200000;
    if (n >= 2 && b->var_off[1] == nv / n && b->expr_off[1] == ne / n && b->var_off[n - 1] == (uint64_t)(n - 1) * b->var_off[1] &&
        b->expr_off[n - 1] == (uint64_t)(n - 1) * b->expr_off[1] && nv == (uint64_t)n * b->var_off[1] && ne == (uint64_t)n * b->expr_off[1] &&
        memcmp(b->expr_tag, b->expr_tag + b->expr_off[n - 1], b->expr_off[1] * sizeof(*b->expr_tag)) == 0)
        min_items = 4000000;
    parallel_ranges(n, (uint64_t)ne + nv, [&](uint32_t t, uint32_t s_lo, uint32_t s_hi) {
        Partial& pt = part[t];
        auto bad = [&](int code, uint32_t s, const char* fmt, unsigned a0 = 0, unsigned a1 = 0, unsigned a2 = 0, unsigned a3 = 0) {
            pt.err = code;
            pt.err_system = s;
            snprintf(pt.msg, sizeof(pt.msg), fmt, a0, a1, a2, a3);
        };
        std::vector<int32_t> free_rank;  // per variable of the current system: system-wide free rank
        std::vector<uint32_t> comp_free, comp_rows, comp_pairs, comp_ents, comp_tri;
        pt.first = s_lo;
        // One sketch with many parameter sets is the common batch: a System whose raw structure arrays are those of the
        // System before it takes that System's analysis (two memcmp / memcpy passes instead of the walk below; every
        // statistic of the range is a maximum — unchanged — or a sum of per-System terms kept here).
        uint64_t prev_nnz = 0;
        uint32_t prev_large = 0;
        auto outputs_equal = [&](uint32_t s, uint32_t q, uint32_t nvt, uint32_t net) {
            const uint32_t v0 = b->var_off[s], e0 = b->expr_off[s], pv0 = b->var_off[q], pe0 = b->expr_off[q];
            return memcmp(&p.var_info[v0], &p.var_info[pv0], nvt * sizeof(uint16_t)) == 0 && memcmp(&p.expr_tagx[e0], &p.expr_tagx[pe0], net) == 0 &&
                   memcmp(&p.expr_comp[e0], &p.expr_comp[pe0], net * sizeof(uint16_t)) == 0 &&
                   memcmp(&p.expr_idx16[4 * (size_t)e0], &p.expr_idx16[4 * (size_t)pe0], 4 * (size_t)net * sizeof(uint16_t)) == 0;
        };
        for (uint32_t s = s_lo; s < s_hi && pt.err == FX_OK; ++s) {
            const uint32_t v0 = b->var_off[s], nvt = b->var_off[s + 1] - v0;
            const uint32_t e0 = b->expr_off[s], net = b->expr_off[s + 1] - e0;
            if (s > s_lo) {
                const uint32_t pv0 = b->var_off[s - 1], pe0 = b->expr_off[s - 1];
                if (nvt == v0 - pv0 && net == e0 - pe0 && memcmp(b->var_fixed + v0, b->var_fixed + pv0, nvt) == 0 &&
                    memcmp(b->expr_tag + e0, b->expr_tag + pe0, net * sizeof(*b->expr_tag)) == 0 &&
                    memcmp(b->expr_idx + 4 * (size_t)e0, b->expr_idx + 4 * (size_t)pe0, 4 * (size_t)net * sizeof(uint32_t)) == 0 &&
                    (!b->var_comp || memcmp(b->var_comp + v0, b->var_comp + pv0, nvt * sizeof(*b->var_comp)) == 0) &&
                    (!b->expr_comp || memcmp(b->expr_comp + e0, b->expr_comp + pe0, net * sizeof(*b->expr_comp)) == 0)) {
                    memcpy(&p.var_info[v0], &p.var_info[pv0], nvt * sizeof(uint16_t));
                    memcpy(&p.expr_tagx[e0], &p.expr_tagx[pe0], net);
                    memcpy(&p.expr_comp[e0], &p.expr_comp[pe0], net * sizeof(uint16_t));
                    memcpy(&p.expr_idx16[4 * (size_t)e0], &p.expr_idx16[4 * (size_t)pe0], 4 * (size_t)net * sizeof(uint16_t));
                    p.sys_ncomp[s] = p.sys_ncomp[s - 1];
                    p.sys_large[s] = p.sys_large[s - 1];
                    p.same_as_prev[s] = 1;
                    pt.nnz += prev_nnz;
                    pt.n_large += prev_large;
                    continue;
                }
            }
            const uint64_t nnz_before = pt.nnz;
            const uint32_t large_before = pt.n_large;
            if (nvt > FX_MAX_LARGE_SYSTEM_VARS) {
                bad(FX_ERR_TOO_LARGE, s, "system %u has %u variables (limit %u)", s, nvt, FX_MAX_LARGE_SYSTEM_VARS);
                break;
            }
            bool large = nvt > FX_MAX_SYSTEM_VARS;
            pt.max_vars_all = std::max(pt.max_vars_all, nvt);
            pt.max_exprs_all = std::max(pt.max_exprs_all, net);

            uint32_t ncomp = 0;
            free_rank.assign(nvt, -1);
            int32_t rank = 0;
            for (uint32_t i = 0; i < nvt; ++i) {
                uint16_t c = b->var_comp ? b->var_comp[v0 + i] : 0;
                bool fixed = b->var_fixed[v0 + i] != 0;
                uint16_t info;
                if (c == FX_NO_COMPONENT) {
                    info = fx::VAR_COMP_NONE;
                } else {
                    if (c >= fx::VAR_COMP_NONE) {
                        bad(FX_ERR_INVALID, s, "system %u: component id %u too large", s, c);
                        break;
                    }
                    info = c;
                    ncomp = std::max<uint32_t>(ncomp, c + 1u);
                    if (!fixed) free_rank[i] = rank++;
                }
                if (fixed) info |= fx::VAR_FIXED_BIT;
                p.var_info[v0 + i] = info;
            }
            for (uint32_t i = 0; i < net && pt.err == FX_OK; ++i) {
                uint16_t c = b->expr_comp ? b->expr_comp[e0 + i] : 0;
                if (c == FX_NO_COMPONENT) {
                    c = fx::VAR_COMP_NONE;  // never selected by any component loop
                } else {
                    if (c >= fx::VAR_COMP_NONE) {
                        bad(FX_ERR_INVALID, s, "system %u: component id %u too large", s, c);
                        break;
                    }
                    ncomp = std::max<uint32_t>(ncomp, c + 1u);
                }
                p.expr_comp[e0 + i] = c;
            }
            if (pt.err != FX_OK) break;
            p.sys_ncomp[s] = (uint16_t)ncomp;

            comp_free.assign(ncomp, 0);
            comp_rows.assign(ncomp, 0);
            comp_pairs.assign(ncomp, 0);
            comp_ents.assign(ncomp, 0);
            comp_tri.assign(ncomp, 0);
            for (uint32_t i = 0; i < nvt; ++i) {
                uint16_t info = p.var_info[v0 + i];
                uint16_t c = info & fx::VAR_COMP_MASK;
                if (c != fx::VAR_COMP_NONE && !(info & fx::VAR_FIXED_BIT)) comp_free[c] += 1;
            }

            for (uint32_t i = 0; i < net; ++i) {
                const uint32_t e = e0 + i;
                const int tag = b->expr_tag[e];
                if (tag < 0 || tag >= n_tags) {
                    bad(FX_ERR_INVALID, s, "expression %u of system %u: bad tag %d", i, s, (unsigned)tag);
                    break;
                }
                const uint32_t* f = b->expr_idx + 4 * (size_t)e;
                uint32_t vars8[8];
                int k = fx::expand_vars<true>(tag, f, vars8);
                bool in_range = true;
                for (int q = 0; q < k; ++q) {
                    if (vars8[q] >= nvt) {
                        bad(FX_ERR_INVALID, s, "expression %u of system %u reads variable %u >= %u", i, s, vars8[q], nvt);
                        in_range = false;
                        break;
                    }
                }
                if (!in_range) break;
                for (int q = 0; q < 4; ++q) p.expr_idx16[4 * (size_t)e + q] = (uint16_t)(f[q] < nvt ? f[q] : 0);
                p.expr_tagx[e] = (uint8_t)tag;
                bool dup = false, all_free = true, distinct = true;
                for (int q = 0; q < k; ++q) {
                    all_free = all_free && free_rank[vars8[q]] >= 0;
                    for (int u = q + 1; u < k; ++u) {
                        distinct = distinct && vars8[q] != vars8[u];
                        dup = dup || (vars8[q] == vars8[u] && free_rank[vars8[q]] >= 0);
                    }
                }
                if (dup) p.expr_tagx[e] |= 0x80;
                uint16_t c = p.expr_comp[e];
                if (c != fx::VAR_COMP_NONE) {
                    comp_rows[c] += 1;
                    uint32_t kf = 0;  // entries with a free variable (an upper bound inside the kernel's component)
                    for (int q = 0; q < k; ++q) kf += free_rank[vars8[q]] >= 0;
                    comp_pairs[c] += kf * kf;
                    comp_ents[c] += kf;
                    // the lower triangle only: unordered pairs, plus once more for two entries on one column
                    uint32_t twice = 0;
                    for (int q = 0; q < k; ++q)
                        for (int u = q + 1; u < k; ++u) twice += vars8[q] == vars8[u] && free_rank[vars8[q]] >= 0;
                    comp_tri[c] += kf * (kf + 1u) / 2u + twice;
                }
                if (all_free && distinct) {
                    pt.nnz += (uint64_t)k;
                } else {
                    int32_t cols[8];
                    pt.nnz += (uint64_t)row_columns(vars8, k, free_rank.data(), cols, nullptr);
                }
            }
            if (pt.err != FX_OK) break;
            uint32_t cp_max = 0, ce_max = 0, ct_max = 0;
            for (uint32_t c = 0; c < ncomp; ++c) {
                cp_max = std::max(cp_max, comp_pairs[c]);
                ce_max = std::max(ce_max, comp_ents[c]);
                ct_max = std::max(ct_max, comp_tri[c]);
            }
            bool wide = false;  // more than one wavefront's columns, but still an LDS-resident dense problem
            uint32_t cf_max = 0, cr_max = 0;
            for (uint32_t c = 0; c < ncomp; ++c) {
                cf_max = std::max(cf_max, comp_free[c]);
                cr_max = std::max(cr_max, comp_rows[c]);
            }
            if (!large && cr_max <= FX_MAX_ROWS && cf_max > FX_MAX_FREE_VARS && cf_max <= FX_MAX_WIDE_FREE_VARS) wide = true;
            large = large || cf_max > FX_MAX_FREE_VARS || cr_max > FX_MAX_ROWS;
            if (large) {
                p.sys_large[s] = wide ? 2 : 1;
                pt.n_large += 1;
                pt.max_pairs_large = std::max(pt.max_pairs_large, cp_max);
                pt.max_ents_large = std::max(pt.max_ents_large, ce_max);
                if (wide) {
                    pt.w_max_free = std::max(pt.w_max_free, cf_max);
                    pt.w_max_rows = std::max(pt.w_max_rows, cr_max);
                    pt.w_max_vars = std::max(pt.w_max_vars, nvt);
                }
            } else {  // LDS layout and kernel instantiation are sized by the one-wavefront systems only
                pt.max_vars = std::max(pt.max_vars, nvt);
                pt.max_exprs = std::max(pt.max_exprs, net);
                pt.max_pairs = std::max(pt.max_pairs, cp_max);
                pt.max_ents = std::max(pt.max_ents, ce_max);
                pt.max_pairs_tri = std::max(pt.max_pairs_tri, ct_max);
                for (uint32_t c = 0; c < ncomp; ++c) {
                    pt.max_free = std::max(pt.max_free, comp_free[c]);
                    pt.max_rows = std::max(pt.max_rows, comp_rows[c]);
                }
            }
            prev_nnz = pt.nnz - nnz_before;
            prev_large = pt.n_large - large_before;
            if (s > s_lo && pt.chain_same)
                pt.chain_same = nvt == v0 - b->var_off[s - 1] && net == e0 - b->expr_off[s - 1] && outputs_equal(s, s - 1, nvt, net);
        }
    }, &nr, min_items);
    for (uint32_t t = 0; t < nr; ++t)  // ranges are in system order: the first failing System is reported
        if (part[t].err != FX_OK) return fail(part[t].err, "%s", part[t].msg);
    for (uint32_t t = 0; t < nr; ++t) {
        p.nnz += part[t].nnz;
        p.n_large += part[t].n_large;
        p.max_free = std::max(p.max_free, part[t].max_free);
        p.max_rows = std::max(p.max_rows, part[t].max_rows);
        p.max_vars = std::max(p.max_vars, part[t].max_vars);
        p.max_exprs = std::max(p.max_exprs, part[t].max_exprs);
        p.max_vars_all = std::max(p.max_vars_all, part[t].max_vars_all);
        p.max_exprs_all = std::max(p.max_exprs_all, part[t].max_exprs_all);
        p.max_pairs = std::max(p.max_pairs, part[t].max_pairs);
        p.max_ents = std::max(p.max_ents, part[t].max_ents);
        p.max_pairs_tri = std::max(p.max_pairs_tri, part[t].max_pairs_tri);
        p.max_pairs_large = std::max(p.max_pairs_large, part[t].max_pairs_large);
        p.max_ents_large = std::max(p.max_ents_large, part[t].max_ents_large);
        p.w_max_free = std::max(p.w_max_free, part[t].w_max_free);
        p.w_max_rows = std::max(p.w_max_rows, part[t].w_max_rows);
        p.w_max_vars = std::max(p.w_max_vars, part[t].w_max_vars);
    }
    for (uint32_t s = 0; s < n; ++s)
        if (p.sys_large[s] == 2) p.wide_list.push_back(s);
    // Components of 65 ... 128 columns have two homes. The wide kernel (fx_wide.hip: one wavefront per System, dense packed
    // factor in LDS) holds 4 / 2 / 1 Systems per CU; the team kernels (fx_sparse_team.h: a workgroup of 16 wavefronts per
    // System, sparse factor, the whole solve in one launch) finish ONE such System in half the time (66 variables: 0.17
    // against 0.36 ms) and cost half a microsecond per System at any size, which only the wide kernel's four-per-CU case
    // beats, and only from a thousand Systems on. By cost, measured on the reference's hinged-triangle sketches of 66 / 98 /
    // 126 variables, 1 ... 20 000 per batch (tools/hinged_batch.py):
    //   team  = max(0.18 ms, n (0.49 us + 0.0011 (c - 66)))   below 768 Systems (16 wavefronts per System);
    //           max(0.30 ms, n (0.19 us + 0.0014 (c - 66)))   from there on (2 / 4 wavefronts per System, round 4)
    //   wide  = ceil(n / (256 CUs x Systems per CU)) x (0.37 ms + 0.0123 (c - 66))
    // Both follow the reference's iteration path; they sum in different orders, so which one ran shows in the last bits (as it does
    // for a large System alone / among seven others): fx_ctx_set_wide_routing(ctx, 0 | 1) pins it.
    if (!p.wide_list.empty()) {
        const int routing = g_wide_routing_pinned != -2 ? g_wide_routing_pinned : g_wide_routing;
        bool team = routing == 0;
        if (routing < 0) {
            fx::DeviceBatch probe{};
            probe.w_max_free = p.w_max_free;
            probe.w_max_vars = p.w_max_vars;
            probe.w_max_rows = p.w_max_rows;
            const size_t lds = fx::wide_lds_bytes(probe);
            const double per_cu = lds ? std::min<double>(4., std::floor(160. * 1024. / (double)lds)) : 1.;
            const double c = (double)p.w_max_free - 66., nw = (double)p.wide_list.size();
            // (round 4: from 768 Systems on the team kernels run 2 / 4 wavefronts per System instead of 16 — fx_sparse.hip:
            // team_waves_for — 0.19 us per 66-variable System, 0.27 per 126-variable one)
            const double team_ms = nw < 768. ? std::max(0.18, nw * (0.49e-3 + 1.1e-6 * c)) : std::max(0.30, nw * (0.19e-3 + 1.4e-6 * c));
            const double wide_ms = std::ceil(nw / (256. * std::max(per_cu, 1.))) * (0.37 + 0.0123 * c);
            team = team_ms < wide_ms;
        }
        p.wide_decision = team ? 0 : 1;
        if (team) {
            for (uint32_t s : p.wide_list) p.sys_large[s] = 1;
            p.wide_list.clear();
            p.w_max_free = p.w_max_rows = p.w_max_vars = 0;
        }
    }
    // Same structure everywhere? (sizes, components, fixed flags, kinds and element fields of System 0.) The
    // grouped kernel then builds its per-System lists once per lane row instead of once per System.
    if (n >= 2) {
        const uint32_t nv0 = b->var_off[1] - b->var_off[0], ne0 = b->expr_off[1] - b->expr_off[0];
        bool same = (uint64_t)nv0 * n == nv && (uint64_t)ne0 * n == ne;
        // (inside a range the walk above has compared every System with the one before it: the ranges' first Systems
        // are left; sizes equal all along and the totals above make the offsets regular)
        for (uint32_t t = 0; same && t < nr; ++t) {
            const uint32_t s = part[t].first;
            same = part[t].chain_same && b->var_off[s + 1] - b->var_off[s] == nv0 && b->expr_off[s + 1] - b->expr_off[s] == ne0 &&
                   b->var_off[s] == s * nv0 && b->expr_off[s] == s * ne0 &&
                   memcmp(&p.var_info[(size_t)s * nv0], &p.var_info[0], nv0 * sizeof(uint16_t)) == 0 &&
                   memcmp(&p.expr_tagx[(size_t)s * ne0], &p.expr_tagx[0], ne0) == 0 &&
                   memcmp(&p.expr_comp[(size_t)s * ne0], &p.expr_comp[0], ne0 * sizeof(uint16_t)) == 0 &&
                   memcmp(&p.expr_idx16[4 * (size_t)s * ne0], &p.expr_idx16[0], 4 * (size_t)ne0 * sizeof(uint16_t)) == 0;
        }
        p.uniform = same ? 1u : 0u;
    }
    // Not one structure, but a batch the grouped kernel will take: the structure classes (a few sketches, each with
    // many parameter sets, is the other common batch). Class = first System with the same sizes, components, fixed
    // flags, kinds and element fields: found by a 64-bit hash of those arrays, confirmed by comparing them.
    if (!p.uniform && n >= 1024 && p.max_free > 0 && p.max_free <= 48) {
        auto slices = [&](uint32_t s, const void* ptr[4], size_t len[4]) {
            const uint32_t v0 = b->var_off[s], nvs = b->var_off[s + 1] - v0, e0 = b->expr_off[s], nes = b->expr_off[s + 1] - e0;
            ptr[0] = &p.var_info[v0];            len[0] = nvs * sizeof(uint16_t);
            ptr[1] = &p.expr_tagx[e0];           len[1] = nes;
            ptr[2] = &p.expr_comp[e0];           len[2] = nes * sizeof(uint16_t);
            ptr[3] = &p.expr_idx16[4 * (size_t)e0]; len[3] = 4 * (size_t)nes * sizeof(uint16_t);
        };
        std::vector<uint64_t> hash(n);
        parallel_ranges(n, (uint64_t)nv + ne, [&](uint32_t, uint32_t lo, uint32_t hi) {
            for (uint32_t s = lo; s < hi; ++s) {
                if (s > lo && p.same_as_prev[s]) {
                    hash[s] = hash[s - 1];
                    continue;
                }
                const void* ptr[4];
                size_t len[4];
                slices(s, ptr, len);
                uint64_t h = 0x9E3779B97F4A7C15ull ^ ((uint64_t)len[0] << 32) ^ len[1];
                for (int k = 0; k < 4; ++k) {
                    const unsigned char* q = static_cast<const unsigned char*>(ptr[k]);
                    size_t i = 0;
                    for (; i + 8 <= len[k]; i += 8) {
                        uint64_t w;
                        memcpy(&w, q + i, 8);
                        h = (h ^ w) * 0xFF51AFD7ED558CCDull;
                        h ^= h >> 29;
                    }
                    uint64_t w = 0;
                    if (i < len[k]) memcpy(&w, q + i, len[k] - i);
                    h = (h ^ w ^ (uint64_t)k) * 0xC4CEB9FE1A85EC53ull;
                    h ^= h >> 32;
                }
                hash[s] = h;
            }
        });
        std::unordered_map<uint64_t, uint32_t> first;
        first.reserve(1024);
        p.sys_class.resize(n);
        for (uint32_t s = 0; s < n; ++s) {
            if (s && p.same_as_prev[s]) {
                p.sys_class[s] = p.sys_class[s - 1];
                continue;
            }
            auto it = first.find(hash[s]);
            if (it == first.end()) {
                first.emplace(hash[s], s);
                p.sys_class[s] = s;
                continue;
            }
            const void *pa[4], *pb[4];
            size_t la[4], lb[4];
            slices(s, pa, la);
            slices(it->second, pb, lb);
            bool eq = true;
            for (int k = 0; eq && k < 4; ++k) eq = la[k] == lb[k] && memcmp(pa[k], pb[k], la[k]) == 0;
            p.sys_class[s] = eq ? it->second : s;  // (a colliding hash: the System is its own class)
        }
    }
    return FX_OK;
}

// Every System's raw structure arrays against the first System's (what analyze's walk compares System by System when it is not
// told): threads over ranges of Systems, a pass over the batch's structure bytes.
bool verify_one_structure(const fx_batch* b) {
    const uint32_t n = b->n_systems;
    if (n < 2) return true;
    const uint32_t nv0 = b->var_off[1], ne0 = b->expr_off[1];
    std::atomic<uint32_t> differs{0};
    parallel_ranges(n, (uint64_t)n * (nv0 + ne0), [&](uint32_t, uint32_t lo, uint32_t hi) {
        bool same = true;
        for (uint32_t s = std::max(lo, 1u); s < hi && same; ++s) {
            const size_t v0 = (size_t)s * nv0, e0 = (size_t)s * ne0;
            same = memcmp(b->var_fixed + v0, b->var_fixed, nv0) == 0 && memcmp(b->expr_tag + e0, b->expr_tag, ne0 * sizeof(*b->expr_tag)) == 0 &&
                   memcmp(b->expr_idx + 4 * e0, b->expr_idx, 4 * (size_t)ne0 * sizeof(uint32_t)) == 0 &&
                   (!b->var_comp || memcmp(b->var_comp + v0, b->var_comp, nv0 * sizeof(*b->var_comp)) == 0) &&
                   (!b->expr_comp || memcmp(b->expr_comp + e0, b->expr_comp, ne0 * sizeof(*b->expr_comp)) == 0);
        }
        if (!same) differs.store(1u, std::memory_order_relaxed);
    });
    return differs.load() == 0u;
}

}  // namespace fxh
