// Device residency of a batch: the one block an upload makes (upload_planned), what is built on first use from the
// device's own arrays (residual buffer, CSR structure, SinglePass blocks, QR plans, the component walk), the way back
// (read_back_and_free) and the release of a batch's blocks into the context's cache.
#include "fx_host.h"

namespace fxh {

int bind(fx_ctx* ctx) {
    if (!ctx) return fail(FX_ERR_INVALID, "ctx is NULL");
    FX_HIP(hipSetDevice(ctx->device));
    return FX_OK;
}

// The arrays of the row-parallel kernels (EvalPlan) and the residual buffer, on first use: the structure is read back
// from the device's own compact arrays (nothing is kept on the host for batches that are only ever solved).
int ensure_resid(fx_ctx* ctx, fx_dbatch* db) {
    fx::DeviceBatch& d = db->d;
    if (d.resid) return FX_OK;
    const uint32_t n = d.n_systems;
    std::vector<uint32_t> var_off((size_t)n + 1, 0), expr_off((size_t)n + 1, 0);
    std::vector<uint16_t> var_info(d.n_vars), expr_idx(4 * (size_t)d.n_exprs);
    std::vector<uint8_t> expr_tag(d.n_exprs);
    FX_HIP(hipMemcpyAsync(var_off.data(), d.var_off, var_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipMemcpyAsync(expr_off.data(), d.expr_off, expr_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_vars) FX_HIP(hipMemcpyAsync(var_info.data(), d.var_info, var_info.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_exprs) {
        FX_HIP(hipMemcpyAsync(expr_idx.data(), d.expr_idx, expr_idx.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_tag.data(), d.expr_tag, expr_tag.size(), hipMemcpyDeviceToHost, ctx->stream));
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));
    EvalPlan ep;
    build_eval_plan(n, var_off.data(), expr_off.data(), var_info.data(), expr_tag.data(), expr_idx.data(), ep);
    int rc = dev_alloc_copy(ctx, db, &d.expr_var0, ep.expr_var0.data(), ep.expr_var0.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.row_perm, ep.row_perm.data(), ep.row_perm.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.row_sysoff, ep.row_sysoff.data(), ep.row_sysoff.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.blk_info, ep.blk_info.data(), ep.blk_info.size());
    double* resid = nullptr;
    if (!rc) rc = dev_alloc_copy(ctx, db, &resid, (const double*)nullptr, d.n_exprs);
    if (rc) return rc;
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die with this frame
    d.resid = resid;  // set last: marks the arrays as complete
    return FX_OK;
}

// The CSR Jacobian structure of a resident batch, built on first use from the device's own compact
// arrays (nothing is kept on the host for batches that are only ever solved).
int ensure_csr(fx_ctx* ctx, fx_dbatch* db) {
    fx::DeviceBatch& d = db->d;
    if (d.jrow_ptr) return FX_OK;
    int rc = ensure_resid(ctx, db);
    if (rc) return rc;
    const uint32_t n = d.n_systems;
    std::vector<uint32_t> var_off((size_t)n + 1, 0), expr_off((size_t)n + 1, 0);
    std::vector<uint16_t> var_info(d.n_vars), expr_idx(4 * (size_t)d.n_exprs);
    std::vector<uint8_t> expr_tag(d.n_exprs);
    const size_t nblk = ((size_t)d.n_exprs + 255) / 256;
    std::vector<fx::BlockInfo> blk(nblk);
    FX_HIP(hipMemcpyAsync(var_off.data(), d.var_off, var_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipMemcpyAsync(expr_off.data(), d.expr_off, expr_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_vars) FX_HIP(hipMemcpyAsync(var_info.data(), d.var_info, var_info.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_exprs) {
        FX_HIP(hipMemcpyAsync(expr_idx.data(), d.expr_idx, expr_idx.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_tag.data(), d.expr_tag, expr_tag.size(), hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(blk.data(), d.blk_info, nblk * sizeof(fx::BlockInfo), hipMemcpyDeviceToHost, ctx->stream));
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));
    CsrPlan csr;
    build_csr(n, var_off.data(), expr_off.data(), var_info.data(), expr_tag.data(), expr_idx.data(), csr);
    if (csr.jcol.size() != d.nnz) return fail(FX_ERR_INVALID, "internal: CSR size %zu != counted %llu", csr.jcol.size(), (unsigned long long)d.nnz);
    for (size_t k = 0; k < nblk; ++k) {
        const size_t r0 = k * 256, r1 = std::min<size_t>(r0 + 256, d.n_exprs);
        blk[k].jbase = csr.jrow_ptr[r0];
        blk[k].jcount = csr.jrow_ptr[r1] - csr.jrow_ptr[r0];
    }
    uint32_t* jrow_ptr = nullptr;
    rc = dev_alloc_copy(ctx, db, &jrow_ptr, csr.jrow_ptr.data(), csr.jrow_ptr.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.jcol, csr.jcol.data(), csr.jcol.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.jslot, csr.jslot.data(), csr.jslot.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.jvals, (const double*)nullptr, d.nnz);
    if (rc) return rc;
    if (nblk) FX_HIP(hipMemcpyAsync(d.blk_info, blk.data(), nblk * sizeof(fx::BlockInfo), hipMemcpyHostToDevice, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die with this frame
    d.jrow_ptr = jrow_ptr;  // set last: marks the structure as complete
    return FX_OK;
}

// Builds the SinglePass blocks of every System that runs in the fused kernel (once per batch; the
// structure is read back from the device arrays, so nothing extra is kept on the host for batches
// that never ask for it). Large Systems get theirs inside the sparse path.
int ensure_units(fx_ctx* ctx, fx_dbatch* db) {
    fx::DeviceBatch& d = db->d;
    if (d.sys_unit_off) return FX_OK;
    const uint32_t n = d.n_systems;
    std::vector<uint32_t> var_off((size_t)n + 1), expr_off((size_t)n + 1);
    std::vector<uint16_t> var_info(d.n_vars), expr_idx(4 * (size_t)d.n_exprs);
    std::vector<uint8_t> expr_tag(d.n_exprs), sys_large(n);
    std::vector<uint16_t> sys_ncomp(n);
    if (db->packed_base && (ctx->pinned || db->zero_copy) && db->packed_bytes <= fx_ctx::PINNED_HALF) {
        // a small batch is one block on the device: one copy of it (page-locked, second half of the staging area) instead
        // of seven small ones — the structure arrays are then taken from that image
        // (a zero-copy block is host memory already)
        unsigned char* img = db->zero_copy ? db->zc_image : ctx->pinned + fx_ctx::PINNED_HALF;
        if (!db->zero_copy) FX_HIP(hipMemcpyAsync(img, db->packed_base, db->packed_bytes, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipStreamSynchronize(ctx->stream));
        ctx->stream_synced();
        auto grab = [&](void* dst, const void* dev, size_t bytes) {
            if (bytes) memcpy(dst, img + (reinterpret_cast<const unsigned char*>(dev) - db->packed_base), bytes);
        };
        grab(var_off.data(), d.var_off, var_off.size() * 4);
        grab(expr_off.data(), d.expr_off, expr_off.size() * 4);
        grab(var_info.data(), d.var_info, var_info.size() * 2);
        grab(expr_idx.data(), d.expr_idx, expr_idx.size() * 2);
        grab(expr_tag.data(), d.expr_tag, expr_tag.size());
        grab(sys_large.data(), d.sys_large, n);
        grab(sys_ncomp.data(), d.sys_ncomp, (size_t)n * 2);
    } else {
    FX_HIP(hipMemcpyAsync(var_off.data(), d.var_off, var_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipMemcpyAsync(expr_off.data(), d.expr_off, expr_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_vars) FX_HIP(hipMemcpyAsync(var_info.data(), d.var_info, var_info.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_exprs) {
        FX_HIP(hipMemcpyAsync(expr_idx.data(), d.expr_idx, expr_idx.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_tag.data(), d.expr_tag, expr_tag.size(), hipMemcpyDeviceToHost, ctx->stream));
    }
    if (n) {
        FX_HIP(hipMemcpyAsync(sys_large.data(), d.sys_large, n, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(sys_ncomp.data(), d.sys_ncomp, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));
    }

    std::vector<uint32_t> sys_unit_off((size_t)n + 1, 0);
    std::vector<fx::UnitDesc> desc;
    std::vector<uint32_t> unit_rows;
    std::vector<uint16_t> unit_vars;
    uint32_t max_unit_free = 0, max_unit_rows = 0;
    // Large Systems: if every block fits the one-wavefront limits the System is walked on the device
    // with its System-wide vectors in an HBM scratch area (the GLOBAL instantiation); otherwise it
    // stays with the host-driven sparse path.
    std::vector<uint32_t> g_list, g_off;
    std::vector<uint8_t> g_ok(n, 0);
    uint32_t g_total = 0, max_unit_free_g = 0, max_unit_rows_g = 0;

    // Systems are independent: ranges of them are decomposed on separate host threads into local
    // lists (offsets relative to the range), stitched together in System order afterwards.
    struct Range {
        uint32_t s_lo = 0, s_hi = 0;
        std::vector<fx::UnitDesc> desc;
        std::vector<uint32_t> rows, desc_count;  // desc_count[s - s_lo]: entries of System s
        std::vector<uint16_t> vars;
        std::vector<uint32_t> g_sys, g_nvt;
        uint32_t max_free = 0, max_rows = 0, max_free_g = 0, max_rows_g = 0;
    } ranges[MAX_RANGES];
    uint32_t n_ranges = 1;
    parallel_ranges(n, (uint64_t)d.n_exprs * 4, [&](uint32_t t, uint32_t s_lo, uint32_t s_hi) {
        Range& R = ranges[t];
        R.s_lo = s_lo;
        R.s_hi = s_hi;
        R.desc_count.assign(s_hi - s_lo, 0);
        fx::Incidence inc;
        fx::UnitList units;
        std::vector<uint32_t> free_sorted;
        for (uint32_t s = s_lo; s < s_hi; ++s) {
            const size_t desc_mark = R.desc.size(), rows_mark = R.rows.size(), vars_mark = R.vars.size();
            bool fits = true;
            uint32_t mf = 0, mrw = 0;
            const uint32_t v0 = var_off[s], nvt = var_off[s + 1] - v0;
            const uint32_t e0 = expr_off[s], net = expr_off[s + 1] - e0;
            inc.build(nvt, net, expr_tag.data() + e0, expr_idx.data() + 4 * (size_t)e0);
            fx::SinglePassDecomposer dec(inc);
            for (uint32_t c = 0; c < sys_ncomp[s]; ++c) {
                free_sorted.clear();
                bool any_var = false;
                for (uint32_t i = 0; i < nvt; ++i) {
                    uint16_t info = var_info[v0 + i];
                    if ((info & fx::VAR_COMP_MASK) != c) continue;
                    any_var = true;
                    if (!(info & fx::VAR_FIXED_BIT)) free_sorted.push_back(i);
                }
                if (!any_var) continue;  // skipped by the reference (`elements.is_empty()`)
                dec.run(free_sorted, units);
                if (units.count() == 0) {
                    R.desc.push_back(fx::UnitDesc{0, 0, 0, 0, (uint16_t)c, (uint16_t)(fx::UNIT_FIRST | fx::UNIT_EMPTY)});
                    continue;
                }
                for (uint32_t u = 0; u < units.count(); ++u) {
                    fx::UnitDesc ud{};
                    ud.row_off = (uint32_t)R.rows.size();
                    ud.var_off = (uint32_t)R.vars.size();
                    ud.nrows = (uint16_t)(units.row_off[u + 1] - units.row_off[u]);
                    ud.nvars = (uint16_t)(units.var_off[u + 1] - units.var_off[u]);
                    ud.comp = (uint16_t)c;
                    ud.flags = u == 0 ? fx::UNIT_FIRST : 0;
                    for (uint32_t k = units.row_off[u]; k < units.row_off[u + 1]; ++k) R.rows.push_back(units.rows[k]);
                    for (uint32_t k = units.var_off[u]; k < units.var_off[u + 1]; ++k) R.vars.push_back((uint16_t)units.vars[k]);
                    const uint32_t bf = units.var_off[u + 1] - units.var_off[u], br = units.row_off[u + 1] - units.row_off[u];
                    fits = fits && bf <= FX_MAX_FREE_VARS && br <= FX_MAX_ROWS;
                    mf = std::max(mf, bf);
                    mrw = std::max(mrw, br);
                    R.desc.push_back(ud);
                }
            }
            if (!sys_large[s]) {
                R.max_free = std::max(R.max_free, mf);
                R.max_rows = std::max(R.max_rows, mrw);
            } else if (fits) {
                R.g_sys.push_back(s);
                R.g_nvt.push_back(nvt);
                R.max_free_g = std::max(R.max_free_g, mf);
                R.max_rows_g = std::max(R.max_rows_g, mrw);
            } else {  // some block is itself too large: the System keeps the sparse path, drop its entries
                R.desc.resize(desc_mark);
                R.rows.resize(rows_mark);
                R.vars.resize(vars_mark);
            }
            R.desc_count[s - s_lo] = (uint32_t)(R.desc.size() - desc_mark);
        }
    }, &n_ranges);
    for (uint32_t t = 0; t < n_ranges; ++t) {
        Range& R = ranges[t];
        const uint32_t row_base = (uint32_t)unit_rows.size(), var_base = (uint32_t)unit_vars.size();
        uint32_t at = (uint32_t)desc.size();
        for (uint32_t s = R.s_lo; s < R.s_hi; ++s) {
            sys_unit_off[s] = at;
            at += R.desc_count[s - R.s_lo];
        }
        for (fx::UnitDesc ud : R.desc) {
            ud.row_off += row_base;
            ud.var_off += var_base;
            desc.push_back(ud);
        }
        unit_rows.insert(unit_rows.end(), R.rows.begin(), R.rows.end());
        unit_vars.insert(unit_vars.end(), R.vars.begin(), R.vars.end());
        for (size_t k = 0; k < R.g_sys.size(); ++k) {
            g_ok[R.g_sys[k]] = 1;
            g_list.push_back(R.g_sys[k]);
            g_off.push_back(g_total);
            g_total += R.g_nvt[k];
        }
        max_unit_free = std::max(max_unit_free, R.max_free);
        max_unit_rows = std::max(max_unit_rows, R.max_rows);
        max_unit_free_g = std::max(max_unit_free_g, R.max_free_g);
        max_unit_rows_g = std::max(max_unit_rows_g, R.max_rows_g);
    }
    sys_unit_off[n] = (uint32_t)desc.size();
    int rc = dev_alloc_copy(ctx, db, &d.unit_desc, desc.data(), desc.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.unit_rows, unit_rows.data(), unit_rows.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &d.unit_vars, unit_vars.data(), unit_vars.size());
    uint32_t* off = nullptr;
    if (!rc) rc = dev_alloc_copy(ctx, db, &off, sys_unit_off.data(), sys_unit_off.size());
    if (rc) return rc;
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die with this frame
    if (!g_list.empty()) {
        rc = dev_alloc_copy(ctx, db, &d.g_list, g_list.data(), g_list.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &d.g_off, g_off.data(), g_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &d.g_xs, (const double*)nullptr, 2 * (size_t)g_total);
        if (!rc) rc = dev_alloc_copy(ctx, db, &d.g_vout, (const double*)nullptr, g_total);
        if (!rc) rc = dev_alloc_copy(ctx, db, &d.g_colof, (const int16_t*)nullptr, g_total);
        if (rc) return rc;
        FX_HIP(hipStreamSynchronize(ctx->stream));
        d.n_g = (uint32_t)g_list.size();
        d.max_unit_free_g = max_unit_free_g;
        d.max_unit_rows_g = max_unit_rows_g;
    }
    db->h_units_on_device = g_ok;
    d.max_unit_free = max_unit_free;
    d.max_unit_rows = max_unit_rows;
    if (fx::solve_lds_bytes_units(d) > 160u * 1024u)
        return fail(FX_ERR_TOO_LARGE, "SinglePass blocks need %zu bytes of LDS per wavefront (limit 163840)", fx::solve_lds_bytes_units(d));
    db->n_units = (uint32_t)desc.size();
    db->n_unit_rows = (uint32_t)unit_rows.size();
    db->n_unit_vars = (uint32_t)unit_vars.size();
    d.sys_unit_off = off;  // set last: marks the plan as complete
    return FX_OK;
}

// Builds (once per resident batch and decomposer) the QR plans of every System the one-wavefront kernel takes.
// Systems of one structure share a plan: a batch of one sketch with many parameter sets is analysed once.
int ensure_qr_plans(fx_ctx* ctx, fx_dbatch* db, bool units) {
    fx::DeviceBatch& d = db->d;
    fx::QrPlans& Q = units ? d.qr_units : d.qr_none;
    if (Q.desc) return FX_OK;
    const uint32_t n = d.n_systems;
    std::vector<uint32_t> var_off((size_t)n + 1), expr_off((size_t)n + 1), sys_class;
    std::vector<uint16_t> var_info(d.n_vars), expr_idx(4 * (size_t)d.n_exprs), expr_comp(d.n_exprs), sys_ncomp(n);
    std::vector<uint8_t> expr_tag(d.n_exprs), sys_large(n);
    std::vector<uint32_t> sys_unit_off, unit_rows;
    std::vector<fx::UnitDesc> unit_desc;
    std::vector<uint16_t> unit_vars;
    if (d.sys_class && !units) sys_class.resize(n);
    if (db->packed_base && (ctx->pinned || db->zero_copy) && db->packed_bytes <= fx_ctx::PINNED_HALF) {
        // a small batch is one block on the device: its image in one page-locked copy instead of nine small ones
        // (a zero-copy block is host memory already)
        unsigned char* img = db->zero_copy ? db->zc_image : ctx->pinned + fx_ctx::PINNED_HALF;
        if (!db->zero_copy) FX_HIP(hipMemcpyAsync(img, db->packed_base, db->packed_bytes, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipStreamSynchronize(ctx->stream));
        ctx->stream_synced();
        auto grab = [&](void* dst, const void* dev, size_t bytes) {
            if (bytes) memcpy(dst, img + (reinterpret_cast<const unsigned char*>(dev) - db->packed_base), bytes);
        };
        grab(var_off.data(), d.var_off, var_off.size() * 4);
        grab(expr_off.data(), d.expr_off, expr_off.size() * 4);
        grab(var_info.data(), d.var_info, var_info.size() * 2);
        grab(expr_idx.data(), d.expr_idx, expr_idx.size() * 2);
        grab(expr_tag.data(), d.expr_tag, expr_tag.size());
        grab(expr_comp.data(), d.expr_comp, expr_comp.size() * 2);
        grab(sys_large.data(), d.sys_large, n);
        grab(sys_ncomp.data(), d.sys_ncomp, (size_t)n * 2);
        if (!sys_class.empty()) grab(sys_class.data(), d.sys_class, (size_t)n * 4);
    } else {
    FX_HIP(hipMemcpyAsync(var_off.data(), d.var_off, var_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    FX_HIP(hipMemcpyAsync(expr_off.data(), d.expr_off, expr_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_vars) FX_HIP(hipMemcpyAsync(var_info.data(), d.var_info, var_info.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    if (d.n_exprs) {
        FX_HIP(hipMemcpyAsync(expr_idx.data(), d.expr_idx, expr_idx.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_tag.data(), d.expr_tag, expr_tag.size(), hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(expr_comp.data(), d.expr_comp, expr_comp.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    }
    if (n) {
        FX_HIP(hipMemcpyAsync(sys_large.data(), d.sys_large, n, hipMemcpyDeviceToHost, ctx->stream));
        FX_HIP(hipMemcpyAsync(sys_ncomp.data(), d.sys_ncomp, (size_t)n * 2, hipMemcpyDeviceToHost, ctx->stream));
        if (!sys_class.empty()) FX_HIP(hipMemcpyAsync(sys_class.data(), d.sys_class, (size_t)n * 4, hipMemcpyDeviceToHost, ctx->stream));
    }
    }
    if (units) {
        if (!d.sys_unit_off) return fail(FX_ERR_INVALID, "internal: SinglePass blocks are not built yet");
        sys_unit_off.resize((size_t)n + 1);
        unit_desc.resize(db->n_units);
        unit_rows.resize(db->n_unit_rows);
        unit_vars.resize(db->n_unit_vars);
        FX_HIP(hipMemcpyAsync(sys_unit_off.data(), d.sys_unit_off, sys_unit_off.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (db->n_units) FX_HIP(hipMemcpyAsync(unit_desc.data(), d.unit_desc, unit_desc.size() * sizeof(fx::UnitDesc), hipMemcpyDeviceToHost, ctx->stream));
        if (db->n_unit_rows) FX_HIP(hipMemcpyAsync(unit_rows.data(), d.unit_rows, unit_rows.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
        if (db->n_unit_vars) FX_HIP(hipMemcpyAsync(unit_vars.data(), d.unit_vars, unit_vars.size() * 2, hipMemcpyDeviceToHost, ctx->stream));
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));

    // which System's analysis a System uses (itself, System 0 of a uniform batch, or its structure class)
    auto owner = [&](uint32_t s) -> uint32_t {
        if (units) return s;  // blocks are indexed per block
        if (d.uniform) return 0u;
        if (!sys_class.empty()) return sys_class[s];
        return s;
    };
    struct Range {
        std::vector<fx::QrDesc> desc;
        std::vector<uint16_t> u16;
        std::vector<uint64_t> u64;
        std::vector<uint32_t> sys_first;  // per System of the range: its first desc (relative), or UINT32_MAX when it has none
        uint32_t s_lo = 0, max_m = 0, max_h = 0;
        bool failed = false;
    } ranges[MAX_RANGES];
    uint32_t n_ranges = 1;
    parallel_ranges(n, (uint64_t)d.n_exprs * 64, [&](uint32_t t, uint32_t s_lo, uint32_t s_hi) {
        Range& R = ranges[t];
        R.s_lo = s_lo;
        R.sys_first.assign(s_hi - s_lo, 0xFFFFFFFFu);
        std::vector<uint32_t> rows, free_;
        QrHostPlan hp;
        auto emit = [&](const uint32_t* rw, uint32_t m, const uint32_t* fr, uint32_t nf, uint32_t s, bool skip) {
            const uint32_t e0 = expr_off[s], nvt = var_off[s + 1] - var_off[s];
            fx::QrDesc qd{};
            qd.u16_off = (uint32_t)R.u16.size();
            qd.u64_off = (uint32_t)R.u64.size();
            qd.n = (uint16_t)nf;
            qd.m = (uint16_t)m;
            if (!skip && build_qr_plan(expr_tag.data() + e0, expr_idx.data() + 4 * (size_t)e0, rw, m, fr, nf, nvt, hp)) {
                qd.nnzh = (uint16_t)hp.nnzh;
                qd.ok = 1;
                R.u16.insert(R.u16.end(), hp.u16.begin(), hp.u16.end());
                R.u64.insert(R.u64.end(), hp.u64.begin(), hp.u64.end());
                R.max_m = std::max(R.max_m, m + nf);
                R.max_h = std::max(R.max_h, hp.nnzh);
            } else if (!skip) {
                R.failed = true;
            }
            R.desc.push_back(qd);
        };
        for (uint32_t s = s_lo; s < s_hi; ++s) {
            if (sys_large[s] || owner(s) != s) continue;
            R.sys_first[s - s_lo] = (uint32_t)R.desc.size();
            const uint32_t v0 = var_off[s], nvt = var_off[s + 1] - v0, e0 = expr_off[s], net = expr_off[s + 1] - e0;
            if (units) {
                for (uint32_t u = sys_unit_off[s]; u < sys_unit_off[s + 1]; ++u) {
                    const fx::UnitDesc& ud = unit_desc[u];
                    rows.assign(unit_rows.begin() + ud.row_off, unit_rows.begin() + ud.row_off + ud.nrows);
                    free_.clear();
                    for (uint32_t k = 0; k < ud.nvars; ++k) free_.push_back(unit_vars[ud.var_off + k]);
                    emit(rows.data(), ud.nrows, free_.data(), ud.nvars, s, (ud.flags & fx::UNIT_EMPTY) != 0);
                }
            } else {
                for (uint32_t c = 0; c < sys_ncomp[s]; ++c) {
                    rows.clear();
                    free_.clear();
                    bool any_var = false;  // a component without variables is skipped by the kernel (and by the reference)
                    for (uint32_t i = 0; i < nvt; ++i) {
                        const uint16_t info = var_info[v0 + i];
                        any_var = any_var || (info & fx::VAR_COMP_MASK) == c;
                        if ((info & fx::VAR_COMP_MASK) == c && !(info & fx::VAR_FIXED_BIT)) free_.push_back(i);
                    }
                    for (uint32_t i = 0; i < net; ++i)
                        if (expr_comp[e0 + i] == c) rows.push_back(i);
                    emit(rows.data(), (uint32_t)rows.size(), free_.data(), (uint32_t)free_.size(), s, !any_var);
                }
            }
        }
    }, &n_ranges);
    std::vector<fx::QrDesc> desc;
    std::vector<uint16_t> u16;
    std::vector<uint64_t> u64;
    std::vector<uint32_t> sys_first(n, 0);
    uint32_t max_m = 0, max_h = 0;
    for (uint32_t t = 0; t < n_ranges; ++t) {
        Range& R = ranges[t];
        if (R.failed) return fail(FX_ERR_UNSUPPORTED, "FX_STEP_QR: the symbolic analysis of a component failed (more than 64 columns or a malformed pattern)");
        const uint32_t dbase = (uint32_t)desc.size(), b16 = (uint32_t)u16.size(), b64 = (uint32_t)u64.size();
        for (fx::QrDesc qd : R.desc) {
            qd.u16_off += b16;
            qd.u64_off += b64;
            desc.push_back(qd);
        }
        u16.insert(u16.end(), R.u16.begin(), R.u16.end());
        u64.insert(u64.end(), R.u64.begin(), R.u64.end());
        for (size_t k = 0; k < R.sys_first.size(); ++k)
            if (R.sys_first[k] != 0xFFFFFFFFu) sys_first[R.s_lo + k] = dbase + R.sys_first[k];
        max_m = std::max(max_m, R.max_m);
        max_h = std::max(max_h, R.max_h);
    }
    std::vector<uint32_t> index;
    if (units) {  // one desc per block, in block order
        index.assign(db->n_units, 0);
        for (uint32_t s = 0; s < n; ++s) {
            if (sys_large[s]) continue;
            for (uint32_t u = sys_unit_off[s]; u < sys_unit_off[s + 1]; ++u) index[u] = sys_first[s] + (u - sys_unit_off[s]);
        }
    } else {
        index.assign(n, 0);
        for (uint32_t s = 0; s < n; ++s) index[s] = sys_first[owner(s)];
    }
    fx::QrPlans q;
    {  // the kernel keeps a component's augmented matrix in LDS: does the largest one fit? Asked before anything is uploaded
       // (a batch that fails here fails again on every later FX_STEP_QR solve — nothing may pile up on the device)
        Q.max_m = max_m;
        Q.max_h = max_h;
        const size_t need = fx::solve_lds_bytes_qr(d, units);
        if (need > 160u * 1024u) {
            Q.max_m = Q.max_h = 0;
            return fail(FX_ERR_TOO_LARGE, "FX_STEP_QR keeps the (expressions + free variables) x (free variables + 1) matrix of a component in LDS: %zu bytes needed (limit 163840)",
                        need);
        }
    }
    // a batch of ONE structure with a single component of at most 32 columns: the program of the grouped build as well
    QrgHostProgram prog;
    if (!units && d.uniform && d.u_ncomp == 1u && n >= 1 && !sys_large[0] && sys_ncomp[0] == 1u) {
        const uint32_t nvt = var_off[1] - var_off[0], net = expr_off[1] - expr_off[0];
        std::vector<uint32_t> rows, free_;
        for (uint32_t i = 0; i < nvt; ++i)
            if ((var_info[i] & fx::VAR_COMP_MASK) == 0 && !(var_info[i] & fx::VAR_FIXED_BIT)) free_.push_back(i);
        for (uint32_t i = 0; i < net; ++i)
            if (expr_comp[i] == 0) rows.push_back(i);
        (void)build_qrg_program(expr_tag.data(), expr_idx.data(), rows.data(), (uint32_t)rows.size(), free_.data(), (uint32_t)free_.size(), nvt, prog);
    }
    // Systems beyond one wavefront whose components all fit the wide kernel (at most 128 columns, 256 rows, 512 variables):
    // a program per component for its FX_STEP_QR build (fx_wide.hip). Systems of one structure share theirs. All or nothing
    // for the Systems the analysis marked for the wide kernel (they have no other reference-numerics home), and within a
    // budget of program words — beyond it, as for everything larger, FX_STEP_QR stays FX_STEP_CHOLESKY_REFINED.
    std::vector<uint32_t> w_list, w_comp_off(1, 0), w_prog_off, w_words;
    uint32_t w_nx = 0, w_free = 0, w_vars = 0, w_rows = 0;
    if (!units) {  // (qr_none's list is built here, once: the marks belong to it — a SinglePass solve in between leaves them alone)
        db->h_qr_wide.assign(n, 0);
        bool all_marked = true;
        // per owner System (the first of its structure): the first entry of its row in w_prog_off, NOT_YET before it was
        // planned, NO_FIT when its planning failed — every member of a structure that does not fit is then refused at once
        // instead of repeating the symbolic analysis
        constexpr uint32_t NOT_YET = 0xFFFFFFFFu, NO_FIT = 0xFFFFFFFEu;
        std::vector<uint32_t> first_of(n, NOT_YET);
        constexpr size_t WORD_BUDGET = size_t(48) << 20;       // 192 MB of tables per batch
        for (uint32_t s = 0; s < n && w_words.size() <= WORD_BUDGET && all_marked; ++s) {
            if (!sys_large[s]) continue;
            const uint32_t v0 = var_off[s], nvt = var_off[s + 1] - v0, e0 = expr_off[s], net = expr_off[s + 1] - e0;
            bool fits = nvt <= 512u && sys_ncomp[s] >= 1u;
            const uint32_t own = owner(s);
            uint32_t first = NOT_YET;
            if (fits && own != s) {
                first = first_of[own];
                if (first == NO_FIT) fits = false;
            }
            std::vector<uint32_t> offs;
            if (fits && first == NOT_YET) {
                std::vector<uint32_t> rows, free_;
                for (uint32_t c = 0; c < sys_ncomp[s] && fits; ++c) {
                    rows.clear();
                    free_.clear();
                    bool any_var = false;
                    for (uint32_t i = 0; i < nvt; ++i) {
                        const uint16_t info = var_info[v0 + i];
                        any_var = any_var || (info & fx::VAR_COMP_MASK) == c;
                        if ((info & fx::VAR_COMP_MASK) == c && !(info & fx::VAR_FIXED_BIT)) free_.push_back(i);
                    }
                    for (uint32_t i = 0; i < net; ++i)
                        if (expr_comp[e0 + i] == c) rows.push_back(i);
                    if (!any_var) {  // skipped by the kernel, as by the reference
                        offs.push_back(0xFFFFFFFFu);
                        continue;
                    }
                    QrgHostProgram wp;
                    if (free_.empty() || rows.empty() ||
                        !build_qrg_program(expr_tag.data() + e0, expr_idx.data() + 4 * (size_t)e0, rows.data(), (uint32_t)rows.size(), free_.data(),
                                           (uint32_t)free_.size(), nvt, wp, /*wide=*/true)) {
                        fits = false;
                        break;
                    }
                    offs.push_back((uint32_t)w_words.size());
                    w_words.insert(w_words.end(), wp.words.begin(), wp.words.end());
                    w_nx = std::max(w_nx, wp.nx);
                    w_free = std::max(w_free, wp.n);
                    w_rows = std::max(w_rows, wp.m);
                }
                if (fits) {
                    first = (uint32_t)w_prog_off.size();
                    w_prog_off.insert(w_prog_off.end(), offs.begin(), offs.end());
                }
                first_of[s] = fits ? first : NO_FIT;
                if (own != s && first_of[own] == NOT_YET) first_of[own] = first_of[s];
            }
            if (fits && first != NOT_YET) {
                if (own != s) {  // a System of a structure already planned: its own row of offsets (the same values)
                    const uint32_t nc = sys_ncomp[s];
                    const uint32_t at = (uint32_t)w_prog_off.size();
                    for (uint32_t c = 0; c < nc; ++c) {
                        const uint32_t po = w_prog_off[first + c];
                        w_prog_off.push_back(po);
                    }
                    first = at;
                }
                w_list.push_back(s);
                w_comp_off.push_back(first);
                w_vars = std::max(w_vars, nvt);
                db->h_qr_wide[s] = 1;
            } else if (sys_large[s] == 2) {
                all_marked = false;
            }
        }
        if (!all_marked || w_words.size() > WORD_BUDGET ||
            (!w_list.empty() && fx::wide_qr_lds_bytes(w_free, w_vars, w_rows, w_nx) > 160u * 1024u)) {
            w_list.clear();
            db->h_qr_wide.assign(n, 0);
        }
    }
    unsigned long long* d64 = nullptr;
    int rc = dev_alloc_copy(ctx, db, &q.u16, u16.data(), u16.size());
    if (!rc && !w_list.empty()) {
        // (w_comp_off holds, per listed System, the first entry of ITS components in w_prog_off)
        w_comp_off.erase(w_comp_off.begin());
        rc = dev_alloc_copy(ctx, db, &q.qrw_list, w_list.data(), w_list.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &q.qrw_comp_first, w_comp_off.data(), w_comp_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &q.qrw_prog_off, w_prog_off.data(), w_prog_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &q.qrw_words, w_words.data(), w_words.size());
        q.n_qrw = (uint32_t)w_list.size();
        q.qrw_nx = w_nx;
        q.qrw_free = w_free;
        q.qrw_vars = w_vars;
        q.qrw_rows = w_rows;
    }
    // a batch of SEVERAL structures: the same program per big structure class (the classes of the one-structure Cholesky build,
    // upload_planned) — a launch of the grouped build per class over its member list (fx_solve.cpp: launch_class_qr)
    if (!rc && !units && !d.uniform && !db->classes.empty() && db->qr_class.empty() && !sys_class.empty() && db->class_first.size() == db->classes.size()) {
        std::vector<fx_dbatch::QrClassProg> cls(db->classes.size());
        std::vector<uint8_t> skip(sys_large.begin(), sys_large.end());
        bool any = false;
        for (size_t k = 0; k < db->classes.size() && !rc; ++k) {
            const uint32_t s = db->class_first[k];
            if (s >= n || sys_large[s] || sys_ncomp[s] != 1u) continue;
            const uint32_t v0 = var_off[s], nvt = var_off[s + 1] - v0, e0 = expr_off[s], net = expr_off[s + 1] - e0;
            std::vector<uint32_t> rows, free_;
            for (uint32_t i = 0; i < nvt; ++i)
                if ((var_info[v0 + i] & fx::VAR_COMP_MASK) == 0 && !(var_info[v0 + i] & fx::VAR_FIXED_BIT)) free_.push_back(i);
            for (uint32_t i = 0; i < net; ++i)
                if (expr_comp[e0 + i] == 0) rows.push_back(i);
            QrgHostProgram cp;
            if (!build_qrg_program(expr_tag.data() + e0, expr_idx.data() + 4 * (size_t)e0, rows.data(), (uint32_t)rows.size(), free_.data(), (uint32_t)free_.size(), nvt,
                                   cp) ||
                !cp.ok)
                continue;
            rc = dev_alloc_copy(ctx, db, &cls[k].prog, cp.words.data(), cp.words.size());
            cls[k].words = (uint32_t)cp.words.size();
            cls[k].small_words = cp.words[14];
            cls[k].ng = cp.ng;
            cls[k].nx = cp.nx;
            cls[k].n = cp.n;
            cls[k].m = cp.m;
            any = true;
        }
        if (!rc && any) {
            std::vector<uint8_t> has(n, 0);  // (a class is named by its first System)
            for (size_t k = 0; k < cls.size(); ++k)
                if (cls[k].prog) has[db->class_first[k]] = 1;
            for (uint32_t s = 0; s < n; ++s)
                if (!skip[s] && has[sys_class[s]]) skip[s] = 4;
            rc = dev_alloc_copy(ctx, db, &db->qr_skip, skip.data(), skip.size());
            if (!rc) db->qr_class = std::move(cls);
        }
    }
    if (!rc && prog.ok) {
        rc = dev_alloc_copy(ctx, db, &q.qrg, prog.words.data(), prog.words.size());
        q.qrg_words = (uint32_t)prog.words.size();
        q.qrg_small = prog.words[14];
        q.qrg_ng = prog.ng;
        q.qrg_nx = prog.nx;
        q.qrg_n = prog.n;
        q.qrg_m = prog.m;
    }
    if (!rc) rc = dev_alloc_copy(ctx, db, &d64, reinterpret_cast<const unsigned long long*>(u64.data()), u64.size());
    if (!rc) rc = dev_alloc_copy(ctx, db, &q.index, index.data(), index.size());
    fx::QrDesc* ddesc = nullptr;
    if (!rc) rc = dev_alloc_copy(ctx, db, &ddesc, desc.data(), desc.size());
    if (rc) return rc;
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the host vectors die with this frame
    q.u64 = d64;
    q.max_m = max_m;
    q.max_h = max_h;
    q.desc = ddesc;  // set last: marks the plans as complete
    Q = q;
    return FX_OK;
}

// Decomposer::None on a System too large for LDS but made of components that each fit one wavefront (a
// sketch of many separate features): the GLOBAL block walker takes the components as its blocks — rows
// and free variables ascending, snapshot restore after each (quirk Q2) — instead of the host-driven
// sparse path going through them one by one.
int ensure_component_walk(fx_ctx* ctx, fx_dbatch* db) {
    if (db->comp_walk_built) return FX_OK;
    const fx_batch& hb = db->h_batch;
    const uint32_t n = db->d.n_systems;
    std::vector<uint32_t> unit_off((size_t)n + 1, 0), unit_rows, g_list, g_off;
    std::vector<uint16_t> unit_vars;
    std::vector<fx::UnitDesc> desc;
    db->h_comp_walk.assign(n, 0);
    uint32_t g_total = 0, mf = 0, mr = 0, mp = 0, me = 0;
    for (uint32_t s = 0; s < n; ++s) {
        unit_off[s] = (uint32_t)desc.size();
        if (db->h_sys_large.empty() || db->h_sys_large[s] != 1) continue;
        const uint32_t v0 = hb.var_off[s], nvt = hb.var_off[s + 1] - v0;
        const uint32_t e0 = hb.expr_off[s], net = hb.expr_off[s + 1] - e0;
        uint32_t ncomp = 0;
        for (uint32_t i = 0; i < nvt; ++i) {
            uint16_t c = hb.var_comp ? hb.var_comp[v0 + i] : 0;
            if (c != FX_NO_COMPONENT) ncomp = std::max<uint32_t>(ncomp, c + 1u);
        }
        std::vector<std::vector<uint32_t>> rows(ncomp), fvars(ncomp);
        std::vector<uint8_t> has_var(ncomp, 0);
        std::vector<uint32_t> pairs(ncomp, 0), ents(ncomp, 0);
        for (uint32_t i = 0; i < nvt; ++i) {
            uint16_t c = hb.var_comp ? hb.var_comp[v0 + i] : 0;
            if (c == FX_NO_COMPONENT) continue;
            has_var[c] = 1;
            if (!hb.var_fixed[v0 + i]) fvars[c].push_back(i);
        }
        for (uint32_t i = 0; i < net; ++i) {
            uint16_t c = hb.expr_comp ? hb.expr_comp[e0 + i] : 0;
            if (c == FX_NO_COMPONENT || c >= ncomp) continue;
            rows[c].push_back(i);
            uint32_t k = (uint32_t)fx::tag_nvars<true>((int)hb.expr_tag[e0 + i]);
            pairs[c] += k * k;
            ents[c] += k;
        }
        bool fits = nvt <= 0xFFFFu;
        for (uint32_t c = 0; c < ncomp && fits; ++c) fits = fvars[c].size() <= FX_MAX_FREE_VARS && rows[c].size() <= FX_MAX_ROWS;
        if (!fits) continue;
        for (uint32_t c = 0; c < ncomp; ++c) {
            if (!has_var[c]) continue;  // skipped by the reference
            fx::UnitDesc ud{};
            ud.row_off = (uint32_t)unit_rows.size();
            ud.var_off = (uint32_t)unit_vars.size();
            ud.nrows = (uint16_t)rows[c].size();
            ud.nvars = (uint16_t)fvars[c].size();
            ud.comp = (uint16_t)c;
            ud.flags = (uint16_t)(fx::UNIT_FIRST | fx::UNIT_RESTORE);
            unit_rows.insert(unit_rows.end(), rows[c].begin(), rows[c].end());
            for (uint32_t v : fvars[c]) unit_vars.push_back((uint16_t)v);
            desc.push_back(ud);
            mf = std::max<uint32_t>(mf, ud.nvars);
            mr = std::max<uint32_t>(mr, ud.nrows);
            mp = std::max(mp, pairs[c]);
            me = std::max(me, ents[c]);
        }
        db->h_comp_walk[s] = 1;
        g_list.push_back(s);
        g_off.push_back(g_total);
        g_total += nvt;
    }
    unit_off[n] = (uint32_t)desc.size();
    fx::DeviceBatch w = db->d;
    w.n_g = 0;
    if (!g_list.empty()) {
        int rc = dev_alloc_copy(ctx, db, &w.unit_desc, desc.data(), desc.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.unit_rows, unit_rows.data(), unit_rows.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.unit_vars, unit_vars.data(), unit_vars.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.sys_unit_off, unit_off.data(), unit_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_list, g_list.data(), g_list.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_off, g_off.data(), g_off.size());
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_xs, (const double*)nullptr, 2 * (size_t)g_total);
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_vout, (const double*)nullptr, g_total);
        if (!rc) rc = dev_alloc_copy(ctx, db, &w.g_colof, (const int16_t*)nullptr, g_total);
        if (rc) return rc;
        FX_HIP(hipStreamSynchronize(ctx->stream));
        w.n_g = (uint32_t)g_list.size();
        w.max_unit_free_g = mf;
        w.max_unit_rows_g = mr;
        w.max_pairs_g = mp;
        w.max_ents_g = me;
    }
    db->comp_walk = w;
    db->comp_walk_built = true;
    return FX_OK;
}

// Systems [s0, s1) of an analysed batch onto the device (the whole batch, or one chunk of solve_host_chunked: the limits
// that size kernels and LDS are the whole batch's either way, so a chunk runs the very kernels the whole batch would).
int upload_planned(fx_ctx* ctx, const fx_batch* batch, const HostPlan& p, uint32_t s0, uint32_t s1, fx_dbatch** out, bool one_shot,
                   uint32_t defer) {
    *out = nullptr;
    int rc = FX_OK;
    const bool whole = s0 == 0 && s1 == p.n_systems;
    const uint32_t n_sys = s1 - s0;
    const uint32_t v0 = n_sys ? batch->var_off[s0] : 0, e0 = n_sys ? batch->expr_off[s0] : 0;
    const uint32_t n_vars = n_sys ? batch->var_off[s1] - v0 : 0, n_exprs = n_sys ? batch->expr_off[s1] - e0 : 0;
    fx_dbatch* db = new (std::nothrow) fx_dbatch();
    if (!db) return fail(FX_ERR_NOMEM, "out of host memory");
    db->resident = true;
    BatchHolder hold(ctx, db);  // freed on every early way out, by error code or by exception
    fx::DeviceBatch& d = db->d;
    d.n_systems = n_sys;
    d.n_vars = n_vars;
    d.n_exprs = n_exprs;
    d.nnz = p.nnz;  // (of the whole batch: an upper bound for a chunk, which never builds its CSR)
    d.max_free = p.max_free;
    d.max_rows = p.max_rows;
    d.max_vars = p.max_vars;
    d.max_exprs = p.max_exprs;
    d.max_vars_all = p.max_vars_all;
    d.max_exprs_all = p.max_exprs_all;
    d.max_pairs = p.max_pairs;
    d.max_ents = p.max_ents;
    d.max_pairs_tri = p.max_pairs_tri;
    d.uniform = p.uniform && n_sys >= 2 ? 1u : 0u;
    // (one System::solve on a sketch of at most eight variables and expressions — the reference's own bench at one triangle — is a
    // batch of one structure too: it gets the program of fx_grouped_tiny.hip's build instead of the general kernel's list building)
    if (whole && n_sys == 1u && p.n_large == 0 && p.sys_ncomp[0] == 1u && n_vars <= 8u && n_exprs <= 8u && n_exprs >= 1u && p.max_free >= 1u) d.uniform = 1u;
    d.u_nvars = d.uniform ? batch->var_off[1] : 0;
    d.u_nexprs = d.uniform ? batch->expr_off[1] : 0;
    d.u_ncomp = d.uniform ? p.sys_ncomp[0] : 0;
    d.max_pairs_g = p.max_pairs_large;  // blocks of a large System hold at most its components' products
    d.max_ents_g = p.max_ents_large;
    const uint32_t zero_off[1] = {0};
    const uint32_t* voff = n_sys ? batch->var_off : zero_off;
    const uint32_t* eoff = n_sys ? batch->expr_off : zero_off;
    std::vector<uint32_t> voff_r, eoff_r, class_r;
    if (!whole) {  // a chunk's offsets start at 0; a structure class is the first System OF THE CHUNK with the structure
        voff_r.resize((size_t)n_sys + 1);
        eoff_r.resize((size_t)n_sys + 1);
        for (uint32_t s = 0; s <= n_sys; ++s) {
            voff_r[s] = batch->var_off[s0 + s] - v0;
            eoff_r[s] = batch->expr_off[s0 + s] - e0;
        }
        voff = voff_r.data();
        eoff = eoff_r.data();
        if (!p.sys_class.empty()) {
            class_r.resize(n_sys);
            std::unordered_map<uint32_t, uint32_t> first;
            for (uint32_t s = 0; s < n_sys; ++s) {
                if (s && p.sys_class[s0 + s] == p.sys_class[s0 + s - 1]) {
                    class_r[s] = class_r[s - 1];
                    continue;
                }
                class_r[s] = first.emplace(p.sys_class[s0 + s], s).first->second;
            }
        }
    }
    const uint32_t* sys_class = p.sys_class.empty() ? nullptr : whole ? p.sys_class.data() : class_r.data();
    // Every array of the batch: (device pointer to set, host source or NULL for zeros, bytes) — ONE device block.
    //  - a small batch (one System::solve) goes up with ONE copy from the page-locked staging area: a dozen-and-a-half
    //    separate hipMemcpy calls cost more than its solve;
    //  - up to PINNED_HALF the same, but `vars` (the working copy of vars0) and the zeroed `results` at the block's end
    //    are made on the device (a copy, a memset) instead of crossing the bus;
    //  - a big batch: one copy per array straight from the caller's memory (no extra pass over 100 MB on the host).
    // (`period`: the array repeats with this period — the structure arrays of a batch of one structure —, so a big batch sends
    // its first `period` bytes over the bus and the device fills in the rest: 45 % of a ring16 batch's bytes)
    struct Req { void** dst; const void* src; size_t bytes; size_t period; bool later = false; };
    std::vector<Req> reqs;
#define FX_UP(field, host, count) \
    reqs.push_back({reinterpret_cast<void**>(&d.field), static_cast<const void*>(host), (size_t)(count) * sizeof(*d.field), 0});
#define FX_UP_PERIODIC(field, host, count, per_system) \
    reqs.push_back({reinterpret_cast<void**>(&d.field), static_cast<const void*>(host), (size_t)(count) * sizeof(*d.field), \
                    d.uniform ? (size_t)(per_system) * sizeof(*d.field) : 0});
    FX_UP(var_off, voff, (size_t)n_sys + 1)
    FX_UP(expr_off, eoff, (size_t)n_sys + 1)
    FX_UP(sys_ncomp, p.sys_ncomp.data() + s0, n_sys)
    FX_UP(sys_large, p.sys_large.data() + s0, n_sys)
    FX_UP(vars0, (const double*)batch->vars + v0, n_vars)
    reqs.back().later = (defer & FX_DEFER_VARS) != 0;  // (room only: the solve kernel or fill_deferred brings the values)
    // (a hinted batch — HostPlan::hinted — holds the first System's structure only: such a batch is big, goes up by periods, and
    // the programs below read one System)
    const uint32_t pv0 = p.hinted ? 0u : v0, pe0 = p.hinted ? 0u : e0;
    FX_UP_PERIODIC(var_info, p.var_info.data() + pv0, n_vars, d.u_nvars)
    FX_UP_PERIODIC(expr_tag, p.expr_tagx.data() + pe0, n_exprs, d.u_nexprs)
    FX_UP_PERIODIC(expr_comp, p.expr_comp.data() + pe0, n_exprs, d.u_nexprs)
    FX_UP_PERIODIC(expr_idx, p.expr_idx16.data() + 4 * (size_t)pe0, 4 * (size_t)n_exprs, 4 * (size_t)d.u_nexprs)
    FX_UP(expr_param, batch->expr_param + e0, n_exprs)
    reqs.back().later = (defer & FX_DEFER_PARAMS) != 0;
    FX_UP(work_counter, (const uint32_t*)nullptr, 16)  // (the batch's queue head, then those of up to eight structure classes and of the rest: launch_class_solves)
    if (sys_class) FX_UP(sys_class, sys_class, n_sys)
    // A batch of one structure gets the program of one of the grouped kernel's builds for such batches, when the structure
    // qualifies: the sparse build (fx_grouped_s.hip) for components beyond a register-resident factor — and from 33 free variables
    // on when the factor is sparse (at most a quarter of the dense triangle: the reference's bench sketch of 11 triangles, 46
    // variables, 201 of 1 081 entries: 1.52 ms per 100 000 against 3.14 in the 48-column register build) —, the register build
    // (fx_grouped_c.hip) otherwise.
    GsHostProgram gs;
    GcHostProgram gc;
    bool sparse_build = false;
    if (d.uniform && d.u_ncomp == 1u && d.u_nvars <= 255u && d.u_nexprs <= 255u &&
        build_gs_program(p.var_info.data() + pv0, p.expr_tagx.data() + pe0, p.expr_comp.data() + pe0, p.expr_idx16.data() + 4 * (size_t)pe0, d.u_nvars,
                         d.u_nexprs, gs))
        sparse_build = gs.nfree > 48u || 4u * gs.nl <= gs.nfree * (gs.nfree + 1u) / 2u;
    if (sparse_build) {
        FX_UP(gs_tab, gs.words.data(), gs.words.size())
        d.gs_words = (uint32_t)gs.words.size();
        d.gs_nl = gs.nl;
        d.gs_ng = gs.ng;
        d.gs_nfree = gs.nfree;
    } else if (d.uniform && d.u_ncomp == 1u && p.n_large == 0 && p.max_free >= 1u && p.max_free <= 48u &&
               build_gc_program(p.var_info.data() + pv0, p.expr_tagx.data() + pe0, p.expr_comp.data() + pe0, p.expr_idx16.data() + 4 * (size_t)pe0,
                                d.u_nvars, d.u_nexprs, p.max_free, gc)) {
        FX_UP(gc_tab, gc.words.data(), gc.words.size())
        d.gc_words = gc.words_f64;
        d.gc_words_all = (uint32_t)gc.words.size();
        d.gc_nslots = gc.nslots;
        d.gc_ng = gc.ng;
        d.gc_nc = gc.nc;
        d.gc_rc = gc.rc;
    }
    // Several structures: the classes with 2 048 members and more (at most eight, the largest first; those of the largest one's
    // build — columns per lane) get a program each
    std::vector<uint32_t> cl_words_h, cl_lists_h;
    if (!d.uniform && sys_class && p.n_large == 0 && p.max_free >= 1u && p.max_free <= 48u) {
        constexpr uint32_t MAX_CLASSES = 8u;
        // a class from 256 members on — when the classes hold three quarters of the batch: a launch over the classes plus a launch
        // of the general build over a large rest is two slow ends instead of one (six structures x 1 500 Systems with one class of
        // 3 000: 1.00 ms, 0.68 with all six as classes; tools/probes/class_min_probe.py)
        static const uint32_t CLASS_MIN = [] { const char* e = std::getenv("FIKSI_AMD_CLASS_MIN"); return e ? (uint32_t)atoi(e) : 256u; }();
        std::vector<uint32_t> count(n_sys, 0u);  // (a class is named by its first System)
        for (uint32_t s = 0; s < n_sys; ++s) count[sys_class[s]] += 1u;
        std::vector<std::pair<uint32_t, uint32_t>> big;  // (members, first System)
        for (uint32_t f = 0; f < n_sys; ++f)
            if (count[f] >= CLASS_MIN) big.push_back({count[f], f});
        std::sort(big.begin(), big.end(), [](auto& a, auto& b2) { return a.first != b2.first ? a.first > b2.first : a.second < b2.second; });
        std::vector<uint32_t>& class_slot = count;  // first System -> index into db->classes (reusing the array: NO_SLOT elsewhere)
        constexpr uint32_t NO_SLOT = 0xFFFFFFFFu;
        std::fill(class_slot.begin(), class_slot.end(), NO_SLOT);
        for (auto& pr : big) {
            if (db->classes.size() >= MAX_CLASSES) break;
            const uint32_t f = pr.second;
            if (p.sys_ncomp[s0 + f] != 1u) continue;
            const uint32_t vf = v0 + voff[f], ef = e0 + eoff[f], nvt = voff[f + 1] - voff[f], net = eoff[f + 1] - eoff[f];
            uint32_t nfree = 0;
            for (uint32_t i = 0; i < nvt; ++i) nfree += (p.var_info[vf + i] & fx::VAR_FIXED_BIT) ? 0u : 1u;
            GcHostProgram cp;
            if (!build_gc_program(p.var_info.data() + vf, p.expr_tagx.data() + ef, p.expr_comp.data() + ef, p.expr_idx16.data() + 4 * (size_t)ef, nvt, net,
                                  nfree, cp))
                continue;
            if (db->classes.empty()) {
                db->cl_nc = cp.nc;
                db->cl_rc = cp.rc;
            }
            if (cp.nc != db->cl_nc || cp.rc != db->cl_rc) continue;
            fx::GcClass cl;
            cl.prog_off = (uint32_t)cl_words_h.size();
            cl.words = cp.words_f64;
            cl.words_all = (uint32_t)cp.words.size();
            cl.list_off = 0;
            cl.count = pr.first;
            db->cl_max_words = std::max(db->cl_max_words, cl.words);
            db->cl_max_words_all = std::max(db->cl_max_words_all, cl.words_all);
            db->cl_max_slots = std::max(db->cl_max_slots, cp.nslots);
            db->cl_max_ng = std::max(db->cl_max_ng, cp.ng);
            cl_words_h.insert(cl_words_h.end(), cp.words.begin(), cp.words.end());
            class_slot[f] = (uint32_t)db->classes.size();
            db->classes.push_back(cl);
            db->class_first.push_back(f);  // (of this batch or chunk)
        }
        {
            uint64_t covered = 0;
            for (const auto& cl : db->classes) covered += cl.count;
            if (4u * covered < 3u * (uint64_t)n_sys) {  // (too much of the batch outside the classes: the general build for all)
                db->classes.clear();
                db->class_first.clear();
                cl_words_h.clear();
                db->cl_max_words = db->cl_max_words_all = db->cl_max_slots = db->cl_max_ng = 0;
            }
        }
        if (!db->classes.empty()) {
            uint32_t at = 0;
            for (auto& cl : db->classes) {
                cl.list_off = at;
                at += cl.count;
            }
            db->cl_systems = at;
            db->rest_off = at;
            cl_lists_h.resize(n_sys);
            std::vector<uint32_t> fill(db->classes.size(), 0);
            uint32_t nrest = 0;
            for (uint32_t s = 0; s < n_sys; ++s) {
                const uint32_t slot = class_slot[sys_class[s]];
                if (slot == NO_SLOT) cl_lists_h[db->rest_off + nrest++] = s;
                else cl_lists_h[db->classes[slot].list_off + fill[slot]++] = s;
            }
            db->rest_count = nrest;
            reqs.push_back({reinterpret_cast<void**>(&db->cl_words), cl_words_h.data(), cl_words_h.size() * 4, 0});
            reqs.push_back({reinterpret_cast<void**>(&db->cl_lists), cl_lists_h.data(), cl_lists_h.size() * 4, 0});
            reqs.push_back({reinterpret_cast<void**>(&db->cl_desc), db->classes.data(), db->classes.size() * sizeof(fx::GcClass), 0});
        }
    }
    FX_UP(w_list, p.wide_list.data(), whole ? p.wide_list.size() : 0)
    const size_t n_front = reqs.size();  // the two below end the block, side by side: a one-shot solve reads them back in one copy
    FX_UP(vars, (const double*)batch->vars + v0, n_vars)
    reqs.back().later = (defer & FX_DEFER_VARS) != 0;
    FX_UP(results, (const fx_result*)nullptr, n_sys)
#undef FX_UP
#undef FX_UP_PERIODIC
    auto room_of = [](const Req& r) { return (std::max<size_t>(r.bytes, 1) + 255u) & ~size_t(255); };
    size_t packed = 0, front = 0;
    for (size_t i = 0; i < reqs.size(); ++i) {
        packed += room_of(reqs[i]);
        if (i + 1 == n_front) front = packed;
    }
    hipError_t e1 = hipSuccess;
    // A one-shot solve of a few small Systems (System::solve on one sketch) makes no copy call: the host writes the block's
    // image into the context's host-coherent region, one small kernel pulls it over in a burst and another pushes `vars` and
    // `results` back when the solve is done (2 us each; a solve kernel working on the region in place pays a PCIe round trip
    // per dependent load, and waits for its stores: 40 / 30 us instead of 10 — both measured).
    const bool zero_copy = one_shot && whole && !defer && packed <= fx_ctx::ZC_BYTES && p.n_large == 0 && p.wide_list.empty() && ctx->ensure_zc();
    db->allocations.reserve(db->allocations.size() + 1);
    unsigned char* base = static_cast<unsigned char*>(ctx->take(packed, e1));
    if (!base) {
        return fail(e1 == hipErrorOutOfMemory ? FX_ERR_NOMEM : FX_ERR_HIP, "hipMalloc(%zu): %s", packed, hipGetErrorString(e1));
    }
    db->allocations.push_back({base, packed});
    db->zero_copy = zero_copy;
    db->zc_image = zero_copy ? ctx->zc : nullptr;
    {
        size_t at = 0;
        for (size_t i = 0; i < reqs.size(); ++i) {
            *reqs[i].dst = base + at;
            at += room_of(reqs[i]);
        }
    }
    const bool on_device_tail = !zero_copy && packed > (size_t(256) << 10);  // vars and results made on the device
    auto upload = [&]() -> int {
        if (zero_copy) {
            db->packed_base = base;
            db->packed_bytes = packed;
            unsigned char* img = ctx->zc;
            size_t at = 0;
            for (const Req& r : reqs) {
                const size_t room = room_of(r);
                if (r.src && r.bytes) {
                    if (p.hinted && r.period && r.period < r.bytes) {
                        for (size_t o2 = 0; o2 < r.bytes; o2 += r.period) memcpy(img + at + o2, r.src, std::min(r.period, r.bytes - o2));
                    } else {
                        memcpy(img + at, r.src, r.bytes);
                    }
                    memset(img + at + r.bytes, 0, room - r.bytes);
                } else {
                    memset(img + at, 0, room);
                }
                at += room;
            }
            FX_HIP(fx::launch_pull(base, img, packed, ctx->stream));
            db->zc_front = front;
            db->upload_pending = true;  // (the image outlives this frame: nothing to wait for below)
            return FX_OK;
        }
        if (packed <= fx_ctx::PINNED_HALF) {
            db->packed_base = base;
            db->packed_bytes = packed;
            const size_t staged = on_device_tail ? front : packed;
            std::vector<unsigned char> stage;
            unsigned char* st = nullptr;
            if (ctx->ensure_pinned()) {
                ctx->wait_pinned();  // an earlier upload may still be reading the staging area
                st = ctx->pinned;
                db->upload_pending = true;
            } else {
                stage.resize(staged);
                st = stage.data();
            }
            size_t at = 0;
            for (size_t i = 0; i < reqs.size() && at < staged; ++i) {
                const Req& r = reqs[i];
                const size_t room = room_of(r);
                if (r.src && r.bytes && !r.later) {
                    if (p.hinted && r.period && r.period < r.bytes) {
                        for (size_t o2 = 0; o2 < r.bytes; o2 += r.period) memcpy(st + at + o2, r.src, std::min(r.period, r.bytes - o2));
                    } else {
                        memcpy(st + at, r.src, r.bytes);
                    }
                    memset(st + at + r.bytes, 0, room - r.bytes);
                } else {
                    memset(st + at, 0, room);
                }
                at += room;
            }
            FX_HIP(hipMemcpyAsync(base, st, staged, hipMemcpyHostToDevice, ctx->stream));
            if (db->upload_pending) {
                (void)hipEventRecord(ctx->ev_pinned, ctx->stream);
                ctx->pinned_stream = ctx->stream;
                ctx->pinned_busy = true;
            } else {
                FX_HIP(hipStreamSynchronize(ctx->stream));  // `stage` goes away with this frame
            }
        } else {
            for (size_t i = 0; i < n_front; ++i) {
                const Req& r = reqs[i];
                if (r.later) continue;
                if (r.src && r.bytes && r.period && r.period < r.bytes) {
                    FX_HIP(hipMemcpyAsync(*r.dst, r.src, r.period, hipMemcpyHostToDevice, ctx->stream));
                    FX_HIP(fx::launch_replicate(*r.dst, r.period, r.bytes, ctx->stream));
                } else if (r.src && r.bytes) {
                    FX_HIP(hipMemcpyAsync(*r.dst, r.src, r.bytes, hipMemcpyHostToDevice, ctx->stream));
                } else {
                    FX_HIP(hipMemsetAsync(*r.dst, 0, room_of(r), ctx->stream));
                }
            }
        }
        if (on_device_tail) {
            if (n_vars && !(defer & FX_DEFER_VARS))
                FX_HIP(hipMemcpyAsync(d.vars, d.vars0, (size_t)n_vars * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            FX_HIP(hipMemsetAsync(d.results, 0, room_of(reqs.back()), ctx->stream));
        }
        return FX_OK;
    };
    rc = upload();
    if (rc) {
        return rc;
    }
    d.n_wide = whole ? (uint32_t)p.wide_list.size() : 0u;
    d.w_max_free = p.w_max_free;
    d.w_max_vars = p.w_max_vars;
    d.w_max_rows = p.w_max_rows;
    // jrow_ptr / jcol / jslot / jvals (the CSR Jacobian) and resid are only needed by the standalone
    // evaluation entry points: they are built on first use (ensure_csr / ensure_resid), so a plain
    // solve neither computes nor uploads them.
#undef FX_UP
    // the host plan lives on this stack frame: finish the copies before returning (a small batch went through the
    // context's page-locked staging area, which outlives the call: its copy is left in flight)
    if (!db->upload_pending) {
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) {
            return fail(FX_ERR_HIP, "upload failed: %s", hipGetErrorString(e));
        }
    }
    if (p.n_large && !whole) {
        return fail(FX_ERR_INVALID, "internal: a batch with Systems beyond one wavefront is not cut into chunks");
    }
    if (p.n_large) {
        const uint32_t nv = p.n_vars, ne = p.n_exprs, n = p.n_systems;
        db->n_large = p.n_large;
        db->h_sys_large.assign(p.sys_large.begin(), p.sys_large.end());
        db->h_var_off.assign(batch->var_off, batch->var_off + n + 1);
        db->h_expr_off.assign(batch->expr_off, batch->expr_off + n + 1);
        db->h_vars.assign(batch->vars, batch->vars + nv);
        db->h_var_fixed.assign(batch->var_fixed, batch->var_fixed + nv);
        db->h_expr_tag.assign(batch->expr_tag, batch->expr_tag + ne);
        db->h_expr_idx.assign(batch->expr_idx, batch->expr_idx + 4 * (size_t)ne);
        db->h_expr_param.assign(batch->expr_param, batch->expr_param + ne);
        if (batch->var_comp) db->h_var_comp.assign(batch->var_comp, batch->var_comp + nv);
        if (batch->expr_comp) db->h_expr_comp.assign(batch->expr_comp, batch->expr_comp + ne);
        fx_batch& hb = db->h_batch;
        hb.n_systems = n;
        hb.var_off = db->h_var_off.data();
        hb.expr_off = db->h_expr_off.data();
        hb.vars = db->h_vars.data();
        hb.var_fixed = db->h_var_fixed.data();
        hb.expr_tag = db->h_expr_tag.data();
        hb.expr_idx = db->h_expr_idx.data();
        hb.expr_param = db->h_expr_param.data();
        hb.var_comp = batch->var_comp ? db->h_var_comp.data() : nullptr;
        hb.expr_comp = batch->expr_comp ? db->h_expr_comp.data() : nullptr;
    }
    if (fx::solve_lds_bytes(d) > 160u * 1024u) {
        return fail(FX_ERR_TOO_LARGE, "batch needs %zu bytes of LDS per wavefront (limit 163840)", fx::solve_lds_bytes(d));
    }
    *out = hold.release();
    return FX_OK;
}

// What upload_planned left out on request (`defer`): the values, by ordinary copies — the solve did not take the route that reads
// them in place after all.
int fill_deferred(fx_ctx* ctx, fx_dbatch* db, const fx_batch* batch, uint32_t defer) {
    fx::DeviceBatch& d = db->d;
    if ((defer & FX_DEFER_VARS) && d.n_vars) {
        FX_HIP(hipMemcpyAsync(d.vars0, batch->vars, (size_t)d.n_vars * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        FX_HIP(hipMemcpyAsync(d.vars, d.vars0, (size_t)d.n_vars * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    }
    if ((defer & FX_DEFER_PARAMS) && d.n_exprs)
        FX_HIP(hipMemcpyAsync(d.expr_param, batch->expr_param, (size_t)d.n_exprs * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    FX_HIP(hipStreamSynchronize(ctx->stream));  // (pageable sources: the caller may not outlive an asynchronous copy's staging)
    ctx->stream_synced();
    return FX_OK;
}

void free_batch(fx_ctx* ctx, fx_dbatch* db, bool stream_idle) {
    if (!db) return;
    if (ctx && !stream_idle) {
        (void)hipSetDevice(ctx->device);
        (void)hipStreamSynchronize(ctx->stream);
        ctx->stream_synced();
    }
    for (auto& kv : db->sparse_plans) fx::sparse_cache_free(kv.second.plan);
    for (auto& blk : db->allocations) {
        if (ctx) ctx->give_back(blk.p, blk.size);  // the stream is idle: the blocks can be handed out again
        else (void)hipFree(blk.p);
    }
    delete db;
}

// After a one-shot solve: solved variables and results back to the caller, then the batch is freed. A small batch sits in
// one block on the device: both come back in ONE copy through the page-locked staging area and the call waits on the
// stream once.
int read_back_and_free(fx_ctx* ctx, fx_dbatch* db, const fx_batch* batch, fx_result* results, int rc) {
    if (db->zero_copy) {  // vars and results, the end of the block, pushed into the host-coherent image by one small kernel
        if (!rc) {
            hipError_t e = fx::launch_pull(db->zc_image + db->zc_front, db->packed_base + db->zc_front, db->packed_bytes - db->zc_front, ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = fail(FX_ERR_HIP, "solve failed: %s", hipGetErrorString(e));
            else ctx->stream_synced();
        }
        if (!rc && batch->n_systems) {
            const unsigned char* dev0 = db->packed_base;
            memcpy(batch->vars, db->zc_image + (reinterpret_cast<const unsigned char*>(db->d.vars) - dev0), (size_t)db->d.n_vars * sizeof(double));
            if (results)
                memcpy(results, db->zc_image + (reinterpret_cast<const unsigned char*>(db->d.results) - dev0), (size_t)db->d.n_systems * sizeof(fx_result));
        }
        free_batch(ctx, db, /*stream_idle=*/rc == FX_OK);
        return rc;
    }
    if (!rc && batch->n_systems && db->packed_base && ctx->pinned && db->packed_bytes <= fx_ctx::PINNED_HALF) {
        const unsigned char* lo = reinterpret_cast<const unsigned char*>(db->d.vars);
        const unsigned char* hi = reinterpret_cast<const unsigned char*>(db->d.results + db->d.n_systems);
        if (lo >= db->packed_base && hi <= db->packed_base + db->packed_bytes && lo < hi) {
            unsigned char* back = ctx->pinned + fx_ctx::PINNED_HALF;
            auto run = [&]() -> int {
                FX_HIP(hipMemcpyAsync(back, lo, (size_t)(hi - lo), hipMemcpyDeviceToHost, ctx->stream));
                FX_HIP(hipStreamSynchronize(ctx->stream));
                return FX_OK;
            };
            rc = run();
            ctx->stream_synced();
            if (!rc) {
                memcpy(batch->vars, back, (size_t)db->d.n_vars * sizeof(double));
                if (results) memcpy(results, back + (reinterpret_cast<const unsigned char*>(db->d.results) - lo), (size_t)db->d.n_systems * sizeof(fx_result));
            }
            free_batch(ctx, db, /*stream_idle=*/rc == FX_OK);
            return rc;
        }
    }
    if (!rc && batch->n_systems) rc = fx_batch_get_vars(ctx, db, batch->vars);
    if (!rc && results && batch->n_systems) rc = fx_batch_get_results(ctx, db, results);
    fx_batch_free(ctx, db);
    return rc;
}

}  // namespace fxh
