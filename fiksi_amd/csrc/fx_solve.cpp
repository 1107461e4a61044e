// Routing and launches: which kernel a resident batch's solve takes (the grouped kernel and its one-structure builds, one
// System per wavefront, the wide kernel, the block walker, the team kernels by structure group), the scheduled hand-out
// (presort), and the host-buffer calls (one device block per call, big batches in chunks on three streams).
#include "fx_host.h"

namespace fxh {

// launch_solve, with the Systems of a big batch of small Systems handed out longest-first (fx_presort.hip) unless the
// caller chose a schedule (fx_batch_schedule_by_last_solve) or switched it off (fx_ctx_set_presort)
// A batch of several structures whose big classes have programs (upload_planned): ONE launch of the one-structure build over
// all of them — a wavefront works through the queue of its class, then loads the next class's program and helps there
// (fx_grouped_c.hip) —, and the general build over everyone else. Results are each System's own: the bits of the general build.
// Returns false when the batch or the options do not qualify (nothing launched).
// The member lists of a batch's structure classes (and of the rest), each longest-first as a whole batch would be (fx_presort.hip):
// the scout pass once, a ranking per list. *lists: where they are (the unsorted ones when the presort is off or the batch small).
static int presorted_class_lists(fx_ctx* ctx, fx_dbatch* db, uint32_t** lists) {
    fx::DeviceBatch& d = db->d;
    *lists = db->cl_lists;
    if (!ctx->presort || d.n_systems < ctx->presort_min_systems) return FX_OK;
    const uint32_t n = d.n_systems;
    if (!db->ps_keys) {
        db->ps_temp_bytes = fx::presort_temp_bytes(n);
        int r2 = dev_alloc_copy(ctx, db, &db->ps_keys, (const float*)nullptr, 2 * (size_t)n);
        if (!r2) r2 = dev_alloc_copy(ctx, db, &db->ps_ids, (const uint32_t*)nullptr, 2 * (size_t)n);
        if (!r2) r2 = dev_alloc_copy(ctx, db, &db->ps_temp, (const unsigned char*)nullptr, db->ps_temp_bytes);
        if (r2) return r2;
    }
    std::vector<uint32_t> offs, counts;
    for (const fx::GcClass& cl : db->classes) {
        offs.push_back(cl.list_off);
        counts.push_back(cl.count);
    }
    offs.push_back(db->rest_off);
    counts.push_back(db->rest_count);
    hipError_t e0 = fx::launch_presort_lists(d, db->ps_keys, db->cl_lists, offs.data(), counts.data(), (uint32_t)offs.size(), db->ps_ids + n, ctx->stream);
    if (e0 != hipSuccess) return fail(FX_ERR_HIP, "presort launch failed: %s", hipGetErrorString(e0));
    *lists = db->ps_ids + n;
    return FX_OK;
}

static bool launch_class_solves(fx_ctx* ctx, fx_dbatch* db, const fx::LmParams& p, int* rc) {
    fx::DeviceBatch& d = db->d;
    *rc = FX_OK;
    if (db->classes.empty() || d.order || !p.grouped_one_structure || !fx::grouped_applies(d, p)) return false;
    fx::DeviceBatch dc = d;
    dc.gc_tab = db->cl_words;
    dc.gc_words = db->cl_max_words;
    dc.gc_words_all = db->cl_max_words_all;
    dc.gc_nslots = db->cl_max_slots;
    dc.gc_ng = db->cl_max_ng;
    dc.gc_nc = db->cl_nc;
    dc.gc_rc = db->cl_rc;
    dc.gc_classes = db->cl_desc;
    dc.gc_nclasses = (uint32_t)db->classes.size();
    dc.order = db->cl_lists;
    dc.n_systems = db->cl_systems;
    dc.work_counter = d.work_counter + 1;
    if (!fx::grouped_c_applies(dc, p)) return false;  // (f32 beyond the 32-column shape, the stamped build, ...: the general build for all)
    // every list longest-first, as a whole batch would be (fx_presort.hip): the scout pass once, a ranking per list
    uint32_t* lists = db->cl_lists;
    {
        const int r2 = presorted_class_lists(ctx, db, &lists);
        if (r2) {
            *rc = r2;
            return true;
        }
        dc.order = lists;
    }
    hipError_t e = fx::launch_solve_grouped_c(dc, p, ctx->stream);
    if (e == hipSuccess && db->rest_count) {
        fx::DeviceBatch dr = d;
        dr.order = lists + db->rest_off;
        dr.n_systems = db->rest_count;
        dr.work_counter = d.work_counter + 1 + dc.gc_nclasses;
        e = fx::launch_solve_grouped_general(dr, p, ctx->stream);
    }
    if (e != hipSuccess) *rc = fail(FX_ERR_HIP, "class launch failed: %s", hipGetErrorString(e));
    return true;
}

// FX_STEP_QR on a batch of several structures: the grouped QR build (four Systems per wavefront, fx_grouped.hip) once per big
// structure class that has a program (ensure_qr_plans), over the class's member list; then the one-wavefront QR kernel over the
// whole batch, which passes those Systems by (DeviceBatch::sys_large = fx_dbatch::qr_skip). Same operations per System either way.
static bool launch_class_qr(fx_ctx* ctx, fx_dbatch* db, const fx::LmParams& p, int* rc) {
    fx::DeviceBatch& d = db->d;
    *rc = FX_OK;
    static const bool on = [] { const char* e = std::getenv("FIKSI_AMD_QR_CLASSES"); return !e || atoi(e) != 0; }();
    if (!on || p.lm.solver != FX_STEP_QR || db->qr_class.empty() || !db->qr_skip || d.order || d.uniform || !p.grouped_one_structure || p.route_grouped == 0)
        return false;
    bool any = false;
    uint32_t* lists = db->cl_lists;
    if ((*rc = presorted_class_lists(ctx, db, &lists)) != FX_OK) return true;
    for (size_t k = 0; k < db->qr_class.size(); ++k) {
        const fx_dbatch::QrClassProg& c = db->qr_class[k];
        if (!c.prog) continue;
        fx::DeviceBatch dc = d;
        dc.qr_none.qrg = c.prog;
        dc.qr_none.qrg_words = c.words;
        dc.qr_none.qrg_small = c.small_words;
        dc.qr_none.qrg_ng = c.ng;
        dc.qr_none.qrg_nx = c.nx;
        dc.qr_none.qrg_n = c.n;
        dc.qr_none.qrg_m = c.m;
        dc.order = lists + db->classes[k].list_off;
        dc.n_systems = db->classes[k].count;
        dc.work_counter = d.work_counter + 1 + (uint32_t)k;
        if (!fx::grouped_qr_class_applies(dc, p)) {
            if (any) {
                *rc = fail(FX_ERR_INTERNAL, "FX_STEP_QR: a structure class with a program does not qualify for the grouped build");
                return true;
            }
            return false;  // (the first class decides for all: the layout depends on the batch's maxima only)
        }
        // (one after the other on the context's stream: side by side on two streams the two persistent kernels halve each other's
        // share of the chip from the start — 25.7 against 22.1 ms on 100 000 ring16 sketches of two structures)
        hipError_t e = fx::launch_grouped_qr_class(dc, p, ctx->stream);
        if (e != hipSuccess) {
            *rc = fail(FX_ERR_HIP, "grouped QR launch over a structure class failed: %s", hipGetErrorString(e));
            return true;
        }
        any = true;
    }
    if (!any) return false;
    fx::DeviceBatch dq = d;
    dq.sys_large = db->qr_skip;
    hipError_t e = fx::launch_solve(dq, p, ctx->stream);
    if (e != hipSuccess) *rc = fail(FX_ERR_HIP, "solve kernel launch failed: %s", hipGetErrorString(e));
    return true;
}

int launch_solve_scheduled(fx_ctx* ctx, fx_dbatch* db, const fx::LmParams& p) {
    fx::DeviceBatch& d = db->d;
    // (every System beyond one wavefront — one System::solve on a large sketch is such a batch: the one-wavefront kernels would be
    // launched to find nothing of theirs; solve_beyond_one_wavefront has them all)
    // (SinglePass: the launch below also starts the block walker of large Systems whose blocks fit one wavefront)
    if (db->n_large && db->n_large == d.n_systems && !(p.mode & fx::MODE_UNITS)) return FX_OK;
    {
        int rc = FX_OK;
        if (launch_class_solves(ctx, db, p, &rc)) return rc;
        if (launch_class_qr(ctx, db, p, &rc)) return rc;
    }
    // (a batch solved in place on the caller's arrays has no start values on the device yet for the scout pass to rank by)
    // (... and the tiny build has no queue to hand Systems out from: eight consecutive Systems per wavefront)
    const bool tiny = d.uniform && fx::grouped_applies(d, p) && fx::grouped_c_applies(d, p) && fx::grouped_tiny_applies(d, p);
    if (ctx->presort && !db->in_place && !tiny && !d.order && !p.prof && d.n_systems >= ctx->presort_min_systems && fx::grouped_applies(d, p)) {
        const uint32_t n = d.n_systems;
        if (!db->ps_keys) {
            db->ps_temp_bytes = fx::presort_temp_bytes(n);
            int rc = dev_alloc_copy(ctx, db, &db->ps_keys, (const float*)nullptr, 2 * (size_t)n);
            if (!rc) rc = dev_alloc_copy(ctx, db, &db->ps_ids, (const uint32_t*)nullptr, 2 * (size_t)n);
            if (!rc) rc = dev_alloc_copy(ctx, db, &db->ps_temp, (const unsigned char*)nullptr, db->ps_temp_bytes);
            if (rc) return rc;
        }
        FX_HIP(fx::launch_presort(d, db->ps_keys, db->ps_ids, db->ps_temp, db->ps_temp_bytes, ctx->stream));
        fx::DeviceBatch dd = d;
        dd.order = db->ps_ids + n;
        FX_HIP(fx::launch_solve(dd, p, ctx->stream));
        return FX_OK;
    }
    if (tiny && !d.order && d.n_systems >= 64u) {  // the tiny build with a place for the stragglers it hands over (fx_grouped_tiny.hip)
        if (!db->tiny_left) {
            int rc = dev_alloc_copy(ctx, db, &db->tiny_left, (const uint32_t*)nullptr, (size_t)d.n_systems);
            if (rc) return rc;
        }
        fx::DeviceBatch dd = d;
        dd.order = db->tiny_left;
        dd.queue_len = d.work_counter + 15;  // (the last of the sixteen counters: the queue heads sit at the front)
        FX_HIP(fx::launch_solve(dd, p, ctx->stream));
        return FX_OK;
    }
    FX_HIP(fx::launch_solve(d, p, ctx->stream));
    return FX_OK;
}


// Systems beyond the one-wavefront limits: host-driven LM with device numerics (fx_sparse.hip).
// the wide kernel covers Levenberg-Marquardt without a decomposer. Every path for Systems beyond the one-wavefront
// limits computes in f64 (the sparse path always did): a request for f32 compute keeps its options (ftol, max_outer)
// and gets the f64 device kernels here rather than the host-driven loop.
bool wide_kernel_applies(const fx::LmParams& p) {
    return !(p.mode & (fx::MODE_UNITS | fx::MODE_LBFGS));
}

int solve_large_systems(fx_ctx* ctx, fx_dbatch* db, const fx::LmParams& p) {
    if (!db->n_large) return FX_OK;
    // cluster problems with pose rows: only the one-wavefront pose build and the sparse path evaluate those
    const bool pose = db->d.has_pose != 0;
    if (!pose && wide_kernel_applies(p) && db->d.n_wide && !db->qr_wide_active) {  // (FX_STEP_QR: the QR build has solved them)
        hipError_t e = fx::launch_solve_wide(db->d, p, ctx->stream);
        if (e != hipSuccess) return fail(FX_ERR_HIP, "wide kernel launch failed: %s", hipGetErrorString(e));
    }
    const bool device_units = (p.mode & fx::MODE_UNITS) && !(p.mode & fx::MODE_LBFGS);
    // Decomposer::None, f64 LM: large Systems made of small components are walked on the device
    const bool comp_walk = !pose && wide_kernel_applies(p);
    if (comp_walk) {
        int rc = ensure_component_walk(ctx, db);
        if (rc) return rc;
        if (db->comp_walk.n_g) {
            fx::LmParams pw = p;
            pw.mode |= fx::MODE_UNITS;  // the walker's block loop; the RESTORE flag keeps None semantics
            hipError_t e = fx::launch_solve_walk(db->comp_walk, pw, ctx->stream);
            if (e != hipSuccess) return fail(FX_ERR_HIP, "component walk launch failed: %s", hipGetErrorString(e));
        }
    }
    // ---- Systems of one STRUCTURE (fixed flags, tags, fields, components) share a plan and are solved together
    const fx_batch& hb = db->h_batch;
    const bool wide_done = !pose && wide_kernel_applies(p);
    const uint32_t groups_of = (p.mode & (fx::MODE_UNITS | fx::MODE_LBFGS)) | (wide_done ? 0x100u : 0u) | (device_units ? 0x200u : 0u) |
                               (comp_walk ? 0x400u : 0u) | (pose ? 0x800u : 0u) | (db->qr_wide_active ? 0x1000u : 0u);
    std::vector<fx_dbatch::StructureGroup> local_groups;
    const bool cached = db->resident && db->large_groups.count(groups_of) != 0;
    std::vector<fx_dbatch::StructureGroup>& groups = db->resident ? db->large_groups[groups_of] : local_groups;
    if (!cached) {
        std::vector<uint32_t> todo;
        for (uint32_t s = 0; s < db->d.n_systems; ++s) {
            if (!db->h_sys_large[s]) continue;
            if (db->h_sys_large[s] == 2 && wide_done) continue;                                            // done by the wide kernel
            if (db->qr_wide_active && s < db->h_qr_wide.size() && db->h_qr_wide[s]) continue;             // done by its QR build
            if (device_units && s < db->h_units_on_device.size() && db->h_units_on_device[s]) continue;  // done by the kernel
            if (comp_walk && s < db->h_comp_walk.size() && db->h_comp_walk[s]) continue;                  // done by the walker
            todo.push_back(s);
        }
        auto slices = [&](uint32_t s, const void* ptr[5], size_t len[5]) {
            const uint32_t v0 = hb.var_off[s], nvt = hb.var_off[s + 1] - v0, e0 = hb.expr_off[s], net = hb.expr_off[s + 1] - e0;
            ptr[0] = hb.var_fixed + v0;                  len[0] = nvt;
            ptr[1] = hb.expr_tag + e0;                   len[1] = net;
            ptr[2] = hb.expr_idx + 4 * (size_t)e0;       len[2] = 4 * (size_t)net * sizeof(uint32_t);
            ptr[3] = hb.var_comp ? hb.var_comp + v0 : nullptr;   len[3] = hb.var_comp ? nvt * sizeof(uint16_t) : 0;
            ptr[4] = hb.expr_comp ? hb.expr_comp + e0 : nullptr; len[4] = hb.expr_comp ? net * sizeof(uint16_t) : 0;
        };
        auto structure_key = [&](uint32_t s) {
            const void* ptr[5];
            size_t len[5];
            slices(s, ptr, len);
            std::vector<unsigned char> key;
            const uint32_t head[4] = {p.mode & (fx::MODE_UNITS | fx::MODE_LBFGS), 0u, hb.var_off[s + 1] - hb.var_off[s],
                                      hb.expr_off[s + 1] - hb.expr_off[s]};
            key.reserve(sizeof(head) + len[0] + len[1] + len[2] + len[3] + len[4]);
            key.insert(key.end(), reinterpret_cast<const unsigned char*>(head), reinterpret_cast<const unsigned char*>(head) + sizeof(head));
            for (int k = 0; k < 5; ++k)
                if (len[k]) key.insert(key.end(), static_cast<const unsigned char*>(ptr[k]), static_cast<const unsigned char*>(ptr[k]) + len[k]);
            return key;
        };
        auto same_structure = [&](uint32_t x, uint32_t y) {  // the raw arrays of two Systems, slice by slice
            const void *px[5], *py[5];
            size_t lx[5], ly[5];
            slices(x, px, lx);
            slices(y, py, ly);
            for (int k = 0; k < 5; ++k)
                if (lx[k] != ly[k] || (lx[k] && memcmp(px[k], py[k], lx[k]) != 0)) return false;
            return true;
        };
        auto hash_of = [](const std::vector<unsigned char>& key) {  // eight bytes at a time
            uint64_t h = 1469598103934665603ull;
            size_t i = 0;
            for (; i + 8 <= key.size(); i += 8) {
                uint64_t w;
                memcpy(&w, key.data() + i, 8);
                h = (h ^ w) * 0xFF51AFD7ED558CCDull;
                h ^= h >> 29;
            }
            for (; i < key.size(); ++i) h = (h ^ key[i]) * 1099511628211ull;
            return h;
        };
        // groups in order of their first System; a System with the structure of the one before it joins that one's group
        // (one sketch, many parameter sets: two memcmp passes instead of a key), otherwise a 64-bit hash finds the candidates
        // and the bytes decide
        std::map<uint64_t, std::vector<size_t>> by_hash;
        size_t last_group = 0;
        uint32_t last_system = 0;
        bool have_last = false;
        for (uint32_t s : todo) {
            if (have_last && same_structure(s, last_system)) {
                groups[last_group].systems.push_back(s);
                continue;
            }
            std::vector<unsigned char> key = structure_key(s);
            const uint64_t h = hash_of(key);
            std::vector<size_t>& cand = by_hash[h];
            size_t g = groups.size();
            for (size_t i : cand)
                if (groups[i].key == key) g = i;
            if (g == groups.size()) {
                cand.push_back(g);
                groups.emplace_back();
                groups.back().key = std::move(key);
                groups.back().hash = h;
            }
            groups[g].systems.push_back(s);
            last_group = g;
            last_system = s;
            have_last = true;
        }
    }
    if (groups.empty()) return FX_OK;
    // plans: a resident batch keeps its own (per structure and decomposer mode); one-shot calls share the context's
    // (fx_ctx::plan_for — entries this call has touched are never evicted under it). Cluster problems of
    // RecursiveAssembly differ from step to step: nothing to keep.
    const uint64_t call_clock = ctx->plan_clock;
    std::vector<fx::SparsePlanCache*> group_plan(groups.size(), nullptr);
    for (size_t g = 0; g < groups.size(); ++g) {
        if (db->resident) {
            const uint64_t h = groups[g].hash ^ ((p.mode & fx::MODE_UNITS) ? 0x9E3779B97F4A7C15ull : 0ull);
            auto range = db->sparse_plans.equal_range(h);
            fx_dbatch::ResidentPlan* found = nullptr;
            for (auto it = range.first; it != range.second; ++it)
                if (it->second.key == groups[g].key) found = &it->second;
            if (!found) {
                found = &db->sparse_plans.emplace(h, fx_dbatch::ResidentPlan{groups[g].key, nullptr})->second;
                found->plan = fx::sparse_cache_new();  // (after the entry exists: free_batch finds it whatever happens next)
                fx::sparse_cache_keep_slab(found->plan, SIZE_MAX);  // the batch is there to be solved again: its slab goes with it
            }
            group_plan[g] = found->plan;
        } else if (!pose) {
            group_plan[g] = ctx->plan_for(std::vector<unsigned char>(groups[g].key), call_clock);
        }
    }
    // every launch covers a whole group, control flow on the device (fx_sparse_team.h): Levenberg-Marquardt or L-BFGS.
    // Groups are independent (their own plans, slabs and Systems): several of them run side by side, a host thread and a
    // stream each (fx_ctx_set_host_threads) — one structure's launches leave most of the chip idle (32 different
    // 150-variable sketches: 9.5 ms one after the other).
    const uint32_t n_workers = (uint32_t)std::min<size_t>(std::max(1u, ctx->host_threads), groups.size());
    if (n_workers <= 1) {
        for (size_t g = 0; g < groups.size(); ++g) {
            hipError_t e = fx::sparse_solve_group(&hb, db->d, groups[g].systems.data(), (uint32_t)groups[g].systems.size(), p, ctx->stream,
                                                  group_plan[g], /*stay_async=*/group_plan[g] != nullptr);
            if (e != hipSuccess)
                return fail(FX_ERR_HIP, "sparse path failed on the group of system %u: %s", groups[g].systems[0], hipGetErrorString(e));
        }
        return FX_OK;
    }
    while (ctx->worker_streams.size() + 1 < n_workers) {
        hipStream_t st = nullptr;
        FX_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        ctx->worker_streams.push_back(st);
    }
    FX_HIP(hipStreamSynchronize(ctx->stream));  // the batch's upload and whatever else the context's stream still holds
    ctx->stream_synced();
    std::atomic<size_t> next{0};
    std::vector<hipError_t> errs(n_workers, hipSuccess);
    std::vector<size_t> err_group(n_workers, 0);
    auto work = [&](uint32_t w) {
        (void)hipSetDevice(ctx->device);
        hipStream_t st = w == 0 ? ctx->stream : ctx->worker_streams[w - 1];
        for (;;) {
            const size_t g = next.fetch_add(1);
            if (g >= groups.size() || errs[w] != hipSuccess) break;
            errs[w] = fx::sparse_solve_group(&hb, db->d, groups[g].systems.data(), (uint32_t)groups[g].systems.size(), p, st, group_plan[g]);
            err_group[w] = g;
        }
    };
    fx::run_workers(n_workers, work);  // (a worker that cannot be started: the others empty the queue)
    for (uint32_t w = 0; w < n_workers; ++w)
        if (errs[w] != hipSuccess)
            return fail(FX_ERR_HIP, "sparse path failed on the group of system %u: %s", groups[err_group[w]].systems[0], hipGetErrorString(errs[w]));
    return FX_OK;
}

// Systems beyond one wavefront. FX_STEP_QR: those whose components have at most 128 columns run the wide kernel's QR build
// (the reference's numerics; ensure_qr_plans listed them), everything larger takes the refined step on the sparse path.
int solve_beyond_one_wavefront(fx_ctx* ctx, fx_dbatch* db, fx::LmParams p) {
    if (p.lm.solver == FX_STEP_QR) {
        const bool qr_wide = !(p.mode & (fx::MODE_UNITS | fx::MODE_LBFGS)) && db->d.qr_none.n_qrw != 0 && !db->d.has_pose;
        if (qr_wide) {
            hipError_t e = fx::launch_solve_wide_qr(db->d, p, ctx->stream);
            if (e != hipSuccess) return fail(FX_ERR_HIP, "wide QR kernel launch failed: %s", hipGetErrorString(e));
        }
        p.lm.solver = FX_STEP_CHOLESKY_REFINED;
        db->qr_wide_active = qr_wide;
        const int rc = solve_large_systems(ctx, db, p);
        db->qr_wide_active = false;
        return rc;
    }
    return solve_large_systems(ctx, db, p);
}

// ---- host-buffer entry points ---------------------------------------------------------------


// A big batch of one-wavefront Systems, analysed as a whole, goes up and is solved in chunks of some megabytes: chunk k + 1
// is copied up (second stream) while chunk k is being solved; the read-backs follow in order. Every System is solved on its
// own and every chunk runs the kernels the whole batch would, so the cut changes nothing in the results (the tests compare
// the bits). FIKSI_AMD_HOST_CHUNKS=0 switches it off, =k sets the number of chunks.
constexpr int FX_HINT_REFUSED = 1;  // (internal to this file: the hinted batch failed its verification — never returned to a caller)
static int solve_host_chunked(fx_ctx* ctx, const fx_batch* batch, const HostPlan& p, uint32_t n_chunks, const fx_solving_opts* sopts,
                              const fx_lm_opts* lopts, bool system_level, fx_result* results) {
    const uint32_t n = p.n_systems;
    struct Chunk {
        fx_batch b{};
        fx_dbatch* db = nullptr;
        hipStream_t solve_stream = nullptr;
        uint32_t s0 = 0;
    };
    std::vector<Chunk> chunks(n_chunks);
    hipStream_t const main_stream = ctx->stream;
    PhaseTrace tr;
    int rc = FX_OK;
    for (uint32_t k = 0; k < n_chunks && rc == FX_OK; ++k) {
        Chunk& c = chunks[k];
        const uint32_t s0 = (uint32_t)((uint64_t)n * k / n_chunks), s1 = (uint32_t)((uint64_t)n * (k + 1) / n_chunks);
        c.s0 = s0;
        c.b.n_systems = s1 - s0;
        c.b.vars = batch->vars + batch->var_off[s0];  // (all the read-back needs of the chunk's host side)
        // the copies on the second stream (never behind a solve), the solve on the context's own, after them
        ctx->stream = ctx->stream2;
        try {
            rc = upload_planned(ctx, batch, p, s0, s1, &c.db);
        } catch (...) {  // (the context's stream is borrowed here and earlier chunks are in flight: no unwinding through this frame)
            rc = fx::translate_exception();
        }
        c.solve_stream = (k & 1u) ? ctx->stream3 : main_stream;
        ctx->stream = c.solve_stream;
        if (rc) break;
        c.db->resident = false;
        if (hipEventRecord(ctx->ev_chunk, ctx->stream2) != hipSuccess || hipStreamWaitEvent(c.solve_stream, ctx->ev_chunk, 0) != hipSuccess) {
            rc = fail(FX_ERR_HIP, "event between the copy and the solve stream failed");
            break;
        }
        rc = system_level ? fx_system_solve_device(ctx, c.db, sopts) : fx_lm_solve_device(ctx, c.db, lopts);
    }
    tr.stamp("chunks: up + launched", n);
    if (rc == FX_OK && p.hinted && !verify_one_structure(batch)) rc = FX_HINT_REFUSED;  // (beside the device's work; nothing is back yet)
    if (p.hinted) tr.stamp("one structure: verified", n);
    if (rc != FX_OK) {  // nothing has been written to the caller's arrays yet
        (void)hipStreamSynchronize(ctx->stream2);
        for (Chunk& c : chunks)
            if (c.db) {
                ctx->stream = c.solve_stream ? c.solve_stream : main_stream;
                free_batch(ctx, c.db, /*stream_idle=*/false);
            }
        ctx->stream = main_stream;
        return rc;
    }
    for (Chunk& c : chunks) {
        ctx->stream = c.solve_stream;
        int r = read_back_and_free(ctx, c.db, &c.b, results ? results + c.s0 : nullptr, FX_OK);
        if (r && !rc) rc = r;
    }
    ctx->stream = main_stream;
    tr.stamp("chunks: read back", n);
    return rc;
}

// A batch of one structure whose value arrays the caller has page-locked (fx_host_register): the structure goes up as ever (one
// period per array, a few hundred KB of offsets), the values do not — the one-structure build of the grouped kernel reads a System's
// start values and parameters from the caller's arrays when the System's turn comes, and writes the solved variables and the result
// record back there when it is done, so the transfers are spread over the solve instead of standing before and after it (100 000
// ring16 sketches: 51 MB up, 35 MB down). The device keeps its own copy of everything (vars0 included: a refused hint
// puts the caller's start values back from there). A solve that would not take that build fills the values in by ordinary copies.
static int solve_host_in_place(fx_ctx* ctx, const fx_batch* batch, const HostPlan& p, void* dev_vars, void* dev_params, void* dev_results,
                               const fx_solving_opts* sopts, const fx_lm_opts* lopts, bool system_level, fx_result* results, PhaseTrace& tr) {
    fx_dbatch* db = nullptr;
    const uint32_t defer = FX_DEFER_VARS | FX_DEFER_PARAMS;
    int rc = upload_planned(ctx, batch, p, 0, p.n_systems, &db, /*one_shot=*/true, defer);
    if (rc) return rc;
    BatchHolder hold(ctx, db);
    db->resident = false;
    tr.stamp("upload (structure only)", p.n_systems);
    if (!takes_one_structure_build(ctx, db, sopts, lopts, system_level)) {
        rc = fill_deferred(ctx, db, batch, defer);
        tr.stamp("values, by copies", p.n_systems);
        if (rc) return rc;
        rc = system_level ? fx_system_solve_device(ctx, db, sopts) : fx_lm_solve_device(ctx, db, lopts);
        if (p.hinted && !rc && !verify_one_structure(batch)) return FX_HINT_REFUSED;  // (nothing is back yet; the holder frees the batch)
        return read_back_and_free(ctx, hold.release(), batch, results, rc);
    }
    fx::DeviceBatch& d = db->d;
    db->in_place = true;
    d.vars_in = static_cast<const double*>(dev_vars);
    d.param_in = static_cast<const double*>(dev_params);
    d.vars_out = static_cast<double*>(dev_vars);
    d.results_out = static_cast<fx_result*>(dev_results);
    rc = system_level ? fx_system_solve_device(ctx, db, sopts) : fx_lm_solve_device(ctx, db, lopts);
    tr.stamp("solve (launched)", p.n_systems);
    if (rc) return rc;  // (refused before any launch: the caller's arrays are as they were)
    const bool kept = !p.hinted || verify_one_structure(batch);  // beside the device's work
    if (p.hinted) tr.stamp("one structure: verified", p.n_systems);
    hipError_t e = hipStreamSynchronize(ctx->stream);
    ctx->stream_synced();
    tr.stamp("wait", p.n_systems);
    if (e != hipSuccess) return fail(FX_ERR_HIP, "solve failed: %s", hipGetErrorString(e));
    if (!kept) {  // the promise was not kept: the caller's start values back from the device's copy, then the ordinary way
        FX_HIP(hipMemcpy(batch->vars, d.vars0, (size_t)d.n_vars * sizeof(double), hipMemcpyDeviceToHost));
        return FX_HINT_REFUSED;
    }
    if (results && !dev_results) rc = fx_batch_get_results(ctx, db, results);
    return rc;
}

int solve_host(fx_ctx* ctx, const fx_batch* batch, const fx_solving_opts* sopts, const fx_lm_opts* lopts,
               bool system_level, fx_result* results, bool no_hint) {
    int rc = bind(ctx);
    if (rc) return rc;
    fx_dbatch* db = nullptr;
    PhaseTrace tr;
    bool hinted = false;
    {
        HostPlan p;
        g_wide_routing = ctx->wide_routing;
        g_hint_one_structure = (ctx->batch_hints & FX_HINT_ONE_STRUCTURE) != 0 && !no_hint;
        try {
            rc = analyze(batch, &p);
        } catch (...) {
            g_hint_one_structure = false;
            throw;
        }
        g_hint_one_structure = false;
        if (rc) return rc;
        hinted = p.hinted;
        tr.stamp("analysis", p.n_systems);
        // Two chunks from 65 536 Systems on: measured on 100 000 ring16 sketches (tools/host_path.py), 2 chunks
        // 6.3 ms, 3 and 4 chunks 6.8 ms, 8 chunks 8.7 ms, uncut 7.6 ms — every chunk pays its own dozen copies and the slow
        // end of its own solve, so more chunks lose what the earlier start of the first solve wins
        static const bool in_place_on = [] { const char* e = std::getenv("FIKSI_AMD_IN_PLACE"); return !e || atoi(e) != 0; }();
        if (in_place_on && p.uniform && p.n_large == 0 && p.wide_list.empty() && p.n_systems >= 16384u) {
            void* dv = registered_range(batch->vars, (size_t)p.n_vars * sizeof(double));
            void* dp = registered_range(batch->expr_param, (size_t)p.n_exprs * sizeof(double));
            void* dr = results ? registered_range(results, (size_t)p.n_systems * sizeof(fx_result)) : nullptr;
            if (dv && dp) {
                rc = solve_host_in_place(ctx, batch, p, dv, dp, dr, sopts, lopts, system_level, results, tr);
                return rc == FX_HINT_REFUSED ? solve_host(ctx, batch, sopts, lopts, system_level, results, /*no_hint=*/true) : rc;
            }
        }
        static const int forced = [] { const char* e = std::getenv("FIKSI_AMD_HOST_CHUNKS"); return e ? atoi(e) : -1; }();
        uint32_t n_chunks = p.n_systems >= 65536u ? 2u : 0u;
        if (forced >= 0) n_chunks = std::min<uint32_t>((uint32_t)forced, p.n_systems / 2u);
        // (the copy stream at the highest priority: a chunk's small copies and period fills are kernels, and behind the workgroups of
        // the chunk before's solve one of them took 0.5 ms instead of 5 us — profiles/round5_host_path.md)
        auto copy_stream = [&]() {
            if (ctx->stream2) return true;
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) greatest = 0;
            return hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, greatest) == hipSuccess;
        };
        if (n_chunks >= 2 && p.n_large == 0 && copy_stream() &&
            (ctx->stream3 || hipStreamCreateWithFlags(&ctx->stream3, hipStreamNonBlocking) == hipSuccess) &&
            (ctx->ev_chunk || hipEventCreateWithFlags(&ctx->ev_chunk, hipEventDisableTiming) == hipSuccess))
        {
            rc = solve_host_chunked(ctx, batch, p, n_chunks, sopts, lopts, system_level, results);
            // the batch was not of one structure after all: again, the ordinary way (the caller's arrays are untouched)
            return rc == FX_HINT_REFUSED ? solve_host(ctx, batch, sopts, lopts, system_level, results, /*no_hint=*/true) : rc;
        }
        rc = upload_planned(ctx, batch, p, 0, p.n_systems, &db, /*one_shot=*/true);
    }
    if (rc) return rc;
    BatchHolder hold(ctx, db);
    tr.stamp("upload", batch->n_systems);
    db->resident = false;  // solved once and freed: no point in keeping plans
    rc = system_level ? fx_system_solve_device(ctx, db, sopts) : fx_lm_solve_device(ctx, db, lopts);
    tr.stamp("solve (launches)", batch->n_systems);
    if (hinted && !rc && !verify_one_structure(batch))  // (beside the device's work; the holder frees the batch, nothing is back yet)
        return solve_host(ctx, batch, sopts, lopts, system_level, results, /*no_hint=*/true);
    rc = read_back_and_free(ctx, hold.release(), batch, results, rc);
    tr.stamp("wait + read back", batch->n_systems);
    return rc;
}

}  // namespace fxh
