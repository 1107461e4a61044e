// Large-component path (fx_sparse.hip): host structure + device numeric sparse Cholesky LM.
#pragma once
#ifdef FX_HOST_ONLY
#include "fx_hip_shim.h"
#else
#include <hip/hip_runtime.h>
#endif

#include "fx_device.h"

namespace fx {

// `cache` (may be NULL): the plan of a structure in one decomposer mode, from an earlier solve; filled on the first
// solve, reused afterwards.
struct SparsePlanCache;
SparsePlanCache* sparse_cache_new();
void sparse_cache_free(SparsePlanCache* c);
bool sparse_cache_ready(const SparsePlanCache* c);  // filled by a completed solve: later solves only read it
// how much of the value slab a solve leaves with the plan for the next one (default 256 MB; a resident batch's plans: all of it)
void sparse_cache_keep_slab(SparsePlanCache* c, size_t bytes);
// Levenberg-Marquardt or L-BFGS (prm.mode) for Systems systems[0 .. n) of the host batch `b`, which all have the structure of the first one
// (fixed flags, tags, fields, components): one plan, every launch covers the whole group, results and solved variables
// go straight to the resident batch `d` (d.vars, d.results). Synchronises `stream` before it returns — unless stay_async is set (the
// caller's next work goes to the same stream) and the plan is a resident batch's that has seen this group before: such a solve
// uploads nothing and gives nothing back, and returns with its launches in flight.
hipError_t sparse_solve_group(const fx_batch* b, const DeviceBatch& d, const uint32_t* systems, uint32_t n, const LmParams& prm,
                              hipStream_t stream, SparsePlanCache* cache, bool stay_async = false);

}  // namespace fx
