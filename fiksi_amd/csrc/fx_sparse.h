// Large-component path (fx_sparse.hip): host structure + device numeric sparse Cholesky LM.
#pragma once
#ifdef FX_HOST_ONLY
#include "fx_hip_shim.h"
#else
#include <hip/hip_runtime.h>
#endif

#include "fx_device.h"

namespace fx {

// Solves System `s` of the host batch `b` (all of its connected components) on `stream`; the
// solved variables go to the device buffer `d_vars_out` (the System's slice, nvt doubles).
// `cache` (may be NULL): the System's plan from an earlier solve of the same resident batch in the same
// decomposer mode; filled on the first solve, reused afterwards.
struct SparsePlanCache;
SparsePlanCache* sparse_cache_new();
void sparse_cache_free(SparsePlanCache* c);
bool sparse_cache_ready(const SparsePlanCache* c);  // filled by a completed solve: later solves only read it
// Levenberg-Marquardt for Systems systems[0 .. n) of the host batch `b`, which all have the structure of the first one
// (fixed flags, tags, fields, components): one plan, every launch covers the whole group, results and solved variables
// go straight to the resident batch `d` (d.vars, d.results). Synchronises `stream` before it returns.
hipError_t sparse_solve_group(const fx_batch* b, const DeviceBatch& d, const uint32_t* systems, uint32_t n, const LmParams& prm,
                              hipStream_t stream, SparsePlanCache* cache);
// Optimizer::LBfgs, one System:
hipError_t sparse_solve_system(const fx_batch* b, uint32_t s, const LmParams& prm, hipStream_t stream,
                               double* d_vars_out, fx_result* result, SparsePlanCache* cache);

}  // namespace fx
