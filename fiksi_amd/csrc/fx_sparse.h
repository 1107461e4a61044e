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
hipError_t sparse_solve_system(const fx_batch* b, uint32_t s, const LmParams& prm, hipStream_t stream,
                               double* d_vars_out, fx_result* result, SparsePlanCache* cache);

}  // namespace fx
