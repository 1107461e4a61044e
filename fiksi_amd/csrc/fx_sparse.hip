// Large-component path: Systems whose connected components exceed the one-wavefront limits of the
// fused kernel (more than 64 free variables or 256 expressions) — e.g. BASELINE cfg2, one sketch of
// 5 000 points / 10 000 constraints.
//
// Same algorithm as the fused kernel (fiksi/src/assemble/mod.rs:46-167, fiksi/src/solve/lm.rs:21-193,
// LM step in normal-equation form), organised for one big sparse problem instead of many small dense
// ones:
//   host   : structure only — free-column list, CSR pattern of J, pattern of A = JtJ, a
//            reverse-Cuthill-McKee fill-reducing order (the reference runs COLAMD on the host too,
//            qr.rs:118-206), symbolic Cholesky (pattern of L via elimination-tree merging) and, for
//            every non-zero of A and of L, the list of products that define it ("gather lists").
//            The LM accept/reject decisions are taken on the host from three scalars per trial.
//   device : all arithmetic — scale/perturb (K0), residual + Jacobian rows (K1/K2, one thread per
//            row), A = JtJ and g = -Jt r by deterministic gathers (K3), numeric sparse Cholesky and
//            the two triangular solves column by column in one wavefront (K4), trial update, SSE.
// No floating-point work happens on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <queue>
#include <vector>

#include "fx_decompose.h"
#include "fx_device.h"
#include "fx_expr.h"
#include "fx_lbfgs.h"
#include "fx_sparse.h"
#include "fx_wave.h"

namespace fx {

namespace {

// ------------------------------------------------------------------------------------------------
// device kernels
// ------------------------------------------------------------------------------------------------
struct SpRows {                 // expression data of the whole System (device)
    const uint8_t* tag;         // [net]
    const uint16_t* idx;        // [4*net]
    const double* param;        // [net] unscaled
    double* sparam;             // [net] scaled (distance parameters * 1/scale)
    uint32_t net;
    uint32_t has_pose;          // the System holds pose rows (cluster problems): the POSE builds of the row kernels run it
};

// state of a device-controlled LM loop (the kernels that use it come further down)
struct SpLm {
    double lambda, sse, sse_t, dn2, sse_start;
    uint32_t cur, accepted, trials, outer, exit_code, done, need_form, flag;
};
__device__ __forceinline__ uint32_t sp_lm_cur(const SpLm* st) { return st->cur; }
__device__ __forceinline__ bool sp_lm_done(const SpLm* st) { return st->done != 0; }
struct SpBufs {  // the two generations of the trial vectors
    double* xs[2];
    double* r[2];
    double* j[2];
};
// scal[0] = scale, scal[1] = 1/scale. Sequential summation in the reference's order (assemble/mod.rs:32-44: all
// variables, then the distance parameters, each ascending) so the scale is bit-identical — the sum itself cannot be
// split, but everything around it can: 256 threads square 4096 values at a time into LDS (+0.0 for expressions without
// a distance: exact), then ONE thread adds them up in order, sixteen LDS values in flight per step. (The first
// version added 64 values per step through v_readlane pairs: 0.83 ms for cfg2's 20 000 values; this one ~0.1 ms.)
__global__ __launch_bounds__(256) void sp_scale_kernel(const double* __restrict__ vars0, uint32_t nvt, SpRows rows,
                                                      double* __restrict__ scal, int do_scale) {
    constexpr uint32_t CH = 4096;
    __shared__ double sq[CH];
    __shared__ uint32_t cnt_s[256];
    double sum = 0.0;
    uint32_t ndist = 0;
    const uint32_t total = do_scale ? nvt + rows.net : 0u;
    for (uint32_t base = 0; base < total; base += CH) {
        const uint32_t n = min(CH, total - base);
        uint32_t mine = 0;
        for (uint32_t i = threadIdx.x; i < n; i += 256u) {
            const uint32_t g = base + i;
            double v = 0.0;
            if (g < nvt) {
                v = vars0[g];
            } else {
                const int tag = rows.tag[g - nvt] & 0x7F;
                if (tag == 1 || tag == 4) {  // FX_TAG_PPD, FX_TAG_PLD
                    v = rows.param[g - nvt];
                    mine += 1;
                }
            }
            sq[i] = v * v;
        }
        cnt_s[threadIdx.x] = mine;
        __syncthreads();
        if (threadIdx.x == 0) {
            for (uint32_t i = 0; i < n; i += 16u) {
                double t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = (i + u < n) ? sq[i + u] : 0.0;
#pragma unroll
                for (int u = 0; u < 16; ++u) sum += t[u];
            }
            for (uint32_t i = 0; i < 256u; ++i) ndist += cnt_s[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double scale = do_scale ? ::sqrt(sum / (double)(nvt + ndist)) : 1.0;
        scal[0] = scale;
        scal[1] = 1.0 / scale;
    }
}

__global__ void sp_init_kernel(const double* __restrict__ vars0, uint32_t nvt, SpRows rows, const double* __restrict__ scal,
                               double* __restrict__ xs_a, double* __restrict__ xs_b, int do_scale) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const double recip = scal[1];
    if (i < nvt) {
        double v = vars0[i];
        double x = do_scale ? v * recip : v;
        xs_a[i] = x;
        xs_b[i] = x;
    }
    if (i < rows.net) {
        int tag = rows.tag[i] & 0x7F;
        double p = rows.param[i];
        if (do_scale && (tag == FX_TAG_PPD || tag == FX_TAG_PLD)) p = recip * p;
        rows.sparam[i] = p;
    }
}

// LCG skip-ahead: state after n steps of s <- s*A + C is s*A^n + C*(A^n-1)/(A-1) (mod 2^32),
// computed by squaring the affine map.
__device__ __forceinline__ uint32_t lcg_skip(uint32_t s, uint32_t n) {
    uint32_t a = 1664525u, c = 1013904223u;  // fiksi/src/rand.rs:24-30
    uint32_t acc_a = 1u, acc_c = 0u;
    while (n) {
        if (n & 1u) {
            acc_c = acc_c * a + c;
            acc_a = acc_a * a;
        }
        c = c * a + c;
        a = a * a;
        n >>= 1;
    }
    return s * acc_a + acc_c;
}

// assemble/mod.rs:113-124: free variable k (ascending) takes draws 2k and 2k+1 of the shared Rng.
__global__ void sp_perturb_kernel(const uint32_t* __restrict__ fvar, uint32_t nv, uint32_t rng_state,
                                  double* __restrict__ xs_a, double* __restrict__ xs_b) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nv) return;
    uint32_t st = lcg_skip(rng_state, 2u * k);
    st = st * 1664525u + 1013904223u;
    double f1 = (1.0 / 4294967295.0) * (double)st;
    st = st * 1664525u + 1013904223u;
    double f2 = (1.0 / 4294967295.0) * (double)st;
    uint32_t vi = fvar[k];
    double x = xs_a[vi];
    x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
    xs_a[vi] = x;
    xs_b[vi] = x;
}

struct SpJac {                   // J of one component (device)
    const uint32_t* rows;        // [m] expression id of row
    const uint32_t* jrow_ptr;    // [m+1]
    const uint32_t* jslot;       // [m] 8 x 4-bit slot of each gradient entry (0xF = not a free column)
    uint32_t m;
    int overwrite;               // entries sharing a column: 0 = summed (sparse J, sparse_col_mat.rs:710-711),
                                 // 1 = the last one wins (dense J of L-BFGS, expressions.rs:993-1008) — quirk Q4
};

// K1/K2 for one component: thread per row (subsystem.rs:93-166).
template <bool WANT_J, bool POSE = false>
__global__ __launch_bounds__(256) void sp_eval_kernel(SpRows rows, SpJac jac, const double* __restrict__ xs,
                                                      double* __restrict__ r, double* __restrict__ jvals) {
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= jac.m) return;
    uint32_t e = jac.rows[row];
    int tag = rows.tag[e] & 0x7F;
    ushort4 f4 = reinterpret_cast<const ushort4*>(rows.idx)[e];
    uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
    uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    expand_vars<POSE>(tag, ff, vars8);
    double v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = xs[vars8[q]];
    r[row] = eval_expression<double, WANT_J, false, POSE>(tag, v, rows.sparam[e], g);
    if (WANT_J) {
        uint32_t slots = jac.jslot[row];
        uint32_t base = jac.jrow_ptr[row];
        uint32_t cnt = jac.jrow_ptr[row + 1] - base;
        double out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            uint32_t sl = (slots >> (4 * q)) & 0xFu;
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                if (jac.overwrite) out[t] = (sl == (uint32_t)t) ? g[q] : out[t];
                else out[t] += (sl == (uint32_t)t) ? g[q] : 0.0;
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            if ((uint32_t)t < cnt) jvals[base + t] = out[t];
        }
    }
}

// the same at the trial point of a device-controlled LM loop: generation cur ^ 1 of the vectors
template <bool POSE = false>
__global__ __launch_bounds__(256) void sp_eval_dc_kernel(SpRows rows, SpJac jac, double* xs0, double* xs1, double* r0, double* r1, double* j0,
                                                         double* j1, const SpLm* __restrict__ st) {
    if (sp_lm_done(st)) return;
    const uint32_t t = sp_lm_cur(st) ^ 1u;
    const double* xs = t ? xs1 : xs0;
    double* r = t ? r1 : r0;
    double* jvals = t ? j1 : j0;
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= jac.m) return;
    uint32_t e = jac.rows[row];
    int tag = rows.tag[e] & 0x7F;
    ushort4 f4 = reinterpret_cast<const ushort4*>(rows.idx)[e];
    uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
    uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    expand_vars<POSE>(tag, ff, vars8);
    double v[8], g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = xs[vars8[q]];
    r[row] = eval_expression<double, true, false, POSE>(tag, v, rows.sparam[e], g);
    uint32_t slots = jac.jslot[row];
    uint32_t base = jac.jrow_ptr[row];
    uint32_t cnt = jac.jrow_ptr[row + 1] - base;
    double out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        uint32_t sl = (slots >> (4 * q)) & 0xFu;
#pragma unroll
        for (int u = 0; u < 8; ++u) out[u] += (sl == (uint32_t)u) ? g[q] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if ((uint32_t)u < cnt) jvals[base + u] = out[u];
    }
}

// out[0] = sum v[i]^2 (fixed-shape tree: deterministic)
__global__ __launch_bounds__(1024) void sp_sumsq_kernel(const double* __restrict__ v, uint32_t n, double* __restrict__ out) {
    __shared__ double part[1024];
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) s += v[i] * v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

// K3a: A[k] = sum over its gather list of J[a]*J[b] (lower triangle of JtJ, permuted order)
__global__ void sp_form_a_kernel(const uint32_t* __restrict__ pair_ptr, const uint32_t* __restrict__ pairs,
                                 const double* __restrict__ jvals, uint32_t nnz_a, double* __restrict__ a) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz_a) return;
    double s = 0.0;
    for (uint32_t p = pair_ptr[k]; p < pair_ptr[k + 1]; ++p) s += jvals[pairs[2 * p]] * jvals[pairs[2 * p + 1]];
    a[k] = s;
}

// K3b: b[c] = -sum_{rows of column c} J * r   (permuted column order)
template <bool NEGATE>
__global__ void sp_rhs_kernel(const uint32_t* __restrict__ cptr, const uint32_t* __restrict__ cidx,
                              const uint32_t* __restrict__ crow, const double* __restrict__ jvals,
                              const double* __restrict__ r, uint32_t nv, double* __restrict__ b) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nv) return;
    double s = 0.0;
    for (uint32_t p = cptr[c]; p < cptr[c + 1]; ++p) s += jvals[cidx[p]] * (NEGATE ? -r[crow[p]] : r[crow[p]]);
    b[c] = s;
}

// ---- Optimizer::LBfgs on one large block: vectors in (permuted) column space, one workgroup ----------
__device__ __forceinline__ double block_sum_1024(double v, double* sh) {
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    double out = sh[0];
    __syncthreads();
    return out;
}

// out[0] = sum a_i b_i (fixed-shape tree: deterministic)
__global__ __launch_bounds__(1024) void sp_dot_kernel(const double* __restrict__ a, const double* __restrict__ b, uint32_t n,
                                                      double* __restrict__ out) {
    __shared__ double sh[1024];
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) s += a[i] * b[i];
    s = block_sum_1024(s, sh);
    if (threadIdx.x == 0) out[0] = s;
}

__global__ void sp_scaled_copy_kernel(const double* __restrict__ x, double alpha, uint32_t n, double* __restrict__ out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = alpha * x[i];
}

// The two-loop recursion (lbfgs.rs:86-139) for iteration k: dir = -H_k grad, with the reference's ring
// indexing (k + i) % 5 (unwritten slots are zero) and the gamma scaling from the previous pair.
__global__ __launch_bounds__(1024) void sp_lbfgs_direction_kernel(uint32_t k, uint32_t n, const double* __restrict__ S,
                                                                 const double* __restrict__ Y, const double* __restrict__ rho,
                                                                 const double* __restrict__ grad, double* __restrict__ dir) {
    __shared__ double sh[1024];
    const uint32_t hl = k < 5u ? k : 5u;
    double alpha[5] = {0., 0., 0., 0., 0.};
    for (uint32_t i = threadIdx.x; i < n; i += 1024) dir[i] = grad[i];
    __syncthreads();
    for (int i = 4; i >= 0; --i) {
        if ((uint32_t)i >= hl) continue;
        const uint32_t h = (k + (uint32_t)i) % 5u;
        double part = 0.0;
        for (uint32_t j = threadIdx.x; j < n; j += 1024) part += S[(size_t)h * n + j] * dir[j];
        alpha[i] = rho[h] * block_sum_1024(part, sh);
        for (uint32_t j = threadIdx.x; j < n; j += 1024) dir[j] -= alpha[i] * Y[(size_t)h * n + j];
        __syncthreads();
    }
    if (k > 0) {
        const uint32_t h = (k - 1u) % 5u;
        double p1 = 0.0, p2 = 0.0;
        for (uint32_t j = threadIdx.x; j < n; j += 1024) {
            double yv = Y[(size_t)h * n + j];
            p1 += S[(size_t)h * n + j] * yv;
            p2 += yv * yv;
        }
        double s_dot_y = block_sum_1024(p1, sh), y_dot_y = block_sum_1024(p2, sh);
        if (y_dot_y > 0.) {
            double scale = s_dot_y / y_dot_y;
            for (uint32_t j = threadIdx.x; j < n; j += 1024) dir[j] *= scale;
        }
        __syncthreads();
    }
    for (int i = 0; i < 5; ++i) {
        if ((uint32_t)i >= hl) continue;
        const uint32_t h = (k + (uint32_t)i) % 5u;
        double part = 0.0;
        for (uint32_t j = threadIdx.x; j < n; j += 1024) part += Y[(size_t)h * n + j] * dir[j];
        double beta = rho[h] * block_sum_1024(part, sh);
        for (uint32_t j = threadIdx.x; j < n; j += 1024) dir[j] += S[(size_t)h * n + j] * (alpha[i] - beta);
        __syncthreads();
    }
    for (uint32_t j = threadIdx.x; j < n; j += 1024) dir[j] *= -1.;
}

// s_k = step * dir, y_k = grad - (old gradient parked in Y[h]), rho_k = 1 / s_k.y_k   (lbfgs.rs:168-180)
__global__ __launch_bounds__(1024) void sp_lbfgs_update_kernel(uint32_t h, uint32_t n, double step, const double* __restrict__ dir,
                                                              const double* __restrict__ grad, double* __restrict__ S,
                                                              double* __restrict__ Y, double* __restrict__ rho) {
    __shared__ double sh[1024];
    double part = 0.0;
    for (uint32_t j = threadIdx.x; j < n; j += 1024) {
        double sk = step * dir[j];
        double yk = grad[j] - Y[(size_t)h * n + j];
        S[(size_t)h * n + j] = sk;
        Y[(size_t)h * n + j] = yk;
        part += sk * yk;
    }
    double s_dot_y = block_sum_1024(part, sh);
    if (threadIdx.x == 0) rho[h] = 1.0 / s_dot_y;
}

__device__ __forceinline__ double ld_l2(const double* p) {  // L2-coherent load (bypasses the CU's L1)
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

struct SpChol {                  // symbolic factor (device)
    const uint32_t* lcolptr;     // [nv+1]; first entry of every column is its diagonal
    const uint32_t* lrow;        // [nnzL]
    const int32_t* l2a;          // [nnzL] index into A or -1 (fill-in)
    const uint32_t* lpair_ptr;   // [nnzL+1]
    const uint32_t* lpairs;      // [2*npairs] indices into L
    const uint32_t* lpair_k;     // [npairs] the entry of L a product belongs to (the inverse of lpair_ptr)
    const uint8_t* coop;         // [nv] 1 = the column's gather lists are long: the whole wavefront sums each one
    uint32_t nv;
};

// Work lists of the elimination-tree schedule: list q holds columns cols[ptr[q] .. ptr[q+1]) in
// ascending order; one wavefront walks one list. Lists are grouped in levels: level 0 are whole
// subtrees, the columns above them form chains whose level is one more than the deepest list below.
// One launch covers the lists [first, first + gridDim.x) of one level, which are independent.
struct ColLists {
    const uint32_t* ptr;
    const uint32_t* cols;
    uint32_t first;
};

__device__ __forceinline__ void lds_add_f64(double* p, double v) {
    __builtin_amdgcn_ds_atomic_fadd_f64((__attribute__((address_space(3))) double*)p, v);
}
__device__ __forceinline__ double wave_sum64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

struct SpRowsOfL {               // L by rows (strictly lower part), for the forward sweep
    const uint32_t* rptr;        // [nv+1]
    const uint32_t* ridx;        // index of the entry in L's value array
    const uint32_t* rcol;        // its column
};

// K4a + K4b: numeric sparse Cholesky of A + lambda I, left-looking by gather lists, with the forward
// sweep L y = b riding along: once column j is factored, row j of L is complete (it only holds
// columns below j in the elimination tree), so y_j = (b_j - sum_{k<j} L_jk y_k) / L_jj follows at
// once. flag[0] |= 1 when a pivot is not positive and finite. Columns of up to 64 entries (the usual
// case) take one pass: the pivot travels by v_readlane. Everything a column reads was written either
// by this wavefront or by an earlier launch (a lower level). y overwrites b.
__global__ __launch_bounds__(64) void sp_factor_forward_kernel(SpChol c, SpRowsOfL lr, ColLists cl,
                                                               const double* __restrict__ a, double lambda,
                                                               double* __restrict__ l, double* __restrict__ b,
                                                               uint32_t* __restrict__ flag, const SpLm* __restrict__ st) {
    __shared__ double acc[64];
    if (st) {  // device-controlled loop: lambda and the stop flag live on the device
        if (st->done) return;
        lambda = st->lambda;
    }
    const int lane = threadIdx.x;
    const uint32_t list = cl.first + blockIdx.x;
    bool bad = false;
    for (uint32_t q = cl.ptr[list]; q < cl.ptr[list + 1]; ++q) {
        const uint32_t j = cl.cols[q];
        const uint32_t beg = c.lcolptr[j], end = c.lcolptr[j + 1];
        // forward-sweep gather for row j (all of its columns are final already)
        double part = 0.0;
        for (uint32_t p = lr.rptr[j] + lane; p < lr.rptr[j + 1]; p += 64)
            part = fma(ld_l2(l + lr.ridx[p]), ld_l2(b + lr.rcol[p]), part);
        double d;
        if (end - beg <= 64u) {
            const uint32_t k = beg + lane;
            double s = 0.0;
            if (k < end) {
                int32_t ai = c.l2a[k];
                s = ai >= 0 ? a[ai] : 0.0;
                if (k == beg) s += lambda;
            }
            // All products of the column in one flat, lane-strided sweep (they are contiguous: lpair_ptr
            // is a prefix over the entries), summed per entry with LDS atomics. A lane walking its own
            // list one product at a time pays an L2 round trip per product — 85 in a row for the top
            // separators; flat, the whole column is a handful of passes.
            if (lane < 64) acc[lane] = 0.0;
            __syncthreads();
            for (uint32_t p = c.lpair_ptr[beg] + lane; p < c.lpair_ptr[end]; p += 64) {
                const double v = -ld_l2(l + c.lpairs[2 * p]) * ld_l2(l + c.lpairs[2 * p + 1]);
                lds_add_f64(&acc[c.lpair_k[p] - beg], v);
            }
            __syncthreads();
            if (k < end) s += acc[lane];
            int lo = __builtin_amdgcn_readfirstlane(__double2loint(s));
            int hi = __builtin_amdgcn_readfirstlane(__double2hiint(s));
            double piv = __hiloint2double(hi, lo);
            bad = bad || !(piv > 0.0) || !(piv < 1.0e300);
            d = ::sqrt(piv);
            if (k < end) l[k] = (k == beg) ? d : s / d;
        } else {
            for (uint32_t base = beg; base < end; base += 64) {
                uint32_t k = base + lane;
                if (k < end) {
                    int32_t ai = c.l2a[k];
                    double s = ai >= 0 ? a[ai] : 0.0;
                    if (k == beg) s += lambda;
                    for (uint32_t p = c.lpair_ptr[k]; p < c.lpair_ptr[k + 1]; ++p)
                        s = fma(-ld_l2(l + c.lpairs[2 * p]), ld_l2(l + c.lpairs[2 * p + 1]), s);
                    l[k] = s;
                }
            }
            __syncthreads();
            double piv = ld_l2(l + beg);
            bad = bad || !(piv > 0.0) || !(piv < 1.0e300);
            d = ::sqrt(piv);
            for (uint32_t base = beg; base < end; base += 64) {
                uint32_t k = base + lane;
                if (k < end) {
                    double raw = ld_l2(l + k);
                    l[k] = (k == beg) ? d : raw / d;
                }
            }
        }
        part = wave_sum64(part);
        if (lane == 0) b[j] = (ld_l2(b + j) - part) / d;
        __syncthreads();  // waits for the stores: the next column of this list may read them
    }
    if (lane == 0 && bad) atomicOr(flag, 1u);
}

// K4c: Lt x = y by column gathers, lists walked backwards: x_j = (y_j - sum_{i>j} L_ij x_i) / L_jj
// reads only ancestors of j — the levels run from the top down. x overwrites y.
__global__ __launch_bounds__(64) void sp_backward_kernel(SpChol c, ColLists cl, const double* __restrict__ l,
                                                         double* __restrict__ b, const SpLm* __restrict__ st) {
    if (st && st->done) return;
    const int lane = threadIdx.x;
    const uint32_t list = cl.first + blockIdx.x;
    for (uint32_t q = cl.ptr[list + 1]; q-- > cl.ptr[list];) {
        const uint32_t j = cl.cols[q];
        const uint32_t beg = c.lcolptr[j], end = c.lcolptr[j + 1];
        double part = 0.0;
        for (uint32_t k = beg + 1 + lane; k < end; k += 64) part = fma(l[k], ld_l2(b + c.lrow[k]), part);
        part = wave_sum64(part);
        if (lane == 0) b[j] = (ld_l2(b + j) - part) / l[beg];
        __syncthreads();
    }
}

// ---- LM control on the device (lm.rs:108-191). The host used to read three scalars back after every trial and
// decide; now the decisions are taken by a one-thread kernel on this state, every kernel of a trial looks at it (which
// buffer is current, lambda, whether anything is left to do), and the host enqueues trials in chunks without waiting —
// one read-back per chunk instead of a synchronisation per trial.
__global__ void sp_lm_init_kernel(SpLm* st, fx_lm_opts o) {
    const double sse = st->sse;  // written by the start point's sum of squares
    st->sse_start = sse;
    st->lambda = o.lambda0;
    st->cur = 0;
    st->accepted = st->trials = st->outer = 0;
    st->exit_code = FX_EXIT_MAX_OUTER;
    st->need_form = 1;
    st->flag = 0;
    st->done = 0;
    if (!(sse == sse) || !(sse < 1.0e300)) {
        st->exit_code = FX_EXIT_NAN;
        st->done = 1;
    } else if (o.max_outer == 0) {
        st->done = 1;
    } else if (sse < o.sse_tol) {  // lm.rs:110-112
        st->exit_code = FX_EXIT_SSE;
        st->done = 1;
    } else if (o.max_trials == 0) {
        st->exit_code = FX_EXIT_TRIAL_CAP;
        st->done = 1;
    }
}
// after a trial: accept / reject / stop, exactly the host loop this replaces
__global__ void sp_lm_control_kernel(SpLm* st, fx_lm_opts o) {
    if (st->done) return;
    st->trials += 1;
    bool check_cap = true;
    if (st->flag) {  // lm.rs:134-137
        st->lambda *= o.singular_factor;
        if (!(st->lambda < 1.0e300)) {
            st->exit_code = FX_EXIT_NAN;
            st->done = 1;
        }
    } else {
        const double dn2 = st->dn2, sse_t = st->sse_t, sse = st->sse;
        if (!(dn2 == dn2)) {
            st->exit_code = FX_EXIT_NAN;
            st->done = 1;
            check_cap = false;
        } else if (dn2 < o.step_tol) {  // lm.rs:139-142
            st->exit_code = FX_EXIT_STEP;
            st->done = 1;
            check_cap = false;
        } else if (sse_t < sse) {  // accept, lm.rs:151-186
            double lam = st->lambda * o.accept_factor;
            if (lam < o.lambda_min) lam = o.lambda_min;
            st->lambda = lam;
            st->cur ^= 1u;
            st->accepted += 1;
            const double rel = (sse - sse_t) / sse;
            st->sse = sse_t;
            if (rel <= o.ftol) {
                st->exit_code = FX_EXIT_FTOL;
                st->done = 1;
                check_cap = false;
            } else {
                st->need_form = 1;
                st->outer += 1;
                if (st->outer >= o.max_outer) {
                    st->done = 1;  // exit_code is still FX_EXIT_MAX_OUTER
                    check_cap = false;
                } else if (sse_t < o.sse_tol) {
                    st->exit_code = FX_EXIT_SSE;
                    st->done = 1;
                    check_cap = false;
                }
            }
        } else {  // reject, lm.rs:187-190
            st->lambda *= o.reject_factor;
            if (!(sse_t == sse_t) && !(st->lambda < 1.0e300)) {
                st->exit_code = FX_EXIT_NAN;
                st->done = 1;
                check_cap = false;
            }
        }
    }
    if (check_cap && !st->done && st->trials >= o.max_trials) {
        st->exit_code = FX_EXIT_TRIAL_CAP;
        st->done = 1;
    }
    st->flag = 0;
}
__global__ void sp_form_a_dc_kernel(const uint32_t* __restrict__ pair_ptr, const uint32_t* __restrict__ pairs, SpBufs bf, uint32_t nnz_a,
                                    double* __restrict__ a, const SpLm* __restrict__ st) {
    if (st->done || !st->need_form) return;
    const double* jvals = bf.j[st->cur];
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nnz_a) return;
    double s = 0.0;
    for (uint32_t p = pair_ptr[k]; p < pair_ptr[k + 1]; ++p) s += jvals[pairs[2 * p]] * jvals[pairs[2 * p + 1]];
    a[k] = s;
}
__global__ void sp_rhs_dc_kernel(const uint32_t* __restrict__ cptr, const uint32_t* __restrict__ cidx, const uint32_t* __restrict__ crow,
                                 SpBufs bf, uint32_t nv, double* __restrict__ b, const SpLm* __restrict__ st) {
    if (st->done || !st->need_form) return;
    const double* jvals = bf.j[st->cur];
    const double* r = bf.r[st->cur];
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nv) return;
    double s = 0.0;
    for (uint32_t p = cptr[c]; p < cptr[c + 1]; ++p) s += jvals[cidx[p]] * -r[crow[p]];
    b[c] = s;
}
// delta <- right-hand side (the factorization kernel solves in place); the formed flag is cleared
__global__ void sp_begin_dc_kernel(const double* __restrict__ rhs, double* __restrict__ delta, uint32_t nv, SpLm* __restrict__ st) {
    if (st->done) return;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nv) delta[i] = rhs[i];
}
__global__ void sp_formed_dc_kernel(SpLm* st) { st->need_form = 0; }
__global__ void sp_trial_dc_kernel(const uint32_t* __restrict__ fvar, const uint32_t* __restrict__ perm, uint32_t nv,
                                   const double* __restrict__ delta, SpBufs bf, const SpLm* __restrict__ st) {
    if (st->done) return;
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nv) return;
    const uint32_t v = fvar[perm[k]];
    bf.xs[st->cur ^ 1u][v] = bf.xs[st->cur][v] + delta[k];
}
// out[0] = sum v[i]^2 over the buffer of generation cur ^ sel (fixed-shape tree: deterministic)
__global__ __launch_bounds__(1024) void sp_sumsq_dc_kernel(const double* v0, const double* v1, uint32_t sel, uint32_t n, double* __restrict__ out,
                                                           const SpLm* __restrict__ st) {
    if (st->done) return;
    const double* v = ((st->cur ^ sel) & 1u) ? v1 : v0;
    __shared__ double part[1024];
    double s = 0.0;
    for (uint32_t i = threadIdx.x; i < n; i += 1024) s += v[i] * v[i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = part[0];
}

// ---- FX_STEP_CHOLESKY_REFINED: one refinement step on the least-squares problem itself (corrected semi-normal
// equations), as in the fused kernel: t = -r - J delta from the Jacobian rows, (JtJ + lambda I) e = Jt t - lambda delta
// with the factor at hand, delta += e. All sums in a fixed order.
__global__ void sp_refine_t_kernel(const uint32_t* __restrict__ jrow_ptr, const uint32_t* __restrict__ jcol, SpBufs bf,
                                   const double* __restrict__ delta, uint32_t m, double* __restrict__ t, const SpLm* __restrict__ st) {
    if (st->done) return;
    const double* jvals = bf.j[st->cur];
    const double* r = bf.r[st->cur];
    uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    double acc = -r[row];
    for (uint32_t p = jrow_ptr[row]; p < jrow_ptr[row + 1]; ++p) acc -= jvals[p] * delta[jcol[p]];
    t[row] = acc;
}
__global__ void sp_refine_rhs_kernel(const uint32_t* __restrict__ cptr, const uint32_t* __restrict__ cidx,
                                     const uint32_t* __restrict__ crow, SpBufs bf, const double* __restrict__ t,
                                     const double* __restrict__ delta, uint32_t nv, double* __restrict__ out, const SpLm* __restrict__ st) {
    if (st->done) return;
    const double* jvals = bf.j[st->cur];
    const double lambda = st->lambda;
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nv) return;
    double s = 0.0;
    for (uint32_t p = cptr[c]; p < cptr[c + 1]; ++p) s += jvals[cidx[p]] * t[crow[p]];
    out[c] = s - lambda * delta[c];
}
__global__ void sp_add_kernel(const double* __restrict__ e, uint32_t n, double* __restrict__ x, const SpLm* __restrict__ st) {
    if (st->done) return;
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) x[i] += e[i];
}
// L y = b with the stored factor (the factorization kernel does this sweep on the fly; the refinement needs it again):
// row gathers, lists in ascending order, levels bottom up. y overwrites b.
__global__ __launch_bounds__(64) void sp_forward_kernel(SpChol c, SpRowsOfL lr, ColLists cl, const double* __restrict__ l,
                                                        double* __restrict__ b, const SpLm* __restrict__ st) {
    if (st && st->done) return;
    const int lane = threadIdx.x;
    const uint32_t list = cl.first + blockIdx.x;
    for (uint32_t q = cl.ptr[list]; q < cl.ptr[list + 1]; ++q) {
        const uint32_t j = cl.cols[q];
        double part = 0.0;
        for (uint32_t p = lr.rptr[j] + lane; p < lr.rptr[j + 1]; p += 64) part = fma(l[lr.ridx[p]], ld_l2(b + lr.rcol[p]), part);
        part = wave_sum64(part);
        if (lane == 0) b[j] = (ld_l2(b + j) - part) / l[c.lcolptr[j]];
        __syncthreads();
    }
}

// trial point: xs_dst[fvar[perm[k]]] = xs_src[...] + delta[k]
__global__ void sp_trial_kernel(const uint32_t* __restrict__ fvar, const uint32_t* __restrict__ perm, uint32_t nv,
                                const double* __restrict__ delta, const double* __restrict__ xs_src,
                                double* __restrict__ xs_dst) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nv) return;
    uint32_t vi = fvar[perm[k]];
    xs_dst[vi] = xs_src[vi] + delta[k];
}

__global__ void sp_copy_free_kernel(const uint32_t* __restrict__ fvar, uint32_t nv, const double* __restrict__ src,
                                    double* __restrict__ dst) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < nv) dst[fvar[k]] = src[fvar[k]];
}

// assemble/mod.rs:161-166
__global__ void sp_writeback_kernel(const uint32_t* __restrict__ fvar, uint32_t nv, const double* __restrict__ xs,
                                    const double* __restrict__ scal, int do_scale, double* __restrict__ vars_out) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nv) return;
    uint32_t vi = fvar[k];
    vars_out[vi] = do_scale ? scal[0] * xs[vi] : xs[vi];
}

// residual of every expression on unscaled variables (constraints/mod.rs:96-109)
template <bool POSE = false>
__global__ void sp_identity_residual_kernel(SpRows rows, const double* __restrict__ x, double* __restrict__ out) {
    uint32_t e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows.net) return;
    int tag = rows.tag[e] & 0x7F;
    ushort4 f4 = reinterpret_cast<const ushort4*>(rows.idx)[e];
    uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
    uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    expand_vars<POSE>(tag, ff, vars8);
    double v[8], g[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = x[vars8[q]];
    out[e] = eval_expression<double, false, false, POSE>(tag, v, rows.param[e], g);
}

// ------------------------------------------------------------------------------------------------
// host: structure
// ------------------------------------------------------------------------------------------------
template <typename T>
struct DevArr {
    T* p = nullptr;
    size_t n = 0;
};

// Device memory of one System's solve: a few large hipMalloc chunks handed out by bumping a pointer.
// hipMalloc / hipFree synchronise the whole device, and a SinglePass solve creates dozens of small arrays
// for each of thousands of blocks — and several Systems may be solved at once from different host
// threads — so the per-array calls are what must go.
struct Arena {
    struct Chunk { char* base; size_t size; };
    struct Mark { size_t cur, used; };
    std::vector<Chunk> chunks;
    size_t cur = 0, used = 0;
    void* take(size_t bytes, hipError_t& err) {
        bytes = (std::max<size_t>(bytes, 1) + 255u) & ~size_t(255);
        for (; cur < chunks.size(); ++cur, used = 0)
            if (used + bytes <= chunks[cur].size) {
                void* p = chunks[cur].base + used;
                used += bytes;
                return p;
            }
        void* d = nullptr;
        const size_t size = std::max<size_t>(bytes, size_t(4) << 20);
        err = hipMalloc(&d, size);
        if (err != hipSuccess) return nullptr;
        chunks.push_back({static_cast<char*>(d), size});
        cur = chunks.size() - 1;
        used = bytes;
        return d;
    }
    Mark mark() const { return {cur, used}; }
    void reset(Mark m) { cur = m.cur; used = m.used; }
    ~Arena() {
        for (auto& c : chunks) (void)hipFree(c.base);
    }
};

struct Pool {  // the arrays of one scope (System or block); released together when it ends (stream synced by then)
    Arena* arena;
    Arena::Mark start;
    hipStream_t stream = nullptr;
    hipError_t err = hipSuccess;
    explicit Pool(Arena* a) : arena(a), start(a->mark()) {}
    Pool(const Pool&) = delete;
    Pool& operator=(const Pool&) = delete;
    template <typename T>
    T* up(const std::vector<T>& h) {
        T* d = alloc<T>(h.size());
        if (d && !h.empty() && err == hipSuccess)
            err = hipMemcpyAsync(d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, stream);
        return d;
    }
    template <typename T>
    T* alloc(size_t n) {
        if (err != hipSuccess) return nullptr;
        return static_cast<T*>(arena->take(n * sizeof(T), err));
    }
    ~Pool() { arena->reset(start); }
};

// reverse Cuthill-McKee order of the column graph of A (adjacency given as sorted lists)
std::vector<uint32_t> rcm_order(const std::vector<std::vector<uint32_t>>& adj) {
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
std::vector<uint32_t> nd_order(const std::vector<std::vector<uint32_t>>& adj, uint32_t leaf = 48) {
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
    std::vector<uint8_t> coop;                         // per column: sum its gather lists cooperatively
    std::vector<uint32_t> rptr, ridx, rcol;            // strictly lower part of L by rows
    std::vector<uint32_t> list_ptr, list_cols;         // work lists of the elimination-tree schedule
    std::vector<uint32_t> level_ptr;                   // lists of level v: [level_ptr[v], level_ptr[v+1])
};

// Structure of one block on the device (index arrays only; what plan_component produced, uploaded).
struct BlockOnDevice {
    ComponentPlan P;
    uint32_t* d_fvar = nullptr;
    uint32_t* d_perm = nullptr;
    SpJac jac{};
    uint32_t *d_apair_ptr = nullptr, *d_apairs = nullptr, *d_cptr = nullptr, *d_cidx = nullptr, *d_crow = nullptr;
    uint32_t* d_jcol = nullptr;
    SpChol chol{};
    SpRowsOfL lrows{};
    ColLists lists{};
    double plan_ms = 0.0;
};

}  // namespace

// Plans of a resident System (one per decomposer mode), kept between solves: the structure never changes
// on a resident batch — only start values and parameters do — so ordering, symbolic factorisation,
// gather lists and their upload (12 ms for cfg2) are paid once.
struct SparsePlanCache {
    Arena arena;                                     // holds the blocks' index arrays
    std::unique_ptr<Pool> pool;
    std::vector<UnitList> units;                     // per visited component
    std::vector<std::unique_ptr<BlockOnDevice>> blocks;  // in visiting order
    bool ready = false;
};
SparsePlanCache* sparse_cache_new() { return new SparsePlanCache(); }
void sparse_cache_free(SparsePlanCache* c) { delete c; }
bool sparse_cache_ready(const SparsePlanCache* c) { return c && c->ready; }

namespace {

// Builds every index structure of one component. `colof[v]` = free column of system variable v
// (ascending rank among the component's free variables) or -1.
void plan_component(const fx_batch* b, uint32_t s, const std::vector<uint32_t>& rows,
                    const std::vector<uint32_t>& fvar, ComponentPlan& P) {
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
    P.perm = nd_order(adj);
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

    // --- how a column sums its gather lists: one lane per entry (cost ~ the longest list), or the whole
    // wavefront on one list after the other (cost ~ sum over entries of ceil(len / 64) + a reduction)
    P.coop.assign(P.nv, 0);
    for (uint32_t j = 0; j < P.nv; ++j) {
        uint64_t longest = 0, coop_cost = 0;
        for (uint32_t k = P.lcolptr[j]; k < P.lcolptr[j + 1]; ++k) {
            uint64_t len = P.lpair_ptr[k + 1] - P.lpair_ptr[k];
            longest = std::max(longest, len);
            coop_cost += (len + 63) / 64 + 4;
        }
        P.coop[j] = (P.lcolptr[j + 1] - P.lcolptr[j] <= 64u && longest > 2 * coop_cost) ? 1 : 0;
    }

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

    // --- schedule: a column depends only on its descendants in the elimination tree (parent = first
    // sub-diagonal row). Subtrees whose work fits under a cap become level-0 lists, one wavefront each.
    // The columns above them are cut into chains (a column joins the chain of its only child above the
    // cap; a column where several such chains meet starts a new one), and a chain's level is one more
    // than the deepest list feeding it. Levels run as consecutive launches, lists of a level in
    // parallel. The cap minimising the critical path (largest subtree + per level the longest chain +
    // a launch overhead per level) is picked from a geometric ladder.
    std::vector<uint64_t> work(P.nv, 0), subtree(P.nv, 0);
    uint64_t total = 0;
    for (uint32_t j = 0; j < P.nv; ++j) {
        // critical-path cost of the column in "list elements": lanes work in parallel, so what counts is
        // the longest list (per-lane mode) or the strided passes over all lists (cooperative mode)
        uint64_t longest = 0, coop_cost = 0;
        for (uint32_t k = P.lcolptr[j]; k < P.lcolptr[j + 1]; ++k) {
            uint64_t len = P.lpair_ptr[k + 1] - P.lpair_ptr[k];
            longest = std::max(longest, len);
            coop_cost += (len + 63) / 64 + 4;
        }
        uint64_t w = 8 + (P.coop[j] ? coop_cost : longest);  // 8 = per-column latency floor
        work[j] = w;
        total += w;
    }
    constexpr uint32_t NOPARENT = 0xFFFFFFFFu;
    constexpr uint64_t LAUNCH_COST = 24;  // in the same units: about three columns
    std::vector<uint32_t> parent(P.nv, NOPARENT);
    for (uint32_t j = 0; j < P.nv; ++j)
        if (lcol[j].size() > 1) parent[j] = lcol[j][1];
    for (uint32_t j = 0; j < P.nv; ++j) subtree[j] = work[j];
    for (uint32_t j = 0; j < P.nv; ++j)
        if (parent[j] != NOPARENT) subtree[parent[j]] += subtree[j];  // children come before parents

    std::vector<uint32_t> list_of(P.nv), list_level, upper_children(P.nv), feeder(P.nv), below(P.nv);
    std::vector<uint64_t> list_work;
    // builds the lists for a cap; returns the critical-path estimate
    auto build = [&](uint64_t cap) -> uint64_t {
        list_level.clear();
        list_work.clear();
        std::fill(upper_children.begin(), upper_children.end(), 0u);
        std::fill(below.begin(), below.end(), 0u);  // deepest level among the lists feeding column j, plus one
        for (uint32_t j = 0; j < P.nv; ++j)
            if (subtree[j] > cap && parent[j] != NOPARENT) {
                upper_children[parent[j]]++;
                feeder[parent[j]] = j;
            }
        // level-0 lists: subtrees under the cap, numbered from their roots downwards
        for (uint32_t j = P.nv; j-- > 0;) {
            if (subtree[j] > cap) continue;
            uint32_t pa = parent[j];
            if (pa == NOPARENT || subtree[pa] > cap) {
                list_of[j] = (uint32_t)list_level.size();
                list_level.push_back(0);
                list_work.push_back(subtree[j]);
                if (pa != NOPARENT) below[pa] = std::max(below[pa], 1u);
            } else {
                list_of[j] = list_of[pa];
            }
        }
        // chains above the cap, bottom-up (children have smaller numbers)
        for (uint32_t j = 0; j < P.nv; ++j) {
            if (subtree[j] <= cap) continue;
            uint32_t q;
            if (upper_children[j] == 1) {
                // extends its only upper child's chain (its other children are level-0 subtrees, and a
                // chain is never below level 1)
                q = list_of[feeder[j]];
                list_work[q] += work[j];
            } else {
                uint32_t lvl = below[j];
                q = (uint32_t)list_level.size();
                list_level.push_back(lvl ? lvl : 1u);
                list_work.push_back(work[j]);
            }
            list_of[j] = q;
            if (parent[j] != NOPARENT) below[parent[j]] = std::max(below[parent[j]], list_level[q] + 1);
        }
        uint32_t nlevels = 0;
        for (uint32_t v : list_level) nlevels = std::max(nlevels, v + 1);
        std::vector<uint64_t> longest(nlevels, 0);
        for (size_t q = 0; q < list_level.size(); ++q) longest[list_level[q]] = std::max(longest[list_level[q]], list_work[q]);
        uint64_t cost = 0;
        for (uint64_t w : longest) cost += w + LAUNCH_COST;
        return cost;
    };
    uint64_t best_cap = total, best_cost = ~0ull;
    for (uint64_t cap = total; cap >= 32; cap = cap * 3 / 4) {
        uint64_t cost = build(cap);
        if (cost < best_cost) {
            best_cost = cost;
            best_cap = cap;
        }
    }
    build(best_cap);
    // lists sorted by level (stable), columns ascending within a list
    const uint32_t nlists = (uint32_t)list_level.size();
    uint32_t nlevels = 0;
    for (uint32_t v : list_level) nlevels = std::max(nlevels, v + 1);
    P.level_ptr.assign((size_t)nlevels + 1, 0);
    for (uint32_t v : list_level) P.level_ptr[v + 1]++;
    for (uint32_t v = 0; v < nlevels; ++v) P.level_ptr[v + 1] += P.level_ptr[v];
    std::vector<uint32_t> new_id(nlists);
    {
        std::vector<uint32_t> fill(P.level_ptr.begin(), P.level_ptr.end() - 1);
        for (uint32_t q = 0; q < nlists; ++q) new_id[q] = fill[list_level[q]]++;
    }
    P.list_ptr.assign((size_t)nlists + 1, 0);
    for (uint32_t j = 0; j < P.nv; ++j) P.list_ptr[new_id[list_of[j]] + 1]++;
    for (uint32_t q = 0; q < nlists; ++q) P.list_ptr[q + 1] += P.list_ptr[q];
    P.list_cols.assign(P.nv, 0);
    {
        std::vector<uint32_t> fill(P.list_ptr.begin(), P.list_ptr.end() - 1);
        for (uint32_t j = 0; j < P.nv; ++j) P.list_cols[fill[new_id[list_of[j]]]++] = j;
    }
}

inline dim3 grid_for(uint32_t n, uint32_t block = 256) { return dim3((n + block - 1) / block ? (n + block - 1) / block : 1); }

}  // namespace

// ------------------------------------------------------------------------------------------------
// host: LM driver for one System (all of its components), numerics on the device
// ------------------------------------------------------------------------------------------------
hipError_t sparse_solve_system(const fx_batch* b, uint32_t s, const LmParams& prm, hipStream_t stream,
                               double* d_vars_out /* device, n_vars of the System */, fx_result* result,
                               SparsePlanCache* cache) {
    const bool reuse = cache && cache->ready;  // structure from an earlier solve of this resident System
    if (cache && !cache->ready) {  // first solve (or an earlier attempt failed half-way): start clean
        cache->units.clear();
        cache->blocks.clear();
    }
    if (cache && !cache->pool) cache->pool.reset(new Pool(&cache->arena));
    if (cache && !reuse) cache->pool->stream = stream;  // the index arrays go up on the filling call's stream
    size_t comp_at = 0, block_at = 0;
    const bool single_pass = (prm.mode & MODE_UNITS) != 0;
    const bool lbfgs = (prm.mode & MODE_LBFGS) != 0;
    const bool trace = std::getenv("FIKSI_AMD_TRACE") != nullptr;  // diagnostics on stderr
    const uint32_t v0 = b->var_off[s], nvt = b->var_off[s + 1] - v0;
    const uint32_t e0 = b->expr_off[s], net = b->expr_off[s + 1] - e0;
    const fx_lm_opts o = prm.lm;
    const int do_scale = (prm.mode & 1u) ? 1 : 0;
    Arena arena;
    Pool pool(&arena);
    pool.stream = stream;

    // ---- System-wide device data
    std::vector<uint8_t> tags(b->expr_tag + e0, b->expr_tag + e0 + net);
    std::vector<uint16_t> idx16(4 * (size_t)net);
    for (size_t q = 0; q < idx16.size(); ++q) idx16[q] = (uint16_t)b->expr_idx[4 * (size_t)e0 + q];
    std::vector<double> params(b->expr_param + e0, b->expr_param + e0 + net);
    std::vector<double> vars0(b->vars + v0, b->vars + v0 + nvt);
    SpRows rows;
    rows.tag = pool.up(tags);
    rows.idx = pool.up(idx16);
    rows.param = pool.up(params);
    rows.sparam = pool.alloc<double>(net);
    rows.net = net;
    rows.has_pose = 0;
    for (uint8_t t : tags)
        if ((t & 0x7F) >= FX_TAG_POSE_X) rows.has_pose = 1;
    double* d_vars0 = pool.up(vars0);
    double* d_xs[2] = {pool.alloc<double>(nvt), pool.alloc<double>(nvt)};
    double* d_snap = pool.alloc<double>(nvt);  // pre-solve snapshot (quirk Q2)
    double* d_scal = pool.alloc<double>(8);       // scale, 1/scale, sse, dn2, sse_unscaled
    double* d_runs = pool.alloc<double>(std::max(net, 1u));
    if (pool.err != hipSuccess) return pool.err;

    hipLaunchKernelGGL(sp_scale_kernel, dim3(1), dim3(256), 0, stream, d_vars0, nvt, rows, d_scal, do_scale);
    hipLaunchKernelGGL(sp_init_kernel, grid_for(std::max(nvt, net)), dim3(256), 0, stream, d_vars0, nvt, rows, d_scal,
                       d_xs[0], d_xs[1], do_scale);
    hipError_t e = hipMemcpyAsync(d_vars_out, d_vars0, nvt * sizeof(double), hipMemcpyDeviceToDevice, stream);
    if (e != hipSuccess) return e;

    // ---- components in order (assemble/mod.rs:81)
    uint32_t ncomp = 0;
    for (uint32_t i = 0; i < nvt; ++i) {
        uint16_t c = b->var_comp ? b->var_comp[v0 + i] : 0;
        if (c != FX_NO_COMPONENT) ncomp = std::max<uint32_t>(ncomp, c + 1u);
    }
    for (uint32_t i = 0; i < net; ++i) {
        uint16_t c = b->expr_comp ? b->expr_comp[e0 + i] : 0;
        if (c != FX_NO_COMPONENT) ncomp = std::max<uint32_t>(ncomp, c + 1u);
    }
    fx_result res{};
    res.exit = FX_EXIT_SSE;
    uint32_t rng = 42u;  // Rng::from_seed(42), shared by the components (:47)
    double host3[4];

    Incidence inc;
    std::unique_ptr<SinglePassDecomposer> decomposer;

    for (uint32_t c = 0; c < ncomp; ++c) {
        std::vector<uint32_t> crow_ids, fvar;
        bool any_var = false;
        for (uint32_t i = 0; i < nvt; ++i) {
            uint16_t vc = b->var_comp ? b->var_comp[v0 + i] : 0;
            if (vc != c) continue;
            any_var = true;
            if (!b->var_fixed[v0 + i]) fvar.push_back(i);
        }
        if (!any_var) continue;
        for (uint32_t i = 0; i < net; ++i)
            if ((b->expr_comp ? b->expr_comp[e0 + i] : 0) == c) crow_ids.push_back(i);

        // ---- the component's perturbation (:91-111), before any of its blocks
        if (prm.mode & 2u) {
            const uint32_t nfv = (uint32_t)fvar.size();
            if (nfv) {
                Pool tmp(&arena);
                tmp.stream = stream;
                uint32_t* d_all = tmp.up(fvar);
                if (tmp.err != hipSuccess) return tmp.err;
                hipLaunchKernelGGL(sp_perturb_kernel, grid_for(nfv), dim3(256), 0, stream, d_all, nfv, rng, d_xs[0], d_xs[1]);
                e = hipStreamSynchronize(stream);  // tmp is released at the end of this scope
                if (e != hipSuccess) return e;
            }
            for (uint32_t k = 0; k < 2 * nfv; ++k) rng = rng * 1664525u + 1013904223u;  // integer bookkeeping only
        }
        e = hipMemcpyAsync(d_snap, d_xs[0], nvt * sizeof(double), hipMemcpyDeviceToDevice, stream);
        if (e != hipSuccess) return e;

        // ---- the blocks to solve: the whole component, or its SinglePass decomposition
        UnitList units_local;
        if (reuse) {
            // decided on the first solve
        } else if (single_pass) {
            if (!decomposer) {
                inc.build(nvt, net, b->expr_tag + e0, b->expr_idx + 4 * (size_t)e0);
                decomposer.reset(new SinglePassDecomposer(inc));
            }
            decomposer->run(fvar, units_local);
        } else {
            units_local.rows = crow_ids;
            units_local.vars = fvar;
            units_local.row_off.push_back((uint32_t)crow_ids.size());
            units_local.var_off.push_back((uint32_t)fvar.size());
        }
        if (cache && !reuse) cache->units.push_back(units_local);
        const UnitList& units = cache ? cache->units[comp_at] : units_local;
        comp_at += 1;
        res.ncomp += 1;
        res.exit = FX_EXIT_SSE;

        for (uint32_t u = 0; u < units.count(); ++u) {
        Pool pool(&arena);  // device memory of this block only
        pool.stream = stream;
        // the block's structure: planned and uploaded now, or kept from the first solve
        BlockOnDevice local_block;
        BlockOnDevice* blk = &local_block;
        if (reuse) {
            blk = cache->blocks[block_at].get();
        } else {
            if (cache) {
                cache->blocks.emplace_back(new BlockOnDevice());
                blk = cache->blocks.back().get();
            }
            Pool& sp = cache ? *cache->pool : pool;  // index arrays outlive the solve only when cached
            const auto t_plan0 = std::chrono::steady_clock::now();
            plan_component(b, s,
                           std::vector<uint32_t>(units.rows.begin() + units.row_off[u], units.rows.begin() + units.row_off[u + 1]),
                           std::vector<uint32_t>(units.vars.begin() + units.var_off[u], units.vars.begin() + units.var_off[u + 1]),
                           blk->P);
            blk->plan_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_plan0).count();
            const ComponentPlan& Q = blk->P;
            blk->d_fvar = sp.up(Q.fvar);
            blk->d_perm = sp.up(Q.perm);
            blk->jac.rows = sp.up(Q.rows);
            blk->jac.jrow_ptr = sp.up(Q.jrow_ptr);
            blk->jac.jslot = sp.up(Q.jslot);
            blk->jac.m = Q.m;
            blk->d_apair_ptr = sp.up(Q.apair_ptr);
            blk->d_apairs = sp.up(Q.apairs);
            blk->d_cptr = sp.up(Q.cptr);
            blk->d_cidx = sp.up(Q.cidx);
            blk->d_crow = sp.up(Q.crow);
            blk->d_jcol = sp.up(Q.jcol);
            blk->chol.lcolptr = sp.up(Q.lcolptr);
            blk->chol.lrow = sp.up(Q.lrow);
            blk->chol.l2a = sp.up(Q.l2a);
            blk->chol.lpair_ptr = sp.up(Q.lpair_ptr);
            blk->chol.lpairs = sp.up(Q.lpairs);
            blk->chol.lpair_k = sp.up(Q.lpair_k);
            blk->chol.coop = sp.up(Q.coop);
            blk->chol.nv = Q.nv;
            blk->lrows.rptr = sp.up(Q.rptr);
            blk->lrows.ridx = sp.up(Q.ridx);
            blk->lrows.rcol = sp.up(Q.rcol);
            blk->lists.ptr = sp.up(Q.list_ptr);
            blk->lists.cols = sp.up(Q.list_cols);
            blk->lists.first = 0;
            if (sp.err != hipSuccess) return sp.err;
        }
        block_at += 1;
        const ComponentPlan& P = blk->P;
        const uint32_t m = P.m, nv = P.nv;
        uint32_t* d_fvar = blk->d_fvar;
        uint32_t* d_perm = blk->d_perm;
        SpJac jac = blk->jac;
        jac.overwrite = lbfgs ? 1 : 0;
        uint32_t* d_apair_ptr = blk->d_apair_ptr;
        uint32_t* d_apairs = blk->d_apairs;
        uint32_t* d_cptr = blk->d_cptr;
        uint32_t* d_cidx = blk->d_cidx;
        uint32_t* d_crow = blk->d_crow;
        const SpChol chol = blk->chol;
        const SpRowsOfL lrows = blk->lrows;
        const ColLists lists = blk->lists;
        const auto t_plan1 = std::chrono::steady_clock::now();
        const uint32_t nlevels = (uint32_t)P.level_ptr.size() - 1;
        double* d_r[2] = {pool.alloc<double>(m), pool.alloc<double>(m)};
        double* d_j[2] = {pool.alloc<double>(P.nnz_j), pool.alloc<double>(P.nnz_j)};
        double* d_a = pool.alloc<double>(P.nnz_a);
        double* d_l = pool.alloc<double>(P.nnz_l);
        double* d_rhs = pool.alloc<double>(nv);
        double* d_delta = pool.alloc<double>(nv);
        const bool refined = !lbfgs && o.solver == FX_STEP_CHOLESKY_REFINED;
        double* d_t = refined ? pool.alloc<double>(std::max(m, 1u)) : nullptr;
        double* d_e = refined ? pool.alloc<double>(std::max(nv, 1u)) : nullptr;
        if (pool.err != hipSuccess) return pool.err;

        auto eval = [&](int buf, bool want_j, double* sse_out) -> hipError_t {
            if (m) {
                if (want_j)
                    hipLaunchKernelGGL((rows.has_pose ? sp_eval_kernel<true, true> : sp_eval_kernel<true, false>), grid_for(m), dim3(256), 0, stream, rows, jac, d_xs[buf], d_r[buf], d_j[buf]);
                else
                    hipLaunchKernelGGL((rows.has_pose ? sp_eval_kernel<false, true> : sp_eval_kernel<false, false>), grid_for(m), dim3(256), 0, stream, rows, jac, d_xs[buf], d_r[buf], d_j[buf]);
            }
            hipLaunchKernelGGL(sp_sumsq_kernel, dim3(1), dim3(1024), 0, stream, d_r[buf], m, d_scal + 2);
            hipError_t er = hipMemcpyAsync(sse_out, d_scal + 2, sizeof(double), hipMemcpyDeviceToHost, stream);
            if (er == hipSuccess) er = hipStreamSynchronize(stream);
            return er;
        };
        int cur = 0;
        double sse = 0.0;
        e = eval(0, true, &sse);
        if (e != hipSuccess) return e;
        const double sse_start = sse;
        uint32_t accepted = 0, trials = 0, exit_code = FX_EXIT_MAX_OUTER;
        if (lbfgs) {
            // ---- Optimizer::LBfgs (lbfgs.rs:20-193): the host runs the line-search state machine on two
            // scalars per evaluation, everything else is on the device
            double* d_grad = pool.alloc<double>(nv);
            double* d_dir = pool.alloc<double>(nv);
            double* d_s = pool.alloc<double>(5 * (size_t)nv);
            double* d_y = pool.alloc<double>(5 * (size_t)nv);
            double* d_rho = pool.alloc<double>(8);
            if (pool.err != hipSuccess) return pool.err;
            e = hipMemsetAsync(d_s, 0, 5 * (size_t)nv * sizeof(double), stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_y, 0, 5 * (size_t)nv * sizeof(double), stream);
            if (e == hipSuccess) e = hipMemsetAsync(d_rho, 0, 8 * sizeof(double), stream);
            if (e != hipSuccess) return e;
            auto gradient = [&](int buf) {
                if (nv) hipLaunchKernelGGL(sp_rhs_kernel<false>, grid_for(nv), dim3(256), 0, stream, d_cptr, d_cidx, d_crow, d_j[buf],
                                           d_r[buf], nv, d_grad);
            };
            trials = 1;
            double prev = sse;
            if (!(prev == prev)) {
                exit_code = FX_EXIT_NAN;
            } else if (prev < LbfgsConst::START_THRESHOLD) {
                exit_code = FX_EXIT_SSE;
            } else {
                gradient(0);
                for (uint32_t k = 0; k < LbfgsConst::MAX_ITERATIONS; ++k) {
                    const uint32_t h = k % 5u;
                    hipLaunchKernelGGL(sp_lbfgs_direction_kernel, dim3(1), dim3(1024), 0, stream, k, nv, d_s, d_y, d_rho, d_grad, d_dir);
                    e = hipMemcpyAsync(d_y + (size_t)h * nv, d_grad, nv * sizeof(double), hipMemcpyDeviceToDevice, stream);
                    if (e != hipSuccess) return e;
                    hipLaunchKernelGGL(sp_dot_kernel, dim3(1), dim3(1024), 0, stream, d_grad, d_dir, nv, d_scal + 6);
                    e = hipMemcpyAsync(host3, d_scal + 6, sizeof(double), hipMemcpyDeviceToHost, stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(stream);
                    if (e != hipSuccess) return e;
                    HzMachine hz;
                    double step = hz.start(prev, host3[0]);
                    HzParam acc_pt{0., 0., 0.};
                    for (;;) {  // calculate_phi (lbfgs.rs:270-284): xs[1] = xs[0] + step * dir
                        if (nv) {
                            hipLaunchKernelGGL(sp_scaled_copy_kernel, grid_for(nv), dim3(256), 0, stream, d_dir, step, nv, d_delta);
                            hipLaunchKernelGGL(sp_trial_kernel, grid_for(nv), dim3(256), 0, stream, d_fvar, d_perm, nv, d_delta, d_xs[0], d_xs[1]);
                        }
                        if (m) hipLaunchKernelGGL((rows.has_pose ? sp_eval_kernel<true, true> : sp_eval_kernel<true, false>), grid_for(m), dim3(256), 0, stream, rows, jac, d_xs[1], d_r[1], d_j[1]);
                        hipLaunchKernelGGL(sp_sumsq_kernel, dim3(1), dim3(1024), 0, stream, d_r[1], m, d_scal + 5);
                        gradient(1);
                        hipLaunchKernelGGL(sp_dot_kernel, dim3(1), dim3(1024), 0, stream, d_grad, d_dir, nv, d_scal + 6);
                        e = hipMemcpyAsync(host3, d_scal + 5, 2 * sizeof(double), hipMemcpyDeviceToHost, stream);
                        if (e == hipSuccess) e = hipStreamSynchronize(stream);
                        if (e != hipSuccess) return e;
                        trials += 1;
                        if (hz.feed(HzParam{step, host3[0], host3[1]}, step, acc_pt)) break;
                    }
                    if (nv) hipLaunchKernelGGL(sp_copy_free_kernel, grid_for(nv), dim3(256), 0, stream, d_fvar, nv, d_xs[1], d_xs[0]);
                    hipLaunchKernelGGL(sp_lbfgs_update_kernel, dim3(1), dim3(1024), 0, stream, h, nv, acc_pt.p, d_dir, d_grad, d_s, d_y, d_rho);
                    accepted += 1;
                    sse = acc_pt.phi;
                    if (hz.capped) {
                        exit_code = FX_EXIT_TRIAL_CAP;
                        break;
                    }
                    if (!(acc_pt.phi == acc_pt.phi)) {
                        exit_code = FX_EXIT_NAN;
                        break;
                    }
                    if (std::fabs(prev - acc_pt.phi) < LbfgsConst::CONVERGENCE_THRESHOLD) {
                        exit_code = FX_EXIT_FTOL;
                        break;
                    }
                    if (acc_pt.phi < LbfgsConst::RESIDUAL_THRESHOLD) {
                        exit_code = FX_EXIT_SSE;
                        break;
                    }
                    prev = acc_pt.phi;
                }
            }
        } else {
        // ---- Levenberg-Marquardt (lm.rs:108-191), controlled on the device: the host enqueues trials in chunks and
        // reads the state back once per chunk
        SpLm* d_lm = pool.alloc<SpLm>(1);
        if (pool.err != hipSuccess) return pool.err;
        SpBufs bf{{d_xs[0], d_xs[1]}, {d_r[0], d_r[1]}, {d_j[0], d_j[1]}};
        e = hipMemcpyAsync(&d_lm->sse, d_scal + 2, sizeof(double), hipMemcpyDeviceToDevice, stream);  // the start point's SSE
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(sp_lm_init_kernel, dim3(1), dim3(1), 0, stream, d_lm, o);
        auto enqueue_trial = [&]() {
            if (P.nnz_a) hipLaunchKernelGGL(sp_form_a_dc_kernel, grid_for(P.nnz_a), dim3(256), 0, stream, d_apair_ptr, d_apairs, bf, P.nnz_a, d_a, d_lm);
            if (nv) hipLaunchKernelGGL(sp_rhs_dc_kernel, grid_for(nv), dim3(256), 0, stream, d_cptr, d_cidx, d_crow, bf, nv, d_rhs, d_lm);
            hipLaunchKernelGGL(sp_formed_dc_kernel, dim3(1), dim3(1), 0, stream, d_lm);
            if (nv) hipLaunchKernelGGL(sp_begin_dc_kernel, grid_for(nv), dim3(256), 0, stream, d_rhs, d_delta, nv, d_lm);
            for (uint32_t v = 0; v < nlevels; ++v) {
                ColLists cl = lists;
                cl.first = P.level_ptr[v];
                hipLaunchKernelGGL(sp_factor_forward_kernel, dim3(P.level_ptr[v + 1] - P.level_ptr[v]), dim3(64), 0, stream,
                                   chol, lrows, cl, d_a, 0.0, d_l, d_delta, &d_lm->flag, d_lm);
            }
            for (uint32_t v = nlevels; v-- > 0;) {
                ColLists cl = lists;
                cl.first = P.level_ptr[v];
                hipLaunchKernelGGL(sp_backward_kernel, dim3(P.level_ptr[v + 1] - P.level_ptr[v]), dim3(64), 0, stream, chol, cl, d_l, d_delta, d_lm);
            }
            if (refined && nv && m) {
                hipLaunchKernelGGL(sp_refine_t_kernel, grid_for(m), dim3(256), 0, stream, jac.jrow_ptr, blk->d_jcol, bf, d_delta, m, d_t, d_lm);
                hipLaunchKernelGGL(sp_refine_rhs_kernel, grid_for(nv), dim3(256), 0, stream, d_cptr, d_cidx, d_crow, bf, d_t, d_delta, nv, d_e, d_lm);
                for (uint32_t v = 0; v < nlevels; ++v) {
                    ColLists cl = lists;
                    cl.first = P.level_ptr[v];
                    hipLaunchKernelGGL(sp_forward_kernel, dim3(P.level_ptr[v + 1] - P.level_ptr[v]), dim3(64), 0, stream, chol, lrows, cl, d_l, d_e, d_lm);
                }
                for (uint32_t v = nlevels; v-- > 0;) {
                    ColLists cl = lists;
                    cl.first = P.level_ptr[v];
                    hipLaunchKernelGGL(sp_backward_kernel, dim3(P.level_ptr[v + 1] - P.level_ptr[v]), dim3(64), 0, stream, chol, cl, d_l, d_e, d_lm);
                }
                hipLaunchKernelGGL(sp_add_kernel, grid_for(nv), dim3(256), 0, stream, d_e, nv, d_delta, d_lm);
            }
            hipLaunchKernelGGL(sp_sumsq_dc_kernel, dim3(1), dim3(1024), 0, stream, d_delta, d_delta, 0u, nv, &d_lm->dn2, d_lm);
            if (nv) hipLaunchKernelGGL(sp_trial_dc_kernel, grid_for(nv), dim3(256), 0, stream, d_fvar, d_perm, nv, d_delta, bf, d_lm);
            if (m) hipLaunchKernelGGL((rows.has_pose ? sp_eval_dc_kernel<true> : sp_eval_dc_kernel<false>), grid_for(m), dim3(256), 0, stream, rows, jac, d_xs[0], d_xs[1], d_r[0], d_r[1], d_j[0], d_j[1], d_lm);
            hipLaunchKernelGGL(sp_sumsq_dc_kernel, dim3(1), dim3(1024), 0, stream, d_r[0], d_r[1], 1u, m, &d_lm->sse_t, d_lm);
            hipLaunchKernelGGL(sp_lm_control_kernel, dim3(1), dim3(1), 0, stream, d_lm, o);
        };
        SpLm h_lm{};
        for (uint32_t chunk = 4;; chunk = std::min<uint32_t>(2 * chunk, 16)) {
            e = hipMemcpyAsync(&h_lm, d_lm, sizeof(SpLm), hipMemcpyDeviceToHost, stream);
            if (e == hipSuccess) e = hipStreamSynchronize(stream);
            if (e != hipSuccess) return e;
            if (h_lm.done) break;
            for (uint32_t t = 0; t < chunk; ++t) enqueue_trial();
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
        cur = (int)h_lm.cur;
        sse = h_lm.sse;
        accepted = h_lm.accepted;
        trials = h_lm.trials;
        exit_code = h_lm.exit_code;
        }  // optimizer
        if (nv) {
            hipLaunchKernelGGL(sp_writeback_kernel, grid_for(nv), dim3(256), 0, stream, d_fvar, nv, d_xs[cur], d_scal, do_scale, d_vars_out);
            if (single_pass) {
                // SinglePass also updates the working vector (:201-207): later blocks see this one
                hipLaunchKernelGGL(sp_copy_free_kernel, grid_for(nv), dim3(256), 0, stream, d_fvar, nv, d_xs[cur], d_xs[cur ^ 1]);
            } else {
                // accepted point -> output only; later components see the pre-solve snapshot (quirk Q2)
                hipLaunchKernelGGL(sp_copy_free_kernel, grid_for(nv), dim3(256), 0, stream, d_fvar, nv, d_snap, d_xs[0]);
                hipLaunchKernelGGL(sp_copy_free_kernel, grid_for(nv), dim3(256), 0, stream, d_fvar, nv, d_snap, d_xs[1]);
            }
        }
        e = hipStreamSynchronize(stream);  // the block's device memory is released when `pool` goes out of scope
        if (e != hipSuccess) return e;
        if (trace) {
            for (size_t v = 0; v + 1 < P.level_ptr.size(); ++v) {
                uint32_t maxc = 0, totc = 0, maxl = 0, maxr = 0;
                for (uint32_t q = P.level_ptr[v]; q < P.level_ptr[v + 1]; ++q) {
                    uint32_t nc = P.list_ptr[q + 1] - P.list_ptr[q];
                    maxc = std::max(maxc, nc);
                    totc += nc;
                    for (uint32_t t = P.list_ptr[q]; t < P.list_ptr[q + 1]; ++t) {
                        uint32_t j = P.list_cols[t];
                        maxr = std::max(maxr, P.rptr[j + 1] - P.rptr[j]);
                        for (uint32_t k = P.lcolptr[j]; k < P.lcolptr[j + 1]; ++k) maxl = std::max(maxl, P.lpair_ptr[k + 1] - P.lpair_ptr[k]);
                    }
                }
                fprintf(stderr, "[fiksi_amd]   level %zu: %u lists, %u columns (longest list %u), longest product list %u, longest L row %u\n",
                        v, P.level_ptr[v + 1] - P.level_ptr[v], totc, maxc, maxl, maxr);
            }
            const auto t_end = std::chrono::steady_clock::now();
            auto ms = [](auto a, auto b2) { return std::chrono::duration<double, std::milli>(b2 - a).count(); };
            fprintf(stderr, "[fiksi_amd] sparse block: %u rows, %u cols, nnz J %u A %u L %u (%zu products, %u cooperative columns), "
                            "%zu lists in %zu levels; plan %.2f ms, upload+LM %.2f ms (%u trials)\n",
                    m, nv, P.nnz_j, P.nnz_a, P.nnz_l, P.lpairs.size() / 2,
                    (unsigned)std::count(P.coop.begin(), P.coop.end(), (uint8_t)1), P.list_ptr.size() - 1,
                    P.level_ptr.size() - 1, reuse ? 0.0 : blk->plan_ms,
                    ms(t_plan1, t_end), trials);
        }
        res.accepted += accepted;
        res.trials += trials;
        res.exit = exit_code;
        res.sse0 += sse_start;
        res.sse += sse;
        }  // blocks
    }

    // ---- post-solve check on the unscaled variables
    if (net) hipLaunchKernelGGL((rows.has_pose ? sp_identity_residual_kernel<true> : sp_identity_residual_kernel<false>), grid_for(net), dim3(256), 0, stream, rows, d_vars_out, d_runs);
    hipLaunchKernelGGL(sp_sumsq_kernel, dim3(1), dim3(1024), 0, stream, d_runs, net, d_scal + 4);
    e = hipMemcpyAsync(host3, d_scal, sizeof(double), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(host3 + 1, d_scal + 4, sizeof(double), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) return e;
    res.scale = host3[0];
    res.sse_unscaled = host3[1];
    if (result) *result = res;
    if (cache) cache->ready = true;
    e = hipGetLastError();
    return e;
}

}  // namespace fx
