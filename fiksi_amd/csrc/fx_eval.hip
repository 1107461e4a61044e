// K1 / K2 of the hot path over a whole batch (Subsystem::calculate_residuals_and_sparse_jacobian,
// fiksi/src/subsystem.rs:126-166, and the COO -> sorted / merged sparse Jacobian of solvi/src/sparse_col_mat.rs:690-737):
// one thread per expression row, residual + CSR Jacobian values. HBM-streaming: this is the kernel bench.py's
// `roofline` is quoted on.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>

#include "fx_device.h"
#include "fx_expr.h"

namespace fx {

// ------------------------------------------------------------------------------------------
// K1 / K2 over the whole batch: one thread per expression row
// ------------------------------------------------------------------------------------------
// x: n_vars values (vars0 or vars). Rows gather from the owning System's block of x; fixed and
// free variables alike are read from x (IndexSetVariableMap with the snapshot == x).
// CSR slot of every gradient entry of a row whose variables are all free and distinct, computed
// from the element fields instead of being read from the jslot table: fields sorted by variable
// index, a point is 2 columns wide, a scalar 1. Returns the packed 8 x 4-bit slots; cnt = row nnz.
__device__ __forceinline__ uint32_t compute_slots(int tag, const uint16_t f[4], uint32_t& cnt) {
    // widths of the four element fields (expressions.rs:48-182)
    const int k = tag_nvars(tag);
    uint32_t w[4];
    w[0] = (tag == FX_TAG_VVE) ? 1u : 2u;
    w[1] = (tag == FX_TAG_VVE) ? 1u : 2u;
    w[2] = (k >= 6) ? 2u : (k == 5 ? 1u : 0u);
    w[3] = (k == 8) ? 2u : (k == 7 ? 1u : 0u);
    uint32_t sf[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        uint32_t acc = 0;
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (c != a) acc += (f[c] < f[a]) ? w[c] : 0u;
        sf[a] = acc;
    }
    cnt = w[0] + w[1] + w[2] + w[3];
    uint32_t slots = 0;
    if (tag == FX_TAG_VVE) {
        slots = sf[0] | (sf[1] << 4) | 0xFFFFFF00u;
    } else {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            uint32_t lo = (w[a] >= 1u) ? sf[a] : 0xFu;
            uint32_t hi = (w[a] == 2u) ? sf[a] + 1u : 0xFu;
            slots |= (lo | (hi << 4)) << (8 * a);
        }
    }
    return slots;
}

// NT: the results leave with non-temporal stores (the shipped form). NT = false exists for the A / B measurement only
// (FIKSI_AMD_K1_STORES=plain, tools/k1_stores_ab.py).
// uni_period: 0, or — for a batch of ONE structure (one sketch, many parameter sets: b.uniform) — the number of blocks after
// which the tag-sorted order of a block's rows repeats (lcm of the rotation's 4 and u_nexprs / gcd(256, u_nexprs)). Such a
// batch reads its structure from the first System's rows and the first uni_period blocks' orders: a few hundred bytes that
// stay in L1, instead of 11 bytes per row streamed from HBM in two more dependent round trips. What is left in front of a
// block's arithmetic is ONE trip to HBM (parameters and variables, both addressed from the cached structure).
template <bool WANT_J, bool NT = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void eval_rows_kernel(DeviceBatch b, const double* __restrict__ x, uint32_t uni_period) {
    // Block = 256 consecutive rows. Threads take the block's rows in tag-sorted order (host-built
    // permutation) so that a wavefront sees as few expression kinds as possible (divergence: the
    // angle rows cost ~8x a distance row). Residuals and the block's CSR values (contiguous in the
    // value array) are staged in LDS and streamed out with coalesced stores.
    // "Simple" blocks (every row: all variables free and distinct — the common case) derive the CSR
    // slots and row offsets in-kernel (field sort + block scan) instead of reading 12 B/row of tables.
    __shared__ double jstage[WANT_J ? 256 * 8 : 1];
    __shared__ double rstage[256];
    __shared__ uint32_t scnt[WANT_J ? 256 : 1];
    __shared__ uint32_t swave[4];
    const uint32_t row0 = blockIdx.x * 256u;
    const uint32_t nrows = min(256u, b.n_exprs - row0);
    const bool uni = uni_period != 0u && nrows == 256u;
    const BlockInfo bi = b.blk_info[blockIdx.x];
    const bool simple = (bi.flags & 1u) != 0;
    const uint32_t jbase = bi.jbase;
    const bool live = threadIdx.x < nrows;
    uint32_t lrow = 0, slots = 0xFFFFFFFFu, cnt = 0, base = 0;
    bool dup = false;
    double g[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (live) {
        lrow = b.row_perm[(uni ? (blockIdx.x % uni_period) * 256u : row0) + threadIdx.x];
        const uint32_t row = row0 + lrow;
        const uint32_t sys = uni ? row / b.u_nexprs : 0u;      // (the System of the row and the row inside it)
        const uint32_t srow = uni ? row - sys * b.u_nexprs : row;
        const int tagx = b.expr_tag[srow];
        const int tag = tagx & 0x7F;
        dup = (tagx & 0x80) != 0;
        ushort4 f4 = reinterpret_cast<const ushort4*>(b.expr_idx)[srow];
        uint16_t ff[4] = {f4.x, f4.y, f4.z, f4.w};
        const uint32_t v0 = uni ? sys * b.u_nvars : simple ? b.var_off[bi.sys0 + b.row_sysoff[row]] : b.expr_var0[row];
        uint32_t vars8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        expand_vars(tag, ff, vars8);
        double v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = x[v0 + vars8[e]];
        rstage[lrow] = eval_expression<double, WANT_J>(tag, v, b.expr_param[row], g);
        if (WANT_J) {
            if (simple && uni) {
                // (one structure: the row's place among the block's values from the first System's row pointers — cached
                // — instead of the scan over the block's row lengths and its four barriers)
                slots = compute_slots(tag, ff, cnt);
                base = sys * b.jrow_ptr[b.u_nexprs] + b.jrow_ptr[srow] - jbase;
            } else if (simple) {
                slots = compute_slots(tag, ff, cnt);
                scnt[lrow] = cnt;
            } else {
                slots = b.jslot[row];
                base = b.jrow_ptr[row] - jbase;
                cnt = b.jrow_ptr[row + 1] - jbase - base;
            }
        }
    }
    if (WANT_J && simple && !uni) {
        // exclusive scan of the row lengths in row order: wave scan + 4 wave totals
        if (!live) scnt[threadIdx.x] = 0;  // rows beyond the batch end
        __syncthreads();
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        uint32_t val = scnt[threadIdx.x], incl = val;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            uint32_t t = __shfl_up(incl, off, 64);
            if (lane >= off) incl += t;
        }
        if (lane == 63) swave[wv] = incl;
        __syncthreads();
        uint32_t woff = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) woff += (q < wv) ? swave[q] : 0u;
        __syncthreads();
        scnt[threadIdx.x] = woff + incl - val;
        __syncthreads();
        base = scnt[lrow];
    }
    if (WANT_J && live) {
        if (!dup) {
            // every free variable of the row is distinct: partial e goes to its CSR slot;
            // 0xF = fixed variable, dropped (subsystem.rs:159-164)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                uint32_t sl = (slots >> (4 * e)) & 0xFu;
                if (sl != 0xFu) jstage[base + sl] = g[e];
            }
        } else {
            // the same variable appears twice (quirk Q4): duplicates are summed in gradient order
            double out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                uint32_t sl = (slots >> (4 * e)) & 0xFu;
#pragma unroll
                for (int q = 0; q < 8; ++q) out[q] += (sl == (uint32_t)q) ? g[e] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if ((uint32_t)q < cnt) jstage[base + q] = out[q];
            }
        }
    }
    __syncthreads();
    // results are written once and not read back by this kernel: streaming (non-temporal) stores
    if (live) {
        if (NT) __builtin_nontemporal_store(rstage[threadIdx.x], &b.resid[row0 + threadIdx.x]);
        else b.resid[row0 + threadIdx.x] = rstage[threadIdx.x];
    }
    if (WANT_J) {
        // 16-byte stores on the 16-byte-aligned body of [jbase, jend), scalar head / tail
        const uint32_t n = bi.jcount;
        const uint32_t jend = jbase + n;
        const uint32_t head = jbase & 1u;  // jvals is 16-byte aligned at index 0
        if (head && threadIdx.x == 0 && n > 0) b.jvals[jbase] = jstage[0];
        const uint32_t pairs = (n - min(n, head)) >> 1;
        double2* dst = reinterpret_cast<double2*>(b.jvals + jbase + head);
        for (uint32_t i = threadIdx.x; i < pairs; i += 256u) {
            typedef double v2d __attribute__((ext_vector_type(2)));
            v2d t = {jstage[head + 2 * i], jstage[head + 2 * i + 1]};
            if (NT) __builtin_nontemporal_store(t, reinterpret_cast<v2d*>(dst + i));
            else *reinterpret_cast<v2d*>(dst + i) = t;
        }
        if (((n - min(n, head)) & 1u) && threadIdx.x == 1) b.jvals[jend - 1] = jstage[n - 1];
    }
}

hipError_t launch_eval(const DeviceBatch& b, const double* x, bool want_jacobian, hipStream_t stream) {
    if (b.n_exprs == 0) return hipSuccess;
    dim3 grid((b.n_exprs + 255u) / 256u), block(256);
    static const bool plain_stores = [] {
        const char* s = getenv("FIKSI_AMD_K1_STORES");
        return s && s[0] == 'p';
    }();
    static const bool no_uniform = [] {  // (A / B switch of the measurement in DESIGN.md 6)
        const char* s = getenv("FIKSI_AMD_K1_UNIFORM");
        return s && s[0] == '0';
    }();
    uint32_t period = 0;
    if (b.uniform && b.u_nexprs && !no_uniform) {
        uint32_t g = 256u, r = b.u_nexprs;  // gcd
        while (r) {
            const uint32_t t = g % r;
            g = r;
            r = t;
        }
        period = b.u_nexprs / g;
        while (period % 4u) period *= 2u;  // lcm with the rotation of the sorted order (fx_analyze.cpp: build_eval_plan)
    }
    if (want_jacobian && plain_stores) {
        hipLaunchKernelGGL((eval_rows_kernel<true, false>), grid, block, 0, stream, b, x, period);
    } else if (want_jacobian) {
        hipLaunchKernelGGL(eval_rows_kernel<true>, grid, block, 0, stream, b, x, period);
    } else {
        hipLaunchKernelGGL(eval_rows_kernel<false>, grid, block, 0, stream, b, x, period);
    }
    return hipGetLastError();
}


}  // namespace fx
