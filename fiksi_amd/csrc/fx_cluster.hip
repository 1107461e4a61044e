// Device side of Decomposer::RecursiveAssembly around its cluster solves (the solves themselves are the pose
// instantiations of lm_solve_body in fx_kernels.hip):
//
//   system_prepare_kernel   what assemble::solve does to a System before any decomposer arm runs
//                           (fiksi/src/assemble/mod.rs:58-124): divide every variable by the system scale,
//                           then nudge the free variables of each connected component, in component order, with
//                           two draws of the shared LCG each. One wavefront per System; the scale and the
//                           nudges are bit-identical to the reference's (same sums, same order).
//   pose_transform_kernel   Pose2D::transform_point (constraints/expressions.rs:1120-1134) on the points a solved
//                           cluster carries along (assemble/mod.rs:238-275).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fx_device.h"
#include "fx_wave.h"

namespace fx {

__global__ __launch_bounds__(64) void system_prepare_kernel(DeviceBatch b, uint32_t mode, double* __restrict__ out_vars,
                                                            double* __restrict__ out_params, double* __restrict__ out_scale) {
    const uint32_t s = blockIdx.x;
    const int lane = threadIdx.x;
    const uint32_t v0 = b.var_off[s], nvt = b.var_off[s + 1] - v0;
    const uint32_t e0 = b.expr_off[s], net = b.expr_off[s + 1] - e0;
    double scale = 1.0, scale_recip = 1.0;
    if (mode & 1u) {
        scale = system_scale_wave(
            nvt, net, lane, [&](uint32_t i) { return b.vars0[v0 + i]; }, [&](uint32_t i) { return (int)(b.expr_tag[e0 + i] & 0x7F); },
            [&](uint32_t i) { return b.expr_param[e0 + i]; });
        scale_recip = 1.0 / scale;
    }
    if (lane == 0) out_scale[s] = scale;
    for (uint32_t i = lane; i < nvt; i += 64) {
        const double v = b.vars0[v0 + i];
        out_vars[v0 + i] = (mode & 1u) ? v * scale_recip : v;
    }
    for (uint32_t i = lane; i < net; i += 64) {  // Expression::transform, expressions.rs:195-211: distances only
        const int tag = (int)(b.expr_tag[e0 + i] & 0x7F);
        const double p = b.expr_param[e0 + i];
        out_params[e0 + i] = ((mode & 1u) && (tag == 1 || tag == 4)) ? scale_recip * p : p;  // FX_TAG_PPD, FX_TAG_PLD
    }
    if (!(mode & 2u)) return;
    // one Rng::from_seed(42) per solve, shared by the components (:47); two draws per free variable, ascending (:113-124)
    uint32_t rng = 42u;
    const uint32_t ncomp = b.sys_ncomp[s];
    for (uint32_t c = 0; c < ncomp; ++c) {
        uint32_t rank0 = 0;
        for (uint32_t base = 0; base < nvt; base += 64) {
            const uint32_t i = base + (uint32_t)lane;
            bool in = false;
            if (i < nvt) {
                const uint16_t info = b.var_info[v0 + i];
                in = ((info & VAR_COMP_MASK) == c) && !(info & VAR_FIXED_BIT);
            }
            const uint64_t mk = __ballot(in);
            if (in) {
                uint32_t st = lcg_jump(rng, 2u * (rank0 + (uint32_t)__popcll(mk & lanemask_lt(lane))));
                st = st * 1664525u + 1013904223u;
                const double f1 = (1.0 / 4294967295.0) * (double)st;
                st = st * 1664525u + 1013904223u;
                const double f2 = (1.0 / 4294967295.0) * (double)st;
                double x = b.vars0[v0 + i];
                if (mode & 1u) x = x * scale_recip;
                x += x * (1.0 / 8196.0) * f1 + (1.0 / 65568.0) * f2;
                out_vars[v0 + i] = x;
            }
            rank0 += (uint32_t)__popcll(mk);
        }
        rng = lcg_jump(rng, 2u * rank0);
    }
}

hipError_t launch_prepare(const DeviceBatch& b, uint32_t mode, double* out_vars, double* out_params, double* out_scale, hipStream_t stream) {
    if (b.n_systems == 0) return hipSuccess;
    hipLaunchKernelGGL(system_prepare_kernel, dim3(b.n_systems), dim3(64), 0, stream, b, mode, out_vars, out_params, out_scale);
    return hipGetLastError();
}

__global__ __launch_bounds__(64) void pose_transform_kernel(const double* __restrict__ poses, const uint32_t* __restrict__ pose_of,
                                                            const uint32_t* __restrict__ idx, uint32_t n, double* __restrict__ vars) {
    const uint32_t i = blockIdx.x * 64u + threadIdx.x;
    if (i >= n) return;
    const double* pose = poses + 3 * (size_t)pose_of[i];
    double sn, cs;
    ::sincos(pose[0], &sn, &cs);
    const double u = vars[idx[i]], v = vars[idx[i] + 1];
    const double uc = u * cs, us = u * sn, vc = v * cs, vs = v * sn;
    vars[idx[i]] = pose[1] + uc - vs;
    vars[idx[i] + 1] = pose[2] + us + vc;
}

hipError_t launch_pose_transform(const double* poses, const uint32_t* pose_of, const uint32_t* idx, uint32_t n, double* vars,
                                 hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(pose_transform_kernel, dim3((n + 63u) / 64u), dim3(64), 0, stream, poses, pose_of, idx, n, vars);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void unscale_kernel(double scale, const double* __restrict__ scaled, const uint8_t* __restrict__ mask,
                                                      double* __restrict__ vars, uint32_t n) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n && mask[i]) vars[i] = scale * scaled[i];
}

// the same for a batch of Systems of one structure (nvars variables each, one mask): System k's scale for its slice
__global__ __launch_bounds__(256) void unscale_strided_kernel(const double* __restrict__ scales, uint32_t nvars, const double* __restrict__ scaled,
                                                              const uint8_t* __restrict__ mask, double* __restrict__ vars, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256u + threadIdx.x;
    if (i < n && mask[i % nvars]) vars[i] = scales[i / nvars] * scaled[i];
}
hipError_t launch_unscale_strided(const double* scales, uint32_t n_systems, uint32_t nvars, const double* scaled, const uint8_t* mask, double* vars,
                                  hipStream_t stream) {
    const uint64_t n = (uint64_t)n_systems * nvars;
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unscale_strided_kernel, dim3((uint32_t)((n + 255u) / 256u)), dim3(256), 0, stream, scales, nvars, scaled, mask, vars, n);
    return hipGetLastError();
}

hipError_t launch_unscale(double scale, const double* scaled, const uint8_t* mask, double* vars, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(unscale_kernel, dim3((n + 255u) / 256u), dim3(256), 0, stream, scale, scaled, mask, vars, n);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void pull_kernel(uint4* __restrict__ dst, const uint4* __restrict__ src, uint32_t n16) {
    const uint32_t i = blockIdx.x * 256u + threadIdx.x;
    if (i < n16) dst[i] = src[i];
}

// dst[i] = dst[i mod period] for period <= i < total: a batch of one structure sends its structure arrays' first System over the
// bus and fills in the others here (units of four bytes when everything is a multiple of four, of one byte otherwise)
template <typename U>
__global__ __launch_bounds__(256) void replicate_kernel(U* __restrict__ dst, uint32_t period, unsigned long long total) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x + period;
    if (i < total) dst[i] = dst[i % period];
}

hipError_t launch_replicate(void* dst, size_t period_bytes, size_t total_bytes, hipStream_t stream) {
    if (period_bytes == 0 || total_bytes <= period_bytes) return hipSuccess;
    if (period_bytes % 4u == 0 && total_bytes % 4u == 0) {
        const unsigned long long total = total_bytes / 4u, todo = total - period_bytes / 4u;
        hipLaunchKernelGGL(replicate_kernel<uint32_t>, dim3((unsigned)((todo + 255ull) / 256ull)), dim3(256), 0, stream, static_cast<uint32_t*>(dst),
                           (uint32_t)(period_bytes / 4u), total);
    } else {
        const unsigned long long todo = total_bytes - period_bytes;
        hipLaunchKernelGGL(replicate_kernel<uint8_t>, dim3((unsigned)((todo + 255ull) / 256ull)), dim3(256), 0, stream, static_cast<uint8_t*>(dst),
                           (uint32_t)period_bytes, (unsigned long long)total_bytes);
    }
    return hipGetLastError();
}

hipError_t launch_pull(void* dst, const void* src, size_t bytes, hipStream_t stream) {
    const uint32_t n16 = (uint32_t)(bytes / 16u);
    if (n16 == 0) return hipSuccess;
    hipLaunchKernelGGL(pull_kernel, dim3((n16 + 255u) / 256u), dim3(256), 0, stream, static_cast<uint4*>(dst), static_cast<const uint4*>(src), n16);
    return hipGetLastError();
}

}  // namespace fx
