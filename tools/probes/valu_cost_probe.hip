// Issue cost (cycles per instruction, one wavefront alone on a SIMD) of the instruction forms the small-System kernels'
// Cholesky is made of: v_fmac_f64_dpp row_newbcast (fx_grouped.hip, RStep::factor), plain v_fma_f64, v_mov_b64_dpp,
// v_readlane + v_fma with an SGPR pair (fx_chol.h), and dependent chains of each. s_memtime around 256 instructions,
// 64 different destination registers (independent) or one (dependent).
//   hipcc --offload-arch=gfx950 -O3 -o tools/probes/valu_cost_probe.bin tools/probes/valu_cost_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>

#define REP4(x) x x x x
#define REP16(x) REP4(REP4(x))
#define REP64(x) REP4(REP16(x))

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }  // s_memtime / shader clock

__global__ void probe(double* out, unsigned long long* t) {
    double a0 = threadIdx.x + 1.0, a1 = a0 * 0.5, a2 = a0 * 0.25, a3 = a0 * 0.125, a4 = a0 + 2, a5 = a0 + 3, a6 = a0 + 4, a7 = a0 + 5;
    double m = 1.0 / (threadIdx.x + 3.0), w = 1e-9 * threadIdx.x;
    unsigned long long t0, t1;
    // 1. independent v_fmac_f64_dpp row_newbcast (8 destinations round-robin), 256 instructions
    t0 = now();
    REP4(REP16(
        asm volatile("v_fmac_f64_dpp %0, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %3, -%8, %9 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(w));))
    t1 = now();
    if (threadIdx.x == 0) t[0] = t1 - t0;
    // 2. independent plain v_fma_f64
    t0 = now();
    REP4(REP16(
        asm volatile("v_fma_f64 %0, -%8, %9, %0\n\tv_fma_f64 %1, -%8, %9, %1\n\tv_fma_f64 %2, -%8, %9, %2\n\tv_fma_f64 %3, -%8, %9, %3"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(w));))
    t1 = now();
    if (threadIdx.x == 0) t[1] = t1 - t0;
    // 3. dependent v_fmac_f64_dpp (one destination)
    t0 = now();
    REP64(asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                       "v_fmac_f64_dpp %0, -%1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a4) : "v"(m), "v"(w));)
    t1 = now();
    if (threadIdx.x == 0) t[2] = t1 - t0;
    // 4. dependent plain v_fma_f64
    t0 = now();
    REP64(asm volatile("v_fma_f64 %0, -%1, %2, %0\n\tv_fma_f64 %0, -%1, %2, %0\n\tv_fma_f64 %0, -%1, %2, %0\n\tv_fma_f64 %0, -%1, %2, %0" : "+v"(a5) : "v"(m), "v"(w));)
    t1 = now();
    if (threadIdx.x == 0) t[3] = t1 - t0;
    // 5. independent v_mov_b64_dpp row_newbcast
    t0 = now();
    REP4(REP16(
        asm volatile("v_mov_b64_dpp %0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %1, %4 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                     "v_mov_b64_dpp %2, %4 row_newbcast:7 row_mask:0xf bank_mask:0xf\n\tv_mov_b64_dpp %3, %4 row_newbcast:9 row_mask:0xf bank_mask:0xf"
                     : "=v"(a0), "=v"(a1), "=v"(a2), "=v"(a3) : "v"(a6));))
    t1 = now();
    if (threadIdx.x == 0) t[4] = t1 - t0;
    // 6. the self form: v_fmac_f64_dpp d, -d, w (reads its own destination through DPP), 4 destinations round-robin
    t0 = now();
    REP4(REP16(
        asm volatile("v_fmac_f64_dpp %0, -%0, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, -%2, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, -%3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w));))
    t1 = now();
    if (threadIdx.x == 0) t[5] = t1 - t0;
    // 7. the pair as RStep::factor issues it: other array first (reads array KA's register through DPP), then the self form
    t0 = now();
    REP4(REP16(
        asm volatile("v_fmac_f64_dpp %0, -%1, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %1, -%1, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, -%3, %4 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f64_dpp %3, -%3, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w), "v"(m));))
    t1 = now();
    if (threadIdx.x == 0) t[6] = t1 - t0;
    // 8. f32 fmac dpp
    float f0 = threadIdx.x, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, fm = 0.3f, fw = 1e-6f;
    t0 = now();
    REP4(REP16(
        asm volatile("v_fmac_f32_dpp %0, -%4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %1, -%4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %2, -%4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\tv_fmac_f32_dpp %3, -%4, %5 row_newbcast:3 row_mask:0xf bank_mask:0xf"
                     : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(fm), "v"(fw));))
    t1 = now();
    if (threadIdx.x == 0) t[7] = t1 - t0;
    out[threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3;
}

int main() {
    double* out;
    unsigned long long *t, h[8];
    hipMalloc((void**)&out, 64 * 8);
    hipMalloc((void**)&t, 64);
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, t);
    hipMemcpy(h, t, sizeof(h), hipMemcpyDeviceToHost);
    const char* names[8] = {"fmac_f64_dpp_independent", "fma_f64_independent", "fmac_f64_dpp_dependent", "fma_f64_dependent", "mov_b64_dpp_independent",
                            "fmac_f64_dpp_self_independent", "fmac_f64_dpp_pair_as_in_factor", "fmac_f32_dpp_independent"};
    printf("{\"unit\": \"readcyclecounter ticks per instruction, 256 instructions, one wavefront\"");
    for (int i = 0; i < 8; ++i) printf(", \"%s\": %.2f", names[i], (double)h[i] / 256.0);
    printf("}\n");
    return 0;
}
