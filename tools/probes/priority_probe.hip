// When do the workgroups of a kernel on a LOW-priority stream start, next to a long kernel on a normal stream whose grid is
// several times what the chip holds? (Would a polling helper kernel stay out of a persistent solve kernel's way until its tail?)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(64) void long_kernel(unsigned long long* t_first, unsigned long long* t_last, unsigned long long ticks) {
    extern __shared__ unsigned char lds[];
    const unsigned long long t0 = wall_clock64();
    if (threadIdx.x == 0) {
        atomicMin(t_first, t0);
        lds[0] = 1;
    }
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) atomicMax(t_last, wall_clock64());
}
__global__ __launch_bounds__(256) void helper_kernel(unsigned long long* starts) {
    extern __shared__ unsigned char lds[];
    if (threadIdx.x == 0) {
        lds[0] = 1;
        starts[blockIdx.x] = wall_clock64();
    }
}
int main() {
    unsigned long long *d_first, *d_last, *d_starts;
    hipMalloc(&d_first, 8); hipMalloc(&d_last, 8); hipMalloc(&d_starts, 256 * 8);
    int least = 0, greatest = 0;
    hipDeviceGetStreamPriorityRange(&least, &greatest);
    printf("priority range: least %d greatest %d\n", least, greatest);
    hipStream_t main_s, low_s, same_s;
    hipStreamCreateWithFlags(&main_s, hipStreamNonBlocking);
    hipStreamCreateWithPriority(&low_s, hipStreamNonBlocking, least);
    hipStreamCreateWithFlags(&same_s, hipStreamNonBlocking);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&helper_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    for (int variant = 0; variant < 3; ++variant) {
        hipStream_t hs = variant == 0 ? low_s : variant == 1 ? same_s : low_s;
        const int helper_first = variant == 2;
        unsigned long long big = ~0ull, zero = 0;
        hipMemcpy(d_first, &big, 8, hipMemcpyHostToDevice);
        hipMemcpy(d_last, &zero, 8, hipMemcpyHostToDevice);
        hipMemset(d_starts, 0, 256 * 8);
        hipDeviceSynchronize();
        // 8192 blocks of one wavefront, 18.6 KB of LDS each (8 per CU: 2048 at a time), 200 us each: four rounds, 0.8 ms
        if (helper_first) hipLaunchKernelGGL(helper_kernel, dim3(256), dim3(256), 62 * 1024, hs, d_starts);
        hipLaunchKernelGGL(long_kernel, dim3(8192), dim3(64), 18608, main_s, d_first, d_last, 20000ull);
        if (!helper_first) hipLaunchKernelGGL(helper_kernel, dim3(256), dim3(256), 62 * 1024, hs, d_starts);
        hipDeviceSynchronize();
        unsigned long long first, last;
        std::vector<unsigned long long> st(256);
        hipMemcpy(&first, d_first, 8, hipMemcpyDeviceToHost);
        hipMemcpy(&last, d_last, 8, hipMemcpyDeviceToHost);
        hipMemcpy(st.data(), d_starts, 256 * 8, hipMemcpyDeviceToHost);
        std::sort(st.begin(), st.end());
        auto us = [&](unsigned long long t) { return ((double)t - (double)first) * 0.01; };
        printf("%s: long kernel %.0f us; helper workgroups start at (us after the long kernel's first): min %.0f, 10th %.0f, median %.0f, 90th %.0f, max %.0f\n",
               variant == 0 ? "helper on the LOW-priority stream, launched after" : variant == 1 ? "helper on a NORMAL-priority stream, launched after"
                                                                                                  : "helper on the LOW-priority stream, launched BEFORE",
               us(last), us(st[0]), us(st[25]), us(st[128]), us(st[230]), us(st[255]));
    }
    return 0;
}
