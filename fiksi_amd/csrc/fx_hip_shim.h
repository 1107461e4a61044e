// FX_HOST_ONLY (the sanitizer build, `make asan`): the handful of HIP runtime names the host-side sources mention,
// as stubs that report "no device". The pure-host entry points (validation, Jacobian structure, SinglePass blocks,
// QR planning, the System builder) run for real under AddressSanitizer / UBSan; anything that needs the GPU fails with
// FX_ERR_NO_DEVICE exactly as the product does on a box without one. Never part of libfiksi_amd.so.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline

typedef int hipError_t;
typedef struct fx_fake_stream* hipStream_t;
typedef struct fx_fake_event* hipEvent_t;
enum : int { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum : unsigned { hipStreamNonBlocking = 1, hipHostMallocMapped = 2, hipHostMallocCoherent = 0x40000000 };
struct hipDeviceProp_t {
    char name[128];
    char gcnArchName[64];
};
// FIKSI_AMD_SHIM_FAKE_DEVICE=1 (tools/abi_fuzz.py): one make-believe device whose memory is the host heap — allocations,
// copies and fills really happen (so the sanitizers check every upload's index arithmetic and sizes), streams and events are
// inert, and every KERNEL launcher still answers "no device" (fx_host_only.cpp): a call runs its host analysis and its
// uploads for real and then fails with FX_ERR_HIP where the first kernel would start.
inline bool fx_shim_fake() {
    static const bool on = [] { const char* e = std::getenv("FIKSI_AMD_SHIM_FAKE_DEVICE"); return e && e[0] == '1'; }();
    return on;
}
inline const char* hipGetErrorString(hipError_t) { return "no HIP runtime in the host-only build"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = fx_shim_fake() ? 1 : 0; return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
inline hipError_t hipSetDevice(int d) { return fx_shim_fake() && d == 0 ? hipSuccess : hipErrorNoDevice; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) {
    if (!fx_shim_fake()) return hipErrorNoDevice;
    std::memset(p, 0, sizeof(*p));
    std::strcpy(p->name, "host-only shim");
    std::strcpy(p->gcnArchName, "gfx950");
    return hipSuccess;
}
// (live blocks of the make-believe device: the allocation-failure sweep checks that a failed call gives everything back)
inline long& fx_shim_live_blocks() {
    static long live = 0;
    return live;
}
inline hipError_t hipMalloc(void** p, size_t n) {
    *p = fx_shim_fake() ? std::malloc(n ? n : 1) : nullptr;
    if (*p) __atomic_add_fetch(&fx_shim_live_blocks(), 1, __ATOMIC_RELAXED);
    return *p ? hipSuccess : (fx_shim_fake() ? hipErrorOutOfMemory : hipErrorNoDevice);
}
inline hipError_t hipFree(void* p) {
    if (fx_shim_fake() && p) {
        std::free(p);
        __atomic_sub_fetch(&fx_shim_live_blocks(), 1, __ATOMIC_RELAXED);
    }
    return hipSuccess;
}
inline hipError_t hipHostMalloc(void** p, size_t n, unsigned) { return hipMalloc(p, n); }
inline hipError_t hipHostFree(void* p) { return hipFree(p); }
constexpr unsigned hipHostRegisterDefault = 0u;
constexpr unsigned hipHostRegisterMapped = 2u;
inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) {
    *least = 0;
    *greatest = -1;
    return hipSuccess;
}
inline hipError_t hipHostRegister(void*, size_t, unsigned) { return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
inline hipError_t hipHostUnregister(void*) { return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
inline hipError_t hipHostGetDevicePointer(void** dev, void* host, unsigned) {  // (the make-believe device's memory is the host's)
    *dev = host;
    return fx_shim_fake() ? hipSuccess : hipErrorNoDevice;
}
inline hipError_t hipGetLastError() { return hipSuccess; }
inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) {
    if (!fx_shim_fake()) return hipErrorNoDevice;
    if (n) std::memmove(d, s, n);
    return hipSuccess;
}
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) {
    if (!fx_shim_fake()) return hipErrorNoDevice;
    if (n) std::memset(d, v, n);
    return hipSuccess;
}
inline hipError_t hipStreamSynchronize(hipStream_t) { return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = nullptr; return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int) { return hipStreamCreateWithFlags(s, flags); }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
constexpr unsigned hipEventDisableTiming = 2u;
inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return fx_shim_fake() ? hipSuccess : hipErrorNoDevice; }
