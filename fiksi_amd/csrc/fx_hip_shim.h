// FX_HOST_ONLY (the sanitizer build, `make asan`): the handful of HIP runtime names the host-side sources mention,
// as stubs that report "no device". The pure-host entry points (validation, Jacobian structure, SinglePass blocks,
// QR planning, the System builder) run for real under AddressSanitizer / UBSan; anything that needs the GPU fails with
// FX_ERR_NO_DEVICE exactly as the product does on a box without one. Never part of libfiksi_amd.so.
#pragma once
#include <cstddef>
#include <cstdint>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline

typedef int hipError_t;
typedef struct fx_fake_stream* hipStream_t;
typedef struct fx_fake_event* hipEvent_t;
enum : int { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNoDevice = 100 };
enum hipMemcpyKind { hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum : unsigned { hipStreamNonBlocking = 1 };
struct hipDeviceProp_t {
    char name[256];
    char gcnArchName[256];
};
inline const char* hipGetErrorString(hipError_t) { return "no HIP runtime in the host-only build"; }
inline hipError_t hipGetDeviceCount(int* n) { *n = 0; return hipErrorNoDevice; }
inline hipError_t hipSetDevice(int) { return hipErrorNoDevice; }
inline hipError_t hipGetDeviceProperties(hipDeviceProp_t*, int) { return hipErrorNoDevice; }
inline hipError_t hipMalloc(void** p, size_t) { *p = nullptr; return hipErrorNoDevice; }
inline hipError_t hipFree(void*) { return hipSuccess; }
inline hipError_t hipHostMalloc(void** p, size_t, unsigned) { *p = nullptr; return hipErrorNoDevice; }
inline hipError_t hipHostFree(void*) { return hipSuccess; }
inline hipError_t hipMemcpy(void*, const void*, size_t, hipMemcpyKind) { return hipErrorNoDevice; }
inline hipError_t hipMemcpyAsync(void*, const void*, size_t, hipMemcpyKind, hipStream_t) { return hipErrorNoDevice; }
inline hipError_t hipMemsetAsync(void*, int, size_t, hipStream_t) { return hipErrorNoDevice; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipErrorNoDevice; }
inline hipError_t hipStreamCreateWithFlags(hipStream_t*, unsigned) { return hipErrorNoDevice; }
inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
inline hipError_t hipEventCreate(hipEvent_t*) { return hipErrorNoDevice; }
constexpr unsigned hipEventDisableTiming = 2u;
inline hipError_t hipEventCreateWithFlags(hipEvent_t*, unsigned) { return hipErrorNoDevice; }
inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipErrorNoDevice; }
inline hipError_t hipEventSynchronize(hipEvent_t) { return hipErrorNoDevice; }
inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipErrorNoDevice; }
inline hipError_t hipEventElapsedTime(float*, hipEvent_t, hipEvent_t) { return hipErrorNoDevice; }
