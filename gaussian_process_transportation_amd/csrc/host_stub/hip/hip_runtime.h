// Host-only stand-in for <hip/hip_runtime.h>, used ONLY by the sanitizer build of the C-ABI's host orchestration
// (`make host-asan`: gpt_api.hip compiled by g++ with -fsanitize=address,undefined against host_stub/).  "Device"
// memory is malloc'ed host memory, so AddressSanitizer sees every staging copy, header read and plan upload the
// orchestration makes; streams and events are inert; kernels are replaced by host_stub/stub_launchers.cpp.
// Not a HIP implementation and never part of libgpt_hip.so.
#pragma once
#include <cstddef>
#include <cstdlib>
#include <cstring>

typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr hipError_t hipErrorInvalidValue = 1;
constexpr hipError_t hipErrorOutOfMemory = 2;
constexpr hipError_t hipErrorPeerAccessAlreadyEnabled = 704;
typedef struct stub_stream_* hipStream_t;
typedef struct stub_event_* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
constexpr unsigned hipStreamNonBlocking = 1, hipEventDisableTiming = 2;
struct hipDeviceProp_t { int multiProcessorCount; };

static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : "stub HIP error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : hipErrorInvalidValue; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { p->multiProcessorCount = 256; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(malloc(1)); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }
template <typename T>
static inline hipError_t hipMalloc(T** p, size_t bytes) {
    *p = static_cast<T*>(malloc(bytes ? bytes : 1));
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void* d, int, const void* s, int, size_t n, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int* can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemcpy2D(void* d, size_t dpitch, const void* s, size_t spitch, size_t width, size_t height, hipMemcpyKind) {
    for (size_t r = 0; r < height; ++r) memcpy(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, width);
    return hipSuccess;
}
