/*
 * hip/hip_runtime.h -- a FAKE, TEST-ONLY HIP runtime: "device" memory is host memory, streams and events are tokens,
 * every "asynchronous" call completes before it returns -- but for a GATED launch of the persistent grid, which fake_device.cpp runs as a
 * thread (its inputs arrive after the launch) and which hipStreamSynchronize / hipDeviceSynchronize join.  It exists so that the HOST side of the library (sw_capi.cpp,
 * sw_batcher.cpp, sw_multi.cpp: chunking, hooks, the per-chunk sort by geometry, sharding, the coalescer's threads) can be
 * built with -fsanitize=address,undefined and -fsanitize=thread on a CPU and driven hard (GPU sanitizers are not available
 * on this pool).  The kernels' place is taken by tests/cpp/fake_device.cpp.  Never on the product's include path.
 */
#ifndef MGL_TEST_FAKE_HIP_RUNTIME_H
#define MGL_TEST_FAKE_HIP_RUNTIME_H

#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1, hipErrorNoDevice = 100, hipErrorHostMemoryAlreadyRegistered = 712 };
typedef struct fakeStream *hipStream_t;
typedef struct fakeEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3 };
enum { hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipHostMallocDefault = 0 };
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 1, hipDeviceAttributeIsLargeBar = 2 };
enum { hipDeviceMallocFinegrained = 1 };

int fake_hip_device_count(void); /* tests/cpp/fake_device.cpp: FAKE_HIP_DEVICES, default 2 */

static inline hipError_t hipGetDeviceCount(int *n) { *n = fake_hip_device_count(); return *n > 0 ? hipSuccess : hipErrorNoDevice; }
int *fake_hip_current_device(void); /* tests/cpp/fake_device.cpp: the calling thread's current device */
static inline hipError_t hipSetDevice(int d)
{
    if (d < 0 || d >= fake_hip_device_count()) return hipErrorInvalidValue;
    *fake_hip_current_device() = d;
    return hipSuccess;
}
static inline hipError_t hipGetDevice(int *d) { *d = *fake_hip_current_device(); return hipSuccess; }
static inline hipError_t hipGetLastError(void) { return hipSuccess; }
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "success" : e == hipErrorOutOfMemory ? "out of memory" : "fake hip error"; }
static inline hipError_t hipMalloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void **p, size_t bytes, unsigned) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipExtMallocWithFlags(void **p, size_t bytes, unsigned) { *p = malloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
void fake_hip_join_stream(hipStream_t s); /* tests/cpp/fake_device.cpp: waits for the gated grids launched on s (NULL: on any stream) */
int fake_hip_copy_should_fail(void);      /* ... fault injection: this asynchronous copy is to fail (fake_hip_fail_copy_in) */
static inline hipError_t hipDeviceSynchronize(void) { fake_hip_join_stream((hipStream_t)0); return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
enum { hipHostRegisterDefault = 0 };
static inline hipError_t hipHostRegister(void *, size_t, unsigned) { return hipSuccess; }
static inline hipError_t hipHostUnregister(void *) { return hipSuccess; }
static inline hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned) { *d = h; return hipSuccess; }
enum hipMemoryType { hipMemoryTypeUnregistered = 0, hipMemoryTypeHost = 1, hipMemoryTypeDevice = 2 };
typedef struct hipPointerAttribute_t { int type; } hipPointerAttribute_t;
static inline hipError_t hipPointerGetAttributes(hipPointerAttribute_t *a, const void *) { a->type = hipMemoryTypeUnregistered; return hipSuccess; } /* (nothing is pinned behind the library's back here) */
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t)
{
    if (fake_hip_copy_should_fail()) return hipErrorInvalidValue;
    memmove(d, s, n);
    return hipSuccess;
}
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = *t = (size_t)8 << 30; return hipSuccess; }
/* (a chip of four CUs, so that a few thousand pairs are many rounds of it; FAKE_HIP_CUS, read per call, gives another count) */
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int)
{
    const char *e = a == hipDeviceAttributeMultiprocessorCount ? getenv("FAKE_HIP_CUS") : 0;
    *v = e ? atoi(e) : 4;
    return hipSuccess;
}
typedef void *hipDeviceptr_t;
static inline hipError_t hipMemsetD32Async(hipDeviceptr_t d, int v, size_t n, hipStream_t) { for (size_t k = 0; k < n; ++k) ((int *)d)[k] = v; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *lo, int *hi) { *lo = 0; *hi = 0; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(1); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (hipStream_t)malloc(1); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t s) { fake_hip_join_stream(s); return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (hipEvent_t)malloc(1); return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.f; return hipSuccess; }

#endif
