// Pageable / registered / pinned host<->device copy rates by transfer size (decides the chunking of the host-buffer entry).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    const size_t total = (size_t)2 << 30;
    char *host = (char *)malloc(total);
    memset(host, 1, total);
    char *dev; CK(hipMalloc(&dev, total));
    char *pinned; CK(hipHostMalloc(&pinned, (size_t)256 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    CK(hipMemcpy(dev, host, 64 << 20, hipMemcpyHostToDevice));
    for (size_t sz : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)64 << 20, (size_t)256 << 20, (size_t)1 << 30, (size_t)2 << 30}) {
        const int reps = (int)std::max<size_t>(1, total / sz);
        double t0 = now();
        for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(dev + (size_t)r * sz, host + (size_t)r * sz, sz, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t1 = now();
        for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(host + (size_t)r * sz, dev + (size_t)r * sz, sz, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        printf("pageable  %6zu MiB x %4d: H2D %6.1f GB/s  D2H %6.1f GB/s\n", sz >> 20, reps, reps * sz / (t1 - t0) / 1e9, reps * sz / (t2 - t1) / 1e9);
    }
    for (size_t sz : {(size_t)1 << 20, (size_t)16 << 20, (size_t)256 << 20}) {
        const int reps = 8;
        double t0 = now();
        for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(dev + (size_t)r * sz, pinned, sz, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t1 = now();
        for (int r = 0; r < reps; ++r) CK(hipMemcpyAsync(pinned, dev + (size_t)r * sz, sz, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        printf("pinned    %6zu MiB x %4d: H2D %6.1f GB/s  D2H %6.1f GB/s\n", sz >> 20, reps, reps * sz / (t1 - t0) / 1e9, reps * sz / (t2 - t1) / 1e9);
    }
    {
        double t0 = now();
        CK(hipHostRegister(host, total, hipHostRegisterDefault));
        double t1 = now();
        CK(hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        double t2 = now();
        CK(hipMemcpyAsync(host, dev, total, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double t3 = now();
        CK(hipHostUnregister(host));
        double t4 = now();
        printf("register %zu MiB: %.1f ms (%.1f GB/s), H2D %.1f GB/s, D2H %.1f GB/s, unregister %.1f ms\n", total >> 20, (t1 - t0) * 1e3,
               total / (t1 - t0) / 1e9, total / (t2 - t1) / 1e9, total / (t3 - t2) / 1e9, (t4 - t3) * 1e3);
    }
    {   // CPU memcpy into pinned staging, single thread
        const size_t sz = (size_t)256 << 20;
        double t0 = now();
        for (int r = 0; r < 8; ++r) memcpy(pinned, host + (size_t)r * sz, sz);
        double t1 = now();
        printf("memcpy host->pinned 1 thread: %.1f GB/s\n", 8.0 * sz / (t1 - t0) / 1e9);
    }
    return 0;
}
