// bar_probe.hip -- can the host store into device memory directly (large BAR), and what does a mailbox in device memory cost
// against one in pinned host memory?  A resident wave polls a word; the host bumps it and waits for the echo.
//   hipcc --offload-arch=gfx950 -O2 -o build/bar_probe scripts/ubench/bar_probe.hip && build/bar_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// req: polled by the wave (device or host memory); payload: 512 bytes the wave reads after seeing a new number; echo: host memory
__global__ void echo_kernel(const unsigned *req, const unsigned *payload, unsigned *echo, unsigned rounds)
{
    unsigned seen = 0, sum = 0;
    for (unsigned n = 1; n <= rounds;) {
        const unsigned v = __hip_atomic_load(req, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (v == seen) { __builtin_amdgcn_s_sleep(2); continue; }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        sum += __hip_atomic_load(payload + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) + __hip_atomic_load(payload + 64 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        seen = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        if (threadIdx.x == 0) __hip_atomic_store(echo, v + (sum & 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ++n;
    }
}

static double run(unsigned *req_host_view, unsigned *req_dev, unsigned *pay_host_view, unsigned *pay_dev, unsigned *echo, unsigned rounds, hipStream_t s)
{
    *echo = 0;
    *req_host_view = 0;
    hipLaunchKernelGGL(echo_kernel, dim3(1), dim3(64), 0, s, req_dev, pay_dev, echo, rounds);
    const auto t0 = std::chrono::steady_clock::now();
    for (unsigned n = 1; n <= rounds; ++n) {
        for (int k = 0; k < 128; ++k) pay_host_view[k] = n + k; // 512 bytes of "sequences"
        __builtin_ia32_sfence(); // (device memory behind the BAR is write-combining for the host: the stores sit in a buffer until a fence)
        __atomic_store_n(req_host_view, n, __ATOMIC_RELEASE);
        __builtin_ia32_sfence();
        while (__atomic_load_n(echo, __ATOMIC_ACQUIRE) != n) __builtin_ia32_pause();
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    (void)hipStreamSynchronize(s);
    return us;
}

int main()
{
    int large_bar = 0;
    CK(hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0));
    printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    unsigned *pin = nullptr, *echo = nullptr;
    CK(hipHostMalloc(reinterpret_cast<void **>(&pin), 4096, hipHostMallocDefault));
    CK(hipHostMalloc(reinterpret_cast<void **>(&echo), 4096, hipHostMallocDefault));
    memset(pin, 0, 4096);
    printf("mailbox in pinned host memory: %.2f us per round trip\n", run(pin, pin, pin + 256, pin + 256, echo, 20000, s));
    if (!large_bar) { printf("no large BAR: device memory is not host-visible\n"); return 0; }
    unsigned *dev = nullptr;
    CK(hipExtMallocWithFlags(reinterpret_cast<void **>(&dev), 4096, hipDeviceMallocFinegrained));
    CK(hipMemset(dev, 0, 4096));
    CK(hipDeviceSynchronize());
    printf("fine-grained device memory at %p: storing from the host ...\n", (void *)dev);
    fflush(stdout);
    printf("mailbox in fine-grained device memory: %.2f us per round trip\n", run(dev, dev, dev + 256, dev + 256, echo, 20000, s));
    return 0;
}
