// placement.hip -- where do the workgroups of a persistent one-wave-per-workgroup grid land, and are they all resident at once?
// N workgroups of one wave with 256 VGPRs (two fit a SIMD) record their hardware place and start time, then stay for `hold_us`.
//   hipcc --offload-arch=gfx950 -O2 -o build/placement scripts/ubench/placement.hip && build/placement [workgroups] [hold_us] [lds_bytes]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

struct Rec {
    unsigned hw_id, xcc_id;
    unsigned long long t_start, t_end;
};

__global__ __launch_bounds__(64, 2) void hold_kernel(Rec *out, unsigned long long hold_ticks, int lds_bytes)
{
    extern __shared__ unsigned lds[];
    asm volatile("v_mov_b32 v255, 0" ::: "v255"); // the register allocation of the kernel this stands for: 256 VGPRs
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lds_bytes > 0 && threadIdx.x == 0) lds[0] = hw;
    while (__builtin_amdgcn_s_memrealtime() - t0 < hold_ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0) out[blockIdx.x] = Rec{hw, xcc, t0, __builtin_amdgcn_s_memrealtime()};
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 2048;
    const unsigned long long hold_us = argc > 2 ? atoll(argv[2]) : 2000;
    const int lds = argc > 3 ? atoi(argv[3]) : 0;
    Rec *d = nullptr;
    CK(hipMalloc(&d, sizeof(Rec) * n));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(hold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(hold_kernel, dim3(n), dim3(64), lds, 0, d, hold_us * 100ull, lds);
        CK(hipDeviceSynchronize());
    }
    std::vector<Rec> h(n);
    CK(hipMemcpy(h.data(), d, sizeof(Rec) * n, hipMemcpyDeviceToHost));
    unsigned long long t_min = ~0ull;
    for (const Rec &r : h) t_min = std::min(t_min, r.t_start);
    std::map<unsigned, int> per_simd, per_cu, per_xcc;
    int late = 0;
    for (const Rec &r : h) {
        // HW_ID (gfx9): wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13 (gfx950: se 16:13?); XCC_ID 3:0
        const unsigned simd = (r.hw_id >> 4) & 3, cu = (r.hw_id >> 8) & 15, sh = (r.hw_id >> 12) & 1, se = (r.hw_id >> 13) & 7, xcc = r.xcc_id & 15;
        const unsigned cu_key = xcc << 16 | se << 8 | sh << 4 | cu;
        per_cu[cu_key]++;
        per_simd[cu_key << 2 | simd]++;
        per_xcc[xcc]++;
        if (r.t_start - t_min > hold_us * 100ull / 2) ++late;
    }
    std::map<int, int> hist_simd, hist_cu;
    for (auto &kv : per_simd) hist_simd[kv.second]++;
    for (auto &kv : per_cu) hist_cu[kv.second]++;
    printf("%d workgroups (one wave, 256 VGPRs, %d B LDS), held %llu us: %d started late (after half the hold time)\n", n, lds, hold_us, late);
    printf("  distinct CUs %zu, distinct SIMDs %zu\n", per_cu.size(), per_simd.size());
    printf("  workgroups per XCC:");
    for (auto &kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
    printf("\n  CUs by number of workgroups:");
    for (auto &kv : hist_cu) printf(" %d x %d", kv.second, kv.first);
    printf("\n  SIMDs by number of workgroups:");
    for (auto &kv : hist_simd) printf(" %d x %d", kv.second, kv.first);
    printf("\n");
    // the first sixteen workgroups: where consecutive block indices go
    printf("  blocks 0..15 -> xcc:");
    for (int k = 0; k < 16 && k < n; ++k) printf(" %u", h[k].xcc_id & 15);
    printf("\n");
    return 0;
}
