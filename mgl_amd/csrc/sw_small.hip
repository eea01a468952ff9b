// sw_small.hip -- small batches (a few dozen to a couple of thousand pairs) are latency bound: what counts is how long ONE pair
// takes from launch to CIGAR, not how many the chip holds.  One wave per pair, one launch, nothing in HBM but the inputs and the
// results: the pair's body is small_pair() of sw_small_pair.h (fill with H kept in LDS, the walk reads every move off the scores).
// Geometry limits (small_supported): tl <= 512, the H matrix + sequences + text within a workgroup's LDS, text CIGARs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdio>
#include <cstring>

#include "sw_small_pair.h"

namespace mgl_sw_dev {

int small_lds_bytes(int max_tl, int max_ql, int cigar_stride, bool wide)
{
    // (small_column_words(tl) <= (tl + 7) / 2 + 1, or tl + 8, for every tl <= max_tl)
    const int64_t kept = (int64_t)(wide ? max_tl + 8 : (max_tl + 7) / 2 + 1) * (max_ql + 1) * 4; // (+ one spare column)
    const int64_t b = kept + ((max_tl + 3) & ~3) + ((max_ql + 3) & ~3) + small_text_cap(max_tl, max_ql, cigar_stride) + 8;
    return b > (1 << 30) ? (1 << 30) : (int)b;
}
bool small_fits_int16(int max_tl, int max_ql, int match, int mismatch, int gopen, int gext)
{
    if (match < 0 || mismatch > match || gopen < 0 || gext < 0) return false;
    const int64_t hi = (int64_t)match * (max_tl < max_ql ? max_tl : max_ql) + ((int64_t)max_tl + max_ql) * gext;
    const int64_t lo = -2 * (int64_t)gopen - ((int64_t)max_tl + max_ql) * gext;
    return hi - lo <= 65000; // (the span; every pair of the batch centres its own)
}

__global__ __launch_bounds__(64) void sw_small_kernel(const TbArgs a, const int max_tl, const int max_ql, const int wide)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int64_t p = a.first + blockIdx.x;
    small_pair<true>(a, a.dest ? a.dest[p] : p, a.t.length(p), a.q.length(p), a.t.off[p], a.q.off[p], lds, wide, (int)threadIdx.x);
}
bool small_supported(int max_tl, int max_ql, int cigar_stride, int match, int mismatch, int gopen, int gext, bool *wide)
{
    if (max_tl > 512 || max_tl < 1 || max_ql < 1 || !small_mul24_ok(match, mismatch, gext)) return false;
    bool w = !small_fits_int16(max_tl, max_ql, match, mismatch, gopen, gext);
    if (small_lds_bytes(max_tl, max_ql, cigar_stride, w) > 160 * 1024) return false;
    if (wide) *wide = w;
    return true;
}

hipError_t launch_small(const TbArgs &a, int max_tl, int max_ql, bool wide, hipStream_t stream)
{
    if (a.count <= 0) return hipSuccess;
    if (a.binary_cigar || a.count > (1 << 30)) return hipErrorInvalidValue;
    const int lds = small_lds_bytes(max_tl, max_ql, a.cigar_stride, wide);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (lds > 64 * 1024) { // the attribute is per device: raised once on each (this launch sits on the latency path of every small batch)
        static std::atomic<unsigned long long> raised{0};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;
        if (!(raised.load(std::memory_order_acquire) & bit)) {
            e = hipFuncSetAttribute(reinterpret_cast<const void *>(sw_small_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            if (e != hipSuccess) return e;
            raised.fetch_or(bit, std::memory_order_release);
        }
    }
    hipLaunchKernelGGL(sw_small_kernel, dim3((unsigned)a.count), dim3(64), (size_t)lds, stream, a, max_tl, max_ql, wide ? 1 : 0);
    return hipGetLastError();
}

} // namespace mgl_sw_dev

#ifdef MGL_SMALL_PHASES
extern "C" void mgl_small_phases_dump()
{
    unsigned long long h[8] = {0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h, HIP_SYMBOL(mgl_sw_dev::mgl_small_phase_ticks), sizeof h);
    static const char *names[5] = {"staging", "fill", "last column + row", "walk", "results"};
    for (int k = 0; k < 5; ++k) fprintf(stderr, "phase %-20s %8.2f us per wave\n", names[k], (double)h[k] / (double)h[7] / 100.0);
    fprintf(stderr, "waves %llu\n", h[7]);
    memset(h, 0, sizeof h);
    hipMemcpyToSymbol(HIP_SYMBOL(mgl_sw_dev::mgl_small_phase_ticks), h, sizeof h);
}
#endif
