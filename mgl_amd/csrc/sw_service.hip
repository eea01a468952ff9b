// sw_service.hip -- one pair per call (the reference's alignNative pattern, ..._MicrosoftSmithWaterman.cpp:44-71) with no launch and no
// synchronisation on the request path.
//
// Every calling thread owns a MAILBOX (ServiceRequest + ServiceReply, sw_device.h); ONE resident grid serves them, workgroup k (one
// wave) mailbox k:
//
//   host:  writes the pair and its parameters into the request -- fine-grained device memory behind the large BAR where the platform
//          has one, else pinned host memory -- then a new sequence number (seq_a, seq_b), and spins on the reply's done_seq;
//   wave:  lanes 0-15 read the request's first line until it shows a number not yet served, then the wave runs the pair --
//          small_pair() of sw_small_pair.h: sequences straight out of the request, H kept in LDS, the walk off the scores, text and
//          results straight into the reply (pinned host memory) -- and stores the number into done_seq behind a system-scope release.
//
// One grid, not one kernel per mailbox: HIP multiplexes its streams onto four hardware queues, and a resident kernel holds up every
// kernel behind it in its queue -- sixteen one-wave kernels on sixteen streams ran four at a time (measured: 85 k pairs/s from 16
// threads, each wave living out its whole lifetime while twelve others waited).  The grid has a stream of its own, created at the
// highest priority so that it shares its queue with none of the library's other streams.
//
// The grid must never outlive its use (a resident kernel blocks every device-wide synchronisation of the process): every turn of
// a wave's poll loop compares the 100 MHz clock with the time of the last request ANY wave has served (ServiceControl, device
// memory) and with the time of the launch, and the loop ends when the service has been quiet for `idle_ticks`, when the grid is
// `life_ticks` old, or when the host asks (quit_gen) -- conditions every wave reaches whatever the host does.  The first wave that
// decides to end latches stop_gen, and the others end at their next look: the grid is never half alive for longer than a poll, so
// the caller that finds its wave gone (state EXITED) can launch the next grid on the same stream at once (sw_service.cpp); the new
// waves pick up whatever numbers are waiting.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#undef MGL_SMALL_PHASES
#include "sw_small_pair.h"

namespace mgl_sw_dev {

namespace {

__device__ __forceinline__ uint32_t load_system(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void store_system(uint32_t *p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

} // namespace

__global__ __launch_bounds__(64) void sw_service_kernel(const ServiceRequest *const requests, ServiceReply *const replies, ServiceControl *const ctl, const uint32_t gen,
                                                        const uint32_t idle_ticks, const uint32_t life_ticks, const int lds_bytes)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
    const int lane = threadIdx.x;
    const ServiceRequest *const rq = requests + blockIdx.x;
    ServiceReply *const rp = replies + blockIdx.x;
    uint32_t served = __builtin_amdgcn_readfirstlane(load_system(&rp->done_seq)); // (the last number a wave of an earlier grid handed back)
    if (lane == 0) store_system(&rp->state, gen << 4 | SERVICE_RUNNING);
    const uint64_t t_start = __builtin_amdgcn_s_memrealtime();
    uint64_t t_own = t_start; // this mailbox's last request
    const uint32_t *line0 = reinterpret_cast<const uint32_t *>(rq);
    for (;;) {
        // the first line of the mailbox, one dword per lane: one read over the link per turn
        const uint32_t w = lane < 16 ? load_system(line0 + lane) : 0u;
        const uint32_t seq_a = __builtin_amdgcn_readlane(w, 0), seq_b = __builtin_amdgcn_readlane(w, 15);
        const uint32_t quit_gen = __builtin_amdgcn_readlane(w, 10);
        if (seq_a == seq_b && seq_a != served && (int)__builtin_amdgcn_readlane(w, 11) > lds_bytes) {
            // a pair that needs more LDS than this grid was launched with (the default leaves room on the CU for other work): the grid
            // ends -- every wave at its next look -- and the caller, who finds its wave gone, launches one with the larger carve
            if (lane == 0) __hip_atomic_fetch_max(&ctl->stop_gen, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            break;
        }
        if (seq_a == seq_b && seq_a != served) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); // the sequences were written before the number
            TbArgs a{};
            a.t = SeqSet{rq->t, nullptr, nullptr, 0, 0};
            a.q = SeqSet{rq->q, nullptr, nullptr, 0, 0};
            const int tl = (int)__builtin_amdgcn_readlane(w, 1), ql = (int)__builtin_amdgcn_readlane(w, 2);
            a.match = (int)__builtin_amdgcn_readlane(w, 3);
            a.mismatch = (int)__builtin_amdgcn_readlane(w, 4);
            a.gopen = (int)__builtin_amdgcn_readlane(w, 5);
            a.gext = (int)__builtin_amdgcn_readlane(w, 6);
            a.strategy = (int)__builtin_amdgcn_readlane(w, 7);
            a.cigar_stride = (int)__builtin_amdgcn_readlane(w, 8);
            const int wide = (int)__builtin_amdgcn_readlane(w, 9);
            a.offset = &rp->offset;
            a.score = &rp->score;
            a.cigar = rp->cigar;
            a.cigar_len = &rp->cigar_len;
            a.status = &rp->status;
            // (the host has checked the bounds: tl <= SERVICE_MAX_TL, ql <= SERVICE_MAX_QL, the stride within the mailbox's text, LDS)
            small_pair<false>(a, 0, tl, ql, 0, 0, lds, wide, lane);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); // results before the number
            if (lane == 0) store_system(&rp->done_seq, seq_a);
            served = seq_a;
            __builtin_amdgcn_wave_barrier();
            t_own = __builtin_amdgcn_s_memrealtime();
            if (lane == 0) __hip_atomic_fetch_max(&ctl->last_activity, (unsigned long long)t_own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            continue;
        }
        const uint32_t stop_gen = __hip_atomic_load(&ctl->stop_gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t heard = __hip_atomic_load(&ctl->last_activity, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint64_t now = __builtin_amdgcn_s_memrealtime();
        // (generations are 28-bit numbers compared modulo 2^28: the counter wraps; the host sets stop_gen to gen - 1 in front of every launch)
        if (service_gen_reached(quit_gen, gen) || service_gen_reached(stop_gen, gen)) break;
        const uint64_t last = heard > t_start ? heard : t_start;
        if ((int64_t)(now - last) > (int64_t)idle_ticks || now - t_start > life_ticks) {
            if (lane == 0) __hip_atomic_fetch_max(&ctl->stop_gen, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // (generations grow: max is "latest")
            break;
        }
        // a mailbox in use is looked at as fast as the link answers; one that has been quiet for 200 us every few microseconds
        if (now - t_own > 20000)
            __builtin_amdgcn_s_sleep(127);
        else
            __builtin_amdgcn_s_sleep(4);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lane == 0) store_system(&rp->state, gen << 4 | SERVICE_EXITED);
}

hipError_t launch_service(const ServiceRequest *requests, ServiceReply *replies, ServiceControl *ctl, int slots, uint32_t gen, uint32_t idle_ticks, uint32_t life_ticks,
                          int lds_bytes, hipStream_t stream)
{
    if (slots < 1 || lds_bytes < 1 || lds_bytes > SERVICE_LDS_BYTES) return hipErrorInvalidValue;
    static std::atomic<unsigned long long> raised{0}; // the attribute is per device: raised once on each
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = dev < 64 ? 1ull << dev : 0ull;
    if (!(raised.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(sw_service_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, SERVICE_LDS_BYTES);
        if (e != hipSuccess) return e;
        raised.fetch_or(bit, std::memory_order_release);
    }
    // stop_gen = the generation before this one: whatever an earlier grid latched (or a wrapped counter left) no longer stops this grid
    e = hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(&ctl->stop_gen), (int)((gen - 1u) & SERVICE_GEN_MASK), 1, stream);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(sw_service_kernel, dim3((unsigned)slots), dim3(64), (size_t)lds_bytes, stream, requests, replies, ctl, gen, idle_ticks, life_ticks, lds_bytes);
    return hipGetLastError();
}

} // namespace mgl_sw_dev
