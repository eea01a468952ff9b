// sw_service.cpp -- host side of the one-pair-per-call service (device side and protocol: sw_service.hip, ServiceMailbox in sw_device.h).
//
// GATK reaches the aligner through alignNative, one pair per call, from a bounded set of threads
// (/root/reference/src/main/java/com/microsoft/mgl/smithwaterman/MicrosoftSmithWaterman.java:66-86, ..._MicrosoftSmithWaterman.cpp:44-71).
// Every calling thread leases a mailbox (pinned host memory) for as long as it lives.  A call copies the pair into the mailbox,
// publishes a new sequence number and spins until the mailbox's resident wave hands the same number back: no launch, no stream
// synchronisation, no other thread involved.  When the caller finds its wave gone (state EXITED: the service was quiet for idle_us,
// or the grid reached its lifetime) or its mailbox beyond the running grid, it launches the grid again -- one workgroup per mailbox
// in use, on the service's own stream -- the only time a call touches the HIP runtime.
//
// What does not fit a mailbox (targets beyond 512 rows, queries beyond 2 048 bases, a score matrix beyond a workgroup's LDS, more
// calling threads than mailboxes) is declined and takes the coalescing front-end (sw_batcher.cpp), as before.
#include "../../include/mgl_sw.h"
#include "sw_device.h"
#include "sw_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include <sched.h>
#include <time.h>

#define MGL_SW_INTERNAL __attribute__((visibility("hidden")))

namespace {

using namespace mgl_sw_dev;
using Clock = std::chrono::steady_clock;

constexpr int MAX_SLOTS = 128;
constexpr int SERVICE_DECLINED = 1 << 20; // what mgl_sw_service_align returns for a call that does not go through a mailbox (sw_capi.cpp takes the coalescer then)

inline uint32_t load_acquire(const uint32_t *p) { return __atomic_load_n(p, __ATOMIC_ACQUIRE); }
inline void store_release(uint32_t *p, uint32_t v) { __atomic_store_n(p, v, __ATOMIC_RELEASE); }

struct Slot {
    ServiceRequest *rq = nullptr; // where the host writes (device memory behind the BAR, or pinned host memory)
    ServiceReply *rp = nullptr;   // where the host reads (pinned host memory)
    int index = 0;
    uint32_t seq = 0;
    bool leased = false;
};

std::atomic<bool> g_pool_alive{false};

// The service's HIP calls (allocation, launches) happen on the CALLER's thread: its current device is put back when they are done -- a
// caller with its own HIP or torch state on another GPU must not find it changed by a one-pair call.
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device)
    {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = hipSetDevice(device) == hipSuccess;
    }
    ~DeviceGuard()
    {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
};

class ServicePool {
  public:
    static ServicePool &instance()
    {
        static ServicePool p;
        return p;
    }
    ServicePool()
    {
        if (const char *e = getenv("MGL_SW_SERVICE_SLOTS")) max_slots_.store(std::max(0, std::min(MAX_SLOTS, atoi(e))));
        if (const char *e = getenv("MGL_SW_SERVICE_IDLE_US")) idle_us_.store(std::max(1, atoi(e)));
        if (const char *e = getenv("MGL_SW_SERVICE_LIFE_MS")) life_ms_.store(std::max(1, atoi(e)));
        if (const char *e = getenv("MGL_SW_DEVICE")) device_ = atoi(e);
        g_pool_alive.store(true);
    }
    // library unload / process exit: ask the grid to end, wait until it has, release everything
    ~ServicePool()
    {
        g_pool_alive.store(false); // (from here on the leases of threads that end do not come back: give_back is not called any more)
        broken_.store(true);       // ... and a call that is still on its way is declined
        int slots = 0;
        uint32_t gen = 0;
        {
            std::lock_guard<std::mutex> lk(mu_);
            if (!reps_) return;
            slots = grid_slots_;
            gen = grid_gen_;
        }
        // (the lock is NOT held across the wait below: a caller inside relaunch() must be able to finish)
        for (int k = 0; k < slots; ++k) __atomic_store_n(&reqs_[k].quit_gen, gen, __ATOMIC_RELEASE);
        flush_stores();
        DeviceGuard dev(device_);
        (void)hipStreamSynchronize(stream_); // (bounded by the waves' own conditions even if the request went unseen)
        std::lock_guard<std::mutex> lk(mu_);
        (void)hipStreamDestroy(stream_);
        (void)hipFree(ctl_);
        if (over_bar_)
            (void)hipFree(reqs_);
        else
            (void)hipHostFree(reqs_);
        (void)hipHostFree(reps_);
        reps_ = nullptr;
    }

    void configure(int slots, int idle_us)
    {
        max_slots_.store(std::max(0, std::min(MAX_SLOTS, slots)));
        if (idle_us > 0) idle_us_.store(idle_us);
    }
    void stats(int64_t *calls, int64_t *launches) const
    {
        *calls = calls_.load();
        *launches = launches_.load();
    }

    // the calling thread's mailbox, leased at its first call and handed back when the thread ends; nullptr: none left (or switched off)
    Slot *lease()
    {
        struct Lease {
            Slot *s = nullptr;
            ~Lease()
            {
                if (s && g_pool_alive.load()) ServicePool::instance().give_back(s);
            }
        };
        static thread_local Lease mine;
        // (a mailbox's wave holds its carve of LDS for as long as the grid lives: never more mailboxes than a quarter of the CUs' worth of
        // default carves -- two per CU on half the CUs -- whatever the setting; cu_cap_ is known once the pool has allocated)
        const int cap = std::min(max_slots_.load(std::memory_order_relaxed), cu_cap_.load(std::memory_order_relaxed));
        if (broken_.load(std::memory_order_relaxed)) return nullptr;
        if (mine.s) {
            if (mine.s->index < cap) return mine.s;
            give_back(mine.s); // (the service has been switched off or cut down since)
            mine.s = nullptr;
        }
        if (cap == 0) return nullptr;
        std::lock_guard<std::mutex> lk(mu_);
        if (broken_) return nullptr;
        if (!reps_ && !allocate_locked()) return nullptr;
        for (int k = 0; k < cap; ++k)
            if (!slots_[k].leased) {
                slots_[k].leased = true;
                leased_.fetch_add(1, std::memory_order_relaxed);
                used_slots_ = std::max(used_slots_, k + 1);
                return mine.s = &slots_[k];
            }
        return nullptr;
    }

    int call(Slot &s, const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen, int gext, int strategy, int stride,
             int wide, int lds_need, char *cigar, int cigar_cap, int *cigar_len, int *offset, mgl_sw_score *ez)
    {
        ServiceRequest &rq = *s.rq;
        ServiceReply &mb = *s.rp;
        memcpy(rq.t, t, (size_t)tl);
        memcpy(rq.q, q, (size_t)ql);
        rq.tl = tl;
        rq.ql = ql;
        rq.match = match;
        rq.mismatch = mismatch;
        rq.gopen = gopen;
        rq.gext = gext;
        rq.strategy = strategy;
        rq.cigar_stride = stride;
        rq.wide = wide;
        rq.lds_need = lds_need;
        // a pair beyond the carve the grids are launched with: the next grid gets the full one (the running grid ends when its wave sees this request)
        if (lds_need > lds_tier_.load(std::memory_order_relaxed)) lds_tier_.store(SERVICE_LDS_BYTES, std::memory_order_relaxed);
        const uint32_t seq = ++s.seq;
        // (device memory behind the BAR is write-combining for the host: stores leave in no particular order, and only when a buffer is
        // evicted or a fence says so -- the pair before the numbers, the numbers at once)
        flush_stores();
        store_release(&rq.seq_a, seq);
        store_release(&rq.seq_b, seq);
        flush_stores();
        calls_.fetch_add(1, std::memory_order_relaxed);
        // the answer is a device-side latency away (tens of microseconds): spin; a thread that has spun for long gives its CPU away between looks
        // With more than two calling threads per CPU a spinning caller only keeps others from posting their pairs: such callers sleep
        // through the wave's work first (a sleep costs a timer's slack on top, ~50 us, but no CPU time -- what a container's CPU
        // quota counts), and the waves work on as many pairs as there are threads, not CPUs.  Measured on a 16-CPU quota, pairs/s
        // from 16 / 32 / 64 threads: spinning 468 k / 509 k / 451 k, yielding between looks - / 428 k / 428 k, sleeping first - / - / 846 k.
        if (leased_.load(std::memory_order_relaxed) > 2 * cpus_) {
            const timespec nap{0, 20000};
            nanosleep(&nap, nullptr);
        }
        const auto t0 = Clock::now();
        for (unsigned spins = 0;; ++spins) {
            if (load_acquire(&mb.done_seq) == seq) break;
            const uint32_t st = load_acquire(&mb.state);
            if ((st & 15u) == SERVICE_EXITED || (st & 15u) == SERVICE_IDLE) {
                if (load_acquire(&mb.done_seq) == seq) break; // (served by the wave's last turn)
                const int rc = relaunch(s, st);
                if (rc != MGL_SW_OK) return rc;
            }
            if (spins < 4096) {
                cpu_relax();
            } else {
                // (more callers than CPUs, or a grid being launched: sleep between looks instead of fighting the others for the CPU)
                const timespec nap{0, spins < 4200 ? 20000 : 100000};
                nanosleep(&nap, nullptr);
                if ((spins & 255) == 0 && Clock::now() - t0 > std::chrono::seconds(20)) {
                    // No answer: the service is switched off for the rest of the process and THIS call goes the way of a declined one (the
                    // coalescer; nothing of the caller's was handed to the device: a late answer lands in the mailbox, which stays leased
                    // to its thread, and is never read).
                    std::lock_guard<std::mutex> lk(mu_);
                    broken_ = true;
                    return SERVICE_DECLINED;
                }
            }
        }
        *cigar_len = mb.cigar_len;
        if (mb.status != 0) return mb.status; // the pair's own status, verbatim
        if (mb.cigar_len > cigar_cap) return MGL_SW_ERR_CIGAR_OVERFLOW; // (the mailbox's slot is the caller's capacity rounded up to a dword)
        memcpy(cigar, mb.cigar, (size_t)mb.cigar_len); // exactly cigar.length() bytes, no terminator (.cpp:65)
        *offset = mb.offset;
        if (ez) memcpy(ez, &mb.score, sizeof(mgl_sw_score));
        return MGL_SW_OK;
    }

  private:
    bool allocate_locked()
    {
        void *p = nullptr, *d = nullptr, *c = nullptr, *r = nullptr, *rd = nullptr;
        int lo = 0, hi = 0, large_bar = 0, cus = 0;
        DeviceGuard dev(device_);
        if (!dev.ok || hipHostMalloc(&p, sizeof(ServiceReply) * MAX_SLOTS, hipHostMallocDefault) != hipSuccess) {
            broken_ = true;
            return false;
        }
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_) == hipSuccess && cus > 0) cu_cap_.store(std::max(1, cus / 2));
        memset(p, 0, sizeof(ServiceReply) * MAX_SLOTS);
        // the requests: device memory the host stores into directly, where the platform allows (MGL_SW_SERVICE_BAR=0: never)
        const char *const bar = getenv("MGL_SW_SERVICE_BAR");
        if ((!bar || atoi(bar) != 0) && hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device_) == hipSuccess && large_bar &&
            hipExtMallocWithFlags(&r, sizeof(ServiceRequest) * MAX_SLOTS, hipDeviceMallocFinegrained) == hipSuccess) {
            if (hipMemset(r, 0, sizeof(ServiceRequest) * MAX_SLOTS) == hipSuccess && hipDeviceSynchronize() == hipSuccess) {
                over_bar_ = true;
                rd = r;
            } else {
                (void)hipFree(r);
                r = nullptr;
            }
        }
        if (!r) {
            if (hipHostMalloc(&r, sizeof(ServiceRequest) * MAX_SLOTS, hipHostMallocDefault) != hipSuccess || hipHostGetDevicePointer(&rd, r, 0) != hipSuccess) {
                if (r) (void)hipHostFree(r);
                (void)hipHostFree(p);
                broken_ = true;
                return false;
            }
            memset(r, 0, sizeof(ServiceRequest) * MAX_SLOTS);
        }
        (void)hipDeviceGetStreamPriorityRange(&lo, &hi); // (hi = the numerically lowest = the highest priority: the library's other streams are normal or lowest)
        if (hipHostGetDevicePointer(&d, p, 0) != hipSuccess || hipMalloc(&c, sizeof(ServiceControl)) != hipSuccess ||
            hipMemset(c, 0, sizeof(ServiceControl)) != hipSuccess || hipStreamCreateWithPriority(&stream_, hipStreamNonBlocking, hi) != hipSuccess) {
            if (c) (void)hipFree(c);
            if (over_bar_)
                (void)hipFree(r);
            else
                (void)hipHostFree(r);
            (void)hipHostFree(p);
            broken_ = true;
            return false;
        }
        reps_ = static_cast<ServiceReply *>(p);
        reps_dev_ = static_cast<ServiceReply *>(d);
        reqs_ = static_cast<ServiceRequest *>(r);
        reqs_dev_ = static_cast<const ServiceRequest *>(rd);
        ctl_ = static_cast<ServiceControl *>(c);
        for (int k = 0; k < MAX_SLOTS; ++k) {
            slots_[k].rq = reqs_ + k;
            slots_[k].rp = reps_ + k;
            slots_[k].index = k;
        }
        return true;
    }
    void flush_stores() const
    {
#if defined(__x86_64__) || defined(__i386__)
        if (over_bar_) __builtin_ia32_sfence();
#endif
    }
    void give_back(Slot *s)
    {
        std::lock_guard<std::mutex> lk(mu_);
        leased_.fetch_sub(1, std::memory_order_relaxed);
        s->leased = false; // (its wave goes on looking at the mailbox, every few microseconds, until the grid ends)
    }
    // The caller's wave is gone (`seen` = the state it read: EXITED of some generation) or its mailbox was never part of a grid (IDLE).
    // One caller launches the next grid -- over every mailbox handed out so far -- the others find that done and go on waiting.
    int relaunch(Slot &s, uint32_t seen)
    {
        std::lock_guard<std::mutex> lk(mu_);
        if (broken_) return MGL_SW_ERR_DEVICE;
        const uint32_t st = load_acquire(&s.rp->state);
        const bool covered = s.index < grid_slots_;
        if (covered && ((st & 15u) == SERVICE_LAUNCHED || (st & 15u) == SERVICE_RUNNING)) return MGL_SW_OK; // somebody else has
        if (covered && (st >> 4) != grid_gen_) return MGL_SW_OK; // (EXITED of an older grid, written late over the LAUNCHED of the current one: its wave is on its way)
        (void)seen;
        DeviceGuard dev(device_);
        if (!dev.ok) return MGL_SW_ERR_DEVICE;
        // the running grid (if this mailbox is beyond it, it may be busy with the others) is asked to end; the next one starts behind it on the stream
        for (int k = 0; k < grid_slots_; ++k) __atomic_store_n(&reqs_[k].quit_gen, grid_gen_, __ATOMIC_RELEASE);
        flush_stores();
        uint32_t gen = (grid_gen_ + 1) & SERVICE_GEN_MASK;
        if (gen == 0) gen = 1; // (0 is what a mailbox that has never been asked to quit holds)
        const int n = used_slots_;
        for (int k = 0; k < n; ++k) store_release(&reps_[k].state, gen << 4 | SERVICE_LAUNCHED);
        const uint32_t idle = (uint32_t)std::min<int64_t>((int64_t)idle_us_.load() * 100, 0x7fffffff);
        const uint32_t life = (uint32_t)std::min<int64_t>((int64_t)life_ms_.load() * 100000, 0x7fffffff);
        const hipError_t e = launch_service(reqs_dev_, reps_dev_, ctl_, n, gen, idle, life, lds_tier_.load(std::memory_order_relaxed), stream_);
        if (e != hipSuccess) {
            for (int k = 0; k < n; ++k) store_release(&reps_[k].state, grid_gen_ << 4 | SERVICE_EXITED);
            broken_ = true;
            return e == hipErrorOutOfMemory ? MGL_SW_ERR_NOMEM : MGL_SW_ERR_DEVICE;
        }
        grid_gen_ = gen;
        grid_slots_ = n;
        launches_.fetch_add(1, std::memory_order_relaxed);
        return MGL_SW_OK;
    }

    std::mutex mu_;
    Slot slots_[MAX_SLOTS];
    ServiceRequest *reqs_ = nullptr;            // MAX_SLOTS requests as the host writes them ...
    const ServiceRequest *reqs_dev_ = nullptr;  // ... and as the waves read them
    ServiceReply *reps_ = nullptr, *reps_dev_ = nullptr; // MAX_SLOTS replies, pinned host memory
    bool over_bar_ = false;                     // the requests are device memory the host stores into over the BAR
    ServiceControl *ctl_ = nullptr;
    hipStream_t stream_ = nullptr;
    // guarded by mu_
    uint32_t grid_gen_ = 0; // generation of the last grid launched (0: none yet)
    int grid_slots_ = 0;    // ... and the mailboxes it covers
    int used_slots_ = 0;    // mailboxes handed out so far (a grid covers all of them: a thread that comes back finds its wave)
    int device_ = 0;
    std::atomic<int> max_slots_{64}, idle_us_{1000}, life_ms_{20};
    std::atomic<int> cu_cap_{MAX_SLOTS};             // half the device's CUs, once known
    std::atomic<int> lds_tier_{SERVICE_LDS_DEFAULT}; // dynamic LDS of the next grid: the default until a pair has needed more
    std::atomic<int> leased_{0};
    const int cpus_ = usable_cpus();
    std::atomic<int64_t> calls_{0}, launches_{0};
    std::atomic<bool> broken_{false}; // a launch failed or a wave never answered: every call is declined from then on
};

} // namespace

// library-internal (mgl_sw_align, sw_capi.cpp): MGL_SW_SERVICE_DECLINED = this call does not go through a mailbox
extern "C" MGL_SW_INTERNAL int mgl_sw_service_align(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen, int gext, int strategy,
                                                    char *cigar, int cigar_cap, int *cigar_len, int *offset, mgl_sw_score *ez)
{
    constexpr int DECLINED = SERVICE_DECLINED;
    if (tl > SERVICE_MAX_TL || ql > SERVICE_MAX_QL) return DECLINED;
    mgl_sw_normalize_params(&match, &mismatch, &gopen, &gext);
    if (((int64_t)match - mismatch + gopen + 2 * (int64_t)gext) * ((int64_t)tl + ql) >= (1ll << 30)) return DECLINED; // (run_device's bound for 32-bit scores)
    if (!small_mul24_ok(match, mismatch, gext)) return DECLINED; // (small_pair()'s 24-bit products)
    const int stride = std::min((cigar_cap + 3) & ~3, (2 * (tl + ql) + 4 + 3) & ~3); // no CIGAR of this pair is longer than that
    const bool wide = !small_fits_int16(tl, ql, match, mismatch, gopen, gext);
    const int lds_need = small_lds_bytes(tl, ql, stride, wide);
    if (stride > SERVICE_TEXT_BYTES || lds_need > SERVICE_LDS_BYTES) return DECLINED;
    try { // (nothing may leave through the C ABI: the caller may be a JVM)
        ServicePool &pool = ServicePool::instance();
        Slot *s = pool.lease();
        if (!s) return DECLINED;
        return pool.call(*s, t, tl, q, ql, match, mismatch, gopen, gext, strategy, stride, wide ? 1 : 0, lds_need, cigar, cigar_cap, cigar_len, offset, ez);
    } catch (...) {
        return MGL_SW_ERR_DEVICE;
    }
}

extern "C" {

int mgl_sw_set_service(int slots, int idle_us)
{
    if (slots < 0 || idle_us < 0) return MGL_SW_ERR_BAD_ARG;
    try {
        ServicePool::instance().configure(slots, idle_us);
    } catch (...) {
        return MGL_SW_ERR_DEVICE;
    }
    return MGL_SW_OK;
}

int mgl_sw_service_stats(int64_t *calls, int64_t *launches)
{
    if (!calls || !launches) return MGL_SW_ERR_BAD_ARG;
    try {
        ServicePool::instance().stats(calls, launches);
    } catch (...) {
        return MGL_SW_ERR_DEVICE;
    }
    return MGL_SW_OK;
}

} // extern "C"
