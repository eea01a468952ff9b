// sw_batcher.cpp -- coalescing front-end for one-pair-per-call callers.
//
// GATK reaches the aligner through alignNative, one pair per call, from many threads
// (/root/reference/src/main/java/com/microsoft/mgl/smithwaterman/MicrosoftSmithWaterman.java:66-86,
// ..._MicrosoftSmithWaterman.cpp:44-71).  One pair per launch cannot feed a GPU, so unless coalescing is switched
// off (mgl_sw_set_coalescing(0, 0) or the environment variable MGL_SW_COALESCE_US=-1) mgl_sw_align -- and therefore the
// JNI export built on it -- parks the calling thread, a dispatcher thread merges the requests that share one
// parameter set and strategy into a device batch (mgl_sw_align_batch), and every caller gets exactly the answer
// the direct call would have produced.  Host code only: queues, one std::thread, condition variables.
// A device round trip is a few dozen microseconds (one launch of sw_small_kernel), the same order as waking a parked thread: callers
// and dispatcher therefore spin for a short while (MGL_SW_COALESCE_SPIN_US, default 200) before they park on their condition variables.
#include "../../include/mgl_sw.h"
#include "sw_host.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <vector>

#include <climits>
#include <linux/futex.h>
#include <sched.h>
#include <sys/syscall.h>
#include <unistd.h>

// sw_capi.cpp (library-internal, not part of include/mgl_sw.h: hidden visibility, not exported from the .so)
#define MGL_SW_INTERNAL __attribute__((visibility("hidden")))
extern "C" MGL_SW_INTERNAL int mgl_sw_stage_buffers(mgl_sw_ctx *ctx, size_t in_bytes, size_t out_bytes, void **in, void **out);
extern "C" MGL_SW_INTERNAL int mgl_sw_align_batch_staged(mgl_sw_ctx *ctx, int n, size_t in_bytes, size_t t_bytes_padded, int max_tl, int max_ql,
                                         int match, int mismatch, int gopen, int gext, int strategy, int cigar_stride,
                                         size_t out_bytes);

namespace {

using Clock = std::chrono::steady_clock;
using Key = std::tuple<int, int, int, int, int>; // normalised match, mismatch, open, extend, strategy

// Where callers sleep when their answer takes longer than they are willing to spin: ONE futex word for all of them.  A finished batch
// sets its requests' `done` flags, bumps the word and wakes every sleeper with a single system call (48 sleeping callers of a batch
// of 64: one FUTEX_WAKE instead of 48 condition-variable notifications, no mutex for the woken threads to queue up on); a sleeper
// whose request was not in that batch goes back to sleep on the new value.
class WakeWord {
  public:
    // sleep until `done` reads true
    void wait(const std::atomic<bool> &done)
    {
        sleepers_.fetch_add(1, std::memory_order_seq_cst);
        for (;;) {
            const uint32_t seen = word_.load(std::memory_order_seq_cst);
            if (done.load(std::memory_order_seq_cst)) break;
            syscall(SYS_futex, reinterpret_cast<uint32_t *>(&word_), FUTEX_WAIT_PRIVATE, seen, nullptr, nullptr, 0); // returns at once if the word moved on
        }
        sleepers_.fetch_sub(1, std::memory_order_seq_cst);
    }
    // after the `done` flags of a batch have been set
    void wake_all()
    {
        word_.fetch_add(1, std::memory_order_seq_cst);
        if (sleepers_.load(std::memory_order_seq_cst) > 0) syscall(SYS_futex, reinterpret_cast<uint32_t *>(&word_), FUTEX_WAKE_PRIVATE, INT_MAX, nullptr, nullptr, 0);
    }

  private:
    alignas(64) std::atomic<uint32_t> word_{0};
    alignas(64) std::atomic<int> sleepers_{0};
};
static_assert(sizeof(std::atomic<uint32_t>) == sizeof(uint32_t), "the futex word is the atomic itself");

struct Request {
    const char *t, *q;
    int tl, ql;
    char *cigar;
    int cigar_cap;
    int *cigar_len, *offset;
    mgl_sw_score *ez;
    Key key;
    Clock::time_point arrived;
    int rc = MGL_SW_OK;
    std::atomic<Request *> next{nullptr};
    std::atomic<bool> done{false}; // the dispatcher's last touch: the request lives on its caller's stack and is gone once this reads true
};

// The submission queue: many callers push, the dispatcher pops (D. Vyukov's intrusive MPSC queue).  A push is ONE exchange on a
// shared cache line; the sixteen callers of a batch all come back within microseconds of each other, and behind a lock (even a
// spinning one: lock word, queue, counters) their arrivals spread over ~1.5 us each on a two-socket host.
class SubmissionQueue {
  public:
    SubmissionQueue() : tail_(&stub_), head_(&stub_) {}
    // (every access is sequentially consistent: the dispatcher raises its `parked` flag and then looks here, a caller links its
    // request in here and then looks at the flag -- one of the two must see the other)
    void push(Request *r)
    {
        r->next.store(nullptr);
        Request *prev = tail_.exchange(r);
        prev->next.store(r); // (until this store `prev` cannot be popped: it is never freed under us)
    }
    // consumer only; nullptr: empty, or a push is half done (its caller is about to finish it)
    Request *pop()
    {
        Request *head = head_, *next = head->next.load();
        if (head == &stub_) {
            if (!next) return nullptr;
            head_ = head = next;
            next = head->next.load();
        }
        if (next) {
            head_ = next;
            return head;
        }
        if (head != tail_.load()) return nullptr;
        push(&stub_);
        next = head->next.load();
        if (!next) return nullptr;
        head_ = next;
        return head;
    }

  private:
    alignas(64) std::atomic<Request *> tail_; // the callers' cache line
    alignas(64) Request *head_;               // the dispatcher's
    Request stub_;
};

class Coalescer {
  public:
    static Coalescer &instance()
    {
        static Coalescer c;
        return c;
    }

    void configure(int max_batch, int max_wait_us)
    {
        std::lock_guard<std::mutex> lk(cfg_mu_);
        max_batch_.store(max_batch);
        max_wait_us_.store(max_wait_us);
        enabled_.store(max_batch > 0 && max_wait_us >= 0);
        if (enabled_ && !worker_.joinable()) worker_ = std::thread([this] { run(); });
        wake_dispatcher();
    }
    bool enabled() const { return enabled_.load(std::memory_order_relaxed); }

    int submit(Request &r)
    {
        static std::atomic<int> threads_seen{0};
        thread_local const int my_index = threads_seen.fetch_add(1);
        r.arrived = Clock::now();
        queue_.push(&r);
        if (parked_.load(std::memory_order_seq_cst)) wake_dispatcher(); // (it re-checks the queue after raising the flag: no arrival is lost)
        // the answer is usually back within a device round trip: spin that long before parking -- the first threads of the process
        // only, as many as leave the dispatcher and the HIP runtime a CPU each (spinning threads beyond the CPUs this process may use
        // only take the time slices of the ones that have work)
        if (spin_us_ > 0 && my_index < spin_slots_) {
            const auto until = Clock::now() + std::chrono::microseconds(spin_us_);
            do {
                for (int k = 0; k < 64; ++k) {
                    if (r.done.load(std::memory_order_acquire)) return r.rc;
                    cpu_relax();
                }
            } while (Clock::now() < until);
        }
        wake_.wait(r.done);
        return r.rc;
    }

    void stats(int64_t *batches, int64_t *pairs)
    {
        *batches = n_batches_.load();
        *pairs = n_pairs_.load();
    }

    ~Coalescer()
    {
        stop_.store(true);
        wake_dispatcher();
        if (worker_.joinable()) worker_.join();
        const int64_t nbi = n_batches_.load();
        if (timing_ && nbi) {
            const double nb = (double)nbi;
            fprintf(stderr, "[mgl_sw] coalescer: %lld batches, mean %.1f pairs; per batch: %.1f us collecting (first caller %.1f us after the "
                            "previous release, last %.1f us after the first), %.1f us processing "
                            "(%.1f layout, %.1f device round trip, %.1f hand-out), %.1f us waking the callers\n",
                    (long long)nbi, (double)n_pairs_.load() / nb, t_wait_ / nb, t_idle_ / nb, t_spread_ / nb, t_process_ / nb, t_layout_ / nb, t_device_ / nb,
                    t_scatter_ / nb, t_release_ / nb);
        }
        if (ctx_) mgl_sw_ctx_destroy(ctx_);
    }

  private:
    void wake_dispatcher()
    {
        {
            std::lock_guard<std::mutex> lk(park_mu_);
            poke_ = true;
        }
        park_cv_.notify_one();
    }

    // move what the callers have pushed into the per-key queues (dispatcher only); returns the number moved
    int drain()
    {
        int n = 0;
        while (Request *r = queue_.pop()) {
            auto &qd = queues_[r->key];
            if (qd.empty()) oldest_[r->key] = r->arrived;
            qd.push_back(r);
            newest_ = r->arrived;
            ++pending_;
            ++n;
        }
        return n;
    }

    // until a caller arrives, `deadline` passes or somebody pokes (configure, destructor): spin while the traffic is hot, then park
    void wait_for_arrivals(Clock::time_point deadline)
    {
        if (spin_us_ > 0) {
            const auto until = std::min(deadline, Clock::now() + std::chrono::microseconds(spin_us_));
            do {
                for (int k = 0; k < 64; ++k) {
                    if (drain() > 0 || stop_.load(std::memory_order_relaxed)) return;
                    cpu_relax();
                }
            } while (Clock::now() < until);
            if (Clock::now() >= deadline) return;
        }
        parked_.store(true, std::memory_order_seq_cst);
        if (drain() == 0) { // (a push that completes from here on sees the flag and pokes)
            std::unique_lock<std::mutex> lk(park_mu_);
            if (deadline == Clock::time_point::max())
                park_cv_.wait(lk, [&] { return poke_; });
            else
                park_cv_.wait_until(lk, deadline, [&] { return poke_; });
            poke_ = false;
        }
        parked_.store(false, std::memory_order_seq_cst);
    }

    void run()
    {
        for (;;) {
            drain();
            if (stop_.load()) {
                fail_all(MGL_SW_ERR_DEVICE);
                return;
            }
            if (pending_ == 0) {
                wait_for_arrivals(Clock::time_point::max());
                continue;
            }
            // pick the queue that is full, or whose oldest request has waited long enough; otherwise wait until the
            // earliest deadline (new arrivals end the wait too)
            const auto now = Clock::now();
            const int max_batch = std::max(1, max_batch_.load()); // coalescing switched off with requests still queued: drain them
            const int max_wait_us = std::max(0, max_wait_us_.load());
            const Key *ready = nullptr;
            auto earliest = now + std::chrono::hours(1);
            for (auto &kv : queues_) {
                if (kv.second.empty()) continue;
                const auto deadline = oldest_[kv.first] + std::chrono::microseconds(max_wait_us);
                // as many callers as the previous batch held are back: nobody else is expected, go at once (a lone caller
                // never waits; sixteen steady callers are dispatched when the sixteenth arrives, not at the deadline)
                const auto ex = expect_.find(kv.first);
                const int expect = ex == expect_.end() ? 1 : ex->second;
                if ((int)kv.second.size() >= std::min(max_batch, expect) || deadline <= now) {
                    ready = &kv.first;
                    break;
                }
                earliest = std::min(earliest, deadline);
            }
            if (!ready) {
                wait_for_arrivals(earliest);
                continue;
            }
            const Key key = *ready;
            auto &qd = queues_[key];
            std::vector<Request *> batch;
            while (!qd.empty() && (int)batch.size() < max_batch) {
                batch.push_back(qd.front());
                qd.pop_front();
            }
            const auto t_first = oldest_[key], t_last = newest_;
            if (!qd.empty()) oldest_[key] = now; // the rest starts a new waiting period
            pending_ -= (int)batch.size();
            expect_[key] = (int)batch.size();
            const auto t_a = Clock::now();
            process(key, batch);
            const auto t_b = Clock::now();
            const int64_t n_in_batch = (int64_t)batch.size();
            release(batch);
            const auto t_c = Clock::now();
            n_pairs_.fetch_add(n_in_batch);
            const int64_t nb = n_batches_.fetch_add(1) + 1;
            if (timing_) { // diagnostic (MGL_SW_DEBUG_COALESCE_TIMING): where a batch's round trip goes, microseconds
                auto us = [](Clock::duration d) { return std::chrono::duration<double, std::micro>(d).count(); };
                t_wait_ += us(now - t_first);
                if (nb > 1) {
                    t_idle_ += us(t_first - released_); // from the previous batch's release to this one's first caller
                    t_spread_ += us(t_last - t_first);  // ... to its last
                }
                released_ = t_c;
                t_process_ += us(t_b - t_a);
                t_release_ += us(t_c - t_b);
            }
        }
    }

    // hand the results back.  `done` is the last touch of a request (it lives on its caller's stack): spinning callers leave at once,
    // the sleeping ones on the one wake-up call behind the loop
    void release(const std::vector<Request *> &batch)
    {
        for (Request *r : batch) r->done.store(true, std::memory_order_seq_cst);
        wake_.wake_all();
    }

    void fail_all(int rc)
    {
        drain();
        std::vector<Request *> all;
        for (auto &kv : queues_)
            for (Request *r : kv.second) {
                r->rc = rc;
                all.push_back(r);
            }
        queues_.clear();
        pending_ = 0;
        release(all);
    }

    void process(const Key &key, std::vector<Request *> &batch)
    {
        const int n = (int)batch.size();
        if (!ctx_) {
            int dev = 0;
            if (const char *e = getenv("MGL_SW_DEVICE")) dev = atoi(e);
            const int rc = mgl_sw_ctx_create(dev, &ctx_);
            if (rc != MGL_SW_OK) {
                for (Request *r : batch) r->rc = rc;
                return;
            }
            mgl_sw_ctx_set_workspace(ctx_, 8ll << 30); // a cap: the workspace grows with what the batches need
            // small batches are latency bound: one workgroup of four waves per pair (sw_dp_coop.hip) finishes a
            // 256 x 150 pair in a quarter of the time of one wave walking its 16 stripes (tests/cpp/coalesce_bench)
            mgl_sw_ctx_set_cooperative(ctx_, 4);
            // ... and where a pair's matrix of scores fits LDS, one wave does everything in one launch (sw_small.hip)
            mgl_sw_ctx_set_small_kernel(ctx_, 2);
        }
        // lay the batch out in the context's pinned staging buffer (see mgl_sw_align_batch_staged)
        int stride = 16, max_tl = 1, max_ql = 1;
        size_t t_bytes = 0, q_bytes = 0;
        for (Request *r : batch) {
            t_bytes += (size_t)r->tl;
            q_bytes += (size_t)r->ql;
            stride = std::max(stride, r->cigar_cap);
            max_tl = std::max(max_tl, r->tl);
            max_ql = std::max(max_ql, r->ql);
        }
        stride = std::min((stride + 3) & ~3, 1 << 16); // a slot larger than any caller's buffer is pointless
        // ... and so is one larger than any CIGAR of these pairs: an element of length n takes at most 2 n characters and the
        // lengths add up to at most tl + ql (the slot is zero-filled over the link)
        stride = std::min(stride, (2 * (max_tl + max_ql) + 4 + 3) & ~3);
        const size_t offs = (size_t)(n + 1) * 8, t_pad = (t_bytes + 7) & ~(size_t)7, q_pad = (q_bytes + 7) & ~(size_t)7;
        const size_t in_bytes = 2 * offs + t_pad + q_pad;
        const size_t out_bytes = (size_t)n * (12 + sizeof(mgl_sw_score)) + (size_t)n * stride;
        void *in = nullptr, *out = nullptr;
        const auto t_0 = std::chrono::steady_clock::now();
        auto t_1 = t_0, t_2 = t_0;
        int rc = mgl_sw_stage_buffers(ctx_, in_bytes, out_bytes, &in, &out);
        if (rc == MGL_SW_OK) {
            int64_t *toff = static_cast<int64_t *>(in), *qoff = toff + (n + 1);
            uint8_t *tb = static_cast<uint8_t *>(in) + 2 * offs, *qb = tb + t_pad;
            toff[0] = qoff[0] = 0;
            for (int k = 0; k < n; ++k) {
                const Request *r = batch[(size_t)k];
                memcpy(tb + toff[k], r->t, (size_t)r->tl);
                memcpy(qb + qoff[k], r->q, (size_t)r->ql);
                toff[k + 1] = toff[k] + r->tl;
                qoff[k + 1] = qoff[k] + r->ql;
            }
            t_1 = std::chrono::steady_clock::now();
            rc = mgl_sw_align_batch_staged(ctx_, n, in_bytes, t_pad, max_tl, max_ql, std::get<0>(key), std::get<1>(key),
                                           std::get<2>(key), std::get<3>(key), std::get<4>(key), stride, out_bytes);
            t_2 = std::chrono::steady_clock::now();
        }
        const int32_t *off_ = static_cast<const int32_t *>(out), *len_ = off_ + n, *status_ = len_ + n;
        const mgl_sw_score *score_ = reinterpret_cast<const mgl_sw_score *>(status_ + n);
        const char *cig_ = reinterpret_cast<const char *>(score_ + n);
        for (int k = 0; k < n; ++k) {
            Request *r = batch[(size_t)k];
            if (rc != MGL_SW_OK) {
                r->rc = rc;
                continue;
            }
            *r->cigar_len = len_[(size_t)k];
            // the pair's own status goes back verbatim (a device-side failure must not read as "CIGAR does not fit");
            // the slot may be wider than this caller's buffer, so a text that fits the slot can still overflow the caller
            if (status_[(size_t)k] != 0) {
                r->rc = status_[(size_t)k];
                continue;
            }
            if (len_[(size_t)k] > r->cigar_cap) {
                r->rc = MGL_SW_ERR_CIGAR_OVERFLOW;
                continue;
            }
            memcpy(r->cigar, cig_ + (size_t)k * stride, (size_t)len_[(size_t)k]);
            *r->offset = off_[(size_t)k];
            if (r->ez) *r->ez = score_[(size_t)k];
            r->rc = MGL_SW_OK;
        }
        if (timing_) {
            auto us = [](std::chrono::steady_clock::duration d) { return std::chrono::duration<double, std::micro>(d).count(); };
            t_layout_ += us(t_1 - t_0);
            t_device_ += us(t_2 - t_1);
            t_scatter_ += us(std::chrono::steady_clock::now() - t_2);
        }
    }

    SubmissionQueue queue_;
    WakeWord wake_;
    // the dispatcher's own (no lock: nobody else touches them)
    std::map<Key, std::deque<Request *>> queues_;
    std::map<Key, Clock::time_point> oldest_;
    std::map<Key, int> expect_; // size of the previous batch of this key
    int pending_ = 0;
    Clock::time_point newest_{}, released_{};
    // parking the dispatcher
    std::mutex park_mu_;
    std::condition_variable park_cv_;
    bool poke_ = false; // guarded by park_mu_
    std::atomic<bool> parked_{false};
    std::atomic<bool> stop_{false};
    std::mutex cfg_mu_;
    std::atomic<int> max_batch_{0}, max_wait_us_{0};
    std::atomic<bool> enabled_{false};
    const int spin_slots_ = [] { const char *e = getenv("MGL_SW_COALESCE_SPIN_SLOTS"); return e ? atoi(e) : std::max(0, usable_cpus() - 2); }();
    const int spin_us_ = [] { const char *e = getenv("MGL_SW_COALESCE_SPIN_US"); return e ? atoi(e) : 200; }();
    std::thread worker_;
    mgl_sw_ctx *ctx_ = nullptr;
    std::atomic<int64_t> n_batches_{0}, n_pairs_{0};
    const bool timing_ = getenv("MGL_SW_DEBUG_COALESCE_TIMING") != nullptr;
    double t_wait_ = 0, t_process_ = 0, t_release_ = 0, t_layout_ = 0, t_device_ = 0, t_scatter_ = 0, t_idle_ = 0, t_spread_ = 0;
};

struct EnvInit {
    EnvInit()
    {
        // on by default (50 us window): one pair per call from many threads is the reference's calling pattern, and an
        // uncoalesced call is a full device round trip per pair; MGL_SW_COALESCE_US=-1 switches it off
        int us = 50, mb = 4096;
        if (const char *e = getenv("MGL_SW_COALESCE_US")) us = atoi(e);
        if (const char *b = getenv("MGL_SW_COALESCE_BATCH")) mb = atoi(b);
        if (us >= 0 && mb > 0) Coalescer::instance().configure(mb, us);
    }
};

} // namespace

// library-internal entry points used by mgl_sw_align (sw_capi.cpp); C linkage, not in include/mgl_sw.h
extern "C" MGL_SW_INTERNAL bool mgl_sw_coalescing_enabled()
{
    static EnvInit once;
    return Coalescer::instance().enabled();
}

extern "C" MGL_SW_INTERNAL int mgl_sw_coalesced_align(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen, int gext,
                           int strategy, char *cigar, int cigar_cap, int *cigar_len, int *offset, mgl_sw_score *ez)
{
    mgl_sw_normalize_params(&match, &mismatch, &gopen, &gext);
    Request r;
    r.t = t;
    r.q = q;
    r.tl = tl;
    r.ql = ql;
    r.cigar = cigar;
    r.cigar_cap = cigar_cap;
    r.cigar_len = cigar_len;
    r.offset = offset;
    r.ez = ez;
    r.key = Key(match, mismatch, gopen, gext, strategy);
    return Coalescer::instance().submit(r);
}

extern "C" {

int mgl_sw_set_coalescing(int max_batch, int max_wait_us)
{
    if (max_batch < 0 || max_wait_us < 0 || max_batch > (1 << 20)) return MGL_SW_ERR_BAD_ARG;
    Coalescer::instance().configure(max_batch, max_batch == 0 ? -1 : max_wait_us);
    return MGL_SW_OK;
}

int mgl_sw_coalescing_stats(int64_t *batches, int64_t *pairs)
{
    if (!batches || !pairs) return MGL_SW_ERR_BAD_ARG;
    Coalescer::instance().stats(batches, pairs);
    return MGL_SW_OK;
}

} // extern "C"
