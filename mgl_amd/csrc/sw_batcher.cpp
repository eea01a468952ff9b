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

// sw_capi.cpp (library-internal, not part of include/mgl_sw.h: hidden visibility, not exported from the .so)
#define MGL_SW_INTERNAL __attribute__((visibility("hidden")))
extern "C" MGL_SW_INTERNAL int mgl_sw_stage_buffers(mgl_sw_ctx *ctx, size_t in_bytes, size_t out_bytes, void **in, void **out);
extern "C" MGL_SW_INTERNAL int mgl_sw_align_batch_staged(mgl_sw_ctx *ctx, int n, size_t in_bytes, size_t t_bytes_padded, int max_tl, int max_ql,
                                         int match, int mismatch, int gopen, int gext, int strategy, int cigar_stride,
                                         size_t out_bytes);

namespace {

// Callers sleep on one of a few condition variables chosen by thread, each with its own mutex: a finished batch then
// wakes its callers shard by shard instead of stampeding every parked thread through the submission lock.
constexpr int kShards = 32;
struct Shard {
    std::mutex mu;
    std::condition_variable cv;
};

struct Request {
    const char *t, *q;
    int tl, ql;
    char *cigar;
    int cigar_cap;
    int *cigar_len, *offset;
    mgl_sw_score *ez;
    int rc = MGL_SW_OK;
    std::atomic<bool> done{false}; // set under the shard's mutex (parked callers), read without it by callers that still spin
    int shard = 0;
};

inline void cpu_relax()
{
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#else
    std::this_thread::yield();
#endif
}

using Key = std::tuple<int, int, int, int, int>; // normalised match, mismatch, open, extend, strategy

class Coalescer {
  public:
    static Coalescer &instance()
    {
        static Coalescer c;
        return c;
    }

    void configure(int max_batch, int max_wait_us)
    {
        std::lock_guard<std::mutex> lk(mu_);
        max_batch_ = max_batch;
        max_wait_us_ = max_wait_us;
        enabled_.store(max_batch > 0 && max_wait_us >= 0);
        if (enabled_ && !worker_.joinable()) worker_ = std::thread([this] { run(); });
        cv_work_.notify_all();
    }
    bool enabled() const { return enabled_.load(std::memory_order_relaxed); }

    int submit(Request &r, const Key &key)
    {
        static std::atomic<unsigned> next_shard{0};
        thread_local const int my_shard = (int)(next_shard.fetch_add(1) % kShards);
        r.shard = my_shard;
        {
            std::lock_guard<std::mutex> lk(mu_);
            auto &qd = queues_[key];
            if (qd.empty()) oldest_[key] = std::chrono::steady_clock::now();
            qd.push_back(&r);
            const auto ex = expect_.find(key);
            const int expect = ex == expect_.end() ? 1 : ex->second;
            // wake the dispatcher when it is idle, when a batch is full, or when the callers it expects are all back
            const int pend = pending_.fetch_add(1, std::memory_order_release) + 1;
            if (parked_ && (pend == 1 || (int)qd.size() >= max_batch_ || (int)qd.size() == expect)) cv_work_.notify_one();
        }
        // the answer is usually back within a device round trip: spin that long before parking
        if (spin_us_ > 0) {
            const auto until = std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us_);
            do {
                for (int k = 0; k < 64; ++k) {
                    if (r.done.load(std::memory_order_acquire)) return r.rc;
                    cpu_relax();
                }
            } while (std::chrono::steady_clock::now() < until);
        }
        Shard &sh = shards_[my_shard];
        std::unique_lock<std::mutex> lk(sh.mu);
        sh.cv.wait(lk, [&] { return r.done.load(std::memory_order_acquire); });
        return r.rc;
    }

    void stats(int64_t *batches, int64_t *pairs)
    {
        std::lock_guard<std::mutex> lk(mu_);
        *batches = n_batches_;
        *pairs = n_pairs_;
    }

    ~Coalescer()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
            cv_work_.notify_all();
        }
        if (worker_.joinable()) worker_.join();
        if (timing_ && n_batches_) {
            const double nb = (double)n_batches_;
            fprintf(stderr, "[mgl_sw] coalescer: %lld batches, mean %.1f pairs; per batch: %.1f us collecting, %.1f us processing "
                            "(%.1f layout, %.1f device round trip, %.1f hand-out), %.1f us waking the callers\n",
                    (long long)n_batches_, (double)n_pairs_ / nb, t_wait_ / nb, t_process_ / nb, t_layout_ / nb, t_device_ / nb,
                    t_scatter_ / nb, t_release_ / nb);
        }
        if (ctx_) mgl_sw_ctx_destroy(ctx_);
    }

  private:
    void run()
    {
        std::unique_lock<std::mutex> lk(mu_);
        for (;;) {
            wait_for_work(lk, [&] { return stop_ || pending_.load(std::memory_order_acquire) > 0; }, std::chrono::steady_clock::time_point::max());
            if (stop_) {
                fail_all(MGL_SW_ERR_DEVICE);
                return;
            }
            // pick the queue that is full, or whose oldest request has waited long enough; otherwise sleep until
            // the earliest deadline (new arrivals wake us up too)
            const auto now = std::chrono::steady_clock::now();
            const int max_batch = std::max(1, max_batch_); // coalescing switched off with requests still queued: drain them
            const Key *ready = nullptr;
            auto earliest = now + std::chrono::hours(1);
            for (auto &kv : queues_) {
                if (kv.second.empty()) continue;
                const auto deadline = oldest_[kv.first] + std::chrono::microseconds(max_wait_us_);
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
                // more callers are expected (or the window is still open): until one arrives or the earliest deadline passes
                const int seen = pending_.load(std::memory_order_acquire);
                wait_for_work(lk, [&] { return stop_ || pending_.load(std::memory_order_acquire) != seen; }, earliest);
                continue;
            }
            const Key key = *ready;
            auto &qd = queues_[key];
            std::vector<Request *> batch;
            while (!qd.empty() && (int)batch.size() < max_batch) {
                batch.push_back(qd.front());
                qd.pop_front();
            }
            if (!qd.empty()) oldest_[key] = now; // the rest starts a new waiting period
            pending_.fetch_sub((int)batch.size(), std::memory_order_release);
            expect_[key] = (int)batch.size();
            const auto t_first = oldest_[key];
            lk.unlock();
            const auto t_a = std::chrono::steady_clock::now();
            process(key, batch);
            const auto t_b = std::chrono::steady_clock::now();
            release(batch);
            const auto t_c = std::chrono::steady_clock::now();
            lk.lock();
            ++n_batches_;
            n_pairs_ += (int64_t)batch.size();
            if (timing_) { // diagnostic (MGL_SW_DEBUG_COALESCE_TIMING): where a batch's round trip goes, microseconds
                auto us = [](std::chrono::steady_clock::duration d) { return std::chrono::duration<double, std::micro>(d).count(); };
                t_wait_ += us(now - t_first);
                t_process_ += us(t_b - t_a);
                t_release_ += us(t_c - t_b);
            }
        }
    }

    // The dispatcher's wait: spin (lock released) while the traffic is hot, then park on the condition variable.  `ready` is
    // evaluated with the lock released while spinning: it may only read atomics.
    template <typename Pred>
    void wait_for_work(std::unique_lock<std::mutex> &lk, Pred ready, std::chrono::steady_clock::time_point deadline)
    {
        if (ready()) return;
        if (spin_us_ > 0) {
            const auto until = std::min(deadline, std::chrono::steady_clock::now() + std::chrono::microseconds(spin_us_));
            lk.unlock();
            bool ok = false;
            do {
                for (int k = 0; k < 64 && !ok; ++k) {
                    ok = ready();
                    if (!ok) cpu_relax();
                }
            } while (!ok && std::chrono::steady_clock::now() < until);
            lk.lock();
            if (ok || ready() || std::chrono::steady_clock::now() >= deadline) return;
        }
        parked_ = true;
        if (deadline == std::chrono::steady_clock::time_point::max())
            cv_work_.wait(lk, ready);
        else
            cv_work_.wait_until(lk, deadline, ready);
        parked_ = false;
    }

    // hand the results back: per shard, mark its requests done under the shard's mutex, then wake that shard
    // (a request lives on its caller's stack and is gone the moment that caller sees done: sort the batch by shard
    // first, then touch every request exactly once, under its shard's mutex)
    void release(const std::vector<Request *> &batch)
    {
        std::vector<Request *> by_shard[kShards];
        for (Request *r : batch) by_shard[r->shard].push_back(r);
        for (int s = 0; s < kShards; ++s) {
            if (by_shard[s].empty()) continue;
            {
                std::lock_guard<std::mutex> lk(shards_[s].mu);
                for (Request *r : by_shard[s]) r->done.store(true, std::memory_order_release); // (r is gone from here on)
            }
            shards_[s].cv.notify_all();
        }
    }

    void fail_all(int rc)
    {
        std::vector<Request *> all;
        for (auto &kv : queues_)
            for (Request *r : kv.second) {
                r->rc = rc;
                all.push_back(r);
            }
        queues_.clear();
        pending_.store(0, std::memory_order_release);
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

    std::mutex mu_;
    std::condition_variable cv_work_;
    Shard shards_[kShards];
    std::map<Key, std::deque<Request *>> queues_;
    std::map<Key, std::chrono::steady_clock::time_point> oldest_;
    std::map<Key, int> expect_; // size of the previous batch of this key
    std::atomic<int> pending_{0};
    bool parked_ = false; // the dispatcher sleeps on cv_work_ (guarded by mu_): only then a submission has to notify
    const int spin_us_ = [] { const char *e = getenv("MGL_SW_COALESCE_SPIN_US"); return e ? atoi(e) : 200; }();
    int max_batch_ = 0, max_wait_us_ = 0;
    std::atomic<bool> enabled_{false};
    std::atomic<bool> stop_{false};
    std::thread worker_;
    mgl_sw_ctx *ctx_ = nullptr;
    int64_t n_batches_ = 0, n_pairs_ = 0;
    const bool timing_ = getenv("MGL_SW_DEBUG_COALESCE_TIMING") != nullptr;
    double t_wait_ = 0, t_process_ = 0, t_release_ = 0, t_layout_ = 0, t_device_ = 0, t_scatter_ = 0;
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
    return Coalescer::instance().submit(r, Key(match, mismatch, gopen, gext, strategy));
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
