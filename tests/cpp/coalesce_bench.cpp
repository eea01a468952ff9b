// coalesce_bench.cpp -- the calling pattern of GATK's alignNative (one pair per call, many threads;
// /root/reference/src/main/java/com/microsoft/mgl/smithwaterman/MicrosoftSmithWaterman.java:66-86) from native
// threads, against mgl_sw_align: through the mailbox service (mgl_amd/csrc/sw_service.cpp; MGL_SW_SERVICE_SLOTS=0 switches it off),
// through the coalescing front-end (sw_batcher.cpp), or as direct calls.
//   coalesce_bench <threads> <calls per thread> <coalesce wait us | -1 = direct> [tl] [ql]
// Every thread checks its results against a direct call made once at start-up (same pair set for all threads).
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mgl_sw.h"

int main(int argc, char **argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 64, calls = argc > 2 ? atoi(argv[2]) : 200;
    const int wait_us = argc > 3 ? atoi(argv[3]) : 50, tl = argc > 4 ? atoi(argv[4]) : 256, ql = argc > 5 ? atoi(argv[5]) : 150;
    const int distinct = 64;
    std::mt19937 rng(7);
    std::vector<std::string> ts(distinct), qs(distinct), want(distinct);
    std::vector<int> want_off(distinct);
    for (int k = 0; k < distinct; ++k) {
        ts[k].resize(tl);
        for (auto &c : ts[k]) c = "ACGT"[rng() & 3];
        const int s = (int)(rng() % (unsigned)(tl - ql + 1));
        qs[k] = ts[k].substr(s, ql);
        for (int e = 0; e < ql / 50 + 1; ++e) qs[k][rng() % ql] = "ACGT"[rng() & 3];
        char cigar[1024];
        int len = 0;
        if (mgl_sw_align(ts[k].data(), tl, qs[k].data(), ql, 200, -150, 260, 11, MGL_SW_OS_SOFTCLIP, cigar, sizeof cigar, &len,
                         &want_off[k], nullptr) != MGL_SW_OK) {
            fprintf(stderr, "direct call failed\n");
            return 2;
        }
        want[k].assign(cigar, (size_t)len);
    }
    mgl_sw_set_coalescing(wait_us >= 0 ? 4096 : 0, wait_us >= 0 ? wait_us : 0);
    std::atomic<int> bad{0};
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < threads; ++t)
        pool.emplace_back([&, t] {
            char cigar[1024];
            for (int c = 0; c < calls; ++c) {
                const int k = (t * 31 + c) % distinct;
                int len = 0, off = 0;
                const int rc = mgl_sw_align(ts[k].data(), tl, qs[k].data(), ql, 200, -150, 260, 11, MGL_SW_OS_SOFTCLIP, cigar,
                                            sizeof cigar, &len, &off, nullptr);
                if (rc != MGL_SW_OK || off != want_off[k] || want[k].compare(0, std::string::npos, cigar, (size_t)len) != 0) ++bad;
            }
        });
    for (auto &th : pool) th.join();
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    int64_t batches = 0, pairs = 0, mailbox_calls = 0, launches = 0;
    mgl_sw_coalescing_stats(&batches, &pairs);
    mgl_sw_service_stats(&mailbox_calls, &launches);
    mgl_sw_set_coalescing(0, 0);
    printf("%d threads x %d calls, %dx%d, %s: %.0f pairs/s (%.3f s), %lld through mailboxes (%lld launches), %lld coalesced (mean batch %.1f), wrong results %d\n",
           threads, calls, tl, ql, wait_us >= 0 ? "front-end" : "direct", (double)threads * calls / dt, dt, (long long)mailbox_calls, (long long)launches,
           (long long)pairs, batches ? (double)pairs / batches : 1.0, bad.load());
    return bad ? 1 : 0;
}
