// fake_device.cpp -- stands where the HIP kernels stand when the library's HOST side is built against the fake runtime
// (tests/cpp/fake_hip/) for the CPU sanitizer runs.  TEST CODE, never shipped.  The fill launches do nothing; whatever
// would walk the paths (the traceback launch, or the lane kernel's fused walk) computes each pair with the CPU checker
// (oracle/sw_oracle.c) and writes the results exactly where the kernels write them -- slot, stride, dest map, per-pair
// status -- so the driver can tell whether the host layer chunked, sorted, sharded and scattered correctly.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../mgl_amd/csrc/sw_device.h"
#include "../../oracle/sw_oracle.h"

int fake_hip_device_count(void)
{
    static const int n = [] { const char *e = getenv("FAKE_HIP_DEVICES"); return e ? atoi(e) : 2; }();
    return n;
}

int *fake_hip_current_device(void)
{
    static thread_local int d = 0;
    return &d;
}

namespace mgl_sw_dev {

std::atomic<long long> fake_fill_launches{0}, fake_walk_pairs{0}, fake_packed_pairs{0};
static thread_local int g_match, g_mismatch, g_gopen, g_gext; // parameters of the fill launch(es) of the chunk being walked

static void remember(const DpArgs &a)
{
    g_match = a.match;
    g_mismatch = a.mismatch;
    g_gopen = a.gopen;
    g_gext = a.gext;
    ++fake_fill_launches;
}

static void walk(const TbArgs &a, bool scores_only)
{
    for (int64_t slot = 0; slot < a.count; ++slot) {
        const int64_t p = a.first + slot;
        const int tl = a.t.length(p), ql = a.q.length(p);
        const uint8_t *t = a.t.data + a.t.off[p], *q = a.q.data + a.q.off[p];
        std::vector<uint8_t> tb, qb; // 2-bit packed sets: unpacked to letters for the checker (only equality matters)
        if (a.t.packed2) {
            tb.resize((size_t)tl);
            for (int k = 0; k < tl; ++k) tb[(size_t)k] = (uint8_t)"ACGT"[a.t.at(a.t.off[p], k)];
            t = tb.data();
        }
        if (a.q.packed2) {
            qb.resize((size_t)ql);
            for (int k = 0; k < ql; ++k) qb[(size_t)k] = (uint8_t)"ACGT"[a.q.at(a.q.off[p], k)];
            q = qb.data();
        }
        // the checker is compiled without the thread sanitizer in the TSan build (it is the stand-in for the kernels, not the code under
        // test, and instrumented it took six minutes): read the inputs here, in instrumented code, so that a race between the
        // library's host side and the "device" reading a sequence is still seen
        {
            unsigned sum = 0;
            for (int k = 0; k < tl; ++k) sum += t[k];
            for (int k = 0; k < ql; ++k) sum += q[k];
            asm volatile("" : : "r"(sum)); // (keeps the loops)
        }
        const int64_t o = a.dest ? a.dest[p] : p;
        std::vector<char> text((size_t)(tl + ql + 4) * 12);
        int len = 0, off = 0;
        swo_score ez;
        const int rc = swo_align(t, tl, q, ql, g_match, g_mismatch, g_gopen, g_gext, a.strategy, text.data(), (int)text.size(), &len, &off, &ez, nullptr);
        if (rc != SWO_OK) abort();
        char *slot_out = a.cigar + (size_t)o * a.cigar_stride;
        int status = 0;
        if (ql >= 8 && memcmp(q, "NNNNNNNN", 8) == 0) { // fault injection: this pair's walk "fails on the device" (what the
            memset(slot_out, 0, (size_t)a.cigar_stride);  // traceback kernel reports when a fill kernel gave a pair up)
            status = ERR_DEVICE;
            len = 0;
        } else if (scores_only) {
            off = 0;
            len = 0;
        } else if (a.binary_cigar) {
            abort(); // not exercised by the host-sanitizer driver
        } else if (len > a.cigar_stride) {
            memset(slot_out, 0, (size_t)a.cigar_stride);
            status = ERR_CIGAR_OVERFLOW;
        } else {
            memcpy(slot_out, text.data(), (size_t)len);
            memset(slot_out + len, 0, (size_t)(a.cigar_stride - len));
        }
        a.offset[o] = status ? 0 : off;
        if (a.cigar_len) a.cigar_len[o] = len;
        if (a.status) a.status[o] = status;
        if (a.status_any && status) *a.status_any = *a.status_any > status ? *a.status_any : status;
        if (a.score) {
            Score sc{ez.mqe, ez.mqe_t, ez.max, ez.max_t, ez.max_q, ez.seg_length};
            a.score[o] = sc;
        }
        ++fake_walk_pairs;
        if (a.packed16) ++fake_packed_pairs;
    }
}

int64_t dp_group_bytes(int sps_cap, int rows) { return (int64_t)dp_ring_entries(sps_cap, rows) * 8 + 4ll * dp_qcopy_bytes(sps_cap, rows); }
int dp_lds_bytes(int sps_cap, int waves_per_block, int rows)
{
    const int64_t b = waves_per_block * (64 / rows) * dp_group_bytes(sps_cap, rows);
    return b > (1 << 30) ? (1 << 30) : (int)b;
}
int dp16_lds_bytes(int sps, int waves_per_block) { return waves_per_block * 4 * ((sps + 24) * 8 + (sps + 48) * 4); }
bool dp16_range_ok(int tl, int ql, int match, int mismatch, int gopen, int gext, int)
{
    if (match <= 0 || gopen < gext) return false;
    const int64_t top = (int64_t)match * (tl < ql ? tl : ql) + (int64_t)gext * ((int64_t)tl + ql);
    const int64_t low = -3 * (int64_t)gopen - ((int64_t)match - mismatch) - 2 * (int64_t)gext - 64;
    return 32767 - top + low >= -32768 && (int64_t)match - mismatch <= 30000 && gopen <= 10000 && gext <= 5000;
}
int coop_lds_bytes(int sps_cap, int waves_per_block) { return coop_query_bytes(sps_cap) + waves_per_block * 2048 + 1024; }
bool lane16_supported(const SeqSet &t, const SeqSet &q) { return !t.packed2 && !q.packed2; }
bool lane16_ck_supported(const SeqSet &t, const SeqSet &q) { return (t.packed2 != 0) == (q.packed2 != 0); }

hipError_t launch_dp16(const DpArgs &a, int, hipStream_t) { remember(a); return hipSuccess; }
hipError_t launch_dp(const DpArgs &a, int, int, hipStream_t) { remember(a); return hipSuccess; }
hipError_t launch_dp_coop(const DpArgs &a, int, hipStream_t) { remember(a); return hipSuccess; }
bool coop16_possible(int match, int, int gopen, int gext) { return match > 0 && match <= 1000 && gopen >= gext && gopen <= 1000; }
bool coop16_worthwhile(int match, int mismatch, int gopen, int gext) { return coop16_possible(match, mismatch, gopen, gext) && match <= 300; }
int coop16_lds_bytes(int sps_cap, int waves_per_block) { return coop_lds_bytes(sps_cap, waves_per_block) + coop_query_bytes(sps_cap); }
hipError_t launch_dp_coop16(const DpArgs &a, int, hipStream_t) { remember(a); return hipSuccess; }
int strip16_lds_bytes(int max_ql, int waves) { return strip16_qwords(max_ql) * 4 + waves * 64 + 128; }
int strip16_lds_bytes_codes(int max_ql, int waves) { return strip16_table_words(max_ql) * 4 + waves * 64 + 128; }
int strip16_waves_per_simd(int rows) { return rows >= 27 ? 2 : 3; }
bool strip16_range_ok(int match, int, int gopen, int gext) { return match > 0 && match <= 300 && gopen >= gext && gopen <= 400; }
hipError_t launch_dp16_strip(const DpArgs &a, int, int, hipStream_t) { remember(a); return hipSuccess; }
hipError_t launch_dp16_lane(const DpArgs &a, const TbArgs &w, int, hipStream_t)
{
    remember(a);
    if (w.cigar) walk(w, false);
    return hipSuccess;
}
// sw_dp16_lane_matrix.hip (MGL_SW_FLAG_SHARED_TARGET): the fake device has no substitution-matrix arithmetic -- the host layer is told that the
// kernel's byte table cannot hold the parameters and takes its other path
bool lane16_matrix_params_ok(int, int, int, int) { return false; }
int lane16_matrix_lds_bytes(int) { return 0; }
hipError_t launch_dp16_lane_matrix(const DpArgs &, const TbArgs &, hipStream_t) { return hipErrorInvalidValue; }
hipError_t launch_tile_geometry(const SeqSet &, const SeqSet &, int64_t, int64_t, int32_t *, hipStream_t) { return hipErrorInvalidValue; }
// ---- the persistent grid of sw_dp16_lane_ck.hip, protocol for protocol.  Device resident (no gate): done before the launch returns.
// With a gate (the DIRECT form of the host entries: the grid is launched FIRST and its inputs arrive beside it) the grid is a THREAD
// that goes through the waves' own loop -- wait at the gate for the tile's pairs and the next tile's, leave on a negative gate, give
// up when the word stands still for gate_timeout_ticks, draw the next tile, count itself out, the last wave out zeroes the counter --
// and is joined where the real stream would be synchronised (fake_hip_join_stream, called by the fake hipStreamSynchronize).
std::atomic<long long> fake_gated_grids{0}, fake_gate_leavers{0}, fake_gate_give_ups{0};
namespace {
std::mutex g_async_mu;
std::vector<std::pair<hipStream_t, std::thread>> g_async;

void lane_ck_wave(const DpArgs &a, const TbArgs &w, int64_t tiles, int64_t slots, int64_t slot)
{
    using Clock = std::chrono::steady_clock;
    int64_t arrived = a.gate ? 0 : INT64_MAX;
    unsigned long long *const ctr = reinterpret_cast<unsigned long long *>(a.tile_ctr);
    for (int64_t tile = slot; tile < tiles;) {
        const int64_t need = a.first + std::min(a.count, (tile + 2) * 128);
        if (arrived < need) {
            auto since = Clock::now();
            int64_t last = arrived;
            bool gave_up = false, left_early = false;
            for (;;) {
                arrived = __atomic_load_n(a.gate, __ATOMIC_ACQUIRE);
                if (arrived >= need) break;
                if (arrived < 0) {
                    left_early = true;
                    break;
                }
                const auto now = Clock::now();
                if (arrived != last) {
                    last = arrived;
                    since = now;
                } else if (std::chrono::duration_cast<std::chrono::nanoseconds>(now - since).count() > (long long)a.gate_timeout_ticks * 10) {
                    gave_up = true;
                    break;
                }
                std::this_thread::yield();
            }
            if (gave_up) {
                __atomic_store_n(a.gate_failed, 1, __ATOMIC_RELEASE);
                ++fake_gate_give_ups;
                break;
            }
            if (left_early) {
                ++fake_gate_leavers;
                break;
            }
        }
        if (w.cigar) {
            TbArgs part = w;
            part.first = w.first + tile * 128;
            part.count = std::min<int64_t>(128, w.count - tile * 128);
            walk(part, false);
        }
        if (tiles <= slots) break;
        const unsigned next = (unsigned)__atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED); // (one 64-bit object {draws, waves out}, relaxed: as the kernel)
        if (next >= (unsigned)tiles && a.grid_fault) __atomic_store_n(a.grid_fault, 1, __ATOMIC_RELEASE);
        tile = slots + (int64_t)next;
    }
    if (tiles > slots) {
        const unsigned out = (unsigned)(__atomic_fetch_add(ctr, 1ull << 32, __ATOMIC_RELAXED) >> 32);
        if (out == (unsigned)slots - 1u) {
            __atomic_store_n(ctr, 0ull, __ATOMIC_RELAXED);
        } else if (out >= (unsigned)slots && a.grid_fault) {
            __atomic_store_n(a.grid_fault, 1, __ATOMIC_RELEASE);
        }
    }
}
} // namespace
} // namespace mgl_sw_dev

void fake_hip_join_stream(hipStream_t s)
{
    std::vector<std::thread> mine;
    {
        std::lock_guard<std::mutex> lk(mgl_sw_dev::g_async_mu);
        for (auto it = mgl_sw_dev::g_async.begin(); it != mgl_sw_dev::g_async.end();)
            if (!s || it->first == s) {
                mine.push_back(std::move(it->second));
                it = mgl_sw_dev::g_async.erase(it);
            } else {
                ++it;
            }
    }
    for (auto &t : mine) t.join();
}

// fault injection for the host-sanitizer driver: the k-th asynchronous copy from now fails (0: none does)
static std::atomic<int> g_copy_fail_in{0};
void fake_hip_fail_copy_in(int k) { g_copy_fail_in = k; }
int fake_hip_copy_should_fail(void)
{
    int v = g_copy_fail_in.load();
    while (v > 0 && !g_copy_fail_in.compare_exchange_weak(v, v - 1)) {
    }
    return v == 1;
}

namespace mgl_sw_dev {
hipError_t launch_dp16_lane_ck(const DpArgs &a, const TbArgs &w, hipStream_t stream)
{
    // the persistent grid's contract (sw_dp16_lane_ck.hip): wave slots, and wherever the tiles outnumber them a counter entry that
    // stands at {0, 0} -- the last wave of the launch before it on that entry has seen to that, HOWEVER that launch ended
    const int64_t tiles = ((a.count + 1) / 2 + 63) / 64;
    if (a.lane_slots < 1 || (tiles > a.lane_slots && (!a.tile_ctr || __atomic_load_n(reinterpret_cast<unsigned long long *>(a.tile_ctr), __ATOMIC_ACQUIRE) != 0)))
        return hipErrorInvalidValue;
    const int64_t slots = std::min<int64_t>(tiles, a.lane_slots);
    remember(a);
    if (!a.gate) {
        for (int64_t s = 0; s < slots; ++s) lane_ck_wave(a, w, tiles, slots, s);
        return hipSuccess;
    }
    ++fake_gated_grids;
    std::lock_guard<std::mutex> lk(g_async_mu);
    g_async.emplace_back(stream, std::thread([a, w, tiles, slots, m = g_match, x = g_mismatch, o = g_gopen, e = g_gext] {
        g_match = m, g_mismatch = x, g_gopen = o, g_gext = e;
        // (the waves one after the other: wave 0 draws most of the tiles -- any order of draws is one the real grid may produce)
        for (int64_t s = 0; s < slots; ++s) lane_ck_wave(a, w, tiles, slots, s);
    }));
    return hipSuccess;
}
hipError_t launch_traceback(const TbArgs &a, hipStream_t) { walk(a, false); return hipSuccess; }
hipError_t launch_strip_ck_walk(const TbArgs &a, int, int, hipStream_t) { walk(a, false); return hipSuccess; }
bool small_supported(int max_tl, int max_ql, int, int match, int mismatch, int, int gext, bool *wide)
{
    if (wide) *wide = false;
    return max_tl <= 512 && (int64_t)max_tl * max_ql <= 60000 && small_mul24_ok(match, mismatch, gext);
}
int small_lds_bytes(int tl, int ql, int stride, bool wide) { return tl * ql * (wide ? 4 : 2) + tl + ql + stride; } // (about what sw_small.hip's carve takes)
hipError_t launch_small(const TbArgs &a, int, int, bool, hipStream_t)
{
    g_match = a.match;
    g_mismatch = a.mismatch;
    g_gopen = a.gopen;
    g_gext = a.gext;
    ++fake_fill_launches;
    walk(a, false);
    return hipSuccess;
}
bool small_fits_int16(int, int, int, int, int, int) { return true; }
// the resident wave of sw_service.hip as a detached thread: the same protocol on the same mailbox (what the sanitizers watch is
// the host side of it), the checker where small_pair() stands
std::atomic<long long> fake_service_waves{0}, fake_service_pairs{0};
static std::atomic<unsigned long long> g_service_last{0};
static std::atomic<uint32_t> g_service_stop{0};
std::atomic<int> fake_service_lds_bytes{0}; // the dynamic LDS of the last grid launched
hipError_t launch_service(const ServiceRequest *requests, ServiceReply *replies, ServiceControl *, int slots, uint32_t gen, uint32_t idle_ticks, uint32_t life_ticks,
                          int lds_bytes, hipStream_t)
{
    if (slots < 1 || lds_bytes < 1 || lds_bytes > SERVICE_LDS_BYTES) return hipErrorInvalidValue;
    fake_service_lds_bytes = lds_bytes;
    static std::mutex order;              // "the stream": a grid starts when the one before it has ended
    struct Joiner {                       // (the last grid's threads end by their own conditions: joined when the process ends)
        std::vector<std::thread> v;
        ~Joiner()
        {
            for (auto &t : v) t.join();
        }
    };
    static Joiner joiner;
    std::vector<std::thread> &prev = joiner.v;
    std::lock_guard<std::mutex> lk(order);
    for (auto &t : prev) t.join();
    prev.clear();
    g_service_stop.store((gen - 1u) & SERVICE_GEN_MASK); // (launch_service's memset in front of the grid)
    using Clock = std::chrono::steady_clock;
    static const auto epoch = Clock::now();
    for (int k = 0; k < slots; ++k) {
        ++fake_service_waves;
        prev.emplace_back([mb = requests + k, rp = replies + k, gen, idle_ticks, life_ticks, lds_bytes] {
            auto ticks = [] { return (unsigned long long)std::chrono::duration_cast<std::chrono::nanoseconds>(Clock::now() - epoch).count() / 10; };
            uint32_t served = __atomic_load_n(&rp->done_seq, __ATOMIC_ACQUIRE);
            __atomic_store_n(&rp->state, gen << 4 | (uint32_t)SERVICE_RUNNING, __ATOMIC_RELEASE);
            const unsigned long long t_start = ticks();
            for (;;) {
                const uint32_t seq_a = __atomic_load_n(&mb->seq_a, __ATOMIC_ACQUIRE), seq_b = __atomic_load_n(&mb->seq_b, __ATOMIC_ACQUIRE);
                const uint32_t quit_gen = __atomic_load_n(&mb->quit_gen, __ATOMIC_ACQUIRE);
                if (seq_a == seq_b && seq_a != served && mb->lds_need > lds_bytes) { // the grid's carve is too small for this pair: the grid ends
                    uint32_t seen = g_service_stop.load();
                    while (seen < gen && !g_service_stop.compare_exchange_weak(seen, gen)) {
                    }
                    break;
                }
                if (seq_a == seq_b && seq_a != served) {
                    std::vector<char> text((size_t)(mb->tl + mb->ql + 4) * 12);
                    int len = 0, off = 0;
                    swo_score ez;
                    if (swo_align(mb->t, mb->tl, mb->q, mb->ql, mb->match, mb->mismatch, mb->gopen, mb->gext, mb->strategy, text.data(), (int)text.size(), &len, &off,
                                  &ez, nullptr) != SWO_OK)
                        abort();
                    const bool injected = mb->ql >= 8 && memcmp(mb->q, "NNNNNNNN", 8) == 0; // (the same fault injection as walk())
                    rp->status = injected ? ERR_DEVICE : len > mb->cigar_stride ? ERR_CIGAR_OVERFLOW : 0;
                    rp->cigar_len = injected ? 0 : len;
                    rp->offset = rp->status ? 0 : off;
                    if (!rp->status) memcpy(rp->cigar, text.data(), (size_t)len);
                    rp->score = Score{ez.mqe, ez.mqe_t, ez.max, ez.max_t, ez.max_q, ez.seg_length};
                    ++fake_service_pairs;
                    __atomic_store_n(&rp->done_seq, seq_a, __ATOMIC_RELEASE);
                    served = seq_a;
                    unsigned long long now = ticks(), seen = g_service_last.load();
                    while (seen < now && !g_service_last.compare_exchange_weak(seen, now)) {
                    }
                    continue;
                }
                const unsigned long long now = ticks(), last = std::max(g_service_last.load(), t_start);
                if (service_gen_reached(quit_gen, gen) || service_gen_reached(g_service_stop.load(), gen)) break;
                if ((long long)(now - last) > (long long)idle_ticks || now - t_start > life_ticks) {
                    uint32_t seen = g_service_stop.load();
                    while (seen < gen && !g_service_stop.compare_exchange_weak(seen, gen)) {
                    }
                    break;
                }
                std::this_thread::sleep_for(std::chrono::microseconds(20)); // (32 "waves" and 48 callers share a handful of CPUs here)
            }
            __atomic_store_n(&rp->state, gen << 4 | (uint32_t)SERVICE_EXITED, __ATOMIC_RELEASE);
        });
    }
    return hipSuccess;
}
hipError_t launch_regroup(const RegroupArgs &a, hipStream_t)
{
    // the device's counting sort, sequentially (fake device memory is host memory)
    const int cells = a.max_tl * a.max_ql;
    auto cell = [&](int64_t p) {
        const int tl = std::min(std::max(a.t.length(p), 1), a.max_tl), ql = std::min(std::max(a.q.length(p), 1), a.max_ql);
        return (tl - 1) * a.max_ql + (ql - 1);
    };
    for (int c = 0; c < cells; ++c) a.cnt[c] = 0;
    for (int64_t k = 0; k < a.count; ++k) ++a.cnt[cell(a.first + k)];
    int lane = 0, full = 0, rest = 0;
    for (int c = cells - 1; c >= 0; --c) { // (reverse grid order, as on the device: long queries first)
        a.nlane[c] = a.lane_blocks ? a.cnt[c] & ~127 : 0;
        a.lane_start[c] = lane;
        lane += a.nlane[c];
    }
    full = lane;
    for (int c = 0; c < cells; ++c) {
        a.nfull[c] = a.cnt[c] & ~7;
        a.full_start[c] = full;
        full += a.nfull[c] - a.nlane[c];
    }
    for (int c = 0; c < cells; ++c) {
        a.rest_start[c] = full + rest;
        rest += a.cnt[c] & 7;
        a.cnt[c] = 0;
    }
    a.total[0] = full;
    a.total[1] = lane;
    for (int64_t k = 0; k < a.count; ++k) {
        const int64_t p = a.first + k;
        const int c = cell(p), pos = a.cnt[c]++;
        const int64_t slot = pos < a.nlane[c]   ? (int64_t)a.lane_start[c] + pos // (the fake stores the start itself, not the device's offset from the end)
                             : pos < a.nfull[c] ? (int64_t)a.full_start[c] + (pos - a.nlane[c])
                                                : (int64_t)a.rest_start[c] + (pos - a.nfull[c]);
        a.t_start[slot] = a.t.off[p];
        a.q_start[slot] = a.q.off[p];
        a.dest[slot] = p;
        a.t_len[slot] = a.t.length(p);
        a.q_len[slot] = a.q.length(p);
    }
    return hipSuccess;
}
hipError_t launch_scores_only(const TbArgs &a, hipStream_t) { walk(a, true); return hipSuccess; }
hipError_t launch_iota64(int64_t *dst, int64_t n, int64_t step, hipStream_t)
{
    for (int64_t k = 0; k < n; ++k) dst[k] = k * step;
    return hipSuccess;
}
hipError_t launch_cigar_from_matrix(const int32_t *, int, int, int, const Score &, char *, int, int32_t *, hipStream_t) { return hipErrorInvalidValue; }
hipError_t launch_expand(const uint32_t *, const DpRecord *, int, int, int, int, int, int32_t *, hipStream_t, int) { return hipErrorInvalidValue; }
hipError_t launch_band_fill(const int32_t *, const int32_t *, int, int32_t *, int, int, int, int32_t *, int32_t *, int32_t *, int, int, int, int, int,
                            int32_t *, hipStream_t) { return hipErrorInvalidValue; }

} // namespace mgl_sw_dev
