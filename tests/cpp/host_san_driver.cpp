// host_san_driver.cpp -- drives the HOST side of the library (built against the fake HIP runtime and the fake device
// backend of this directory) under AddressSanitizer + UBSan, and again under ThreadSanitizer.  TEST CODE.
//   * mgl_sw_align_batch_status on a batch of mixed geometries cut into several chunks: the per-chunk sort by geometry on
//     its helper thread, both workspace halves reused, results scattered back through the dest map, per-pair overflow status
//   * a uniform batch through the lane-kernel path (fused walk, one buffer) and through the packed path
//   * the same entry with every array REGISTERED (mgl_sw_register_host_buffer: results copied straight into the caller's arrays,
//     no helper jobs), and mgl_sw_align_batch_2bit (packed bases from host memory) with ascending and with shuffled starts
//   * mgl_sw_align_batch_multi over two (fake) devices: shard boundaries, one host thread per device
//   * 48 threads through mgl_sw_align (the coalescing front-end): parking, batching, wake-up by shard, the caller whose
//     buffer is too small (CIGAR_OVERFLOW) and the pair whose walk fails on the "device" (status handed back verbatim)
// Every answer is compared with the CPU checker called directly.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include "../../include/mgl_sw.h"
#include "../../oracle/sw_oracle.h"

int *fake_hip_current_device(void); // tests/cpp/fake_device.cpp: the calling thread's current HIP device
void fake_hip_fail_copy_in(int k);  // ... the k-th asynchronous copy from now fails (0: none)
namespace mgl_sw_dev {
extern std::atomic<long long> fake_fill_launches, fake_walk_pairs, fake_packed_pairs, fake_service_waves, fake_service_pairs, fake_gated_grids, fake_gate_leavers, fake_gate_give_ups;
extern std::atomic<int> fake_service_lds_bytes;
}

#define CHECK(x)                                                          \
    do {                                                                  \
        if (!(x)) {                                                       \
            std::fprintf(stderr, "host-san: %s:%d: %s\n", __FILE__, __LINE__, #x); \
            std::exit(1);                                                 \
        }                                                                 \
    } while (0)

struct Batch {
    std::vector<uint8_t> t, q;
    std::vector<int64_t> toff{0}, qoff{0};
    int64_t n() const { return (int64_t)toff.size() - 1; }
    void add(const std::string &a, const std::string &b)
    {
        t.insert(t.end(), a.begin(), a.end());
        q.insert(q.end(), b.begin(), b.end());
        toff.push_back((int64_t)t.size());
        qoff.push_back((int64_t)q.size());
    }
};

static std::string rnd(std::mt19937 &g, int n)
{
    std::string s((size_t)n, 'A');
    for (char &c : s) c = "ACGT"[g() & 3];
    return s;
}

struct Expect {
    std::vector<int32_t> off, len;
    std::vector<swo_score> sc;
    std::vector<std::string> cg;
};

static Expect expect(const Batch &b, int strategy)
{
    Expect e;
    for (int64_t k = 0; k < b.n(); ++k) {
        const int tl = (int)(b.toff[k + 1] - b.toff[k]), ql = (int)(b.qoff[k + 1] - b.qoff[k]);
        std::vector<char> buf((size_t)(tl + ql + 4) * 12);
        int len = 0, off = 0;
        swo_score ez;
        CHECK(swo_align(b.t.data() + b.toff[k], tl, b.q.data() + b.qoff[k], ql, 200, -150, 260, 11, strategy, buf.data(), (int)buf.size(), &len, &off, &ez, nullptr) == 0);
        e.off.push_back(off);
        e.len.push_back(len);
        e.sc.push_back(ez);
        e.cg.emplace_back(buf.data(), (size_t)len);
    }
    return e;
}

static void compare(const Batch &b, const Expect &e, const std::vector<int32_t> &off, const std::vector<mgl_sw_score> &sc, const std::vector<char> &cg,
                    int stride, const std::vector<int32_t> &len, const std::vector<int32_t> *status)
{
    for (int64_t k = 0; k < b.n(); ++k) {
        if (status && (*status)[(size_t)k] != 0) {
            CHECK((*status)[(size_t)k] == MGL_SW_ERR_CIGAR_OVERFLOW && e.len[(size_t)k] > stride && len[(size_t)k] == e.len[(size_t)k]);
            continue;
        }
        CHECK(off[(size_t)k] == e.off[(size_t)k] && len[(size_t)k] == e.len[(size_t)k]);
        CHECK(memcmp(&sc[(size_t)k], &e.sc[(size_t)k], sizeof(mgl_sw_score)) == 0);
        CHECK(std::string(cg.data() + (size_t)k * stride, (size_t)len[(size_t)k]) == e.cg[(size_t)k]);
        for (int x = len[(size_t)k]; x < stride; ++x) CHECK(cg[(size_t)k * stride + x] == 0);
    }
}

int main()
{
    std::mt19937 g(12345);
    // ---- a batch of mixed geometries: windows of two sizes, reads of 100 .. 150 bases, a sprinkling of odd ones
    Batch mixed;
    for (int k = 0; k < 6000; ++k) {
        const int tl = (g() & 1) ? 200 : 256, ql = (k % 97 == 0) ? 1 + (int)(g() % 40) : 100 + (int)(g() % 51);
        std::string t = rnd(g, tl), q = t.substr(g() % 40, (size_t)ql);
        q.resize((size_t)ql, 'C');
        q[g() % q.size()] = "ACGT"[g() & 3];
        mixed.add(t, q);
    }
    const Expect em = expect(mixed, MGL_SW_OS_SOFTCLIP);
    mgl_sw_ctx *ctx = nullptr;
    CHECK(mgl_sw_ctx_create(0, &ctx) == 0);
    CHECK(mgl_sw_ctx_set_small_kernel(ctx, 1) == 0); // (these batches of a few thousand pairs are here for the chunked paths, not for the one-wave-per-pair kernel)
    for (int stride : {128, 4}) {
        CHECK(mgl_sw_ctx_set_workspace(ctx, 96ll << 20) == 0); // ~2 200 pairs per half: three chunks, the halves are reused
        const int64_t n = mixed.n();
        std::vector<int32_t> off((size_t)n), len((size_t)n), st((size_t)n);
        std::vector<mgl_sw_score> sc((size_t)n);
        std::vector<char> cg((size_t)n * stride, 1);
        const long long packed0 = mgl_sw_dev::fake_packed_pairs.load();
        CHECK(mgl_sw_align_batch_status(ctx, n, mixed.t.data(), mixed.toff.data(), mixed.q.data(), mixed.qoff.data(), 200, -150, 260, 11,
                                        MGL_SW_OS_SOFTCLIP, off.data(), sc.data(), cg.data(), stride, len.data(), st.data()) == 0);
        compare(mixed, em, off, sc, cg, stride, len, &st);
        CHECK(mgl_sw_dev::fake_packed_pairs.load() - packed0 > n * 8 / 10); // most pairs reached the packed kernel's part
        mgl_sw_timing tm;
        CHECK(mgl_sw_ctx_get_timing(ctx, &tm) == 0 && tm.dp_launches >= 3 && tm.packed16 == 1);
        if (stride == 4) { // without the status array the same batch fails as a whole
            const int rc2 = mgl_sw_align_batch(ctx, n, mixed.t.data(), mixed.toff.data(), mixed.q.data(), mixed.qoff.data(), 200, -150, 260, 11,
                                               MGL_SW_OS_SOFTCLIP, off.data(), sc.data(), cg.data(), stride, len.data());
            if (rc2 != MGL_SW_ERR_CIGAR_OVERFLOW) std::fprintf(stderr, "rc2 = %d (%s)\n", rc2, mgl_sw_last_error(ctx));
            CHECK(rc2 == MGL_SW_ERR_CIGAR_OVERFLOW);
        }
    }
    // ---- the same mixed batch as a DEVICE-resident one (the fake device's memory is host memory): sorted by geometry by the
    // "device" (launch_regroup), two chunks ahead of the fills, four rotating sets of index arrays, six or more chunks
    {
        CHECK(mgl_sw_ctx_set_workspace(ctx, 48ll << 20) == 0);
        const int64_t n = mixed.n();
        const int stride = 128;
        std::vector<int32_t> off((size_t)n), len((size_t)n), st((size_t)n);
        std::vector<mgl_sw_score> sc((size_t)n);
        std::vector<char> cg((size_t)n * stride, 1);
        const long long packed0 = mgl_sw_dev::fake_packed_pairs.load();
        CHECK(mgl_sw_align_batch_device(ctx, nullptr, n, mixed.t.data(), mixed.toff.data(), mixed.q.data(), mixed.qoff.data(), 256, 150, 200, -150,
                                        260, 11, MGL_SW_OS_SOFTCLIP, off.data(), sc.data(), cg.data(), stride, len.data(), st.data(), 0) == 0);
        mgl_sw_timing tm;
        CHECK(mgl_sw_ctx_get_timing(ctx, &tm) == 0 && tm.dp_launches >= 5 && tm.packed16 == 1);
        compare(mixed, em, off, sc, cg, stride, len, &st);
        CHECK(mgl_sw_dev::fake_packed_pairs.load() - packed0 > n / 5); // (chunks of ~1 100 pairs over ~100 geometries: up to seven left-over pairs each)
    }
    // ---- a uniform batch: the lane-kernel path (forced; fused walk, one buffer, chunks of 128), then the packed path
    Batch uni;
    for (int k = 0; k < 3000; ++k) {
        std::string t = rnd(g, 256), q = t.substr(g() % 100, 150);
        q[g() % 150] = 'A';
        uni.add(t, q);
    }
    const Expect eu = expect(uni, MGL_SW_OS_INDEL);
    for (int lane_mode : {2, 3, 1}) { // 2: the lane kernel in its default (checkpointed) form, 3: with stored traceback, 1: never
        CHECK(mgl_sw_ctx_set_lane_kernel(ctx, lane_mode == 3 ? 2 : lane_mode) == 0 && mgl_sw_ctx_set_lane_checkpoint(ctx, lane_mode == 3 ? 1 : 0) == 0 &&
              mgl_sw_ctx_set_workspace(ctx, 24ll << 20) == 0);
        const int64_t n = uni.n();
        std::vector<int32_t> off((size_t)n), len((size_t)n);
        std::vector<mgl_sw_score> sc((size_t)n);
        std::vector<char> cg((size_t)n * 64, 1);
        CHECK(mgl_sw_align_batch(ctx, n, uni.t.data(), uni.toff.data(), uni.q.data(), uni.qoff.data(), 200, -150, 260, 11, MGL_SW_OS_INDEL, off.data(),
                                 sc.data(), cg.data(), 64, len.data()) == 0);
        compare(uni, eu, off, sc, cg, 64, len, nullptr);
        mgl_sw_timing tm;
        CHECK(mgl_sw_ctx_get_timing(ctx, &tm) == 0 && tm.fill_kernel == (lane_mode == 2 ? MGL_SW_KERNEL_LANE16_CK : lane_mode == 3 ? MGL_SW_KERNEL_LANE16 : MGL_SW_KERNEL_DP16));
        int32_t btr_probe[4];
        if (lane_mode == 2) CHECK(mgl_sw_ctx_expand_slot(ctx, 0, 1, 1, btr_probe) == MGL_SW_ERR_UNSUPPORTED); // no stored traceback to expand
    }
    // ---- the same uniform batch as ASCII bases in REGISTERED arrays: the direct form of mgl_sw_align_batch_status (round 5) -- one gated
    // launch, six chunks of bases brought in beside it, the offsets written on the "device" (launch_iota64), results in place
    {
        CHECK(mgl_sw_ctx_set_lane_kernel(ctx, 2) == 0 && mgl_sw_ctx_set_lane_checkpoint(ctx, 0) == 0 && mgl_sw_ctx_set_workspace(ctx, 1ll << 30) == 0);
        const int64_t n = uni.n();
        std::vector<int32_t> off((size_t)n), len((size_t)n), st((size_t)n);
        std::vector<mgl_sw_score> sc((size_t)n);
        std::vector<char> cg((size_t)n * 64, 1);
        void *regs[] = {uni.t.data(), uni.q.data(), off.data(), sc.data(), cg.data(), len.data(), st.data()};
        const size_t bytes[] = {uni.t.size(), uni.q.size(), off.size() * 4, sc.size() * sizeof(mgl_sw_score), cg.size(), len.size() * 4, st.size() * 4};
        for (int i = 0; i < 7; ++i) CHECK(mgl_sw_register_host_buffer(ctx, regs[i], bytes[i]) == 0);
        setenv("MGL_SW_DEBUG_DIRECT_CHUNK", "512", 1);
        const long long grids0 = mgl_sw_dev::fake_gated_grids.load();
        CHECK(mgl_sw_align_batch_status(ctx, n, uni.t.data(), uni.toff.data(), uni.q.data(), uni.qoff.data(), 200, -150, 260, 11, MGL_SW_OS_INDEL, off.data(), sc.data(),
                                        cg.data(), 64, len.data(), st.data()) == 0);
        CHECK(mgl_sw_dev::fake_gated_grids.load() == grids0 + 1);
        compare(uni, eu, off, sc, cg, 64, len, &st);
        // a copy that fails behind the launch: the grid is called off, the call fails, the next one works
        fake_hip_fail_copy_in(5);
        CHECK(mgl_sw_align_batch_status(ctx, n, uni.t.data(), uni.toff.data(), uni.q.data(), uni.qoff.data(), 200, -150, 260, 11, MGL_SW_OS_INDEL, off.data(), sc.data(),
                                        cg.data(), 64, len.data(), st.data()) == MGL_SW_ERR_DEVICE);
        fake_hip_fail_copy_in(0);
        std::fill(off.begin(), off.end(), -3);
        CHECK(mgl_sw_align_batch_status(ctx, n, uni.t.data(), uni.toff.data(), uni.q.data(), uni.qoff.data(), 200, -150, 260, 11, MGL_SW_OS_INDEL, off.data(), sc.data(),
                                        cg.data(), 64, len.data(), st.data()) == 0);
        compare(uni, eu, off, sc, cg, 64, len, &st);
        unsetenv("MGL_SW_DEBUG_DIRECT_CHUNK");
        for (int i = 0; i < 7; ++i) CHECK(mgl_sw_unregister_host_buffer(ctx, regs[i]) == 0);
    }
    // ---- a mixed batch with one dominant geometry: the sorted chunks' whole waves of 128 go through the (fake) lane kernel, the rest
    // through the packed and int32 parts -- both entries, several chunks (threshold lowered: MGL_SW_DEBUG_LANE_GROUP_MIN)
    {
        Batch most;
        for (int k = 0; k < 5200; ++k) { // (a device-resident batch is sorted only when n * 8 >= max_tl * max_ql)
            const bool odd = k % 9 == 0;
            std::string t = rnd(g, odd ? 200 + (int)(g() % 57) : 256), q = t.substr(g() % 40, odd ? 100 + g() % 51 : 150);
            q[g() % q.size()] = 'T';
            most.add(t, q);
        }
        const Expect eo = expect(most, MGL_SW_OS_SOFTCLIP);
        setenv("MGL_SW_DEBUG_LANE_GROUP_MIN", "128", 1);
        CHECK(mgl_sw_ctx_set_lane_kernel(ctx, 0) == 0 && mgl_sw_ctx_set_lane_checkpoint(ctx, 0) == 0 && mgl_sw_ctx_set_workspace(ctx, 64ll << 20) == 0);
        const int64_t n = most.n();
        for (int entry = 0; entry < 2; ++entry) {
            std::vector<int32_t> off((size_t)n), len((size_t)n), st((size_t)n);
            std::vector<mgl_sw_score> sc((size_t)n);
            std::vector<char> cg((size_t)n * 64, 1);
            if (entry == 0)
                CHECK(mgl_sw_align_batch_status(ctx, n, most.t.data(), most.toff.data(), most.q.data(), most.qoff.data(), 200, -150, 260, 11, MGL_SW_OS_SOFTCLIP,
                                                off.data(), sc.data(), cg.data(), 64, len.data(), st.data()) == 0);
            else
                CHECK(mgl_sw_align_batch_device(ctx, nullptr, n, most.t.data(), most.toff.data(), most.q.data(), most.qoff.data(), 256, 150, 200, -150, 260, 11,
                                                MGL_SW_OS_SOFTCLIP, off.data(), sc.data(), cg.data(), 64, len.data(), st.data(), 0) == 0);
            mgl_sw_timing tm;
            CHECK(mgl_sw_ctx_get_timing(ctx, &tm) == 0 && tm.dp_launches >= 2 && tm.fill_kernel == MGL_SW_KERNEL_LANE16_CK);
            compare(most, eo, off, sc, cg, 64, len, &st);
        }
        unsetenv("MGL_SW_DEBUG_LANE_GROUP_MIN");
    }
    // ---- registered arrays: the results of every chunk go straight into the caller's arrays (no pinned ring, no helper jobs)
    {
        CHECK(mgl_sw_ctx_set_lane_kernel(ctx, 0) == 0 && mgl_sw_ctx_set_workspace(ctx, 96ll << 20) == 0);
        const int64_t n = mixed.n();
        const int stride = 128;
        std::vector<int32_t> off((size_t)n), len((size_t)n), st((size_t)n);
        std::vector<mgl_sw_score> sc((size_t)n);
        std::vector<char> cg((size_t)n * stride, 1);
        void *regs[] = {mixed.t.data(), mixed.toff.data(), mixed.q.data(), mixed.qoff.data(), off.data(), sc.data(), cg.data(), len.data(), st.data()};
        const size_t bytes[] = {mixed.t.size(), mixed.toff.size() * 8, mixed.q.size(), mixed.qoff.size() * 8, off.size() * 4, sc.size() * sizeof(mgl_sw_score),
                                cg.size(), len.size() * 4, st.size() * 4};
        for (int i = 0; i < 9; ++i) CHECK(mgl_sw_register_host_buffer(ctx, regs[i], bytes[i]) == 0);
        CHECK(mgl_sw_register_host_buffer(ctx, regs[0], bytes[0]) == 0); // registering twice is harmless
        CHECK(mgl_sw_align_batch_status(ctx, n, mixed.t.data(), mixed.toff.data(), mixed.q.data(), mixed.qoff.data(), 200, -150, 260, 11,
                                        MGL_SW_OS_SOFTCLIP, off.data(), sc.data(), cg.data(), stride, len.data(), st.data()) == 0);
        compare(mixed, em, off, sc, cg, stride, len, &st);
        for (int i = 0; i < 9; ++i) CHECK(mgl_sw_unregister_host_buffer(ctx, regs[i]) == 0);
        CHECK(mgl_sw_unregister_host_buffer(ctx, regs[0]) == MGL_SW_ERR_BAD_ARG);
    }
    // ---- 2-bit packed bases from host memory: reads packed back to back (ascending starts: the packed arrays travel chunk by
    // chunk), then the same pairs in shuffled order (whole arrays first); mixed lengths and, on a uniform subset, no length arrays
    {
        auto pack = [](const std::vector<uint8_t> &bases) {
            std::vector<uint8_t> out((bases.size() + 3) / 4 + 8, 0);
            for (size_t k = 0; k < bases.size(); ++k) {
                const int c = bases[k] == 'A' ? 0 : bases[k] == 'C' ? 1 : bases[k] == 'G' ? 2 : 3;
                out[k >> 2] |= (uint8_t)(c << (2 * (k & 3)));
            }
            return out;
        };
        const std::vector<uint8_t> T = pack(mixed.t), Q = pack(mixed.q);
        const int64_t n = mixed.n();
        for (int order = 0; order < 2; ++order) {
            std::vector<int64_t> idx((size_t)n);
            for (int64_t k = 0; k < n; ++k) idx[(size_t)k] = k;
            if (order == 1) std::shuffle(idx.begin(), idx.end(), g);
            Batch view; // the pairs in the order of `idx`, for the expected values
            std::vector<int64_t> ts((size_t)n), qs((size_t)n);
            std::vector<int32_t> tl((size_t)n), ql((size_t)n);
            for (int64_t k = 0; k < n; ++k) {
                const int64_t s = idx[(size_t)k];
                ts[(size_t)k] = mixed.toff[(size_t)s];
                qs[(size_t)k] = mixed.qoff[(size_t)s];
                tl[(size_t)k] = (int32_t)(mixed.toff[(size_t)s + 1] - mixed.toff[(size_t)s]);
                ql[(size_t)k] = (int32_t)(mixed.qoff[(size_t)s + 1] - mixed.qoff[(size_t)s]);
                view.add(std::string(mixed.t.begin() + mixed.toff[(size_t)s], mixed.t.begin() + mixed.toff[(size_t)s + 1]),
                         std::string(mixed.q.begin() + mixed.qoff[(size_t)s], mixed.q.begin() + mixed.qoff[(size_t)s + 1]));
            }
            const Expect ev = expect(view, MGL_SW_OS_SOFTCLIP);
            CHECK(mgl_sw_ctx_set_workspace(ctx, 96ll << 20) == 0);
            std::vector<int32_t> off((size_t)n), len((size_t)n), st((size_t)n);
            std::vector<mgl_sw_score> sc((size_t)n);
            std::vector<char> cg((size_t)n * 128, 1);
            CHECK(mgl_sw_align_batch_2bit(ctx, n, T.data(), (int64_t)mixed.t.size(), ts.data(), tl.data(), Q.data(), (int64_t)mixed.q.size(), qs.data(), ql.data(),
                                          256, 150, 200, -150, 260, 11, MGL_SW_OS_SOFTCLIP, off.data(), sc.data(), cg.data(), 128, len.data(), st.data(), 0) == 0);
            compare(view, ev, off, sc, cg, 128, len, &st);
            mgl_sw_timing tm;
            CHECK(mgl_sw_ctx_get_timing(ctx, &tm) == 0 && tm.dp_launches >= 3);
        }
        // a LARGE uniform batch (many rounds of the fake chip: four CUs): the chunks grow and shrink -- 1, 2, 4, .. rounds of 4 096 pairs,
        // halving towards the end -- on two streams and two workspace halves; windows of one packed "genome" against packed reads
        {
            const int64_t nu = 40000 + 37;
            const int utl = 64, uql = 40;
            std::vector<uint8_t> genome((size_t)(1 << 16)), reads((size_t)nu * uql);
            for (auto &c : genome) c = (uint8_t)"ACGT"[g() & 3];
            std::vector<int64_t> ts((size_t)nu), qs((size_t)nu);
            Batch view;
            for (int64_t k = 0; k < nu; ++k) {
                const int64_t w = (int64_t)(g() % (genome.size() - utl));
                ts[(size_t)k] = w;
                qs[(size_t)k] = k * uql;
                std::string t(genome.begin() + w, genome.begin() + w + utl), q = t.substr(g() % (utl - uql + 1), (size_t)uql);
                q[g() % uql] = "ACGT"[g() & 3];
                if (k % 7 == 0) q.erase(5, 3).append("ACG"); // a gap
                memcpy(&reads[(size_t)k * uql], q.data(), (size_t)uql);
                view.add(t, q);
            }
            const std::vector<uint8_t> G = pack(genome), Rd = pack(reads);
            const Expect ev = expect(view, MGL_SW_OS_SOFTCLIP);
            CHECK(mgl_sw_ctx_set_workspace(ctx, 1ll << 30) == 0 && mgl_sw_ctx_set_lane_kernel(ctx, 2) == 0);
            std::vector<int32_t> off((size_t)nu), len((size_t)nu), st((size_t)nu);
            std::vector<mgl_sw_score> sc((size_t)nu);
            std::vector<char> cg((size_t)nu * 64, 1);
            for (const char *pyr : {"8", "0"}) { // growing chunks, then all chunks two rounds: the same answers
                setenv("MGL_SW_DEBUG_HOST_PYRAMID", pyr, 1);
                CHECK(mgl_sw_align_batch_2bit(ctx, nu, G.data(), (int64_t)genome.size(), ts.data(), nullptr, Rd.data(), (int64_t)reads.size(), qs.data(), nullptr, utl, uql, 200,
                                              -150, 260, 11, MGL_SW_OS_SOFTCLIP, off.data(), sc.data(), cg.data(), 64, len.data(), st.data(), MGL_SW_FLAG_UNIFORM_GEOMETRY) == 0);
                compare(view, ev, off, sc, cg, 64, len, &st);
                mgl_sw_timing tm;
                CHECK(mgl_sw_ctx_get_timing(ctx, &tm) == 0 && tm.fill_kernel == MGL_SW_KERNEL_LANE16_CK);
                // 9.77 rounds: chunks of 1, 2, 3, 1, 1, 1 rounds and the rest (a chunk is at most half of what is left, in whole rounds); without: five chunks of two rounds
                CHECK(tm.dp_launches == (pyr[0] == '8' ? 7 : 5));
            }
            unsetenv("MGL_SW_DEBUG_HOST_PYRAMID");
            // ---- the DIRECT form of the same entry (every array registered): ONE gated launch of the persistent grid -- here a thread that
            // speaks the waves' protocol (fake_device.cpp) -- while the host checks and copies the chunks beside it.  Every way a gated launch
            // can END EARLY, then more launches on the SAME context than its ring of tile counters has entries: the launch that comes round to
            // an entry a called-off grid has used must find it at zero (round 4's host-side copy of the counters did not: the review's finding).
            {
                void *regs[] = {const_cast<uint8_t *>(G.data()), ts.data(), const_cast<uint8_t *>(Rd.data()), qs.data(), off.data(), sc.data(), cg.data(), len.data(), st.data()};
                const size_t bytes[] = {G.size(), ts.size() * 8, Rd.size(), qs.size() * 8, off.size() * 4, sc.size() * sizeof(mgl_sw_score), cg.size(), len.size() * 4, st.size() * 4};
                for (int i = 0; i < 9; ++i) CHECK(mgl_sw_register_host_buffer(ctx, regs[i], bytes[i]) == 0);
                setenv("MGL_SW_DEBUG_DIRECT_CHUNK", "8192", 1); // (chunks of 4 352, then 8 192 pairs: five gates)
                auto direct = [&](int64_t count) {
                    return mgl_sw_align_batch_2bit(ctx, count, G.data(), (int64_t)genome.size(), ts.data(), nullptr, Rd.data(), (int64_t)reads.size(), qs.data(), nullptr, utl, uql, 200,
                                                   -150, 260, 11, MGL_SW_OS_SOFTCLIP, off.data(), sc.data(), cg.data(), 64, len.data(), st.data(), MGL_SW_FLAG_UNIFORM_GEOMETRY);
                };
                auto wipe = [&] {
                    std::fill(off.begin(), off.end(), -77);
                    std::fill(cg.begin(), cg.end(), (char)1);
                };
                const long long grids0 = mgl_sw_dev::fake_gated_grids.load();
                wipe();
                CHECK(direct(nu) == 0);
                CHECK(mgl_sw_dev::fake_gated_grids.load() == grids0 + 1); // the direct form was taken: one gated launch
                compare(view, ev, off, sc, cg, 64, len, &st);
                mgl_sw_timing tm;
                CHECK(mgl_sw_ctx_get_timing(ctx, &tm) == 0 && tm.fill_kernel == MGL_SW_KERNEL_LANE16_CK && tm.dp_launches == 1);
                // (a) a pair outside its array in the FIRST chunk: no gate ever opens, every wave leaves from its first look
                const long long leavers0 = mgl_sw_dev::fake_gate_leavers.load();
                const int64_t keep_first = ts[100], keep_last = ts[(size_t)nu - 5];
                ts[100] = (int64_t)genome.size() - utl + 1;
                CHECK(direct(nu) == MGL_SW_ERR_BAD_ARG);
                ts[100] = keep_first;
                CHECK(mgl_sw_dev::fake_gate_leavers.load() == leavers0 + 32); // all 32 waves of the four fake CUs
                // (b) ... in the LAST chunk: the grid is called off in the middle of its work
                ts[(size_t)nu - 5] = -1;
                CHECK(direct(nu) == MGL_SW_ERR_BAD_ARG);
                ts[(size_t)nu - 5] = keep_last;
                // (c) an input copy that fails behind the launch: the grid is called off too (round 4 opened the gate to n here, and the
                // grid went through pairs nobody had copied or checked)
                const long long walked0 = mgl_sw_dev::fake_walk_pairs.load();
                fake_hip_fail_copy_in(4);
                CHECK(direct(nu) == MGL_SW_ERR_DEVICE);
                fake_hip_fail_copy_in(0);
                CHECK(mgl_sw_dev::fake_walk_pairs.load() - walked0 < nu); // (... not all of them)
                CHECK(mgl_sw_ctx_check(ctx) == 0);
                // (d) more counter-using launches than the ring has entries (64), each compared with the checker: 38 tiles on 32 wave slots
                const int64_t ns = 32 * 128 + 700;
                Batch head;
                for (int64_t k = 0; k < ns; ++k)
                    head.add(std::string(view.t.begin() + view.toff[(size_t)k], view.t.begin() + view.toff[(size_t)k + 1]),
                             std::string(view.q.begin() + view.qoff[(size_t)k], view.q.begin() + view.qoff[(size_t)k + 1]));
                const Expect eh = expect(head, MGL_SW_OS_SOFTCLIP);
                for (int it = 0; it < 70; ++it) {
                    wipe();
                    CHECK(direct(ns) == 0);
                    compare(head, eh, off, sc, cg, 64, len, &st);
                }
                CHECK(mgl_sw_dev::fake_gated_grids.load() == grids0 + 4 + 70);
                // (e) a gate that "stands still" (a time-out of zero ticks): the waves give up, the call goes the chunked way and is right, the
                // direct form is off for this context from then on -- and the ring is gone round once more, through the device entry
                const long long gave0 = mgl_sw_dev::fake_gate_give_ups.load();
                setenv("MGL_SW_DEBUG_GATE_TIMEOUT_TICKS", "0", 1);
                wipe();
                CHECK(direct(nu) == 0);
                unsetenv("MGL_SW_DEBUG_GATE_TIMEOUT_TICKS");
                CHECK(mgl_sw_dev::fake_gate_give_ups.load() > gave0);
                compare(view, ev, off, sc, cg, 64, len, &st);
                const long long grids1 = mgl_sw_dev::fake_gated_grids.load();
                wipe();
                CHECK(direct(nu) == 0 && mgl_sw_dev::fake_gated_grids.load() == grids1); // (chunked from now on)
                compare(view, ev, off, sc, cg, 64, len, &st);
                for (int it = 0; it < 70; ++it) {
                    wipe();
                    CHECK(mgl_sw_align_batch_device_2bit(ctx, nullptr, ns, G.data(), ts.data(), nullptr, Rd.data(), qs.data(), nullptr, utl, uql, 200, -150, 260, 11, MGL_SW_OS_SOFTCLIP,
                                                         off.data(), sc.data(), cg.data(), 64, len.data(), st.data(), MGL_SW_FLAG_UNIFORM_GEOMETRY) == 0);
                    compare(head, eh, off, sc, cg, 64, len, &st);
                }
                CHECK(mgl_sw_ctx_check(ctx) == 0);
                unsetenv("MGL_SW_DEBUG_DIRECT_CHUNK");
                for (int i = 0; i < 9; ++i) CHECK(mgl_sw_unregister_host_buffer(ctx, regs[i]) == 0);
            }
            CHECK(mgl_sw_ctx_set_lane_kernel(ctx, 0) == 0);
        }
        // ---- MANY short pairs of mixed lengths through the packed host entry (round 5): enough of them (>= 131 072) that the chunks are sized by
        // wave slots and sorted on the "device" (launch_regroup) two chunks ahead of the fills, each chunk's inputs brought in front of its
        // sort; whole waves of one geometry through the lane kernel's grid, the left-overs in pieces; results back in the caller's order
        {
            const int64_t nm = 140000 + 77;
            const int mtl = 24, mql = 16;
            std::vector<uint8_t> genome((size_t)(1 << 14)), reads((size_t)nm * mql);
            for (auto &c : genome) c = (uint8_t)"ACGT"[g() & 3];
            std::vector<int64_t> ts((size_t)nm), qs((size_t)nm);
            std::vector<int32_t> tl((size_t)nm), ql((size_t)nm);
            Batch view;
            for (int64_t k = 0; k < nm; ++k) {
                const int64_t w = (int64_t)(g() % (genome.size() - mtl));
                tl[(size_t)k] = k % 1000 == 3 ? 10 + (int)(g() % 14) : mtl;
                ql[(size_t)k] = 12 + (int)(g() % 5);
                ts[(size_t)k] = w;
                qs[(size_t)k] = k * mql;
                std::string t(genome.begin() + w, genome.begin() + w + tl[(size_t)k]), q = t.substr(0, (size_t)std::min(ql[(size_t)k], tl[(size_t)k]));
                q.resize((size_t)ql[(size_t)k], 'G');
                q[g() % q.size()] = "ACGT"[g() & 3];
                memcpy(&reads[(size_t)k * mql], q.data(), q.size());
                view.add(t, q);
            }
            const std::vector<uint8_t> G = pack(genome), Rd = pack(reads);
            const Expect ev = expect(view, MGL_SW_OS_SOFTCLIP);
            CHECK(mgl_sw_ctx_set_workspace(ctx, 1ll << 30) == 0 && mgl_sw_ctx_set_lane_kernel(ctx, 0) == 0);
            std::vector<int32_t> off((size_t)nm), len((size_t)nm), st((size_t)nm);
            std::vector<mgl_sw_score> sc((size_t)nm);
            std::vector<char> cg((size_t)nm * 64, 1);
            setenv("MGL_SW_DEBUG_LANE_GROUP_MIN", "128", 1); // (a chunk's whole waves of one geometry get their launch of the lane kernel however few they are)
            for (const char *chunk : {"0", "32768"}) { // one chunk, then six on two streams
                setenv("MGL_SW_DEBUG_HOST_SORT_CHUNK", chunk, 1);
                std::fill(off.begin(), off.end(), -5);
                CHECK(mgl_sw_align_batch_2bit(ctx, nm, G.data(), (int64_t)genome.size(), ts.data(), tl.data(), Rd.data(), (int64_t)reads.size(), qs.data(), ql.data(), mtl, mql, 200,
                                              -150, 260, 11, MGL_SW_OS_SOFTCLIP, off.data(), sc.data(), cg.data(), 64, len.data(), st.data(), 0) == 0);
                compare(view, ev, off, sc, cg, 64, len, &st);
                mgl_sw_timing tm;
                CHECK(mgl_sw_ctx_get_timing(ctx, &tm) == 0);
                // (a short first and last chunk: 16 384, 3 x 32 768, 8 960, 16 429 pairs)
                if (tm.fill_kernel != MGL_SW_KERNEL_LANE16_CK || tm.dp_launches != (chunk[0] == '0' ? 1 : 6)) std::fprintf(stderr, "mixed host batch, chunk %s: kernel %d, %d launches\n", chunk, tm.fill_kernel, tm.dp_launches);
                CHECK(tm.fill_kernel == MGL_SW_KERNEL_LANE16_CK && tm.dp_launches == (chunk[0] == '0' ? 1 : 6));
            }
            unsetenv("MGL_SW_DEBUG_HOST_SORT_CHUNK");
            unsetenv("MGL_SW_DEBUG_LANE_GROUP_MIN");
        }
        // a pair outside its array, a length above the stated maximum, missing length arrays without the uniform flag
        int64_t ts1[1] = {(int64_t)mixed.t.size() - 10}, qs1[1] = {0};
        int32_t tl1[1] = {40}, ql1[1] = {20}, off1[1], len1[1];
        char cg1[64];
        CHECK(mgl_sw_align_batch_2bit(ctx, 1, T.data(), (int64_t)mixed.t.size(), ts1, tl1, Q.data(), (int64_t)mixed.q.size(), qs1, ql1, 256, 150, 200, -150, 260, 11,
                                      MGL_SW_OS_SOFTCLIP, off1, nullptr, cg1, 64, len1, nullptr, 0) == MGL_SW_ERR_BAD_ARG);
        ts1[0] = 0;
        tl1[0] = 300;
        CHECK(mgl_sw_align_batch_2bit(ctx, 1, T.data(), (int64_t)mixed.t.size(), ts1, tl1, Q.data(), (int64_t)mixed.q.size(), qs1, ql1, 256, 150, 200, -150, 260, 11,
                                      MGL_SW_OS_SOFTCLIP, off1, nullptr, cg1, 64, len1, nullptr, 0) == MGL_SW_ERR_BAD_ARG);
        CHECK(mgl_sw_align_batch_2bit(ctx, 1, T.data(), (int64_t)mixed.t.size(), ts1, nullptr, Q.data(), (int64_t)mixed.q.size(), qs1, nullptr, 256, 150, 200, -150, 260,
                                      11, MGL_SW_OS_SOFTCLIP, off1, nullptr, cg1, 64, len1, nullptr, 0) == MGL_SW_ERR_BAD_ARG);
    }
    mgl_sw_ctx_destroy(ctx);

    // ---- several devices from one process
    {
        mgl_sw_multi *m = nullptr;
        const int devs[3] = {0, 1, 0};
        CHECK(mgl_sw_multi_create(3, devs, &m) == 0 && mgl_sw_multi_device_count(m) == 3);
        CHECK(mgl_sw_multi_set_workspace(m, 64ll << 20) == 0);
        const int64_t n = mixed.n();
        std::vector<int32_t> off((size_t)n), len((size_t)n), st((size_t)n);
        std::vector<mgl_sw_score> sc((size_t)n);
        std::vector<char> cg((size_t)n * 128, 1);
        CHECK(mgl_sw_align_batch_multi(m, n, mixed.t.data(), mixed.toff.data(), mixed.q.data(), mixed.qoff.data(), 200, -150, 260, 11, MGL_SW_OS_SOFTCLIP,
                                       off.data(), sc.data(), cg.data(), 128, len.data(), st.data()) == 0);
        compare(mixed, em, off, sc, cg, 128, len, &st);
        int64_t first[4];
        CHECK(mgl_sw_multi_last_shards(m, first) == 0 && first[0] == 0 && first[3] == n && first[1] % 8 == 0 && first[1] > n / 4 && first[2] > first[1]);
        mgl_sw_multi_destroy(m);
    }

    // ---- 48 threads, one pair per call (the way GATK drives alignNative): through the coalescing front-end alone, then with the
    // mailbox service in front of it -- 32 mailboxes for 48 threads (sixteen threads keep to the coalescer), a grid that gives up after
    // 3 ms of silence and threads that all pause for 12 ms in the middle, so that calls find their wave gone and launch the grid again
    setenv("FAKE_HIP_CUS", "64", 1); // (the service hands out mailboxes for half the CUs at most: 32 here)
    for (int with_service = 0; with_service < 2; ++with_service) {
        CHECK(mgl_sw_set_coalescing(64, 200) == 0);
        CHECK(mgl_sw_set_service(with_service ? 32 : 0, 3000) == 0);
        const auto t_section = std::chrono::steady_clock::now();
        int64_t batches0 = 0, pairs0 = 0, calls0 = 0, launches0 = 0;
        CHECK(mgl_sw_coalescing_stats(&batches0, &pairs0) == 0 && mgl_sw_service_stats(&calls0, &launches0) == 0);
        std::atomic<int> bad{0}, overflow_seen{0}, device_seen{0};
        auto worker = [&](int id) {
            std::mt19937 r((unsigned)id * 7919u + 1u);
            // the caller has a HIP device of its own selected: the service's launches (which happen on this thread) must leave it as it is
            *fake_hip_current_device() = 1;
            for (int it = 0; it < 120; ++it) {
                if (with_service && it % 40 == 39) std::this_thread::sleep_for(std::chrono::milliseconds(12));
                // (poisoned pairs at it = 7, 47, 87; too-small buffers every tenth call of thread 9; thread 3 sends, once, a pair whose scores
                // need more LDS than the grids are launched with by default: its wave ends the grid, the next one gets the full carve)
                const bool large = id == 3 && it == 60;
                const int tl = large ? 400 : 40 + (int)(r() % 200), ql = large ? 180 : 8 + (int)(r() % 100);
                std::string t = rnd(r, tl), q = rnd(r, ql);
                const bool poison = id == 5 && it % 40 == 7; // the fake device fails this pair (see below)
                if (poison) q.replace(0, 8, "NNNNNNNN");
                const int cap = (id == 9 && it % 10 == 0) ? 2 : 4096;
                std::vector<char> buf((size_t)cap), ref(4096);
                int len = 0, off = 0, rlen = 0, roff = 0;
                mgl_sw_score ez;
                swo_score rz;
                const int rc = mgl_sw_align(t.data(), tl, q.data(), ql, 200, -150, 260, 11, MGL_SW_OS_SOFTCLIP, buf.data(), cap, &len, &off, &ez);
                CHECK(swo_align((const uint8_t *)t.data(), tl, (const uint8_t *)q.data(), ql, 200, -150, 260, 11, MGL_SW_OS_SOFTCLIP, ref.data(), 4096, &rlen,
                                &roff, &rz, nullptr) == 0);
                if (poison) { // a device-side failure of one pair reaches its caller as such, not as "CIGAR does not fit"
                    if (rc != MGL_SW_ERR_DEVICE) ++bad;
                    ++device_seen;
                } else if (rc == MGL_SW_ERR_CIGAR_OVERFLOW && cap < rlen) {
                    if (len != rlen) ++bad;
                    ++overflow_seen;
                } else if (rc != MGL_SW_OK || len != rlen || off != roff || memcmp(buf.data(), ref.data(), (size_t)rlen) != 0 ||
                           memcmp(&ez, &rz, sizeof ez) != 0) {
                    ++bad;
                }
                if (*fake_hip_current_device() != 1) ++bad;
            }
        };
        std::vector<std::thread> th;
        for (int id = 0; id < 48; ++id) th.emplace_back(worker, id);
        for (auto &x : th) x.join();
        int64_t batches = 0, pairs = 0, calls = 0, launches = 0;
        CHECK(mgl_sw_coalescing_stats(&batches, &pairs) == 0 && mgl_sw_service_stats(&calls, &launches) == 0);
        batches -= batches0, pairs -= pairs0, calls -= calls0, launches -= launches0;
        CHECK(pairs + calls == 48 * 120 && batches < pairs);
        if (with_service)
            CHECK(mgl_sw_dev::fake_service_lds_bytes.load() == 160 * 1024 && // (the large pair of thread 3 raised the carve)
                  calls >= 32 * 120 && launches >= 2 && // (a thread that ends early hands its mailbox to one that had none; grids grow as threads arrive)
                  mgl_sw_dev::fake_service_waves.load() >= 32 && calls == mgl_sw_dev::fake_service_pairs.load());
        else
            CHECK(calls == 0 && launches == 0);
        CHECK(bad.load() == 0 && overflow_seen.load() > 0 && device_seen.load() == 3);
        std::fprintf(stderr, "one pair per call, %s: %lld through mailboxes (%lld grids), %lld coalesced in %lld batches, %.1f s\n", with_service ? "32 mailboxes" : "coalescer only",
                     (long long)calls, (long long)launches, (long long)pairs, (long long)batches, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_section).count());
        CHECK(mgl_sw_set_coalescing(0, 0) == 0);
    }
    std::printf("host-san ok: %lld fill launches, %lld pairs walked\n", mgl_sw_dev::fake_fill_launches.load(), mgl_sw_dev::fake_walk_pairs.load());
    return 0;
}
