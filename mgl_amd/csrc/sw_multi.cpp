// sw_multi.cpp -- one process, several GPUs: the library-level multi-device entry.
//
// Every (target, query) pair is a pure function call -- the reference's alignNative holds no state
// (/root/reference/src/main/native/mgl_sw/com_microsoft_mgl_smithwaterman_MicrosoftSmithWaterman.cpp:44-71) -- so a host
// batch is cut into contiguous shards, one per device, balanced by the DP cells (sum tl * ql) they hold, and every
// shard goes through the ordinary host-buffer entry (mgl_sw_align_batch_status) on its own mgl_sw_ctx from its own
// host thread.  Results land directly in the caller's arrays (each shard owns a disjoint slice); there is no
// inter-GPU traffic at all.  Host code only.
#include "../../include/mgl_sw.h"

#include <algorithm>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

struct mgl_sw_multi {
    std::vector<int> devices;
    std::vector<mgl_sw_ctx *> ctx;
    std::vector<int64_t> shard_first; // shards of the last call, n_devices + 1 entries
    std::string err;
    std::mutex mu;
};

extern "C" {

int mgl_sw_shard_by_cells(int64_t n, const int64_t *t_off, const int64_t *q_off, int parts, int64_t align,
                          int64_t *first_out)
{
    if (n < 0 || parts < 1 || align < 1 || !first_out || (n > 0 && (!t_off || !q_off))) return MGL_SW_ERR_BAD_ARG;
    first_out[0] = 0;
    if (n == 0) {
        for (int p = 1; p <= parts; ++p) first_out[p] = 0;
        return MGL_SW_OK;
    }
    // prefix sums of the cells, then the boundary of part p is the first pair index (a multiple of `align`) at which the
    // running total reaches p / parts of the whole
    std::vector<double> pre((size_t)n + 1);
    pre[0] = 0.0;
    for (int64_t k = 0; k < n; ++k) {
        const int64_t tl = t_off[k + 1] - t_off[k], ql = q_off[k + 1] - q_off[k];
        if (tl < 0 || ql < 0) return MGL_SW_ERR_BAD_ARG;
        pre[(size_t)k + 1] = pre[(size_t)k] + (double)tl * (double)ql;
    }
    const double total = pre[(size_t)n];
    for (int p = 1; p < parts; ++p) {
        const double want = total * (double)p / (double)parts;
        int64_t k = std::lower_bound(pre.begin(), pre.end(), want) - pre.begin();
        k = (k + align / 2) / align * align; // whole waves' worth of pairs per shard (the packed kernels work on blocks)
        k = std::max(first_out[p - 1], std::min(k, n));
        first_out[p] = k;
    }
    first_out[parts] = n;
    return MGL_SW_OK;
}

int mgl_sw_multi_create(int n_devices, const int *devices, mgl_sw_multi **out)
{
    if (!out || n_devices < 1 || n_devices > 64) return MGL_SW_ERR_BAD_ARG;
    *out = nullptr;
    mgl_sw_multi *m = new (std::nothrow) mgl_sw_multi;
    if (!m) return MGL_SW_ERR_NOMEM;
    for (int d = 0; d < n_devices; ++d) {
        const int dev = devices ? devices[d] : d;
        mgl_sw_ctx *c = nullptr;
        const int rc = mgl_sw_ctx_create(dev, &c);
        if (rc != MGL_SW_OK) {
            for (mgl_sw_ctx *x : m->ctx) mgl_sw_ctx_destroy(x);
            delete m;
            return rc;
        }
        m->devices.push_back(dev);
        m->ctx.push_back(c);
    }
    m->shard_first.assign((size_t)n_devices + 1, 0);
    *out = m;
    return MGL_SW_OK;
}

void mgl_sw_multi_destroy(mgl_sw_multi *m)
{
    if (!m) return;
    for (mgl_sw_ctx *c : m->ctx) mgl_sw_ctx_destroy(c);
    delete m;
}

int mgl_sw_multi_device_count(const mgl_sw_multi *m) { return m ? (int)m->ctx.size() : 0; }

mgl_sw_ctx *mgl_sw_multi_ctx(mgl_sw_multi *m, int index)
{
    return (m && index >= 0 && index < (int)m->ctx.size()) ? m->ctx[(size_t)index] : nullptr;
}

const char *mgl_sw_multi_last_error(const mgl_sw_multi *m) { return m ? m->err.c_str() : ""; }

int mgl_sw_multi_set_workspace(mgl_sw_multi *m, int64_t bytes_per_device)
{
    if (!m) return MGL_SW_ERR_BAD_ARG;
    for (mgl_sw_ctx *c : m->ctx) {
        const int rc = mgl_sw_ctx_set_workspace(c, bytes_per_device);
        if (rc != MGL_SW_OK) return rc;
    }
    return MGL_SW_OK;
}

int mgl_sw_multi_last_shards(mgl_sw_multi *m, int64_t *first_out)
{
    if (!m || !first_out) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    memcpy(first_out, m->shard_first.data(), m->shard_first.size() * sizeof(int64_t));
    return MGL_SW_OK;
}

int mgl_sw_align_batch_multi(mgl_sw_multi *m, int64_t n, const uint8_t *targets, const int64_t *t_off,
                             const uint8_t *queries, const int64_t *q_off, int match, int mismatch, int gopen,
                             int gext, int strategy, int32_t *offset_out, mgl_sw_score *score_out, char *cigar_out,
                             int cigar_stride, int32_t *cigar_len_out, int32_t *status_out)
{
    if (!m) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(m->mu);
    m->err.clear();
    if (n == 0) return MGL_SW_OK;
    if (n < 0 || !targets || !t_off || !queries || !q_off || !offset_out || !cigar_out || cigar_stride < 1) {
        m->err = "mgl_sw_align_batch_multi: bad argument";
        return MGL_SW_ERR_BAD_ARG;
    }
    const int parts = (int)m->ctx.size();
    // shards are multiples of eight pairs: a uniform or grouped batch (blocks of eight with one geometry) stays one
    int rc = mgl_sw_shard_by_cells(n, t_off, q_off, parts, 8, m->shard_first.data());
    if (rc != MGL_SW_OK) {
        m->err = "mgl_sw_align_batch_multi: offsets are not ascending";
        return rc;
    }
    // No C++ exception may cross the C ABI, and none may escape while worker threads are joinable (std::terminate): a shard whose
    // vectors cannot be allocated reports MGL_SW_ERR_NOMEM like any other failure of that device, and threads that cannot be
    // started leave their shards to the calling thread.
    std::vector<int> status;
    try {
        status.assign((size_t)parts, MGL_SW_OK);
    } catch (const std::exception &) {
        m->err = "mgl_sw_align_batch_multi: out of memory";
        return MGL_SW_ERR_NOMEM;
    }
    auto work = [&](int p) noexcept {
        const int64_t lo = m->shard_first[(size_t)p], hi = m->shard_first[(size_t)p + 1], cnt = hi - lo;
        if (cnt == 0) return;
        // the host entry wants offsets that start at 0: rebase this shard's two offset arrays (16 bytes per pair)
        std::vector<int64_t> to, qo;
        try {
            to.resize((size_t)cnt + 1);
            qo.resize((size_t)cnt + 1);
        } catch (const std::exception &) {
            status[(size_t)p] = MGL_SW_ERR_NOMEM;
            return;
        }
        const int64_t t0 = t_off[lo], q0 = q_off[lo];
        for (int64_t k = 0; k <= cnt; ++k) {
            to[(size_t)k] = t_off[lo + k] - t0;
            qo[(size_t)k] = q_off[lo + k] - q0;
        }
        status[(size_t)p] = mgl_sw_align_batch_status(
            m->ctx[(size_t)p], cnt, targets + t0, to.data(), queries + q0, qo.data(), match, mismatch, gopen, gext, strategy,
            offset_out + lo, score_out ? score_out + lo : nullptr, cigar_out + (size_t)lo * (size_t)cigar_stride, cigar_stride,
            cigar_len_out ? cigar_len_out + lo : nullptr, status_out ? status_out + lo : nullptr);
    };
    std::vector<std::thread> th;
    int started = 1;
    try {
        th.reserve((size_t)parts);
        for (; started < parts; ++started) th.emplace_back(work, started);
    } catch (const std::exception &) { // (no more threads: the shards from `started` on run here, one after the other)
    }
    work(0); // the calling thread drives the first device
    for (int p = started; p < parts; ++p) work(p);
    for (std::thread &t : th) t.join();
    for (int p = 0; p < parts; ++p)
        if (status[(size_t)p] != MGL_SW_OK) {
            m->err = std::string("device ") + std::to_string(m->devices[(size_t)p]) + ": " + mgl_sw_last_error(m->ctx[(size_t)p]);
            return status[(size_t)p];
        }
    return MGL_SW_OK;
}

} // extern "C"
