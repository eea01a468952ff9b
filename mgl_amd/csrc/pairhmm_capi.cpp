// pairhmm_capi.cpp -- host side of the C ABI in include/mgl_pairhmm.h: lookup tables, buffers, the two passes.
// There is no CPU compute path here: without a HIP device every entry fails with MGL_PAIRHMM_ERR_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/mgl_pairhmm.h"
#include "pairhmm_device.h"

using namespace mgl_ph_dev;

namespace {

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        size_t want = bytes + bytes / 4 + 256;
        hipError_t e = hipMalloc(&p, want);
        if (e != hipSuccess) return e;
        cap = want;
        return hipSuccess;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

// Context<NUMBER> tables, Context.h:13-190, built once per context on the host exactly as the reference
// builds them at library load (double arithmetic, then the cast to NUMBER)
constexpr double JAC_TOL = 8.0, JAC_STEP = 0.0001, JAC_INV_STEP = 1.0 / JAC_STEP;

template <typename T>
struct Tables {
    std::vector<T> ph2pr, m2m, jacobian;
    static int fast_round(T d) { return (d > (T)0.0) ? (int)(d + (T)0.5) : (int)(d - (T)0.5); } // :57-59
    T approx_log10_sum(T small, T big) const                                                    // :61-86
    {
        if (small > big) std::swap(small, big);
        if (std::isinf(small) || std::isinf(big)) return big;
        const T diff = big - small;
        if (diff >= (T)JAC_TOL) return big;
        return big + jacobian[(size_t)fast_round((T)(diff * ((T)JAC_INV_STEP)))];
    }
    void build()
    {
        const int jac_size = (int)(JAC_TOL / JAC_STEP) + 1; // :11
        jacobian.resize((size_t)jac_size);
        for (int k = 0; k < jac_size; k++) jacobian[(size_t)k] = (T)(std::log10(1.0 + std::pow(10.0, -((double)k) * JAC_STEP))); // :41-46
        m2m.resize(M2M_SIZE);
        const double INV_LN10 = 1.0 / std::log(10);
        for (int i = 0, offset = 0; i <= MAX_QUAL; offset += ++i) // :49-59
            for (int j = 0; j <= i; j++) {
                const double log10Sum = approx_log10_sum(-0.1f * i, -0.1f * j);
                const double m2mLog10 = std::log1p(-std::min(1.0, std::pow(10, log10Sum))) * INV_LN10;
                m2m[(size_t)(offset + j)] = (T)(std::pow(10, m2mLog10));
            }
        ph2pr.resize(128);
        for (int x = 0; x < 128; x++)
            ph2pr[(size_t)x] = sizeof(T) == 4 ? (T)powf(10.f, -((float)x) / 10.f) : (T)std::pow(10.0, -((double)x) / 10.0); // :144,:105
    }
};

} // namespace

struct mgl_pairhmm_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::mutex mu;
    std::string err;
    int use_double = 0;
    int profiling = 0;
    int stripe_rows = 0; // 0 = per batch, 16 / 32 / 64 = forced (mgl_pairhmm_set_stripe_rows)
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ev_valid = false, ran_float = false;
    mgl_pairhmm_timing timing{};
    DevBuf t_ph2pr_f, t_m2m_f, t_ph2pr_d, t_m2m_d;
    DevBuf d_reads, d_read_off, d_haps, d_hap_off, d_pr, d_ph, d_out, d_need;
    // small calls (one active region, the JNI entry): one pinned, device-mapped buffer: copied in once, results written in place
    void *pin = nullptr;
    size_t pin_cap = 0;
};

namespace {

int fail(mgl_pairhmm_ctx *ctx, int code, const char *what)
{
    if (ctx) ctx->err = what;
    return code;
}
int hip_fail(mgl_pairhmm_ctx *ctx, hipError_t e, const char *what)
{
    if (ctx) ctx->err = std::string(what) + ": " + hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? MGL_PAIRHMM_ERR_NOMEM : MGL_PAIRHMM_ERR_DEVICE;
}
#define HIP_TRY(ctx, call)                                     \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

int max_hap_len_for(int elem_bytes)
{
    int lo = 1, hi = 1 << 20;
    while (lo < hi) {
        const int mid = (lo + hi + 1) / 2;
        if (ph_lds_bytes(mid, 64, elem_bytes) <= 160 * 1024)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

// enqueue the float pass (unless use_double) and the double pass on `stream`
int run_device(mgl_pairhmm_ctx *ctx, hipStream_t stream, int64_t n_pairs, const uint8_t *d_reads, const int64_t *d_read_off,
               const uint8_t *d_haps, const int64_t *d_hap_off, const int32_t *d_pr, const int32_t *d_ph, int max_read_len,
               int max_hap_len, double *d_out, int32_t *d_need)
{
    if (n_pairs == 0) return MGL_PAIRHMM_OK;
    if (n_pairs < 0 || !d_reads || !d_read_off || !d_haps || !d_hap_off || !d_pr || !d_ph || !d_out || max_read_len < 1 ||
        max_hap_len < 1)
        return fail(ctx, MGL_PAIRHMM_ERR_BAD_ARG, "mgl_pairhmm: bad argument");
    if (max_hap_len > mgl_pairhmm_max_haplotype_len(1))
        return fail(ctx, MGL_PAIRHMM_ERR_UNSUPPORTED, "haplotype longer than mgl_pairhmm_max_haplotype_len()");
    if (n_pairs > (int64_t)0x7fffffff * 4) return fail(ctx, MGL_PAIRHMM_ERR_UNSUPPORTED, "too many pairs for one call");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!d_need) {
        HIP_TRY(ctx, ctx->d_need.reserve((size_t)n_pairs * 4));
        d_need = static_cast<int32_t *>(ctx->d_need.p);
    }
    PhArgs a;
    a.n_pairs = n_pairs;
    a.reads = d_reads;
    a.read_off = d_read_off;
    a.haps = d_haps;
    a.hap_off = d_hap_off;
    a.pair_read = d_pr;
    a.pair_hap = d_ph;
    a.ph2pr_f = static_cast<const float *>(ctx->t_ph2pr_f.p);
    a.m2m_f = static_cast<const float *>(ctx->t_m2m_f.p);
    a.ph2pr_d = static_cast<const double *>(ctx->t_ph2pr_d.p);
    a.m2m_d = static_cast<const double *>(ctx->t_m2m_d.p);
    a.log10_initial_f = (double)log10f(ldexpf(1.f, 120)); // Context.h:148 (a float), …PairHmm.cc:185
    a.log10_initial_d = std::log10(std::ldexp(1.0, 1020)); // Context.h:109
    a.hap_cap = max_hap_len;
    a.out = d_out;
    a.need_double = d_need;
    a.rescue_only = ctx->use_double ? 0 : 1;
    // lanes per pair: 16 rows x four pairs per wave wastes the fewest lanes, but its four carry rings (16 bytes per
    // haplotype base each) can leave a CU with one or two waves; 64 rows x one pair keeps the CU full
    int rows = ctx->stripe_rows;
    if (!rows) {
        if (rows != 16 && rows != 21 && rows != 32 && rows != 64) {
            rows = (max_read_len >= 48 && ph_lds_bytes(max_hap_len, 16, 4) > 12 * 1024) ? 64 : 16;
            // Reads of 24 .. 160 bases, more pairs than SIMDs: several pairs per wave -- 32 lanes x up to 5 rows (two pairs) or
            // 21 lanes x up to 5 rows (three pairs, reads up to 105), whichever wastes fewer row slots -- instead of one pair
            // on 64 lanes: 150 bases fill 150 of 160 slots instead of 150 of 192 and the pipeline is 32 (21) steps deep
            // instead of 64.  Measured (1.6 M pairs): 150 x 300 2 560 vs 1 908 GCUPS; 100 x 250 2 470 (21 lanes) vs 1 964
            // (32) vs 1 639 (64); 76 x 100 1 739 vs 1 436 (16 lanes); 64 x 200 1 837 (32) vs 1 805 (21); 150 x 1000 2 051
            // vs 2 076 (64): the carry rings of long haplotypes cost occupancy, hence the LDS bounds.
            // A call of up to 1 024 pairs (one active region) is latency bound instead: one pair per wave has the fewest
            // rows per lane, i.e. the shortest dependent chain per step (800 pairs: 74 vs 80 us).
            if (n_pairs > 1024 && max_read_len >= 24 && max_read_len <= 160) {
                const int slots21 = 21 * ((max_read_len + 20) / 21), slots32 = 32 * ((max_read_len + 31) / 32);
                const bool ok21 = max_read_len <= 105 && ph_lds_bytes(max_hap_len, 21, 4) <= 24 * 1024;
                const bool ok32 = ph_lds_bytes(max_hap_len, 32, 4) <= 24 * 1024;
                if (ok21 && (slots21 <= slots32 || !ok32))
                    rows = 21;
                else if (ok32)
                    rows = 32;
                else if (max_read_len > 64)
                    rows = 64;
            }
        }
    }
    if (rows == 32 && (max_read_len > 160 || ph_lds_bytes(max_hap_len, 32, 4) > 160 * 1024)) rows = 64; // one stripe of 32 x 5 rows
    if (rows == 21 && (max_read_len > 105 || ph_lds_bytes(max_hap_len, 21, 4) > 160 * 1024)) rows = 64; // one stripe of 21 x 5 rows
    if (ph_lds_bytes(max_hap_len, rows, 4) > 160 * 1024) rows = 64; // four rings would not fit LDS at all
    const int rows_d = ph_lds_bytes(max_hap_len, 16, 8) <= 160 * 1024 && rows == 16 ? 16 : 64;
    const int rows_per_lane = std::min(4, (max_read_len + 63) / 64);           // one pair per wave (and the double pass)
    const int rows_per_lane_f = rows == 32 ? std::max(3, (max_read_len + 31) / 32) : rows == 21 ? std::max(3, (max_read_len + 20) / 21) : rows_per_lane;
    const bool prof = ctx->profiling != 0;
    ctx->ev_valid = false;
    ctx->ran_float = !ctx->use_double;
    if (!ctx->use_double) {
        if (prof) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], stream));
        HIP_TRY(ctx, launch_pairhmm_float(a, rows, rows_per_lane_f, stream));
        if (prof) HIP_TRY(ctx, hipEventRecord(ctx->ev[1], stream));
    }
    if (prof) HIP_TRY(ctx, hipEventRecord(ctx->ev[2], stream));
    HIP_TRY(ctx, launch_pairhmm_double(a, rows_d, rows_per_lane, stream));
    if (prof) HIP_TRY(ctx, hipEventRecord(ctx->ev[3], stream));
    ctx->ev_valid = prof;
    return MGL_PAIRHMM_OK;
}

} // namespace

extern "C" {

int mgl_pairhmm_version(void) { return MGL_PAIRHMM_VERSION; }

const char *mgl_pairhmm_strerror(int status)
{
    switch (status) {
    case MGL_PAIRHMM_OK: return "ok";
    case MGL_PAIRHMM_ERR_BAD_ARG: return "bad argument";
    case MGL_PAIRHMM_ERR_NOMEM: return "out of memory";
    case MGL_PAIRHMM_ERR_DEVICE: return "no HIP device or HIP runtime error";
    case MGL_PAIRHMM_ERR_UNSUPPORTED: return "geometry not supported";
    default: return "unknown status";
    }
}

const char *mgl_pairhmm_last_error(const mgl_pairhmm_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

int mgl_pairhmm_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mgl_pairhmm_max_haplotype_len(int use_double)
{
    (void)use_double; // a float batch may need the double rescue, so the double carve bounds both
    static const int cap = max_hap_len_for(8);
    return cap;
}

int mgl_pairhmm_ctx_create(int device, mgl_pairhmm_ctx **out)
{
    if (!out) return MGL_PAIRHMM_ERR_BAD_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) return MGL_PAIRHMM_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MGL_PAIRHMM_ERR_DEVICE;
    mgl_pairhmm_ctx *ctx = new (std::nothrow) mgl_pairhmm_ctx;
    if (!ctx) return MGL_PAIRHMM_ERR_NOMEM;
    ctx->device = device;
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    for (auto &e : ctx->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    Tables<float> tf;
    Tables<double> td;
    tf.build();
    td.build();
    auto up = [&](DevBuf &b, const void *src, size_t bytes) {
        ok = ok && b.reserve(bytes) == hipSuccess && hipMemcpy(b.p, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    up(ctx->t_ph2pr_f, tf.ph2pr.data(), tf.ph2pr.size() * 4);
    up(ctx->t_m2m_f, tf.m2m.data(), tf.m2m.size() * 4);
    up(ctx->t_ph2pr_d, td.ph2pr.data(), td.ph2pr.size() * 8);
    up(ctx->t_m2m_d, td.m2m.data(), td.m2m.size() * 8);
    if (!ok) {
        mgl_pairhmm_ctx_destroy(ctx);
        return MGL_PAIRHMM_ERR_DEVICE;
    }
    *out = ctx;
    return MGL_PAIRHMM_OK;
}

void mgl_pairhmm_ctx_destroy(mgl_pairhmm_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (DevBuf *b : {&ctx->t_ph2pr_f, &ctx->t_m2m_f, &ctx->t_ph2pr_d, &ctx->t_m2m_d, &ctx->d_reads, &ctx->d_read_off, &ctx->d_haps,
                      &ctx->d_hap_off, &ctx->d_pr, &ctx->d_ph, &ctx->d_out, &ctx->d_need})
        b->release();
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->pin) (void)hipHostFree(ctx->pin);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int mgl_pairhmm_initialize(mgl_pairhmm_ctx *ctx, int use_double, int max_threads)
{
    (void)max_threads; // ignored by the reference as well (…PairHmm.cc:50-70)
    if (!ctx) return MGL_PAIRHMM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->use_double = use_double ? 1 : 0;
    return MGL_PAIRHMM_OK;
}

int mgl_pairhmm_set_stripe_rows(mgl_pairhmm_ctx *ctx, int rows)
{
    if (!ctx || (rows != 0 && rows != 16 && rows != 21 && rows != 32 && rows != 64)) return MGL_PAIRHMM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->stripe_rows = rows;
    return MGL_PAIRHMM_OK;
}

int mgl_pairhmm_set_profiling(mgl_pairhmm_ctx *ctx, int enable)
{
    if (!ctx) return MGL_PAIRHMM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->profiling = enable ? 1 : 0;
    return MGL_PAIRHMM_OK;
}

int mgl_pairhmm_get_timing(mgl_pairhmm_ctx *ctx, mgl_pairhmm_timing *out)
{
    if (!ctx || !out) return MGL_PAIRHMM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->ev_valid) {
        HIP_TRY(ctx, hipEventSynchronize(ctx->ev[3]));
        ctx->timing.float_ms = 0.f;
        if (ctx->ran_float) HIP_TRY(ctx, hipEventElapsedTime(&ctx->timing.float_ms, ctx->ev[0], ctx->ev[1]));
        HIP_TRY(ctx, hipEventElapsedTime(&ctx->timing.double_ms, ctx->ev[2], ctx->ev[3]));
        ctx->ev_valid = false;
    }
    *out = ctx->timing;
    return MGL_PAIRHMM_OK;
}

int mgl_pairhmm_compute_pairs_device(mgl_pairhmm_ctx *ctx, void *stream, int64_t n_pairs, const uint8_t *d_reads_data,
                                     const int64_t *d_read_off, const uint8_t *d_haps_data, const int64_t *d_hap_off,
                                     const int32_t *d_pair_read, const int32_t *d_pair_hap, int max_read_len, int max_hap_len,
                                     double *d_out, int32_t *d_used_double)
{
    if (!ctx) return MGL_PAIRHMM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->timing = mgl_pairhmm_timing{};
    ctx->timing.rescued = -1;
    return run_device(ctx, stream ? static_cast<hipStream_t>(stream) : ctx->stream, n_pairs, d_reads_data, d_read_off, d_haps_data,
                      d_hap_off, d_pair_read, d_pair_hap, max_read_len, max_hap_len, d_out, d_used_double);
}

int mgl_pairhmm_compute_pairs(mgl_pairhmm_ctx *ctx, int64_t n_pairs, int64_t n_reads, const uint8_t *reads_data,
                              const int64_t *read_off, int64_t n_haps, const uint8_t *haps_data, const int64_t *hap_off,
                              const int32_t *pair_read, const int32_t *pair_hap, double *out)
{
    if (!ctx) return MGL_PAIRHMM_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (n_pairs == 0) return MGL_PAIRHMM_OK;
    if (n_pairs < 0 || n_reads < 1 || n_haps < 1 || !reads_data || !read_off || !haps_data || !hap_off || !pair_read || !pair_hap ||
        !out || read_off[0] != 0 || hap_off[0] != 0)
        return fail(ctx, MGL_PAIRHMM_ERR_BAD_ARG, "mgl_pairhmm_compute_pairs: bad argument");
    int64_t max_r = 0, max_h = 0;
    for (int64_t r = 0; r < n_reads; ++r) {
        const int64_t len = read_off[r + 1] - read_off[r];
        // an empty read or haplotype divides by zero / indexes row -1 in the reference (compute_prob_scalar.cc:101,214)
        if (len < 1 || len > 0x3fffffff) return fail(ctx, MGL_PAIRHMM_ERR_BAD_ARG, "read length < 1 or too large");
        max_r = std::max(max_r, len);
    }
    for (int64_t h = 0; h < n_haps; ++h) {
        const int64_t len = hap_off[h + 1] - hap_off[h];
        if (len < 1 || len > 0x3fffffff) return fail(ctx, MGL_PAIRHMM_ERR_BAD_ARG, "haplotype length < 1 or too large");
        max_h = std::max(max_h, len);
    }
    int64_t cells = 0;
    for (int64_t k = 0; k < n_pairs; ++k) {
        const int32_t r = pair_read[k], h = pair_hap[k];
        if (r < 0 || r >= n_reads || h < 0 || h >= n_haps) return fail(ctx, MGL_PAIRHMM_ERR_BAD_ARG, "pair index out of range");
        cells += (read_off[r + 1] - read_off[r]) * (hap_off[h + 1] - hap_off[h]);
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const size_t rbytes = (size_t)read_off[n_reads] * 5, hbytes = (size_t)hap_off[n_haps];
    hipStream_t st = ctx->stream;
    ctx->timing = mgl_pairhmm_timing{};
    ctx->timing.cells = cells;

    // ---- small calls (one active region per call is how GATK drives the JNI entry): latency, not bandwidth.  Everything
    // goes into ONE pinned, device-mapped, coherent buffer: one copy command in, results written in place by the kernels,
    // one synchronisation (the eight copies of the general path cost ~45 us of launch latency per call: 800 pairs of
    // 150 x 300 take 82 us per call instead of 127)
    auto up8 = [](size_t x) { return (x + 7) & ~(size_t)7; };
    const size_t o_roff = 0, o_hoff = o_roff + (size_t)(n_reads + 1) * 8, o_out = o_hoff + (size_t)(n_haps + 1) * 8,
                 o_pr = o_out + (size_t)n_pairs * 8, o_ph = o_pr + up8((size_t)n_pairs * 4), o_need = o_ph + up8((size_t)n_pairs * 4),
                 o_reads = o_need + up8((size_t)n_pairs * 4), o_haps = o_reads + up8(rbytes), total = o_haps + up8(hbytes);
    constexpr size_t zero_copy_max = (size_t)(1u << 20);
    if (total <= zero_copy_max) {
        if (total > ctx->pin_cap) {
            if (ctx->pin) (void)hipHostFree(ctx->pin);
            ctx->pin = nullptr;
            ctx->pin_cap = 0;
            const size_t want = std::max<size_t>(2 * total, 256 * 1024);
            HIP_TRY(ctx, hipHostMalloc(&ctx->pin, want, hipHostMallocDefault));
            ctx->pin_cap = want;
        }
        char *h = static_cast<char *>(ctx->pin);
        memcpy(h + o_roff, read_off, (size_t)(n_reads + 1) * 8);
        memcpy(h + o_hoff, hap_off, (size_t)(n_haps + 1) * 8);
        memcpy(h + o_pr, pair_read, (size_t)n_pairs * 4);
        memcpy(h + o_ph, pair_hap, (size_t)n_pairs * 4);
        memcpy(h + o_reads, reads_data, rbytes);
        memcpy(h + o_haps, haps_data, hbytes);
        void *dv = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&dv, ctx->pin, 0));
        char *d = static_cast<char *>(dv);   // the results are written in place
        // the inputs go through ONE copy into HBM (every read is used by all haplotypes: reading them in place is no
        // faster, 74-163 vs 71-146 us per region)
        HIP_TRY(ctx, ctx->d_reads.reserve(ctx->pin_cap));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_reads.p, ctx->pin, total, hipMemcpyHostToDevice, st));
        const char *di = static_cast<const char *>(ctx->d_reads.p);
        const int rc = run_device(ctx, st, n_pairs, reinterpret_cast<const uint8_t *>(di + o_reads), reinterpret_cast<const int64_t *>(di + o_roff),
                                  reinterpret_cast<const uint8_t *>(di + o_haps), reinterpret_cast<const int64_t *>(di + o_hoff),
                                  reinterpret_cast<const int32_t *>(di + o_pr), reinterpret_cast<const int32_t *>(di + o_ph), (int)max_r,
                                  (int)max_h, reinterpret_cast<double *>(d + o_out), reinterpret_cast<int32_t *>(d + o_need));
        const hipError_t es = hipStreamSynchronize(st);
        if (rc != MGL_PAIRHMM_OK) return rc;
        if (es != hipSuccess) return hip_fail(ctx, es, "hipStreamSynchronize");
        memcpy(out, h + o_out, (size_t)n_pairs * 8);
        int64_t rescued = 0;
        if (!ctx->use_double) {
            const int32_t *need = reinterpret_cast<const int32_t *>(h + o_need);
            for (int64_t k = 0; k < n_pairs; ++k) rescued += need[k] != 0;
        }
        ctx->timing.rescued = ctx->use_double ? n_pairs : rescued;
        return MGL_PAIRHMM_OK;
    }

    HIP_TRY(ctx, ctx->d_reads.reserve(rbytes));
    HIP_TRY(ctx, ctx->d_read_off.reserve((size_t)(n_reads + 1) * 8));
    HIP_TRY(ctx, ctx->d_haps.reserve(hbytes));
    HIP_TRY(ctx, ctx->d_hap_off.reserve((size_t)(n_haps + 1) * 8));
    HIP_TRY(ctx, ctx->d_pr.reserve((size_t)n_pairs * 4));
    HIP_TRY(ctx, ctx->d_ph.reserve((size_t)n_pairs * 4));
    HIP_TRY(ctx, ctx->d_out.reserve((size_t)n_pairs * 8));
    HIP_TRY(ctx, ctx->d_need.reserve((size_t)n_pairs * 4));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_reads.p, reads_data, rbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_read_off.p, read_off, (size_t)(n_reads + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_haps.p, haps_data, hbytes, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_hap_off.p, hap_off, (size_t)(n_haps + 1) * 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_pr.p, pair_read, (size_t)n_pairs * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_ph.p, pair_hap, (size_t)n_pairs * 4, hipMemcpyHostToDevice, st));
    const int rc = run_device(ctx, st, n_pairs, static_cast<const uint8_t *>(ctx->d_reads.p),
                              static_cast<const int64_t *>(ctx->d_read_off.p), static_cast<const uint8_t *>(ctx->d_haps.p),
                              static_cast<const int64_t *>(ctx->d_hap_off.p), static_cast<const int32_t *>(ctx->d_pr.p),
                              static_cast<const int32_t *>(ctx->d_ph.p), (int)max_r, (int)max_h,
                              static_cast<double *>(ctx->d_out.p), static_cast<int32_t *>(ctx->d_need.p));
    if (rc != MGL_PAIRHMM_OK) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    std::vector<int32_t> need((size_t)n_pairs);
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_out.p, (size_t)n_pairs * 8, hipMemcpyDeviceToHost, st));
    if (!ctx->use_double) HIP_TRY(ctx, hipMemcpyAsync(need.data(), ctx->d_need.p, (size_t)n_pairs * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    int64_t rescued = 0;
    if (!ctx->use_double)
        for (int64_t k = 0; k < n_pairs; ++k) rescued += need[(size_t)k] != 0;
    ctx->timing.rescued = ctx->use_double ? n_pairs : rescued;
    return MGL_PAIRHMM_OK;
}

int mgl_pairhmm_compute_likelihoods(mgl_pairhmm_ctx *ctx, const int32_t *lengths, const uint8_t *reads, const uint8_t *haps,
                                    double *likelihoods)
{
    if (!ctx) return MGL_PAIRHMM_ERR_BAD_ARG;
    if (!lengths || !reads || !haps || !likelihoods) return fail(ctx, MGL_PAIRHMM_ERR_BAD_ARG, "mgl_pairhmm_compute_likelihoods: null buffer");
    // …PairHmm.cc:84-118: {n_reads, read lengths, n_haps, haplotype lengths}
    const int32_t n_reads = lengths[0];
    if (n_reads < 0) return fail(ctx, MGL_PAIRHMM_ERR_BAD_ARG, "negative read count");
    const int32_t n_haps = lengths[1 + n_reads];
    if (n_haps < 0) return fail(ctx, MGL_PAIRHMM_ERR_BAD_ARG, "negative haplotype count");
    if (n_reads == 0 || n_haps == 0) return MGL_PAIRHMM_OK;
    std::vector<int64_t> read_off((size_t)n_reads + 1, 0), hap_off((size_t)n_haps + 1, 0);
    for (int32_t r = 0; r < n_reads; ++r) read_off[(size_t)r + 1] = read_off[(size_t)r] + lengths[1 + r];
    for (int32_t h = 0; h < n_haps; ++h) hap_off[(size_t)h + 1] = hap_off[(size_t)h] + lengths[2 + n_reads + h];
    const int64_t n_pairs = (int64_t)n_reads * n_haps;
    std::vector<int32_t> pr((size_t)n_pairs), ph((size_t)n_pairs);
    for (int32_t r = 0; r < n_reads; ++r)
        for (int32_t h = 0; h < n_haps; ++h) {
            pr[(size_t)r * n_haps + h] = r; // likelihoodArray[read_idx * hapCount + hap_idx], :188
            ph[(size_t)r * n_haps + h] = h;
        }
    return mgl_pairhmm_compute_pairs(ctx, n_pairs, n_reads, reads, read_off.data(), n_haps, haps, hap_off.data(), pr.data(), ph.data(),
                                     likelihoods);
}

} // extern "C"
