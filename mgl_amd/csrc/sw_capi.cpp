// sw_capi.cpp -- the C ABI of include/mgl_sw.h over the gfx950 kernels of sw_kernels.hip.
//
// Host-side work only: argument checks, the parameter normalisation of the reference's JNI
// boundary, workspace management, chunking of a batch so that its traceback fits the
// workspace, and kernel launches.  No alignment arithmetic happens on the CPU and there is
// no CPU fallback: without a HIP device every compute entry returns MGL_SW_ERR_DEVICE.
#include "../../include/mgl_sw.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <functional>
#include <future>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "sw_device.h"

using namespace mgl_sw_dev;

static_assert(sizeof(mgl_sw_score) == sizeof(Score), "mgl_sw_score layout");

// library-internal entry points shared with sw_batcher.cpp: C linkage, but not exported from the .so
#define MGL_SW_INTERNAL __attribute__((visibility("hidden")))

namespace {

constexpr int64_t kDefaultWorkspace = 4ll << 30;

// what the caller knows about the pair geometries of a batch
enum { GEOM_MIXED = 0,    // anything
       GEOM_UNIFORM = 1,  // every pair exactly max_tl x max_ql
       GEOM_GROUPED = 2 };// every aligned block of eight pairs has one (tl, ql) <= (max_tl, max_ql)

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes)
    {
        if (bytes <= cap) return hipSuccess;
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
};

} // namespace

struct mgl_sw_ctx {
    int device = 0;
    int n_cus = 256;
    hipStream_t stream = nullptr; // used by the host-buffer entry points
    int64_t ws_limit = kDefaultWorkspace;
    // kernel workspace, two halves: the traceback of chunk k (aux stream) overlaps the fill of chunk k+1
    DevBuf tb[2], rec[2], bnd[2], diag, scratch;
    // sw_dp16_lane_ck_kernel's tile counters: a ring of entries {draws, waves out}, one per launch, a 64-byte line each (zeroed once; an
    // entry comes round again after kTileCounters launches of this context, which are ordered behind each other by then: same half,
    // same stream).  The host keeps NO copy of where an entry stands: the grid's last wave out puts it back to zero, however the launch
    // ended (sw_dp16_lane_ck.hip) -- round 4's host-side copy went out of step when a gated launch was called off.
    DevBuf tile_ctr;
    uint64_t tile_seq = 0;
    int32_t *pin_fault = nullptr; // pinned host word: a wave drew a tile number no launch of its size can draw (DpArgs::grid_fault); sticky, see grid_fault_check()
    // the DIRECT form of the host entries (mgl_sw_align_batch_2bit with every array page-locked by the caller): ONE launch of the persistent
    // grid while the copy engines bring the inputs in, gated by a word in pinned host memory; the results are written by the waves straight
    // into the caller's arrays (whole lines out of LDS)
    int64_t *pin_gate = nullptr;   // [0] pairs arrived, [1] (as int32) a wave gave up waiting
    std::vector<hipEvent_t> gate_ev;
    const int64_t *cur_gate = nullptr; // set around run_device by the direct form: device view of pin_gate (null: no gate)
    int64_t *cur_gate_dev = nullptr;   // ... and the gate's mirror in device memory (DpArgs::gate_dev: two words behind d_any's status word, zeroed with it)
    bool direct_broken = false;    // a gate once timed out on this context: the direct form is not tried again
    int32_t *direct_status_any = nullptr; // ... and the device word that collects the largest per-pair status of its launch
    void *pin_tiles = nullptr;          // run_shared_target: the tiles' geometries on their way in, their order and the slots' regions on their way out (page-locked)
    size_t pin_tiles_cap = 0;
    // small-batch entry of the coalescing front-end: one pinned host buffer each way, mirrored on the device
    void *pin_in = nullptr, *pin_out = nullptr;
    size_t pin_in_cap = 0, pin_out_cap = 0;
    DevBuf stage_in, stage_out;
    // host entry, batches of mixed geometries: per workspace half the chunk's index arrays (start / length / caller index per
    // slot, sorted by geometry), built in pinned memory and mirrored on the device
    // host-buffer entry: results travel device -> pinned ring (asynchronous, own stream) -> the caller's arrays (helper jobs)
    void *pin_res[3] = {nullptr, nullptr, nullptr};
    size_t pin_res_cap[3] = {0, 0, 0};
    hipEvent_t res_copied[3] = {nullptr, nullptr, nullptr};
    hipStream_t d2h = nullptr;
    hipStream_t h2d_hi = nullptr, d2h_hi = nullptr; // the same two at the highest priority (queues of their own): mgl_sw_align_batch_2bit swaps them in
    void *pin_grp[2] = {nullptr, nullptr};
    size_t pin_grp_cap[2] = {0, 0};
    DevBuf d_grp[2];
    DevBuf d_srt[4];               // device-side sort by geometry: the index arrays of four consecutive chunks, so that the sort
    hipEvent_t srt_free[4] = {nullptr, nullptr, nullptr, nullptr}; // of chunk k waits for the walk of chunk k-4, long done, not k-2
    hipEvent_t srt_done[4] = {nullptr, nullptr, nullptr, nullptr}; // the sort of chunk k has finished (the host reads its block total)
    DevBuf d_grid;                 // ... four ints per cell of the (tl, ql) grid + the block totals
    int64_t *pin_total = nullptr;  // pinned word the block total is read back through
    hipEvent_t grp_copied[2] = {nullptr, nullptr};
    // substitution matrix + code table: copied here first, so the caller's buffers may go away when the call returns
    void *pin_matrix = nullptr;
    hipEvent_t matrix_copied = nullptr;
    int carry_memory = 0; // 0 = LDS when it fits, 1 = always the HBM scratch (tests)
    int stripe_rows = 0;  // 0 = choose per batch, 16 / 64 = force (tests)
    int cooperative = 0;  // 0 = choose per batch, 1 = never, 2..16 = always, that many waves per pair (tests)
    int strip_kernel = 0; // long reads, one strip per lane-half (sw_dp16_strip.hip): 0 = by size, 1 = never, 2 = whenever eligible (tests)
    int lane_kernel = 0;  // two-pairs-per-lane packed kernel: 0 = large uniform batches, 1 = never, 2 = whenever eligible (tests)
    int small_kernel = 0; // small batches, one wave per pair in one launch (sw_small.hip): 0 = by size on an unforced context, 1 = never, 2 = whenever the bounds allow
    int lane_checkpoint = 0; // ... in its checkpointed form (sw_dp16_lane_ck.hip, no stored traceback): 0 = by default, 1 = never, 2 = always
    int last_rows = 16;
    hipStream_t aux = nullptr;                       // traceback stream
    hipStream_t fill2 = nullptr;                     // host-buffer entry, lane kernel: odd chunks (their tails overlap the next chunk)
    hipStream_t h2d = nullptr;                       // host-buffer entry: input copies of the next chunk
    hipEvent_t in_done = nullptr;
    hipEvent_t out_ready[2] = {nullptr, nullptr};    // host-buffer entry: traceback of chunk k done (k & 1)
    hipEvent_t fill_done[2] = {nullptr, nullptr};    // fill of the chunk in half h finished (caller's stream)
    hipEvent_t tb_done[2] = {nullptr, nullptr};      // traceback of the chunk in half h finished (aux stream)
    int last_half = 0;
    hipEvent_t ws_idle = nullptr;   // recorded when a call's last kernel has been enqueued: the next call (possibly on
    bool ws_idle_set = false;       // another stream) waits for it before it reuses the workspace
    // profiling: event pairs around every fill (caller's stream) and traceback (aux or caller's stream) launch of
    // the last call; read back lazily by mgl_sw_ctx_get_timing so that the run itself is not serialised
    std::vector<hipEvent_t> pool;
    int pool_used = 0;
    int64_t diag_blocks = 0;
    DevBuf d_t, d_toff, d_q, d_qoff, d_off, d_score, d_cig, d_len, d_status, d_btr, d_any, d_matrix, d_tlen, d_qlen; // host-API staging
    int64_t last_stride_words = 0, last_chunk_count = 0; // geometry of the last chunk (for expand_slot)
    int last_packed16 = 0;
    int precision = 0; // 0 = choose per batch, 32 = always the int32 kernels, 16 = try the self-checking 16-bit long-read kernel whenever its constants fit
    int profiling = 0; // 0 off, 1 per-kernel HIP events, 2 also the in-kernel clock probe
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    mgl_sw_timing timing{};
    // host arrays the caller has page-locked for this context (mgl_sw_register_host_buffer): copies from / into them are true
    // asynchronous DMA, so the host entry neither blocks in them nor stages results through its own pinned ring
    std::vector<std::pair<const char *, size_t>> registered;
    bool is_registered(const void *p, size_t bytes) const
    {
        const char *c = static_cast<const char *>(p);
        for (const auto &r : registered)
            if (c >= r.first && c + bytes <= r.first + r.second) return true;
        // ... or page-locked by the caller's own means (hipHostMalloc, a framework's pinned allocator): both ends in pinned host memory
        if (bytes == 0) return false;
        hipPointerAttribute_t a0{}, a1{};
        if (hipPointerGetAttributes(&a0, c) != hipSuccess || hipPointerGetAttributes(&a1, c + bytes - 1) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost;
    }
    std::string err;
    std::mutex mu;
};

namespace {

// The only environment switches the planner reads: DEBUG aids that turn a code path off so that tests and measurements can compare
// against the path it replaced (INTEGRATION.md 7 lists them with the operational settings of the front-end).  Read once per process,
// except MGL_SW_DEBUG_LANE_GROUP_MIN (per chunk: the host-sanitizer driver changes it between calls).
struct DebugKnobs {
    bool auto_group;  // MGL_SW_DEBUG_AUTO_GROUP=0: batches of mixed geometries are not sorted by the library (they take the int32 kernel)
    bool lane_ck;     // MGL_SW_DEBUG_LANE_CK=0: the lane kernel keeps the flags of every cell (sw_dp16_lane_kernel) instead of checkpoints
    bool lane_group;  // MGL_SW_DEBUG_LANE_GROUP=0: sorted chunks never launch the lane kernel for their whole waves of one geometry
    bool host_timing; // MGL_SW_DEBUG_HOST_TIMING=1: the host entry reports where the calling thread waited (stderr)
};
const DebugKnobs &debug_knobs()
{
    static const DebugKnobs k = [] {
        auto on = [](const char *name, bool dflt) { const char *e = getenv(name); return e ? atoi(e) != 0 : dflt; };
        return DebugKnobs{on("MGL_SW_DEBUG_AUTO_GROUP", true), on("MGL_SW_DEBUG_LANE_CK", true), on("MGL_SW_DEBUG_LANE_GROUP", true),
                          on("MGL_SW_DEBUG_HOST_TIMING", false)};
    }();
    return k;
}
constexpr int kTileCounters = 64;
constexpr int kTileCounterWords = 16; // an entry's two words (ONE 64-bit atomic object to the kernel) on a 64-byte line of their own
constexpr int kHostChunks = 32; // a host entry cuts a batch into about this many chunks (the units of its copy / compute pipeline)

int geom_of(int flags)
{
    return (flags & MGL_SW_FLAG_UNIFORM_GEOMETRY) ? GEOM_UNIFORM : (flags & (MGL_SW_FLAG_GROUPED_GEOMETRY | MGL_SW_FLAG_SHARED_TARGET)) ? GEOM_GROUPED : GEOM_MIXED; // (tiles of 128 that share a target are blocks of eight of one geometry)
}

int fail(mgl_sw_ctx *ctx, int status, const std::string &what)
{
    if (ctx) ctx->err = what;
    return status;
}

int hip_fail(mgl_sw_ctx *ctx, hipError_t e, const char *where)
{
    std::string msg = std::string(where) + ": " + hipGetErrorString(e);
    (void)hipGetLastError();
    return fail(ctx, e == hipErrorOutOfMemory ? MGL_SW_ERR_NOMEM : MGL_SW_ERR_DEVICE, msg);
}

#define HIP_TRY(ctx, call)                                     \
    do {                                                       \
        hipError_t e_ = (call);                                \
        if (e_ != hipSuccess) return hip_fail(ctx, e_, #call); \
    } while (0)

// A persistent grid whose wave drew a tile number that no launch of its size can draw has left tiles undone (its counter did not stand
// at zero: sw_dp16_lane_ck.hip).  That cannot happen by anything the library does -- which is exactly why it is checked: the flag is
// sticky, every later call on the context and mgl_sw_ctx_check() report it, no result after it is handed out as good.
int grid_fault_check(mgl_sw_ctx *ctx)
{
    if (ctx->pin_fault && __atomic_load_n(ctx->pin_fault, __ATOMIC_ACQUIRE) != 0)
        return fail(ctx, MGL_SW_ERR_DEVICE, "a persistent grid found its tile counter out of range: results of this context since its last good check are not valid");
    return MGL_SW_OK;
}

bool strategy_ok(int s)
{
    return s == MGL_SW_OS_SOFTCLIP || s == MGL_SW_OS_INDEL || s == MGL_SW_OS_LEAD_ID || s == MGL_SW_OS_IGNORE;
}

// waves per block (4 .. 1) that puts the most waves on a CU's 160 KB of LDS with a block within 64 KB; 0 if even one wave
// does not fit.  `extra`: bytes every block adds (the substitution matrix)
int pick_waves_per_block(int sps_cap, int rows, int extra = 0)
{
    // a wave whose carve exceeds 40 KiB would leave fewer than four waves per CU: such queries keep their
    // carry in the HBM scratch instead (return 0)
    if (dp_lds_bytes(sps_cap, 1, rows) > 40 * 1024) return 0;
    int best = -1, pick = 1;
    for (int w = 4; w >= 1; --w) {
        const int lds = dp_lds_bytes(sps_cap, w, rows) + extra;
        if (lds > 64 * 1024) continue;
        const int occ = (160 * 1024 / lds) * w;
        if (best < 0 || occ * 10 > best * 11) { // fewer waves per block only for a real gain (> 10 %): 17 one-wave blocks
            best = occ;                         // per CU instead of 4 x 4 waves measured 7 % slower on 150-base reads
            pick = w;
        }
    }
    return pick;
}

// queries from this length on run one pair per wave (64-row stripes): the pipeline fill/drain is then
// 63/(ql+64) <= 6 %, and a batch needs four times fewer pairs to occupy the machine
constexpr int kRows64MinQuery = 1024;
// uniform batches (launches) from this size on take the two-pairs-per-lane kernel: 640 waves of 128 pairs (measured crossover with
// the eight-pairs-per-wave kernel at 256 x 150, scripts/kernel_crossover.py, round 3: 32 768 pairs 1 243 against 1 948 GCUPS,
// 65 536 pairs 2 442 against 2 425, 131 072 pairs 4 342 against 2 731, 262 144 pairs 4 931 against 2 930; in round 2 it crossed at
// 262 144, with the flags of every cell stored at 524 288)
constexpr int64_t kLaneMinPairs = 128 * 640;
// ... in the form that stores the flags of every cell (three waves per SIMD: a round of the chip is 393 216 pairs) from 524 288 on (round 2's
// measurement); it is what a batch gets whose workspace cannot hold the checkpointed form's regions, or that asks for scores only
constexpr int64_t kLaneStoredMinPairs = 524288;
// ... and a sorted chunk's whole waves of one geometry get a launch of their own from this many pairs on
constexpr int64_t kLaneGroupMinPairs = 128 * 1024;
constexpr int64_t kSideReserve = 16; // wave slots a sorted chunk's persistent grid leaves to the kernels of its left-over pairs (run_device)

int max_lds_query_len()
{
    // largest ql whose one-wave carve fits the 160 KiB LDS (longer queries use the HBM scratch)
    int lo = 1, hi = 1 << 20;
    while (lo < hi) {
        int mid = (lo + hi + 1) / 2;
        if (dp_lds_bytes(sps_for_rows(mid, 64), 1, 64) <= 40 * 1024)
            lo = mid;
        else
            hi = mid - 1;
    }
    return lo;
}

// Host-buffer entry only: per-chunk hooks that move a chunk's inputs in before its fill is launched and its
// results out after the NEXT chunk has been launched (pageable copies block the calling thread, so this order
// is what lets the copies of one chunk overlap the kernels of its neighbours).
// What the host entry hands back for a chunk of a batch of mixed geometries that it has sorted by (tl, ql) itself: the chunk's
// pairs in the new order as (start, length) index arrays over the caller's bases, the caller's pair index of every
// slot, and how many leading slots form full blocks of eight with one geometry each (the packed kernel's food).
struct Regroup {
    const int64_t *d_t_start = nullptr, *d_q_start = nullptr, *d_dest = nullptr;
    const int32_t *d_t_len = nullptr, *d_q_len = nullptr;
    int64_t n_grouped = 0;
    bool lane_blocks = false; // in: a geometry's full blocks of 128 pairs first (whole waves of the checkpointed lane kernel) ...
    int64_t n_lane = 0;       // out: ... slots [0, n_lane)
};
struct ChunkHooks {
    std::function<int(int64_t first, int64_t count, hipStream_t fill_stream)> before_fill;
    std::function<int(int64_t first, int64_t count, hipEvent_t results_ready)> after_traceback;
    // optional: sort the chunk by geometry (slot buffers `half` are free to overwrite when this is called)
    std::function<int(int64_t first, int64_t count, int half, hipStream_t fill_stream, Regroup *out)> regroup;
    // optional (round 5), instead of `regroup`: the chunks of a batch of mixed geometries are sorted ON THE DEVICE (launch_regroup, the
    // device-resident entries' counting sort), two chunks ahead of the fills, and bring_ahead(first, count) enqueues that chunk's inputs
    // on ctx->h2d in front of its sort -- the host never looks at a pair (a host-side sort of 4 M pairs took longer than their
    // alignment: 86.8 ms per call against 23.4 device resident)
    std::function<int(int64_t first, int64_t count)> bring_ahead;
    int32_t *d_status_any = nullptr;   // device word receiving the largest per-pair status
};

// What the planner decides about a batch: which kernel, in what shape, in what chunks.  plan_batch() is PURE -- the context's settings and the
// batch's description in, no HIP call, nothing allocated -- so mgl_sw_explain() can report a plan without a GPU and tests/test_plan.py can pin it;
// run_device() executes it.
struct BatchPlan {
    bool use_lane; // the two-pairs-per-lane kernels (sw_dp16_lane*.hip) ...
    bool lane_ck; // ... in the checkpointed form (no traceback stored, the walk inside the fill kernel)
    bool use16; // the eight-pairs-per-wave packed kernel (sw_dp16.hip)
    bool strip16; // long reads: one strip per lane-half (sw_dp16_strip.hip) ...
    bool coop16; // long reads: one pair per workgroup, packed int16 with a checked window
    int coop_waves; // waves per pair of the workgroup kernels (0: not those)
    int strip_waves; // ... its waves per pair
    int strip_k; // ... strips per kept band (0: every flag stored)
    int strip_passes; // ... passes of 128 W strips over the target (targets beyond 16 384 rows)
    bool auto_group; // a batch of mixed geometries whose chunks the library sorts by (tl, ql)
    bool lane_group; // ... the geometries with whole waves of 128 pairs through the checkpointed lane kernel
    int64_t lane_group_stride; // ... words per wave of that part
    bool score_only; // no traceback, no CIGARs
    bool fused_walk; // the fill kernel walks its own paths (no traceback kernel)
    bool use_scratch; // long queries of the int32 kernel: carry ring and query copies in HBM
    int rows; // target rows per stripe / strip
    int wpb; // waves per workgroup
    int wpb16; // ... of the packed kernel
    int sps_cap; // steps per stripe the LDS carve is sized for
    int sps32; // ... of the int32 part of a sorted chunk
    int64_t stride_words; // traceback words per pair / per two pairs / per wave, by layout
    int64_t stride32_words; // ... of the int32 part of a sorted chunk
    int64_t per_pair; // workspace bytes per pair
    int64_t lane_slots; // the checkpointed lane kernel's persistent grid: wave slots of a launch (= regions of a workspace half)
    int64_t fixed_bytes; // ... the bytes of those regions: workspace of a half that does not grow with the chunk
    bool group_regions;  // a sorted device-resident chunk sized by slots: the lane part's regions, then left_area bytes for the left-over pairs
    int64_t left_area, left_pair16, left_pair32; // ... that area (its first half: the packed kernel's pieces, its second: the int32 kernel's) and a pair's bytes in each
    int64_t chunk; // pairs per chunk (the largest, where the chunks grow and shrink)
    bool dev_sort_host; // a host batch of mixed geometries whose chunks the device sorts (hooks->bring_ahead): a short first and last chunk, two fill streams
    bool pyramid; // the host entry of a large 2-bit batch: chunks of 1, 2, 4, 8 .. 8, 4, 2, 1 rounds of the chip
    int64_t pyr_unit; // ... pairs per round
    bool overlap; // the traceback of chunk k runs beside the fill of chunk k + 1 (two workspace halves)
    bool dual; // consecutive chunks alternate between two fill streams (two workspace halves)
    int halves; // workspace halves in use
    int geom; // the geometry class the batch is planned as (a promise of blocks of eight may be set aside: plan_batch)
};
static int plan_batch_as(mgl_sw_ctx *ctx, int64_t n, const SeqSet &tset, const SeqSet &qset, int max_tl, int max_ql, int match, int mismatch, int gopen, int gext,
                         int strategy, const Score *d_score, const char *d_cigar, int cigar_stride, int geom, bool binary_cigar, const ChunkHooks *hooks,
                         const int8_t *d_matrix, bool score_only_hint, BatchPlan &P)
{
    const bool uniform = geom != GEOM_MIXED;
    (void)cigar_stride;
    (void)binary_cigar;
    // packed-int16 kernel: one geometry per batch (or per block of eight pairs) and a score range that fits 16 bits;
    // four waves per block while their LDS carve fits, else two or one
    const int lds_extra = d_matrix ? MATRIX_DIM * MATRIX_DIM * 2 : 0;
    // waves per block of the packed kernel: the count (4 .. 1) that puts the most waves on a CU's 160 KB of LDS, a block
    // staying within 64 KB (bench batch: 4 x 9.5 KB -> 16 waves per CU either way; 300-residue protein queries: three
    // waves of 16.7 KB + the table -> 9 waves per CU instead of 8, +17 % measured)
    int wpb16 = 1;
    {
        const int per_wave = dp16_lds_bytes(sps_for(max_ql), 1);
        int best = -1;
        for (int w = 4; w >= 1; --w) {
            const int lds = w * per_wave + lds_extra;
            if (lds > 64 * 1024) continue;
            const int occ = (160 * 1024 / lds) * w;
            if (best < 0 || occ * 10 > best * 11) { // fewer waves per block only for a real gain (> 10 %)
                best = occ;
                wpb16 = w;
            }
        }
    }
    while (wpb16 > 1 && dp16_lds_bytes(sps_for(max_ql), wpb16) + lds_extra > 64 * 1024) --wpb16;
    const bool use16_eligible = uniform && ctx->precision != 32 && dp16_lds_bytes(sps_for(max_ql), wpb16) + lds_extra <= 64 * 1024 &&
                       match > 0 && dp16_range_ok(max_tl, max_ql, match, mismatch, gopen, gext, strategy);
    // two pairs per LANE (sw_dp16_lane.hip): uniform ASCII batches with enough pairs to give every lane its own --
    // 128 pairs per wave, so a chunk should hold a few thousand waves; smaller batches fill the chip better with the
    // eight-pairs-per-wave kernel above
    // strips of 32 rows (three waves per SIMD) unless 16-row strips (four waves, twice the carry traffic) save at least a
    // tenth of the rows: strips are whole, rows past tl are computed and thrown away
    const int lane_rows = ((max_tl + 31) / 32 * 32 - (max_tl + 15) / 16 * 16) * 10 >= max_tl ? 16 : 32;
    // (what counts is the size of a launch: a batch that the workspace cuts into small chunks is no better than a small batch.
    // Behind the host-buffer entry the chunks are the units of the copy pipeline: there a large batch goes one ROUND of the
    // chip at a time -- 128 pairs per wave, three (32-row strips) or four waves per SIMD -- on two alternating streams)
    // (the checkpointed form runs two waves per SIMD, the forms that store every flag three (32-row strips) or four)
    const bool lane_ck_on = debug_knobs().lane_ck;
    // The checkpointed form is a PERSISTENT grid (sw_dp16_lane_ck.hip): a launch keeps one region per wave SLOT -- at most the waves the
    // chip holds at two per SIMD, 4 GB on an MI355X at 256 x 150 -- and 32 bytes of record per pair, however many pairs it holds.  The
    // regions take at most three quarters of a workspace part (host entries run two launches side by side: two parts); a workspace that
    // cannot hold one wave per SIMD leaves the batch to the kernels that store their flags.
    const char *const slots_env = getenv("MGL_SW_DEBUG_LANE_SLOTS"); // (measurements: a grid of this many wave slots instead of what the chip holds; read per call)
    const int64_t ck_region = lane_ck_region_bytes(max_tl, max_ql), ck_chip = slots_env && atoll(slots_env) > 0 ? atoll(slots_env) : (int64_t)ctx->n_cus * LANE_CK_WAVES_PER_CU;
    const int64_t ck_part = hooks ? ctx->ws_limit / 2 : ctx->ws_limit;
    const int64_t ck_slots_max = std::min<int64_t>(ck_chip, ck_part / 4 * 3 / ck_region);
    const bool ck_fits = ck_slots_max >= ((ctx->lane_kernel == 2 || ctx->lane_checkpoint == 2) ? 1 : std::min<int64_t>(ck_chip, (int64_t)ctx->n_cus * LANE_CK_WAVES_PER_CU) / 2); // (forced by a test: any number of slots)
    const int64_t ck_launch_max = ck_fits ? (ck_part - ck_slots_max * ck_region) / (int64_t)sizeof(DpRecord) / 128 * 128 : 0;
    const bool lane_ck_ok = lane_rows == 32 && d_cigar != nullptr && ctx->lane_checkpoint != 1 && (ctx->lane_checkpoint == 2 || lane_ck_on) && ck_fits;
    // (a host entry's chunk: one round where the inputs are ASCII -- 0.1 GB per round over the link before the first kernel can
    // start -- two where they are 2-bit packed: 10 M pairs 69.1 ms against 72.4 with one, 72.3 with four; ASCII: 83.1 / 84.5 / 87.2)
    const int64_t lane_round = (int64_t)ctx->n_cus * (lane_ck_ok ? 8 : lane_rows == 16 ? 16 : 12) * 128 * (hooks && tset.packed2 ? 2 : 1);
    constexpr int host_chunks_l = kHostChunks;
    const bool lane_rounds = hooks && n >= 4 * lane_round;
    const int64_t lane_launch = std::min<int64_t>(lane_rounds ? n : hooks ? std::max<int64_t>(n / host_chunks_l, (int64_t)256 * 1024) : n,
                                                  std::min<int64_t>(n, lane_ck_ok ? ck_launch_max : ctx->ws_limit / (lane_tb_words(max_tl, max_ql, lane_rows) * 4 / 128 + 1)));
    // (its checkpointed form stages base codes and takes either wire format; the form that stores every flag reads ASCII only)
    const bool lane_ck_wanted = lane_ck_ok && !score_only_hint; // (a 2-bit batch gets the lane kernel only in this form)
    const bool use_lane = geom == GEOM_UNIFORM && ctx->precision != 32 && !d_matrix && match > 0 && ctx->lane_kernel != 1 &&
                          (ctx->lane_kernel == 2 || lane_launch >= (hooks ? 2 : 1) * (lane_ck_wanted ? kLaneMinPairs : kLaneStoredMinPairs)) && // (the host entry's chunks: as measured before)
                          (lane16_supported(tset, qset) || (lane_ck_wanted && lane16_ck_supported(tset, qset))) &&
                          dp16_range_ok(max_tl, max_ql, match, mismatch, gopen, gext, strategy);
    // a batch of mixed geometries whose chunks the host entry sorts by geometry (hooks->regroup): full blocks of eight pairs
    // with one geometry go through the packed kernel, the few left over through the int32 kernel, results land in the
    // caller's order (TbArgs.dest) -- the reference takes any pair (sw_avx.cpp:6-108), so must the fast path
    // ... and a device-resident batch is sorted on the device (launch_regroup), at the price of one short synchronisation per
    // chunk: the host has to know how many pairs landed in full blocks before it can size the two launches
    const bool auto_group_dev_on = debug_knobs().auto_group;
    // (the sort scans the whole (tl, ql) grid per chunk: worth it only where the pairs outnumber an eighth of its cells -- a few
    // thousand short pairs under a large bound would pay milliseconds for it)
    const bool hooks_dev_sort = hooks && hooks->bring_ahead && !hooks->regroup; // (a host entry whose chunks the DEVICE sorts)
    const bool regroup_dev = (!hooks || hooks_dev_sort) && auto_group_dev_on && n >= 1024 && (int64_t)max_tl * max_ql <= (1ll << 20) && n * 8 >= (int64_t)max_tl * max_ql && !score_only_hint &&
                             (tset.len != nullptr || !tset.packed2) && (qset.len != nullptr || !qset.packed2);
    const bool auto_group = geom == GEOM_MIXED && ((hooks && hooks->regroup) || regroup_dev) && ctx->precision != 32 && !d_matrix && match > 0 &&
                            !ctx->stripe_rows && ctx->cooperative < 2 && ctx->carry_memory == 0 && max_ql < kRows64MinQuery &&
                            dp16_lds_bytes(sps_for(max_ql), wpb16) <= 64 * 1024 && pick_waves_per_block(sps_for_rows(max_ql, 16), 16) > 0 &&
                            dp16_range_ok(max_tl, max_ql, match, mismatch, gopen, gext, strategy);
    // ... of which the geometries with 128 pairs and more go, whole waves of one geometry each, through the checkpointed lane kernel
    const bool lane_group_on = debug_knobs().lane_group;
    const bool lane_group = auto_group && lane_group_on && lane_ck_on && ctx->lane_checkpoint != 1 && ctx->lane_kernel != 1 && lane16_ck_supported(tset, qset);
    const int64_t lane_group_stride = lane_group ? lane_ck_words(max_tl, max_ql) : 0; // words per wave
    const bool use16 = (use16_eligible && !use_lane) || auto_group;
    // MGL_SW_FLAG_SCORE_ONLY is honoured by the packed kernels only; elsewhere the full path runs (a superset of the result)
    const bool score_only = score_only_hint && (use16 || use_lane) && d_score != nullptr && !hooks;
    int rows = use_lane ? lane_rows : use16 ? 16 : ctx->stripe_rows ? ctx->stripe_rows : (max_ql >= kRows64MinQuery ? 64 : 16);
    // substitution-matrix mode: 16 rows x four pairs per wave while that carve fits LDS (queries up to ~800 residues;
    // measured faster than one pair per wave at 300 residues: 1 056 vs about 1 000 GCUPS), else 64 rows x one pair (to ~3 300)
    if (d_matrix && !use16 && !ctx->stripe_rows) rows = pick_waves_per_block(sps_for_rows(max_ql, 16), 16) == 0 ? 64 : 16;
    int sps_cap = use_lane ? max_ql : sps_for_rows(max_ql, rows);
    int wpb = use_lane ? 4 : use16 ? wpb16 : pick_waves_per_block(sps_cap, rows, d_matrix ? 1024 : 0);
    // long reads: one pair per WORKGROUP (sw_dp_coop_kernel), its waves pipelined over the 64-row stripes.  Taken
    // when the one-wave-per-pair carve does not fit LDS, or when forced; needs at least two stripes to share.
    int coop_waves = 0;
    bool coop16 = false;
    if (!use16 && !use_lane && !d_matrix && ctx->cooperative != 1 && ctx->carry_memory == 0 && gopen < 65536 && gopen >= gext && (ctx->stripe_rows == 0 || ctx->cooperative >= 2) &&
        ((rows == 64 && wpb == 0) || ctx->cooperative >= 2) && coop_lds_bytes(coop_sps_for(max_ql), 2) <= 160 * 1024) {
        const int stripes = (max_tl + 63) / 64;
        coop_waves = ctx->cooperative >= 2 ? ctx->cooperative : 16;
        while (coop_waves > 2 && (coop_waves > stripes || coop_lds_bytes(coop_sps_for(max_ql), coop_waves) > 160 * 1024)) --coop_waves;
        rows = 64;
        sps_cap = coop_sps_for(max_ql);
        // packed int16 form (sw_dp_coop16_kernel): 128 rows per wave; it checks its own score window and redoes a pair in
        // 32 bits when that fails, so what is decided here is only whether trying is worthwhile
        if (ctx->precision != 32 && (ctx->precision == 16 ? coop16_possible(match, mismatch, gopen, gext) : coop16_worthwhile(match, mismatch, gopen, gext))) {
            const int sps16 = coop16_sps_for(max_ql), dstripes = (max_tl + 127) / 128;
            int w16 = coop_waves;
            while (w16 > 2 && (w16 > dstripes || coop16_lds_bytes(sps16, w16) > 160 * 1024)) --w16;
            if (coop16_lds_bytes(sps16, w16) <= 160 * 1024) {
                coop16 = true;
                coop_waves = w16;
                sps_cap = sps16;
            }
        }
        wpb = coop_waves;
    }
    // long reads whose targets span a few thousand rows: one strip of 32 rows per lane-half, the lane kernel's cell code with
    // per-strip baselines (sw_dp16_strip.hip); W waves per pair hold 128 W strips, W <= 4 keeps three workgroups' worth of
    // registers per SIMD
    bool strip16 = false;
    int strip_waves = 0, strip_k = 0, strip_passes = 1; // strip_k: strips per kept band of the form without stored flags (0: flags stored); strip_passes: passes of 128 W strips over the target
    {
        // waves per pair: as few as hold the target in strips of 32 rows -- but never three: workgroups of three waves run a
        // quarter slower than those of one, two or four (pairs of 8 / 10 / 12 kb with two or four waves: 3.09 / 2.69 / 2.99
        // TCUPS, with three: 2.40 / 2.43 / 2.58; three waves do not spread evenly over a CU's four SIMDs)
        int sw_ = ((max_tl + 31) / 32 + 127) / 128;
        if (sw_ == 3) sw_ = 4;
        // targets beyond the 512 strips of four waves (16 384 rows): several passes of 512 strips each (the kernels without stored flags,
        // round 4: a later pass takes the row above its first strip from what the pass before kept); before: the workgroup kernels, flags stored
        int sp_ = 1;
        if (sw_ > 4) {
            sp_ = ((max_tl + 31) / 32 + 511) / 512;
            sw_ = 4;
        }
        // worth it when enough of the issued lanes are real cells: strips of the 128 W slots x useful steps of all steps
        // rows per strip: as few as still cover the longest target with these waves (fewer rows = fewer instructions per column)
        // (several passes: 22 rows at least, so that a pass's 512 strips are whole bands of K = 2)
        int sr_ = 32;
        for (int cand = 31; cand >= (sp_ > 1 ? 22 : 17); --cand)
            if ((max_tl + cand - 1) / cand <= 128 * sw_ * sp_) sr_ = cand;
        const double used = (double)((max_tl + sr_ - 1) / sr_) / (128.0 * sw_ * sp_) * strip16_groups(max_ql) / (double)strip16_steps(max_ql, sw_);
        // (measured against the kernels it replaces, pairs of n x n: 1.5 kb 1 354 against 1 124 GCUPS at used = 0.44; 2 kb 1 844 / 1 014;
        // 3 kb 2 474 / 870; 4 kb 2 987 / 1 385; 10 kb 2 480 / 1 824)
        const bool want = ctx->strip_kernel == 2 || (ctx->strip_kernel == 0 && (coop_waves || rows == 64) && ctx->cooperative < 2 && used >= 0.4);
        // (its time-major traceback regions are larger than the workgroup kernel's -- 97 MB against 50 for a 10 kb pair: a workspace
        // that cannot hold one of them per half keeps the workgroup kernel)
        // (without stored flags -- rows of every band of K strips and column checkpoints instead, walked by sw_strip_ck_walk_kernel --
        // a pair takes 8 MB (16 where the entries do not pack): that form is the default where CIGARs are written; mgl_sw_ctx_set_lane_checkpoint(ctx, 1) keeps the flags)
        const int sk_ = (ctx->lane_checkpoint != 1 && d_cigar != nullptr && !score_only_hint) ? 64 / sr_ : 0;
        const bool fits = (sk_ ? strip16_ck_words(max_tl, max_ql, sr_, sk_) : tb_words_strip16(max_ql, sw_)) * 4 + (int64_t)sizeof(DpRecord) <= ctx->ws_limit / 2;
        if (want && fits && !use16 && !use_lane && !d_matrix && ctx->precision != 32 && ctx->carry_memory == 0 && !ctx->stripe_rows && sw_ <= 4 && (sp_ == 1 || sk_ > 0) &&
            strip16_lds_bytes(max_ql, sw_) <= 64 * 1024 && strip16_range_ok(match, mismatch, gopen, gext)) {
            strip16 = true;
            strip_k = sk_;
            strip_passes = sp_;
            strip_waves = sw_;
            coop16 = false;
            coop_waves = 0;
            rows = sr_;
            wpb = sw_;
            sps_cap = strip16_steps(max_ql, sw_);
        }
    }
    // queries too long for the LDS carve: carry ring and query copies in an HBM scratch area instead
    while (d_matrix && !use16 && wpb > 1 && dp_lds_bytes(sps_cap, wpb, rows) + 1024 > 64 * 1024) --wpb; // room for the matrix
    if (d_matrix && !use16 && (wpb == 0 || dp_lds_bytes(sps_cap, wpb, rows) + 1024 > 64 * 1024))
        return fail(ctx, MGL_SW_ERR_UNSUPPORTED, "substitution-matrix scoring: query too long for the LDS carve (about 3 300 residues)");
    const bool use_scratch = !use16 && !use_lane && !coop_waves && !strip16 && !d_matrix && (wpb == 0 || ctx->carry_memory == 1);
    if (use_scratch) wpb = 4;
    if ((int64_t)max_tl * max_ql > (1ll << 34) || max_ql > (1 << 24) || max_tl > (1 << 24))
        return fail(ctx, MGL_SW_ERR_UNSUPPORTED, "matrix larger than 2^34 cells");
    // the kernels hold X + (i + j) * gext in 32 bits; the reference's own int arithmetic overflows beyond this too
    if (((int64_t)match - mismatch + gopen + 2 * (int64_t)gext) * ((int64_t)max_tl + max_ql) >= (1ll << 30))
        return fail(ctx, MGL_SW_ERR_UNSUPPORTED, "scores of this geometry and these parameters leave the 32-bit range");
    // the lane kernel without stored flags (sw_dp16_lane_ck.hip): its walk recomputes the blocks the path crosses; strips of 32 rows,
    // fused walk only (a caller who wants the matrix itself -- mgl_sw_ctx_expand_slot -- switches it off)
    const bool lane_ck = use_lane && !score_only && lane_ck_ok;
    // traceback words per pair (int32 layout) or per group of two pairs (packed16 layout)
    // (packed layout: the step count of a pair is not monotone in tl or ql -- a partial last stripe runs stand-alone, short
    // queries are not chained -- so a grouped batch, whose waves each run their own geometry, is sized by a bound that is)
    // (lane layout: words per WAVE of 128 pairs, plus the wave's carry row)
    const int64_t stride_words = lane_ck ? lane_ck_words(max_tl, max_ql)
                                 : use_lane ? lane_tb_words(max_tl, max_ql, rows)
                                 : use16 ? (geom == GEOM_UNIFORM ? tb_words16_for(max_tl, max_ql) : tb_words16_bound(max_tl, max_ql))
                                 : strip16 ? (strip_k ? strip16_ck_words(max_tl, max_ql, rows, strip_k) : tb_words_strip16(max_ql, strip_waves))
                                 : coop16 ? std::max(tb_words_for(max_tl, coop_sps_for(max_ql), 64), tb_words_coop16(max_tl, max_ql)) // either layout
                                         : tb_words_for(max_tl, sps_cap, rows);
    // (auto-grouped chunks: the packed regions first, the int32 regions of the left-over pairs behind them)
    const int sps32 = sps_for_rows(max_ql, 16);
    const int64_t stride32_words = tb_words_for(max_tl, sps32, 16);
    // A device-resident batch the library sorts itself (round 4): the whole waves of one geometry go through the lane kernel's PERSISTENT
    // grid -- one region per wave slot, however many pairs -- and what is left over (fewer than 128 pairs per geometry, whatever the
    // data) through the packed and int32 kernels in an area of its own, in as many pieces as that area needs.  A chunk is therefore
    // sized by its records alone: 4 M reads of 100-150 bases are ONE chunk in an 8 GiB workspace (round 3 sized every pair for the
    // left-over kernels' traceback, 19 KB: 21 chunks in 8 GiB -- 3 125 GCUPS --, 3 in 72 GiB -- 4 850 --, one in 170 GiB -- 5 320).
    const int64_t left_pair = std::max(stride_words * 2, stride32_words * 4); // bytes of traceback per left-over pair, either kernel
    // (A host batch sorted on the device was also run with consecutive chunks on two streams and two halves of the workspace, so that one
    // grid's end lay beside the next grid's start: 32.8 ms per 4 M mixed reads against 33.1 on one stream, for twice the regions -- what
    // the call waited for was the host's own preparation in front of the first launch, not the grids' ends.  One stream, one set of regions.)
    const int64_t grp_ws = ctx->ws_limit;
    const int64_t grp_slots = std::min<int64_t>(ck_chip, grp_ws / 4 * 3 / ck_region);
    const char *const gre = getenv("MGL_SW_DEBUG_GROUP_REGIONS"); // (0: sorted chunks sized per pair, as before round 4; read per call: the tests compare both)
    const char *const lae = getenv("MGL_SW_DEBUG_LEFT_AREA");     // (bytes of the left-over pairs' area: tests make it small, so that the left-overs take several pieces)
    const int64_t grp_left_area = lae && atoll(lae) > 0 ? std::max<int64_t>(left_pair * 16, atoll(lae)) / 256 * 256 : std::max<int64_t>(left_pair * 64, std::min<int64_t>(std::min<int64_t>(ctx->ws_limit / 8, (int64_t)1 << 30), 2 * left_pair * ((n + 63) / 64 * 64))) / 256 * 256; // (two halves: the packed kernel's pieces, the int32 kernel's)
    const bool group_regions = auto_group && lane_group && (!hooks || hooks_dev_sort) && !score_only && !(gre && atoi(gre) == 0) && grp_slots >= std::min<int64_t>(ck_chip, (int64_t)ctx->n_cus * LANE_CK_WAVES_PER_CU) / 2 &&
                               grp_ws - grp_slots * ck_region - grp_left_area >= (int64_t)sizeof(DpRecord) * kLaneGroupMinPairs &&
                               n >= kLaneGroupMinPairs;
    const int64_t per_pair = lane_ck || group_regions ? (int64_t)sizeof(DpRecord) // (+ fixed_bytes per half: the persistent grid's regions)
                             : use_lane ? ((score_only ? 0 : stride_words * 4) + lane_scratch_bytes(max_tl, max_ql, rows)) / 128 + 1 + (int64_t)sizeof(DpRecord)
                             : auto_group ? std::max(std::max(stride_words * 2, stride32_words * 4), lane_group ? (lane_group_stride * 4 + lane_ck_scratch_bytes(max_tl, max_ql)) / 128 + 1 : 0) + (int64_t)sizeof(DpRecord)
                                          : (score_only ? 0 : stride_words * 4 / (use16 ? 2 : 1)) + (int64_t)sizeof(DpRecord);
    // the workspace is split in two halves so that the traceback of one chunk can run (on ctx->aux)
    // while the next chunk is being filled; a batch that fits one half is a single chunk
    // chunks are whole waves' worth of pairs (8: packed kernel, and the blocks of MGL_SW_FLAG_GROUPED_GEOMETRY; 4: 16-row
    // int32 kernel; 1: one pair per wave or workgroup); the last chunk may be shorter (idle lanes store nothing)
    const int64_t gran = use_lane ? 128 : use16 ? 8 : (rows == 64 || strip16) ? 1 : 4;
    // lane kernel: every lane walks the paths of its own two pairs at the end of its fill -- no traceback kernel, nothing
    // to overlap, so the whole workspace is one buffer and the chunks are twice as large
    const bool fused_walk = use_lane && !score_only;
    // ... except behind the host-buffer entry, whose chunks are the units of its copy pipeline anyway: there consecutive chunks
    // alternate between two streams and two halves, so that the last waves of one launch (the launch's tail, 1-2 ms of a
    // 14 ms chunk with most CUs idle) run beside the first waves of the next
    const bool dual_ok = fused_walk && hooks;
    // (a sorted chunk sized by slots: one buffer as well -- its bulk walks inside its fill kernel, its left-overs are few)
    const int64_t ws_whole = (fused_walk && !dual_ok) || group_regions ? ctx->ws_limit : ctx->ws_limit / 2;
    const int64_t fixed_max = lane_ck ? ck_slots_max * ck_region : group_regions ? grp_slots * ck_region + grp_left_area : 0; // (a chunk of fewer tiles than slots needs fewer regions: fixed_bytes below)
    const int64_t ws_part = ws_whole - fixed_max;
    if (per_pair * gran > ws_part) {
        char msg[192];
        snprintf(msg, sizeof msg, "traceback of %lld pair(s) (%lld bytes) does not fit half the workspace: raise it with "
                                  "mgl_sw_ctx_set_workspace", (long long)gran, (long long)(per_pair * gran));
        return fail(ctx, MGL_SW_ERR_NOMEM, msg);
    }
    int64_t chunk = std::max<int64_t>(gran, ws_part / per_pair / gran * gran);
    // host-buffer entry: the chunks are also the units of the copy / compute pipeline (inputs of chunk k+1 and results of
    // chunk k-1 move while chunk k computes), so a batch is cut into ~32 even when the workspace would hold it whole
    // (10 M pairs: 133 ms in 46 chunks, 143 in 12, scripts/host_entry_probe.py)
    constexpr int host_chunks = kHostChunks;
    if (hooks && !(use_lane && lane_rounds) && !(hooks_dev_sort && group_regions))
        chunk = std::min<int64_t>(chunk, std::max<int64_t>((n / host_chunks + gran - 1) / gran * gran, (int64_t)256 * 1024));
    // ... a host batch sorted on the device and sized by wave slots: a chunk is a launch of the persistent grid (its end leaves slots idle:
    // few chunks), and what cannot hide is the first chunk's way in and the last one's results on their way out (small chunks) -- about a
    // million pairs each, eight chunks at least from 4 M pairs on (MGL_SW_DEBUG_HOST_SORT_CHUNK: measurements; read per call)
    if (hooks_dev_sort && group_regions) {
        const char *const hce = getenv("MGL_SW_DEBUG_HOST_SORT_CHUNK");
        const int64_t want = hce && atoll(hce) > 0 ? atoll(hce) : std::max<int64_t>((int64_t)1 << 20, (n / 8 + 1023) / 1024 * 1024);
        chunk = std::min<int64_t>(chunk, std::max<int64_t>(want / 128 * 128, hce && atoll(hce) > 0 ? 128 : 128 * 1024));
    }
    // lane kernel behind the host entry: one round of the chip per chunk.  A chunk that is not a whole number of rounds leaves
    // most CUs idle during its last one (1.25 M pairs = 3.2 rounds ran as 4), and nothing computes while the first chunk's
    // inputs cross the link, so small is good; the tails of consecutive launches overlap on the two streams (10 M pairs:
    // 8 chunks 122 ms, 13 chunks 115 ms, 26 chunks of one round 112 ms; one stream: 136-148 ms; scripts/host_sweep.sh)
    if (use_lane && hooks && chunk > lane_round) chunk = lane_rounds ? lane_round : chunk / lane_round * lane_round;
    // ... and where the batch is many rounds long, the chunks GROW and SHRINK: what cannot hide is the first chunk's way in (nothing
    // computes until its inputs have crossed the link) and the last chunk's way out, so those are one round each; in between
    // chunks double up to eight rounds -- a launch of eight rounds runs at 95 % of the long-launch rate, one of a single round at
    // 75-80 % -- and halve again towards the end: 1, 2, 4, 8, 8, .., 8, 4, 2, 1 rounds.  `chunk` is the largest.
    const int64_t pyr_unit = (int64_t)ctx->n_cus * (lane_ck_ok ? 8 : lane_rows == 16 ? 16 : 12) * 128; // one round of the chip
    const int64_t pyr_fit = ws_part / per_pair / pyr_unit; // rounds that fit half the workspace
    // (2-bit inputs only: ASCII inputs keep the link busy for longer than the kernels run -- 4.2 GB against 64 ms -- and there a chunk's
    // kernel waits for its whole copy: one round per chunk stays best, registered arrays 85.0 ms against 87.9 with growing chunks)
    const char *const pyr_env = getenv("MGL_SW_DEBUG_HOST_PYRAMID"); // (the largest chunk in rounds, 0: all chunks two rounds; read per call: the host-sanitizer driver compares both)
    const int pyr_max = pyr_env ? atoi(pyr_env) : 8;
    const bool pyramid = use_lane && hooks && lane_rounds && !auto_group && tset.packed2 && pyr_max > 0 && pyr_fit >= 1;
    if (pyramid) chunk = pyr_unit * std::min<int64_t>(pyr_max, pyr_fit);
    // strip kernel: one pair per workgroup of W waves at three waves per SIMD -- a chunk that is not a whole number of rounds of
    // the chip (n_cus * (12 / W) pairs) ends with a round in which most CUs idle (1 582 pairs per chunk ran as two rounds)
    // the eight-pairs-per-wave kernel, a chunk of several rounds of the chip: WHOLE rounds (round 5).  Its workgroups are resident for their
    // pairs' lifetime (LDS: nine waves per CU at 300 residues) and a sorted batch's neighbours take alike, so a launch runs round after
    // round -- and one of 3 379 waves on 2 304 slots runs 1.47 rounds in the time of two (the protein bench, 74 launches: counters in
    // profiles/r05_b_protein_pmc.txt, VALU busy 51 %).  MGL_SW_DEBUG_WHOLE_ROUNDS=0: as before (measurements; read per call)
    if (use16 && !use_lane && !auto_group && !hooks) {
        const char *const wre = getenv("MGL_SW_DEBUG_WHOLE_ROUNDS");
        const int lds_block = dp16_lds_bytes(sps_for(max_ql), wpb16) + lds_extra;
        const int64_t round = (int64_t)ctx->n_cus * std::max(1, 160 * 1024 / std::max(lds_block, 1)) * wpb16 * 8;
        if (!(wre && atoi(wre) == 0) && chunk > round && chunk < n) chunk = chunk / round * round;
    }
    if (strip16) {
        // (workgroups per CU: what the kernel's register budget allows -- and, with the query's tables in LDS, what 160 KB hold)
        const int lds_codes = strip_k > 0 ? strip16_lds_bytes_codes(max_ql, strip_waves) : 0;
        const int by_lds = lds_codes > 0 && lds_codes <= 64 * 1024 ? std::max(1, (160 * 1024) / (lds_codes + 512)) : 64;
        const int64_t round = (int64_t)ctx->n_cus * std::min(4 * strip16_waves_per_simd(rows) / strip_waves, by_lds);
        if (chunk > round) chunk = chunk / round * round;
        // (without stored flags the workspace holds 2 304 pairs of 10 kb as ONE chunk, and that is right: the walk kernel's time is a
        // pair's serial chain of ~200 block recomputations whatever the number of pairs -- three chunks of 768 walked three times as
        // long in total, 75.9 ms against 20.6, and hid behind nothing: 94.4 ms per pass against 72.5)
    }
    chunk = std::min<int64_t>(chunk, n);
    const bool overlap = !fused_walk && !group_regions && n > chunk;
    const bool dual = dual_ok && n > chunk;
    const int halves = overlap || dual ? 2 : 1;
    P.use_lane = use_lane;
    P.lane_ck = lane_ck;
    P.use16 = use16;
    P.strip16 = strip16;
    P.coop16 = coop16;
    P.coop_waves = coop_waves;
    P.strip_waves = strip_waves;
    P.strip_k = strip_k;
    P.strip_passes = strip_passes;
    P.auto_group = auto_group;
    P.lane_group = lane_group;
    P.lane_group_stride = lane_group_stride;
    P.score_only = score_only;
    P.fused_walk = fused_walk;
    P.use_scratch = use_scratch;
    P.rows = rows;
    P.wpb = wpb;
    P.wpb16 = wpb16;
    P.sps_cap = sps_cap;
    P.sps32 = sps32;
    P.stride_words = stride_words;
    P.stride32_words = stride32_words;
    P.per_pair = per_pair;
    P.lane_slots = lane_ck ? std::min<int64_t>(ck_slots_max, (chunk + 127) / 128) : group_regions ? std::min<int64_t>(grp_slots, (chunk + 127) / 128)
                   : lane_group ? std::min<int64_t>(ck_chip, (chunk + 127) / 128) : 0;
    P.fixed_bytes = lane_ck ? P.lane_slots * ck_region : group_regions ? P.lane_slots * ck_region + grp_left_area : 0;
    P.group_regions = group_regions;
    P.left_area = group_regions ? grp_left_area : 0;
    P.left_pair16 = stride_words * 2;
    P.left_pair32 = stride32_words * 4;
    P.chunk = chunk;
    P.dev_sort_host = hooks_dev_sort && group_regions;
    P.pyramid = pyramid;
    P.pyr_unit = pyr_unit;
    P.overlap = overlap;
    P.dual = dual;
    P.halves = halves;
    P.geom = geom;
    return MGL_SW_OK;
}

// A caller's promise of blocks of eight (MGL_SW_FLAG_GROUPED_GEOMETRY) used to pin a large device-resident batch to the eight-pairs-per-
// wave kernel (2.9 TCUPS) although a batch the library sorts itself gets the lane kernel for its whole waves of 128 (4.9): round 3 measured
// a caller who sorts and promises at 3 013 GCUPS against 4 883 for one who does neither.  Where the library's own sort would run and feed
// the lane kernel, the promise is therefore set aside and the batch planned as one of mixed geometries (the counting sort of an already
// sorted chunk adds to each counter once per run of equal pairs and workgroup); everywhere else the promise stands.
static int plan_batch(mgl_sw_ctx *ctx, int64_t n, const SeqSet &tset, const SeqSet &qset, int max_tl, int max_ql, int match, int mismatch, int gopen, int gext,
                      int strategy, const Score *d_score, const char *d_cigar, int cigar_stride, int geom, bool binary_cigar, const ChunkHooks *hooks,
                      const int8_t *d_matrix, bool score_only_hint, BatchPlan &P)
{
    const int rc = plan_batch_as(ctx, n, tset, qset, max_tl, max_ql, match, mismatch, gopen, gext, strategy, d_score, d_cigar, cigar_stride, geom, binary_cigar, hooks,
                                 d_matrix, score_only_hint, P);
    if (rc != MGL_SW_OK || geom != GEOM_GROUPED || hooks || !P.use16 || P.use_lane) return rc;
    BatchPlan Q{};
    const std::string err = ctx->err;
    if (plan_batch_as(ctx, n, tset, qset, max_tl, max_ql, match, mismatch, gopen, gext, strategy, d_score, d_cigar, cigar_stride, GEOM_MIXED, binary_cigar, hooks,
                      d_matrix, score_only_hint, Q) == MGL_SW_OK &&
        Q.auto_group && Q.lane_group && std::min(n, Q.chunk) >= kLaneGroupMinPairs)
        P = Q;
    ctx->err = err;
    return MGL_SW_OK;
}

// Enqueue fill + traceback for a device-resident batch on `stream`.
int run_device(mgl_sw_ctx *ctx, hipStream_t stream, int64_t n, const SeqSet &tset, const SeqSet &qset, int max_tl,
               int max_ql, int match, int mismatch, int gopen, int gext, int strategy, int32_t *d_offset, Score *d_score,
               char *d_cigar, int cigar_stride, int32_t *d_cigar_len, int32_t *d_status, int64_t cells_hint, int geom,
               bool binary_cigar = false, const ChunkHooks *hooks = nullptr, const int8_t *d_matrix = nullptr,
               const uint8_t *d_code = nullptr, bool score_only_hint = false, mgl_sw_plan *explain = nullptr)
{
    if (n == 0) return MGL_SW_OK;
    if (n < 0 || !tset.data || !tset.off || !qset.data || !qset.off || !d_offset || !d_cigar || cigar_stride < 1 ||
        max_tl < 1 || max_ql < 1 || !strategy_ok(strategy))
        return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch_device: bad argument");
    if (!explain) {
        const int frc = grid_fault_check(ctx);
        if (frc != MGL_SW_OK) return frc;
    }
    if (d_matrix) {
        // substitution-matrix scoring: `match` / `mismatch` carry the largest / smallest matrix entry (range check and
        // offset representation of the packed kernel); only the gap penalties go through the sign normalisation
        int m1 = 1, m2 = -1;
        mgl_sw_normalize_params(&m1, &m2, &gopen, &gext);
    } else {
        mgl_sw_normalize_params(&match, &mismatch, &gopen, &gext);
    }

    // small batches are latency bound (the coalesced one-pair-per-call traffic of alignNative): one wave per pair keeps the pair's H
    // matrix in LDS, walks the path off it and writes the text, all in ONE launch that touches no workspace (sw_small.hip)
    {
        bool wide = false;
        // (the crossovers were measured where two pairs' matrices share a CU's LDS: a geometry that leaves room for one halves them)
        const int64_t small_limit = (geom == GEOM_MIXED ? MGL_SW_SMALL_BATCH_PAIRS_MIXED : MGL_SW_SMALL_BATCH_PAIRS) /
                                    (small_lds_bytes(max_tl, max_ql, cigar_stride, !small_fits_int16(max_tl, max_ql, match, mismatch, gopen, gext)) > 80 * 1024 ? 2 : 1);
        const bool unforced = ctx->precision == 0 && ctx->stripe_rows == 0 && ctx->cooperative == 0 && ctx->carry_memory == 0 && ctx->lane_kernel != 2 &&
                              ctx->strip_kernel != 2 && ctx->lane_checkpoint != 1;
        if (!hooks && !d_matrix && !score_only_hint && !binary_cigar && ctx->small_kernel != 1 &&
            (ctx->small_kernel == 2 || (unforced && n <= small_limit)) &&
            ((int64_t)match - mismatch + gopen + 2 * (int64_t)gext) * ((int64_t)max_tl + max_ql) < (1ll << 30) &&
            small_supported(max_tl, max_ql, cigar_stride, match, mismatch, gopen, gext, &wide)) {
            if (explain) {
                mgl_sw_plan &pl = *explain;
                pl = mgl_sw_plan{};
                pl.fill_kernel = MGL_SW_KERNEL_SMALL;
                pl.precision_bits = 32;
                pl.rows = ((max_tl + 63) / 64 + 1) & ~1;
                pl.waves_per_block = 1;
                pl.waves_per_pair = 1;
                pl.traceback = 1;
                pl.fused_walk = 1;
                pl.fill_streams = 1;
                pl.workspace_halves = 1;
                pl.chunk_pairs = n;
                pl.chunks = 1;
                return MGL_SW_OK;
            }
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            if (ctx->profiling == 3) {
                ctx->timing.cells += cells_hint;
            } else {
                ctx->timing = mgl_sw_timing{};
                ctx->timing.cells = cells_hint;
                ctx->pool_used = 0;
            }
            TbArgs ta{};
            ta.t = tset;
            ta.q = qset;
            ta.first = 0;
            ta.count = n;
            ta.strategy = strategy;
            ta.match = match;
            ta.mismatch = mismatch;
            ta.gopen = gopen;
            ta.gext = gext;
            ta.offset = d_offset;
            ta.score = d_score;
            ta.cigar = d_cigar;
            ta.cigar_stride = cigar_stride;
            ta.cigar_len = d_cigar_len;
            ta.status = d_status;
            hipEvent_t pe[4] = {nullptr, nullptr, nullptr, nullptr};
            if (ctx->profiling) {
                while ((int)ctx->pool.size() < ctx->pool_used + 4) {
                    hipEvent_t e = nullptr;
                    HIP_TRY(ctx, hipEventCreate(&e));
                    ctx->pool.push_back(e);
                }
                for (int i = 0; i < 4; ++i) pe[i] = ctx->pool[(size_t)ctx->pool_used + i];
                ctx->pool_used += 4;
                ctx->diag_blocks = 0;
            }
            if (pe[0]) HIP_TRY(ctx, hipEventRecord(pe[0], stream));
            HIP_TRY(ctx, launch_small(ta, max_tl, max_ql, wide, stream));
            for (int i = 1; i < 4; ++i)
                if (pe[i]) HIP_TRY(ctx, hipEventRecord(pe[i], stream));
            ctx->last_chunk_count = 0; // (nothing of the matrix leaves the chip: no slot to expand)
            ctx->timing.dp_launches++;
            ctx->timing.packed16 = 0;
            ctx->timing.fill_kernel = MGL_SW_KERNEL_SMALL;
            return MGL_SW_OK;
        }
    }
    // ---- the plan (plan_batch above: pure), then its execution
    BatchPlan P{};
    {
        const int prc = plan_batch(ctx, n, tset, qset, max_tl, max_ql, match, mismatch, gopen, gext, strategy, d_score, d_cigar, cigar_stride, geom, binary_cigar, hooks,
                                   d_matrix, score_only_hint, P);
        if (prc != MGL_SW_OK) return prc;
    }
    const bool use_lane = P.use_lane;
    const bool lane_ck = P.lane_ck;
    const bool use16 = P.use16;
    const bool strip16 = P.strip16;
    const bool coop16 = P.coop16;
    const int coop_waves = P.coop_waves;
    const int strip_waves = P.strip_waves;
    const int strip_k = P.strip_k;
    const int strip_passes = P.strip_passes;
    const bool auto_group = P.auto_group;
    const bool lane_group = P.lane_group;
    const int64_t lane_group_stride = P.lane_group_stride;
    const bool score_only = P.score_only;
    const bool fused_walk = P.fused_walk;
    const bool use_scratch = P.use_scratch;
    const int rows = P.rows;
    const int wpb = P.wpb;
    const int wpb16 = P.wpb16;
    const int sps_cap = P.sps_cap;
    const int sps32 = P.sps32;
    const int64_t stride_words = P.stride_words;
    const int64_t stride32_words = P.stride32_words;
    const int64_t per_pair = P.per_pair;
    const int64_t lane_slots = P.lane_slots;
    const int64_t chunk = P.chunk;
    const bool pyramid = P.pyramid;
    const int64_t pyr_unit = P.pyr_unit;
    const bool overlap = P.overlap;
    const bool dual = P.dual;
    const int halves = P.halves;
    // the size of the chunk that starts at pair `first` as the k-th of the call
    auto chunk_at = [&](int64_t first, int64_t k) -> int64_t {
        const int64_t left = n - first;
        if (P.dev_sort_host) {
            // what cannot hide is the first chunk's way in (nothing computes until its inputs have crossed the link and its sort is done) and
            // the last chunk's results on their way out: those two are half a chunk
            const int64_t half = std::max<int64_t>(128, chunk / 2 / 128 * 128);
            if (n <= chunk) return left; // (one chunk holds the batch)
            if (k == 0) return half;
            if (left > chunk + half) return chunk;
            return left - half >= chunk / 4 ? (left - half) / 128 * 128 : left; // (no launch for a sliver: it goes with the last chunk)
        }
        if (!pyramid) return std::min(chunk, left);
        const int64_t grow = std::min(chunk, pyr_unit << std::min<int64_t>(k, 8));
        const int64_t half_left = std::max(pyr_unit, left / 2 / pyr_unit * pyr_unit); // (whole rounds; towards the end: half of what is left)
        return std::min(left, std::min(grow, half_left));
    };
    if (explain) { // mgl_sw_explain: everything above is pure (the context's settings and the batch's description); nothing below is
        mgl_sw_plan &pl = *explain;
        pl = mgl_sw_plan{};
        const bool lane_bulk = lane_group && std::min(n, chunk) >= kLaneGroupMinPairs; // (a chunk's whole waves of one geometry get their own launch from there on)
        pl.fill_kernel = lane_ck || (auto_group && lane_bulk) ? MGL_SW_KERNEL_LANE16_CK : use_lane ? MGL_SW_KERNEL_LANE16 : use16 ? MGL_SW_KERNEL_DP16 : strip16 ? MGL_SW_KERNEL_STRIP16
                         : coop16 ? MGL_SW_KERNEL_COOP16 : coop_waves ? MGL_SW_KERNEL_COOP : rows == 64 ? MGL_SW_KERNEL_DP32_64 : MGL_SW_KERNEL_DP32;
        pl.precision_bits = (use16 || use_lane || strip16 || coop16) ? 16 : 32;
        pl.rows = rows;
        pl.waves_per_block = wpb;
        pl.waves_per_pair = strip16 ? strip_waves : coop_waves;
        pl.traceback = score_only ? 2 : lane_ck || (strip16 && strip_k) ? 1 : 0;
        pl.fused_walk = fused_walk ? 1 : 0;
        pl.sorted_by_library = auto_group ? (hooks && hooks->regroup ? 2 : 1) : 0;
        pl.fill_streams = dual ? 2 : 1;
        pl.workspace_halves = halves;
        pl.chunk_pairs = chunk;
        pl.chunks = 0;
        for (int64_t first = 0; first < n; first += chunk_at(first, pl.chunks), ++pl.chunks) {
        }
        pl.workspace_bytes_per_pair = per_pair;
        pl.workspace_fixed_bytes = P.fixed_bytes * halves;
        pl.workspace_bytes = (per_pair * chunk + P.fixed_bytes) * halves;
        pl.resident_waves = lane_slots;
        return MGL_SW_OK;
    }

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    for (int h = 0; h < halves; ++h) {
        const size_t regions = (size_t)(lane_ck ? lane_slots : use_lane ? (chunk + 127) / 128 : use16 ? (chunk + 1) / 2 : chunk);
        if (P.group_regions)
            HIP_TRY(ctx, ctx->tb[h].reserve((size_t)lane_slots * (size_t)lane_group_stride * 4 + (size_t)P.left_area + 64));
        else if (auto_group)
            HIP_TRY(ctx, ctx->tb[h].reserve((size_t)chunk * (size_t)(per_pair - (int64_t)sizeof(DpRecord)) + 64));
        else if (!score_only)
            HIP_TRY(ctx, ctx->tb[h].reserve(regions * (size_t)stride_words * 4));
        if (use_lane) HIP_TRY(ctx, ctx->bnd[h].reserve(regions * (size_t)(lane_ck ? lane_ck_scratch_bytes(max_tl, max_ql) : lane_scratch_bytes(max_tl, max_ql, rows))));
        if (lane_group) // (the persistent grid's staging areas: one per wave slot)
            HIP_TRY(ctx, ctx->bnd[h].reserve((size_t)(P.group_regions ? lane_slots + 1 : chunk / 128 + 1) * (size_t)lane_ck_scratch_bytes(max_tl, max_ql)));
        HIP_TRY(ctx, ctx->rec[h].reserve((size_t)chunk * sizeof(DpRecord)));
    }
    if (lane_slots > 0 && !ctx->tile_ctr.p) {
        HIP_TRY(ctx, ctx->tile_ctr.reserve((size_t)kTileCounters * kTileCounterWords * sizeof(unsigned)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->tile_ctr.p, 0, (size_t)kTileCounters * kTileCounterWords * sizeof(unsigned), stream));
    }
    if (lane_slots > 0 && !ctx->pin_fault) {
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->pin_fault), 64, hipHostMallocDefault));
        memset(ctx->pin_fault, 0, 64);
    }

    if (use_scratch) {
        const int64_t groups = (chunk + 15) / 16 * 16; // every group of every launched wave has its own area
        HIP_TRY(ctx, ctx->scratch.reserve((size_t)(groups * dp_group_bytes(sps_cap, rows))));
    }
    if (coop_waves) HIP_TRY(ctx, ctx->scratch.reserve((size_t)chunk * coop_wrap_cols(sps_cap) * 8)); // one carry row per pair
    if (strip16) HIP_TRY(ctx, ctx->scratch.reserve((size_t)chunk * (size_t)strip16_scratch_bytes(max_ql, strip_waves))); // last row, parked last columns

    // the workspace belongs to the context, not to a stream: order this call behind the previous one's kernels
    if (ctx->ws_idle_set) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ws_idle, 0));
    if (ctx->ws_idle_set && dual) HIP_TRY(ctx, hipStreamWaitEvent(ctx->fill2, ctx->ws_idle, 0));
    if (dual) { // whatever the caller enqueued on `stream` before this call comes first on the second stream too
        HIP_TRY(ctx, hipEventRecord(ctx->fill_done[0], stream));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->fill2, ctx->fill_done[0], 0));
    }

    if (ctx->profiling == 3) { // accumulate over calls until mgl_sw_ctx_get_timing reads and clears
        ctx->timing.cells += cells_hint;
    } else {
        ctx->timing = mgl_sw_timing{};
        ctx->timing.cells = cells_hint;
        ctx->pool_used = 0;
    }

    bool tb_pending[2] = {false, false};
    bool side_grid_pending[2] = {false, false}; // a sorted chunk's grid has been enqueued on the fill stream of half h (fill_done[h] follows it)
    int64_t sort_next_first = 0, sorted_chunks = 0; // device sort, two chunks ahead: where the next chunk to sort starts, how many have been enqueued
    bool srt_used[4] = {false, false, false, false};
    int64_t k = 0;
    // result copies trail the launches by two chunks: the traceback of chunk k-2 is what the fill of chunk k waits
    // for anyway, so the (blocking) copy of its results never stalls behind the low-priority traceback stream
    int64_t lane_pairs_last = 0; // sorted chunks: pairs of the last chunk that went through the lane kernel
    struct Pending { int64_t first, count; hipEvent_t ready; } pending[2];
    int n_pending = 0;
    // ---- the device-side sort of a batch of mixed geometries (auto_group without a host-side regroup hook), used inside the loop below
    // the sort of chunk k runs on the copy stream (idle for a device-resident batch) while chunk k-1 is being filled:
    // it is enqueued one chunk ahead, and only that stream is synchronised to read the block total
    auto sort_chunk = [&](int64_t f, int64_t c, int hh, RegroupArgs *out) -> int { // hh = chunk number mod 4
        const size_t cells_n = (size_t)max_tl * max_ql;
        int64_t *d = static_cast<int64_t *>(ctx->d_srt[hh].p);
        int32_t *g = static_cast<int32_t *>(ctx->d_grid.p);
        RegroupArgs ra;
        ra.t = tset;
        ra.q = qset;
        ra.first = f;
        ra.count = c;
        ra.max_tl = max_tl;
        ra.max_ql = max_ql;
        ra.cnt = g;
        ra.nfull = g + cells_n;
        ra.full_start = g + 2 * cells_n;
        ra.rest_start = g + 3 * cells_n;
        ra.nlane = g + 4 * cells_n;
        ra.lane_start = g + 5 * cells_n;
        ra.lane_blocks = lane_group ? 1 : 0;
        ra.total = reinterpret_cast<int64_t *>(g + 6 * cells_n) + 2 * hh;
        ra.t_start = d;
        ra.q_start = d + c;
        ra.dest = d + 2 * c;
        ra.t_len = reinterpret_cast<int32_t *>(d + 3 * c);
        ra.q_len = ra.t_len + c;
        if (out) {
            *out = ra;
            return MGL_SW_OK;
        }
        // these index arrays were last read by the kernels of the chunk four before this one
        if (srt_used[hh]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->h2d, ctx->srt_free[hh], 0));
        if (hooks && hooks->bring_ahead) { // a host batch: this chunk's inputs cross the link in front of its sort, on the same stream
            const int brc = hooks->bring_ahead(f, c);
            if (brc != MGL_SW_OK) return brc;
        }
        HIP_TRY(ctx, launch_regroup(ra, ctx->h2d));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->pin_total + 2 * hh, ra.total, 16, hipMemcpyDeviceToHost, ctx->h2d));
        HIP_TRY(ctx, hipEventRecord(ctx->srt_done[hh], ctx->h2d));
        return MGL_SW_OK;
    };
    // the sorts run two chunks ahead of the fills (beside a fill kernel that has the chip a sort of unsorted reads takes
    // 2-3 ms, longer than one chunk's fill); chunk j starts where chunk j - 1 ended (chunk_at: the chunks need not be alike)
    // (sort_until(j): the sorts of the chunks up to number j are enqueued)
    auto sort_until = [&](int64_t j) -> int {
        while (sorted_chunks <= j && sort_next_first < n) {
            const int64_t f = sort_next_first, c = chunk_at(f, sorted_chunks);
            sort_next_first += c;
            const int src = sort_chunk(f, c, (int)(sorted_chunks & 3), nullptr);
            if (src != MGL_SW_OK) return src;
            ++sorted_chunks;
        }
        return MGL_SW_OK;
    };
    for (int64_t first = 0, count = 0; first < n; first += count, ++k) {
        count = chunk_at(first, k);
        const int h = (int)(k & (halves - 1));
        hipStream_t const fs = dual && (k & 1) ? ctx->fill2 : stream; // this chunk's fill stream (same half => same stream: ordered)
        if (hooks) {
            const int hrc = hooks->before_fill(first, count, fs);
            if (hrc != MGL_SW_OK) return hrc;
        }
        bool sort_more = false;
        // what this chunk launches: one part normally; a chunk the host entry has sorted by geometry has a packed part
        // (full blocks of eight) and an int32 part (the left-over pairs), each with its own kernels and workspace regions
        struct Part {
            bool lane, packed;
            int64_t first, count;
            SeqSet t, q;
            uint32_t *tb;
            int64_t stride;
            DpRecord *rec;
            const int64_t *dest;
            int rows, wpb, sps_cap;
            bool tb_now; // its traceback follows its fill at once, on the fill stream: the next part uses the same workspace area
        };
        std::vector<Part> parts;
        parts.reserve(8);
        uint32_t *const tb_base = static_cast<uint32_t *>(ctx->tb[h].p);
        DpRecord *const rec_base = static_cast<DpRecord *>(ctx->rec[h].p);
        if (!auto_group) {
            parts.push_back(Part{use_lane, use16, first, count, tset, qset, tb_base, stride_words, rec_base, nullptr, rows, wpb, sps_cap, false});
        } else {
            // the index arrays of this half were last read by the kernels of chunk k-2
            if (overlap && tb_pending[h]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->h2d, ctx->tb_done[h], 0));
            Regroup rg;
            rg.lane_blocks = lane_group;
            if (hooks && hooks->regroup) {
                const int hrc = hooks->regroup(first, count, h, fs, &rg);
                if (hrc != MGL_SW_OK) return hrc;
            } else {
                if (k == 0) {
                    const size_t cells_n = (size_t)max_tl * max_ql;
                    sort_next_first = 0;
                    sorted_chunks = 0;
                    for (int b = 0; b < 4; ++b) {
                        HIP_TRY(ctx, ctx->d_srt[b].reserve((size_t)chunk * 32));
                        if (!ctx->srt_free[b]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->srt_free[b], hipEventDisableTiming));
                        if (!ctx->srt_done[b]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->srt_done[b], hipEventDisableTiming));
                    }
                    HIP_TRY(ctx, ctx->d_grid.reserve(cells_n * 24 + 128));
                    if (!ctx->pin_total) HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->pin_total), 64, hipHostMallocDefault));
                    // (the caller's stream may still be producing the inputs)
                    HIP_TRY(ctx, hipEventRecord(ctx->fill_done[0], stream));
                    HIP_TRY(ctx, hipStreamWaitEvent(ctx->h2d, ctx->fill_done[0], 0));
                }
                // A HOST batch: what stands in front of a chunk's sort is the host's own work on that chunk -- a look at its index arrays, its
                // copy commands (bring_ahead) -- and in front of the FIRST launch nothing hides it: the first chunk alone, then the launch, then
                // the chunks ahead (traced: chunks 0, 1 and 2 prepared before the first kernel started, 8 ms into a call of 33).
                {
                    const int src = sort_until(hooks ? k : k + 1);
                    if (src != MGL_SW_OK) return src;
                }
                HIP_TRY(ctx, hipEventSynchronize(ctx->srt_done[k & 3])); // the sort of THIS chunk, enqueued two iterations ago
                RegroupArgs ra;
                (void)sort_chunk(first, count, (int)(k & 3), &ra);
                rg.d_t_start = ra.t_start;
                rg.d_q_start = ra.q_start;
                rg.d_dest = ra.dest;
                rg.d_t_len = ra.t_len;
                rg.d_q_len = ra.q_len;
                rg.n_grouped = ctx->pin_total[2 * (k & 3)];
                rg.n_lane = ctx->pin_total[2 * (k & 3) + 1];
                if (rg.n_grouped < 0 || rg.n_grouped > count || (rg.n_grouped & 7) || rg.n_lane < 0 || rg.n_lane > rg.n_grouped || (rg.n_lane & 127))
                    return fail(ctx, MGL_SW_ERR_DEVICE, "sorting a chunk by geometry failed");
                if (!hooks) { // (device resident: nothing of the host's stands in front of a sort)
                    const int src = sort_until(k + 2);
                    if (src != MGL_SW_OK) return src;
                }
                sort_more = hooks != nullptr; // (a host batch: behind this chunk's launches, below)
            }
            const SeqSet ts{tset.data, rg.d_t_start, rg.d_t_len, max_tl, tset.packed2}, qs{qset.data, rg.d_q_start, rg.d_q_len, max_ql, qset.packed2};
            const int64_t ng = rg.n_grouped;
            // whole waves of one geometry: worth a launch of the lane kernel from two rounds of the chip on (a wave takes as long as 128
            // pairs one after the other however few waves there are: 4 M reads of 100-150 bases in chunks of 1.3 M pairs 2 963 GCUPS
            // against 2 790 without, but the host entry's chunks of 256 k pairs 2 131 against 2 397); fewer go with the blocks of
            // eight, which they also are
            const char *const lgm = getenv("MGL_SW_DEBUG_LANE_GROUP_MIN"); // (tests lower the threshold between calls: read per chunk)
            const int64_t lane_group_min = lgm ? (int64_t)atoll(lgm) : kLaneGroupMinPairs;
            const int64_t nl = lane_group && rg.n_lane >= std::max<int64_t>(lane_group_min, 128) ? rg.n_lane : 0;
            uint32_t *const tb16 = tb_base + (size_t)std::min<int64_t>(nl / 128, lane_slots) * (size_t)lane_group_stride; // (behind the lane part's regions: one per wave slot)
            lane_pairs_last = nl;
            if (nl > 0) parts.push_back(Part{true, false, 0, nl, ts, qs, tb_base, lane_group_stride, rec_base, rg.d_dest, 32, 4, max_ql, false});
            if (P.group_regions) {
                // what is left over, in pieces of what its area holds (one piece each wherever the batch's geometries are few): the packed
                // kernel's pieces in the area's first half, the int32 kernel's in its second; a piece's traceback runs before the
                // next piece's fill overwrites the flags
                uint32_t *const area16 = tb_base + (size_t)lane_slots * (size_t)lane_group_stride, *const area32 = area16 + (size_t)(P.left_area / 2 / 4);
                const int64_t cap16 = std::max<int64_t>(8, P.left_area / 2 / P.left_pair16 / 8 * 8), cap32 = std::max<int64_t>(4, P.left_area / 2 / P.left_pair32 / 4 * 4);
                for (int64_t f = nl; f < ng; f += cap16)
                    parts.push_back(Part{false, true, f, std::min(cap16, ng - f), ts, qs, area16, stride_words, rec_base + f, rg.d_dest, 16, wpb16, sps_for(max_ql), f + cap16 < ng});
                for (int64_t f = ng; f < count; f += cap32)
                    parts.push_back(Part{false, false, f, std::min(cap32, count - f), ts, qs, area32, stride32_words, rec_base + f, rg.d_dest, 16,
                                         pick_waves_per_block(sps32, 16), sps32, f + cap32 < count});
            } else {
                if (ng > nl) parts.push_back(Part{false, true, nl, ng - nl, ts, qs, tb16, stride_words, rec_base + nl, rg.d_dest, 16, wpb16, sps_for(max_ql), false});
                if (count > ng)
                    parts.push_back(Part{false, false, ng, count - ng, ts, qs, tb16 + (size_t)((ng - nl) / 2) * (size_t)stride_words, stride32_words,
                                         rec_base + ng, rg.d_dest, 16, pick_waves_per_block(sps32, 16), sps32, false});
            }
        }
        const int n_parts = (int)parts.size();
        // (a sorted chunk sized by slots: its left-overs run beside its bulk, see the launches below)
        const char *const sre = getenv("MGL_SW_DEBUG_SIDE_RESERVE"); // (wave slots the bulk's grid leaves free for them; 0: one stream, the left-overs behind the grid; read per call)
        const int64_t side_reserve = sre ? atoll(sre) : kSideReserve;
        const bool side = P.group_regions && side_reserve > 0;
        std::vector<DpArgs> das((size_t)n_parts);
        std::vector<TbArgs> tas((size_t)n_parts);
        int64_t n_blocks = 0;
        for (int i = 0; i < n_parts; ++i) {
            const Part &pt = parts[i];
            DpArgs &da = das[i];
            da.t = pt.t;
            da.q = pt.q;
            da.first = pt.first;
            da.count = pt.count;
            da.match = match;
            da.mismatch = mismatch;
            da.gopen = gopen;
            da.gext = gext;
            da.strategy = strategy;
            da.sps_cap = pt.sps_cap;
            da.uni_tl = max_tl;
            da.uni_ql = max_ql;
            da.tb = pt.tb;
            da.tb_stride_words = pt.stride;
            da.rec = pt.rec;
            da.scratch = pt.lane ? static_cast<unsigned char *>(ctx->bnd[h].p) : (use_scratch || coop_waves || strip16) ? static_cast<unsigned char *>(ctx->scratch.p) : nullptr;
            da.diag = nullptr;
            da.matrix = d_matrix;
            da.code = d_code;
            da.matrix_lds_offset = 0;
            da.score_only = score_only ? 1 : 0;
            da.grouped = pt.lane && auto_group ? 1 : 0;
            da.strip_k = strip16 ? strip_k : 0;
            da.strip_passes = strip16 ? strip_passes : 0;
            {
                const char *const sce = getenv("MGL_SW_DEBUG_STRIP_CODES"); // (0: the byte-compare form whatever the sequences; read per call: the tests run both forms)
                da.strip_codes = strip16 && strip_k > 0 && !(sce && atoi(sce) == 0) && strip16_lds_bytes_codes(max_ql, strip_waves) <= 64 * 1024 ? 1 : 0;
            }
            da.lane_slots = 0;
            da.tile_ctr = nullptr;
            da.grid_fault = nullptr;
            da.gate = nullptr;
            da.gate_dev = nullptr;
            da.gate_failed = nullptr;
            da.gate_timeout_ticks = 0;
            if (pt.lane && (lane_ck || da.grouped)) { // the persistent grid: its wave slots and, where the tiles outnumber them, a zeroed counter
                da.lane_slots = (int)std::min<int64_t>(lane_slots - (side && n_parts > 1 ? std::min<int64_t>(side_reserve, lane_slots / 2) : 0), (pt.count + 127) / 128);
                if (ctx->cur_gate && !hooks) { // the direct form of a host entry: the inputs are still arriving
                    da.gate = ctx->cur_gate;
                    da.gate_dev = ctx->cur_gate_dev;
                    da.gate_failed = reinterpret_cast<int32_t *>(const_cast<int64_t *>(ctx->cur_gate) + 1);
                    da.gate_timeout_ticks = 100000000u; // one second without the word moving
                    if (const char *const gte = getenv("MGL_SW_DEBUG_GATE_TIMEOUT_TICKS")) da.gate_timeout_ticks = (unsigned)atoll(gte); // (tests take the give-up path with it; read per call)
                }
                if ((pt.count + 127) / 128 > da.lane_slots) { // (an entry of the ring: at zero, its last launch's last wave has seen to that)
                    da.tile_ctr = static_cast<unsigned *>(ctx->tile_ctr.p) + (size_t)(ctx->tile_seq++ % kTileCounters) * kTileCounterWords;
                    void *fault_dev = nullptr;
                    HIP_TRY(ctx, hipHostGetDevicePointer(&fault_dev, ctx->pin_fault, 0));
                    da.grid_fault = static_cast<int32_t *>(fault_dev);
                }
            }
            const int per_block = pt.lane ? pt.wpb * 128 : pt.packed ? pt.wpb * 8 : strip16 ? 1 : pt.wpb * (64 / pt.rows);
            if (i == 0) n_blocks = da.lane_slots > 0 ? da.lane_slots : (pt.count + per_block - 1) / per_block; // (the persistent grid: one workgroup per wave slot)
            if (ctx->profiling == 2 && i == 0) {
                HIP_TRY(ctx, ctx->diag.reserve((size_t)n_blocks * 16));
                da.diag = static_cast<unsigned long long *>(ctx->diag.p);
            }
            TbArgs &ta = tas[i];
            ta.t = pt.t;
            ta.q = pt.q;
            ta.first = pt.first;
            ta.count = pt.count;
            ta.strategy = strategy;
            ta.tb = pt.tb;
            ta.tb_stride_words = pt.stride;
            ta.packed16 = pt.lane ? 2 : pt.packed ? 1 : strip16 ? (strip_k ? 6 : 4) : coop16 ? 3 : 0;
            ta.match = match;
            ta.mismatch = mismatch;
            ta.gopen = gopen;
            ta.gext = gext;
            ta.strip_rows = pt.rows;
            ta.strip_k = strip16 ? strip_k : 0;
            ta.rows_per_stripe = pt.rows;
            ta.uni_ql = max_ql;
            ta.rec = pt.rec;
            ta.offset = d_offset;
            ta.score = d_score;
            ta.cigar = d_cigar;
            ta.cigar_stride = cigar_stride;
            ta.binary_cigar = binary_cigar ? 1 : 0;
            ta.cigar_len = d_cigar_len;
            ta.status = d_status;
            ta.status_any = hooks ? hooks->d_status_any : ctx->direct_status_any;
            ta.dest = pt.dest;
            // the checkpointed lane kernel hands a tile's results over in whole lines where the arrays allow it (sw_device.h)
            const char *const colds = getenv("MGL_SW_DEBUG_COALESCED_OUT"); // (0: every lane stores its own results, as before round 4; read per call: tests compare)
            ta.coalesced_out = pt.lane && (lane_ck || da.grouped) && !(colds && atoi(colds) == 0) && lane_ck_coalesced_ok(ta) ? 1 : 0;
        }

        hipEvent_t pe[4] = {nullptr, nullptr, nullptr, nullptr};
        if (ctx->profiling) {
            while ((int)ctx->pool.size() < ctx->pool_used + 4) {
                hipEvent_t e = nullptr;
                HIP_TRY(ctx, hipEventCreate(&e));
                ctx->pool.push_back(e);
            }
            for (int i = 0; i < 4; ++i) pe[i] = ctx->pool[(size_t)ctx->pool_used + i];
            ctx->pool_used += 4;
            ctx->diag_blocks = n_blocks;
        }
        hipStream_t tb_stream = overlap ? ctx->aux : fs;
        // this half was last read by the traceback of chunk k-2
        if ((overlap || side) && tb_pending[h]) HIP_TRY(ctx, hipStreamWaitEvent(fs, ctx->tb_done[h], 0));
        if (side && n_parts > 1) {
            // A sorted chunk's left-over pairs -- two short kernels and their walks, a millisecond in a row -- go FIRST and on the second
            // stream, beside the bulk's persistent grid, which leaves them a few wave slots (kSideReserve): behind a grid that holds every
            // slot they would wait for its end.  (Their inputs are complete: the host has waited for this chunk's sort, which waited for the
            // caller's stream.)
            if (k == 0 && ctx->ws_idle_set) HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->ws_idle, 0));
            // ... and, from the second chunk on, behind the grid of the chunk BEFORE (round 5): the chunks share one workspace -- this chunk's
            // left-overs keep their records behind its own whole waves' (rec_base + f), where the last chunk's grid, still running on the other
            // stream, keeps the records of ITS whole waves.  Never met while a chunk held a hundred million pairs (sized by records alone); a
            // host batch sorted on the device is cut into chunks of a million: the first multi-chunk run returned CIGARs that did not fit.
            if (side_grid_pending[h]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->fill_done[h], 0));
            for (int i = 0; i < n_parts; ++i) {
                const Part &pt = parts[i];
                if (pt.lane) continue;
                HIP_TRY(ctx, pt.packed ? launch_dp16(das[i], pt.wpb, ctx->aux) : launch_dp(das[i], pt.wpb, pt.rows, ctx->aux));
                HIP_TRY(ctx, launch_traceback(tas[i], ctx->aux));
            }
            HIP_TRY(ctx, hipEventRecord(ctx->tb_done[h], ctx->aux));
            tb_pending[h] = true;
        }
        if (pe[0]) HIP_TRY(ctx, hipEventRecord(pe[0], fs));
        for (int i = 0; i < n_parts; ++i) {
            const Part &pt = parts[i];
            if (side && n_parts > 1 && !pt.lane) continue; // (on the second stream, above)
            TbArgs walk = tas[i];
            if (!fused_walk && !das[i].grouped) walk.cigar = nullptr; // (the waves of a sorted chunk's lane part walk their own paths too)
            HIP_TRY(ctx, pt.lane ? (lane_ck || das[i].grouped ? launch_dp16_lane_ck(das[i], walk, fs) : launch_dp16_lane(das[i], walk, pt.rows, fs))
                         : pt.packed ? launch_dp16(das[i], pt.wpb, fs)
                         : strip16 ? launch_dp16_strip(das[i], strip_waves, pt.rows, fs)
                         : coop16 ? launch_dp_coop16(das[i], coop_waves, fs) : coop_waves ? launch_dp_coop(das[i], coop_waves, fs) : launch_dp(das[i], pt.wpb, pt.rows, fs));
            if (pt.tb_now) HIP_TRY(ctx, launch_traceback(tas[i], fs));
        }
        if (pe[1]) HIP_TRY(ctx, hipEventRecord(pe[1], fs));
        if (overlap) {
            HIP_TRY(ctx, hipEventRecord(ctx->fill_done[h], fs));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->aux, ctx->fill_done[h], 0));
        }
        if (side) { // (the next chunk's left-overs, on the second stream, wait for this chunk's grid: above)
            HIP_TRY(ctx, hipEventRecord(ctx->fill_done[h], fs));
            side_grid_pending[h] = true;
        }
        if (pe[2]) HIP_TRY(ctx, hipEventRecord(pe[2], tb_stream));
        for (int i = 0; i < n_parts && !fused_walk; ++i) {
            if (das[i].grouped || parts[i].tb_now || (side && n_parts > 1)) continue; // walked inside its fill kernel / behind its fill already
            HIP_TRY(ctx, score_only ? launch_scores_only(tas[i], tb_stream) : tas[i].packed16 == 6 ? launch_strip_ck_walk(tas[i], max_tl, max_ql, tb_stream) : launch_traceback(tas[i], tb_stream));
        }
        if (pe[3]) HIP_TRY(ctx, hipEventRecord(pe[3], tb_stream));
        if (overlap) {
            HIP_TRY(ctx, hipEventRecord(ctx->tb_done[h], ctx->aux));
            tb_pending[h] = true;
        }
        if (auto_group && !(hooks && hooks->regroup)) { // sorted on the device: this chunk's index arrays are free once its walk is done
            if (side && tb_pending[h]) HIP_TRY(ctx, hipStreamWaitEvent(tb_stream, ctx->tb_done[h], 0)); // (... on either stream: the results a host entry copies out are complete behind this too)
            HIP_TRY(ctx, hipEventRecord(ctx->srt_free[k & 3], tb_stream));
            srt_used[k & 3] = true;
        }
        if (sort_more) { // a host batch sorted on the device: the chunks ahead are prepared BEHIND this chunk's launches
            const int src = sort_until(k + 2);
            if (src != MGL_SW_OK) return src;
        }
        if (hooks) {
            if (n_pending == 2) {
                const int hrc = hooks->after_traceback(pending[0].first, pending[0].count, pending[0].ready);
                if (hrc != MGL_SW_OK) return hrc;
                pending[0] = pending[1];
                n_pending = 1;
            }
            // (chunk k-2, just copied out, was the last user of this event)
            HIP_TRY(ctx, hipEventRecord(ctx->out_ready[k & 1], tb_stream));
            pending[n_pending++] = Pending{first, count, ctx->out_ready[k & 1]};
        }
        ctx->last_stride_words = stride_words;
        ctx->last_chunk_count = auto_group ? 0 : count; // (a chunk sorted by geometry has no caller-order slots to expand)
        ctx->last_half = h;
        ctx->last_rows = rows;
        ctx->last_packed16 = lane_ck ? 5 : use_lane ? 2 : use16 ? 1 : strip16 ? (strip_k ? 6 : 4) : coop16 ? 3 : 0;
        ctx->timing.dp_launches++;
        ctx->timing.tb_launches++;
        ctx->timing.tb_bytes += (lane_ck ? std::min<int64_t>(lane_slots, (count + 127) / 128) : use_lane ? (count + 127) / 128 : use16 ? (count + 1) / 2 : count) * stride_words * 4;
        ctx->timing.packed16 = (use16 || use_lane) ? 1 : 0;
        ctx->timing.fill_kernel = lane_ck || lane_pairs_last > 0 ? MGL_SW_KERNEL_LANE16_CK : use_lane ? MGL_SW_KERNEL_LANE16 : use16 ? MGL_SW_KERNEL_DP16 : strip16 ? MGL_SW_KERNEL_STRIP16 : coop16 ? MGL_SW_KERNEL_COOP16 : coop_waves ? MGL_SW_KERNEL_COOP : rows == 64 ? MGL_SW_KERNEL_DP32_64 : MGL_SW_KERNEL_DP32;
    }
    for (int i = 0; hooks && i < n_pending; ++i) {
        const int hrc = hooks->after_traceback(pending[i].first, pending[i].count, pending[i].ready);
        if (hrc != MGL_SW_OK) return hrc;
    }
    // everything this call enqueued is ordered before whatever the caller enqueues next on `stream`
    for (int h = 0; h < 2; ++h)
        if (tb_pending[h]) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->tb_done[h], 0));
    if (dual) {
        HIP_TRY(ctx, hipEventRecord(ctx->tb_done[1], ctx->fill2));
        HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->tb_done[1], 0));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ws_idle, stream));
    ctx->ws_idle_set = true;
    return MGL_SW_OK;
}

// MGL_SW_FLAG_SHARED_TARGET on the substitution-matrix entry: tiles of 128 consecutive pairs that share their target (and their query
// length) go through sw_dp16_lane_matrix_kernel -- ONE launch of a persistent grid whatever the number of pairs, a region of the
// largest geometry per wave slot, the waves walk their own paths.  kNotTaken: the batch is left to run_device (parameters the
// kernel's byte table cannot hold, a score range beyond 16 bits, a workspace that cannot give every SIMD a wave, score-only calls).
constexpr int kNotTaken = -1000;
int run_shared_target(mgl_sw_ctx *ctx, hipStream_t stream, int64_t n, const SeqSet &tset, const SeqSet &qset, int max_tl, int max_ql, int smax, int smin, int gopen,
                      int gext, int strategy, int32_t *d_offset, Score *d_score, char *d_cigar, int cigar_stride, int32_t *d_cigar_len, int32_t *d_status,
                      bool binary_cigar, const int8_t *d_matrix, const uint8_t *d_code, bool score_only)
{
    if (score_only && !d_score) return kNotTaken;
    if (n < 1 || !tset.data || !tset.off || !qset.data || !qset.off || !d_offset || !d_cigar || cigar_stride < 1 || max_tl < 1 || max_ql < 1 || !strategy_ok(strategy))
        return kNotTaken; // (run_device says what is wrong)
    int m1 = 1, m2 = -1;
    mgl_sw_normalize_params(&m1, &m2, &gopen, &gext);
    if (ctx->precision == 32 || ctx->lane_kernel == 1 || tset.packed2 || qset.packed2 || !lane16_matrix_params_ok(smin, smax, gopen, gext) ||
        !dp16_range_ok(max_tl, max_ql, smax, smin, gopen, gext, strategy) || lane16_matrix_lds_bytes(max_tl) > 64 * 1024)
        return kNotTaken;
    const int64_t tiles = (n + 127) / 128;
    const char *const slots_env = getenv("MGL_SW_DEBUG_LANE_SLOTS"); // (tests: a grid of this many wave slots, so that a few tiles already draw from the counter; read per call)
    const int64_t forced = slots_env ? atoll(slots_env) : 0;
    const int64_t chip = forced > 0 ? forced : (int64_t)ctx->n_cus * 12; // three waves per SIMD (168 registers)
    const int frc = grid_fault_check(ctx);
    if (frc != MGL_SW_OK) return frc;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->ws_idle_set) HIP_TRY(ctx, hipStreamWaitEvent(stream, ctx->ws_idle, 0));
    // ---- every tile's geometry comes back (8 bytes per tile; the one synchronisation of this path: the regions are sized by it)
    const size_t geo_bytes = (size_t)tiles * 8, order_off = (geo_bytes + 255) / 256 * 256, off_off = order_off + ((size_t)tiles * 4 + 255) / 256 * 256;
    HIP_TRY(ctx, ctx->d_grid.reserve(off_off + (size_t)(chip + 1) * 8 + 256));
    unsigned char *const dg = static_cast<unsigned char *>(ctx->d_grid.p);
    HIP_TRY(ctx, launch_tile_geometry(tset, qset, 0, n, reinterpret_cast<int32_t *>(dg), stream));
    const size_t pin_need = off_off + (size_t)(chip + 1) * 8 + 256; // (page-locked, the context's: the copies back need no second wait)
    if (ctx->pin_tiles_cap < pin_need) {
        HIP_TRY(ctx, hipStreamSynchronize(stream));
        if (ctx->pin_tiles) (void)hipHostFree(ctx->pin_tiles);
        ctx->pin_tiles = nullptr;
        ctx->pin_tiles_cap = 0;
        HIP_TRY(ctx, hipHostMalloc(&ctx->pin_tiles, pin_need * 2, hipHostMallocDefault));
        ctx->pin_tiles_cap = pin_need * 2;
    }
    unsigned char *const hp = static_cast<unsigned char *>(ctx->pin_tiles);
    const int32_t *const geo = reinterpret_cast<const int32_t *>(hp);
    HIP_TRY(ctx, hipMemcpyAsync(hp, dg, geo_bytes, hipMemcpyDeviceToHost, stream));
    HIP_TRY(ctx, hipStreamSynchronize(stream));
    std::vector<int64_t> need((size_t)tiles);
    int true_max_tl = 1;
    for (int64_t k = 0; k < tiles; ++k) {
        const int tl = geo[2 * (size_t)k], ql = geo[2 * (size_t)k + 1];
        const bool ok = tl >= 1 && ql >= 1 && tl <= max_tl && ql <= max_ql;
        need[(size_t)k] = ok ? lane16_matrix_region_bytes(tl, ql, score_only) : 0; // (a geometry outside the caller's bounds: the kernel refuses the tile)
        if (ok) true_max_tl = std::max(true_max_tl, tl);
    }
    int32_t *const order = reinterpret_cast<int32_t *>(hp + order_off);
    for (int64_t k = 0; k < tiles; ++k) order[(size_t)k] = (int32_t)k;
    std::stable_sort(order, order + tiles, [&](int32_t x, int32_t y) { return need[(size_t)x] > need[(size_t)y]; });
    // wave slot s starts on the s-th largest tile: its region holds exactly that one, and everything it draws later.  As many slots as the
    // workspace has room for (the sum over the largest tiles), at most what the chip holds
    int64_t slots = 0, total = 0;
    int64_t *const slot_off = reinterpret_cast<int64_t *>(hp + off_off);
    slot_off[0] = 0;
    while (slots < std::min<int64_t>(chip, tiles)) {
        const int64_t r = std::max<int64_t>(need[(size_t)order[(size_t)slots]], 256);
        if (total + r > ctx->ws_limit) break;
        total += r;
        slot_off[++slots] = total;
    }
    if (slots < 1 || (forced <= 0 && slots < std::min<int64_t>(tiles, (int64_t)ctx->n_cus * 4))) return kNotTaken; // (fewer than a wave per SIMD: the kernels that keep less per pair do better)
    HIP_TRY(ctx, ctx->tb[0].reserve((size_t)total));
    // (the page-locked buffer is read by these two copies only: the next call waits for this one's kernels -- ws_idle, and its own look at
    // the geometries -- before it writes there again)
    HIP_TRY(ctx, hipMemcpyAsync(dg + order_off, order, (size_t)tiles * 4, hipMemcpyHostToDevice, stream));
    HIP_TRY(ctx, hipMemcpyAsync(dg + off_off, slot_off, (size_t)(slots + 1) * 8, hipMemcpyHostToDevice, stream));
    if (!ctx->tile_ctr.p) {
        HIP_TRY(ctx, ctx->tile_ctr.reserve((size_t)kTileCounters * kTileCounterWords * sizeof(unsigned)));
        HIP_TRY(ctx, hipMemsetAsync(ctx->tile_ctr.p, 0, (size_t)kTileCounters * kTileCounterWords * sizeof(unsigned), stream));
    }
    if (!ctx->pin_fault) {
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->pin_fault), 64, hipHostMallocDefault));
        memset(ctx->pin_fault, 0, 64);
    }
    if (ctx->profiling != 3) {
        ctx->timing = mgl_sw_timing{};
        ctx->pool_used = 0;
    }
    const int64_t tb_words = 0;
    DpArgs da{};
    da.t = tset;
    da.q = qset;
    da.first = 0;
    da.count = n;
    da.match = smax;
    da.mismatch = smin;
    da.gopen = gopen;
    da.gext = gext;
    da.strategy = strategy;
    da.uni_tl = std::min(max_tl, true_max_tl); // (sizes the kernel's LDS copy of a tile's target)
    da.uni_ql = max_ql;
    da.tb = static_cast<uint32_t *>(ctx->tb[0].p);
    da.tb_stride_words = tb_words;
    da.tile_order = reinterpret_cast<const int32_t *>(dg + order_off);
    da.slot_off = reinterpret_cast<const int64_t *>(dg + off_off);
    da.matrix = d_matrix;
    da.code = d_code;
    da.score_only = score_only ? 1 : 0;
    da.lane_slots = (int)slots;
    if (tiles > slots) {
        da.tile_ctr = static_cast<unsigned *>(ctx->tile_ctr.p) + (size_t)(ctx->tile_seq++ % kTileCounters) * kTileCounterWords;
        void *fault_dev = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&fault_dev, ctx->pin_fault, 0));
        da.grid_fault = static_cast<int32_t *>(fault_dev);
    }
    TbArgs ta{};
    ta.t = tset;
    ta.q = qset;
    ta.first = 0;
    ta.count = n;
    ta.strategy = strategy;
    ta.tb = da.tb;
    ta.tb_stride_words = tb_words;
    ta.packed16 = 2;
    ta.rows_per_stripe = 32;
    ta.strip_rows = 32;
    ta.uni_ql = max_ql;
    ta.match = smax;
    ta.mismatch = smin;
    ta.gopen = gopen;
    ta.gext = gext;
    ta.offset = d_offset;
    ta.score = d_score;
    ta.cigar = d_cigar;
    ta.cigar_stride = cigar_stride;
    ta.binary_cigar = binary_cigar ? 1 : 0;
    ta.cigar_len = d_cigar_len;
    ta.status = d_status;
    hipEvent_t pe[4] = {nullptr, nullptr, nullptr, nullptr};
    if (ctx->profiling) {
        while ((int)ctx->pool.size() < ctx->pool_used + 4) {
            hipEvent_t e = nullptr;
            HIP_TRY(ctx, hipEventCreate(&e));
            ctx->pool.push_back(e);
        }
        for (int i = 0; i < 4; ++i) pe[i] = ctx->pool[(size_t)ctx->pool_used + i];
        ctx->pool_used += 4;
        HIP_TRY(ctx, hipEventRecord(pe[0], stream));
    }
    HIP_TRY(ctx, launch_dp16_lane_matrix(da, ta, stream));
    if (pe[0]) {
        HIP_TRY(ctx, hipEventRecord(pe[1], stream));
        HIP_TRY(ctx, hipEventRecord(pe[2], stream)); // (the walk is inside the fill)
        HIP_TRY(ctx, hipEventRecord(pe[3], stream));
    }
    HIP_TRY(ctx, hipEventRecord(ctx->ws_idle, stream));
    ctx->ws_idle_set = true;
    ctx->last_chunk_count = 0; // (no slot to expand: the flags of a tile are gone with the next one)
    ctx->timing.dp_launches++;
    ctx->timing.tb_launches++;
    ctx->timing.tb_bytes += total;
    ctx->timing.packed16 = 1;
    ctx->timing.fill_kernel = MGL_SW_KERNEL_LANE16_MATRIX;
    return MGL_SW_OK;
}

} // namespace

extern "C" {

int mgl_sw_version(void) { return MGL_SW_VERSION; }

const char *mgl_sw_strerror(int status)
{
    switch (status) {
    case MGL_SW_OK: return "ok";
    case MGL_SW_ERR_BAD_ARG: return "bad argument";
    case MGL_SW_ERR_CIGAR_OVERFLOW: return "CIGAR does not fit the caller's buffer";
    case MGL_SW_ERR_NOMEM: return "out of memory";
    case MGL_SW_ERR_DEVICE: return "no usable HIP device / HIP runtime error";
    case MGL_SW_ERR_UNSUPPORTED: return "geometry not supported by this build";
    default: return "unknown status";
    }
}

int mgl_sw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

int mgl_sw_max_query_len(void) { return 1 << 24; }
int mgl_sw_max_lds_query_len(void) { return max_lds_query_len(); }

void mgl_sw_normalize_params(int *match, int *mismatch, int *gopen, int *gext)
{
    // ..._MicrosoftSmithWaterman.cpp:51-55
    if (*match < 0) *match = -*match;
    if (*mismatch > 0) *mismatch = -*mismatch;
    if (*gopen < 0) *gopen = -*gopen;
    if (*gext < 0) *gext = -*gext;
}

int mgl_sw_ctx_create(int device, mgl_sw_ctx **out)
{
    if (!out) return MGL_SW_ERR_BAD_ARG;
    *out = nullptr;
    const int ndev = mgl_sw_device_count();
    if (ndev <= 0 || device < 0 || device >= ndev) return MGL_SW_ERR_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return MGL_SW_ERR_DEVICE;
    mgl_sw_ctx *ctx = new (std::nothrow) mgl_sw_ctx;
    if (!ctx) return MGL_SW_ERR_NOMEM;
    ctx->device = device;
    {
        // default cap of the (grow-only) workspace: a quarter of the device's memory, at least 4 GiB -- 72 GB on an MI355X
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b / 4 > (size_t)kDefaultWorkspace) ctx->ws_limit = (int64_t)(total_b / 4);
    }
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) ctx->n_cus = cus;
    }
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
        delete ctx;
        return MGL_SW_ERR_DEVICE;
    }
    // the traceback stream gets the lowest priority: its waves should only take what the fill kernel leaves
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    bool ok = hipStreamCreateWithPriority(&ctx->aux, hipStreamNonBlocking, prio_lo) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&ctx->h2d, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&ctx->fill2, hipStreamNonBlocking) == hipSuccess;
    ok = ok && hipStreamCreateWithFlags(&ctx->d2h, hipStreamNonBlocking) == hipSuccess;
    // The runtime spreads a process's streams over four hardware queues, and which of this context's four (fills on two, copies in,
    // copies out) end up sharing one depends on what else the process has created before: behind torch.cuda.set_device the packed host
    // entry ran at 5 120 GCUPS, with the context created first at 5 770 (scripts/host_packed_probe.py) -- a copy stream sat behind a fill
    // stream's kernels.  Its copies are small and its pipeline short, so that entry uses copy streams of the HIGHEST priority, which have
    // queues of their own: 5 590-5 610 whatever the order.  (The ASCII / mixed entry keeps the ordinary ones: its copies are the bulk of
    // its time, and its mixed batches lose 3 % behind high-priority copies.)
    ok = ok && hipStreamCreateWithPriority(&ctx->h2d_hi, hipStreamNonBlocking, prio_hi) == hipSuccess;
    ok = ok && hipStreamCreateWithPriority(&ctx->d2h_hi, hipStreamNonBlocking, prio_hi) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->in_done, hipEventDisableTiming) == hipSuccess;
    for (auto &e : ctx->out_ready) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    for (auto &e : ctx->ev) ok = ok && hipEventCreate(&e) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&ctx->ws_idle, hipEventDisableTiming) == hipSuccess;
    for (auto *set : {ctx->fill_done, ctx->tb_done})
        for (int h = 0; h < 2; ++h) ok = ok && hipEventCreateWithFlags(&set[h], hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        mgl_sw_ctx_destroy(ctx);
        return MGL_SW_ERR_DEVICE;
    }
    *out = ctx;
    return MGL_SW_OK;
}

void mgl_sw_ctx_destroy(mgl_sw_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    if (ctx->aux) (void)hipStreamSynchronize(ctx->aux);
    // EVERY stream of the context is idle before it is destroyed (round 5): the copy streams have moved pageable arrays of the caller's,
    // which the runtime page-locks for the time of the copy and lets go of when the queue is next found idle -- a queue that is
    // destroyed first must not take such a pin with it (the intermittent fault, as far as it is understood)
    for (hipStream_t st : {ctx->h2d, ctx->d2h, ctx->h2d_hi, ctx->d2h_hi, ctx->fill2})
        if (st) (void)hipStreamSynchronize(st);
    // the caller's arrays this context page-locked (mgl_sw_register_host_buffer) and the caller never unregistered: every stream that may
    // still copy from or into them is drained first; left registered they would stay pinned, and a later context registering the same array
    // would be refused (hipErrorHostMemoryAlreadyRegistered)
    if (!ctx->registered.empty()) {
        for (hipStream_t st : {ctx->h2d, ctx->d2h, ctx->h2d_hi, ctx->d2h_hi, ctx->fill2})
            if (st) (void)hipStreamSynchronize(st);
        for (const auto &r : ctx->registered) (void)hipHostUnregister(const_cast<char *>(r.first));
        ctx->registered.clear();
    }
    for (DevBuf *b : {&ctx->tb[0], &ctx->tb[1], &ctx->rec[0], &ctx->rec[1], &ctx->bnd[0], &ctx->bnd[1], &ctx->diag, &ctx->scratch, &ctx->tile_ctr, &ctx->d_t, &ctx->d_toff, &ctx->d_q, &ctx->d_qoff, &ctx->d_tlen, &ctx->d_qlen, &ctx->d_off, &ctx->d_score,
                      &ctx->d_cig, &ctx->d_len, &ctx->d_status, &ctx->d_btr, &ctx->d_any, &ctx->d_matrix})
        b->release();
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &e : ctx->pool)
        if (e) (void)hipEventDestroy(e);
    for (auto *set : {ctx->fill_done, ctx->tb_done})
        for (int h = 0; h < 2; ++h)
            if (set[h]) (void)hipEventDestroy(set[h]);
    if (ctx->pin_gate) (void)hipHostFree(ctx->pin_gate);
    if (ctx->pin_fault) (void)hipHostFree(ctx->pin_fault);
    for (auto &e : ctx->gate_ev)
        if (e) (void)hipEventDestroy(e);
    if (ctx->pin_tiles) (void)hipHostFree(ctx->pin_tiles);
    if (ctx->pin_matrix) (void)hipHostFree(ctx->pin_matrix);
    if (ctx->matrix_copied) (void)hipEventDestroy(ctx->matrix_copied);
    if (ctx->ws_idle) (void)hipEventDestroy(ctx->ws_idle);
    for (int r = 0; r < 3; ++r) {
        if (ctx->pin_res[r]) (void)hipHostFree(ctx->pin_res[r]);
        if (ctx->res_copied[r]) (void)hipEventDestroy(ctx->res_copied[r]);
    }
    if (ctx->d2h) (void)hipStreamDestroy(ctx->d2h);
    if (ctx->h2d_hi) (void)hipStreamDestroy(ctx->h2d_hi);
    if (ctx->d2h_hi) (void)hipStreamDestroy(ctx->d2h_hi);
    for (int h = 0; h < 2; ++h) {
        if (ctx->pin_grp[h]) (void)hipHostFree(ctx->pin_grp[h]);
        if (ctx->grp_copied[h]) (void)hipEventDestroy(ctx->grp_copied[h]);
        ctx->d_grp[h].release();
        if (h == 0) {
            for (int b = 0; b < 4; ++b) {
                ctx->d_srt[b].release();
                if (ctx->srt_free[b]) (void)hipEventDestroy(ctx->srt_free[b]);
                if (ctx->srt_done[b]) (void)hipEventDestroy(ctx->srt_done[b]);
            }
            ctx->d_grid.release();
            if (ctx->pin_total) (void)hipHostFree(ctx->pin_total);
            ctx->pin_total = nullptr;
        }
    }
    if (ctx->pin_in) (void)hipHostFree(ctx->pin_in);
    if (ctx->pin_out) (void)hipHostFree(ctx->pin_out);
    ctx->stage_in.release();
    ctx->stage_out.release();
    if (ctx->in_done) (void)hipEventDestroy(ctx->in_done);
    for (auto &e : ctx->out_ready)
        if (e) (void)hipEventDestroy(e);
    if (ctx->h2d) (void)hipStreamDestroy(ctx->h2d);
    if (ctx->fill2) (void)hipStreamDestroy(ctx->fill2);
    if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *mgl_sw_last_error(const mgl_sw_ctx *ctx) { return ctx ? ctx->err.c_str() : ""; }

// Asynchronous faults of the device entries: they return when their kernels are ENQUEUED, so what a kernel finds out about itself
// later (today: a persistent grid whose tile counter was out of range) can only be reported afterwards.  The caller synchronises its
// stream, then asks here; every later call on the context reports the same (sticky).  The host entries ask by themselves.
int mgl_sw_ctx_check(mgl_sw_ctx *ctx)
{
    if (!ctx) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return grid_fault_check(ctx);
}

int mgl_sw_ctx_set_workspace(mgl_sw_ctx *ctx, int64_t bytes)
{
    if (!ctx || bytes < (1 << 20)) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->ws_limit = bytes;
    return MGL_SW_OK;
}

int mgl_sw_ctx_set_precision(mgl_sw_ctx *ctx, int bits)
{
    if (!ctx || (bits != 0 && bits != 16 && bits != 32)) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->precision = bits;
    return MGL_SW_OK;
}

int mgl_sw_ctx_set_strip_kernel(mgl_sw_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->strip_kernel = mode;
    return MGL_SW_OK;
}

int mgl_sw_ctx_set_carry_memory(mgl_sw_ctx *ctx, int mode)
{
    if (!ctx || (mode != 0 && mode != 1)) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->carry_memory = mode;
    return MGL_SW_OK;
}

int mgl_sw_ctx_set_stripe_rows(mgl_sw_ctx *ctx, int rows)
{
    if (!ctx || (rows != 0 && rows != 16 && rows != 64)) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->stripe_rows = rows;
    return MGL_SW_OK;
}

int mgl_sw_ctx_set_cooperative(mgl_sw_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 16) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->cooperative = mode;
    return MGL_SW_OK;
}

int mgl_sw_ctx_set_small_kernel(mgl_sw_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->small_kernel = mode;
    return MGL_SW_OK;
}

int mgl_sw_ctx_set_lane_checkpoint(mgl_sw_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->lane_checkpoint = mode;
    return MGL_SW_OK;
}

int mgl_sw_ctx_set_lane_kernel(mgl_sw_ctx *ctx, int mode)
{
    if (!ctx || mode < 0 || mode > 2) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->lane_kernel = mode;
    return MGL_SW_OK;
}

int mgl_sw_ctx_set_profiling(mgl_sw_ctx *ctx, int enable)
{
    if (!ctx) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    ctx->profiling = enable < 0 ? 0 : enable;
    if (ctx->profiling == 3) { // a new sum starts here
        ctx->timing = mgl_sw_timing{};
        ctx->pool_used = 0;
    }
    return MGL_SW_OK;
}

int mgl_sw_ctx_get_timing(mgl_sw_ctx *ctx, mgl_sw_timing *out)
{
    if (!ctx || !out) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (ctx->pool_used > 0) {
        HIP_TRY(ctx, hipSetDevice(ctx->device));
        float dp = 0.f, tb = 0.f;
        for (int i = 0; i + 3 < ctx->pool_used; i += 4) {
            float a = 0.f, b = 0.f;
            HIP_TRY(ctx, hipEventSynchronize(ctx->pool[(size_t)i + 3]));
            HIP_TRY(ctx, hipEventElapsedTime(&a, ctx->pool[(size_t)i], ctx->pool[(size_t)i + 1]));
            HIP_TRY(ctx, hipEventElapsedTime(&b, ctx->pool[(size_t)i + 2], ctx->pool[(size_t)i + 3]));
            dp += a;
            tb += b;
        }
        ctx->timing.dp_ms = dp;
        ctx->timing.tb_ms = tb;
        if (ctx->profiling == 2 && ctx->diag.p && ctx->diag_blocks > 0) {
            // in-kernel clock of the last chunk's fill launch
            std::vector<unsigned long long> h((size_t)ctx->diag_blocks * 2);
            HIP_TRY(ctx, hipMemcpy(h.data(), ctx->diag.p, h.size() * 8, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int64_t i = 0; i < ctx->diag_blocks; ++i) {
                cyc += (double)h[2 * i];
                real += (double)h[2 * i + 1];
            }
            if (real > 0) ctx->timing.clock_mhz = (int32_t)(cyc / real * 100.0); // s_memrealtime ticks at 100 MHz
        }
        ctx->pool_used = 0;
    }
    *out = ctx->timing;
    if (ctx->profiling == 3) ctx->timing = mgl_sw_timing{}; // the next calls start a new sum
    return MGL_SW_OK;
}

// (a caller whose header is older or newer than the library's: never more than out_size bytes written, include/mgl_sw.h)
int mgl_sw_explain_sized(mgl_sw_ctx *ctx, int64_t workspace_limit, int64_t n, int max_tl, int max_ql, int match, int mismatch, int gopen, int gext,
                         int strategy, int flags, int packed2, int entry, mgl_sw_plan *out, size_t out_size)
{
    if (!out || out_size < 8) return MGL_SW_ERR_BAD_ARG;
    mgl_sw_plan full{};
    const int rc = mgl_sw_explain(ctx, workspace_limit, n, max_tl, max_ql, match, mismatch, gopen, gext, strategy, flags, packed2, entry, &full);
    memset(out, 0, out_size);
    memcpy(out, &full, std::min(out_size, sizeof full));
    return rc;
}

int mgl_sw_explain(mgl_sw_ctx *ctx, int64_t workspace_limit, int64_t n, int max_tl, int max_ql, int match, int mismatch, int gopen, int gext,
                   int strategy, int flags, int packed2, int entry, mgl_sw_plan *out)
{
    if (!out || n < 1 || workspace_limit < 0) return MGL_SW_ERR_BAD_ARG;
    mgl_sw_ctx defaults; // (no device behind it: the planner reads settings only)
    if (!ctx && workspace_limit > 0) defaults.ws_limit = workspace_limit;
    mgl_sw_ctx *const c = ctx ? ctx : &defaults;
    std::unique_lock<std::mutex> lk(c->mu);
    struct Restore { // (a given context explains with another limit for the length of this call only)
        mgl_sw_ctx *c;
        int64_t ws;
        ~Restore() { c->ws_limit = ws; }
    } restore{c, c->ws_limit};
    if (ctx && workspace_limit > 0) c->ws_limit = workspace_limit;
    // stand-ins for the arrays: the planner looks at which of them exist, never at what they hold
    static const uint8_t dummy8[8] = {0};
    static const int64_t dummy64[2] = {0, 0};
    static const int32_t dummy32[2] = {0, 0};
    const bool uniform = (flags & MGL_SW_FLAG_UNIFORM_GEOMETRY) != 0;
    const SeqSet ts{dummy8, dummy64, packed2 && !uniform ? dummy32 : nullptr, max_tl, packed2 ? 1 : 0},
        qs{dummy8, dummy64, packed2 && !uniform ? dummy32 : nullptr, max_ql, packed2 ? 1 : 0};
    ChunkHooks hooks; // a host entry: the hooks exist (they are not called), and the ASCII one sorts mixed batches itself
    if (entry == 1 && !packed2) hooks.regroup = [](int64_t, int64_t, int, hipStream_t, Regroup *) { return MGL_SW_OK; };
    if (entry == 1 && packed2) hooks.bring_ahead = [](int64_t, int64_t) { return MGL_SW_OK; }; // (mgl_sw_align_batch_2bit: mixed geometries are sorted on the device)
    int32_t off_ = 0;
    char cg_ = 0;
    Score sc_{};
    return run_device(c, nullptr, n, ts, qs, max_tl, max_ql, match, mismatch, gopen, gext, strategy, &off_, &sc_, &cg_, 64, nullptr, nullptr, 0, geom_of(flags),
                      (flags & MGL_SW_FLAG_BINARY_CIGAR) != 0, entry == 1 ? &hooks : nullptr, nullptr, nullptr, (flags & MGL_SW_FLAG_SCORE_ONLY) != 0, out);
}

int mgl_sw_align_batch_device(mgl_sw_ctx *ctx, void *stream, int64_t n, const uint8_t *d_targets,
                              const int64_t *d_t_off, const uint8_t *d_queries, const int64_t *d_q_off, int max_tl,
                              int max_ql, int match, int mismatch, int gopen, int gext, int strategy,
                              int32_t *d_offset_out, mgl_sw_score *d_score_out, char *d_cigar_out, int cigar_stride,
                              int32_t *d_cigar_len_out, int32_t *d_status_out, int flags)
{
    if (!ctx) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const SeqSet ts{d_targets, d_t_off, nullptr, max_tl, 0}, qs{d_queries, d_q_off, nullptr, max_ql, 0};
    return run_device(ctx, static_cast<hipStream_t>(stream), n, ts, qs, max_tl, max_ql, match, mismatch, gopen, gext,
                      strategy, d_offset_out, reinterpret_cast<Score *>(d_score_out), d_cigar_out, cigar_stride,
                      d_cigar_len_out, d_status_out, 0, geom_of(flags),
                      (flags & MGL_SW_FLAG_BINARY_CIGAR) != 0, nullptr, nullptr, nullptr, (flags & MGL_SW_FLAG_SCORE_ONLY) != 0);
}

int mgl_sw_align_batch_device_indexed(mgl_sw_ctx *ctx, void *stream, int64_t n, const uint8_t *d_targets,
                                      const int64_t *d_t_start, const int32_t *d_t_len, const uint8_t *d_queries,
                                      const int64_t *d_q_start, const int32_t *d_q_len, int max_tl, int max_ql, int match,
                                      int mismatch, int gopen, int gext, int strategy, int32_t *d_offset_out,
                                      mgl_sw_score *d_score_out, char *d_cigar_out, int cigar_stride,
                                      int32_t *d_cigar_len_out, int32_t *d_status_out, int flags)
{
    if (!ctx) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!d_t_len || !d_q_len) return fail(ctx, MGL_SW_ERR_BAD_ARG, "indexed batch: length arrays are required");
    const SeqSet ts{d_targets, d_t_start, d_t_len, max_tl, 0}, qs{d_queries, d_q_start, d_q_len, max_ql, 0};
    return run_device(ctx, static_cast<hipStream_t>(stream), n, ts, qs, max_tl, max_ql, match, mismatch, gopen, gext,
                      strategy, d_offset_out, reinterpret_cast<Score *>(d_score_out), d_cigar_out, cigar_stride,
                      d_cigar_len_out, d_status_out, 0, geom_of(flags),
                      (flags & MGL_SW_FLAG_BINARY_CIGAR) != 0, nullptr, nullptr, nullptr, (flags & MGL_SW_FLAG_SCORE_ONLY) != 0);
}

int mgl_sw_align_batch_device_matrix(mgl_sw_ctx *ctx, void *stream, int64_t n, const uint8_t *d_targets,
                                     const int64_t *d_t_off, const int32_t *d_t_len, const uint8_t *d_queries,
                                     const int64_t *d_q_off, const int32_t *d_q_len, int max_tl, int max_ql,
                                     const int8_t *matrix, const uint8_t *code, int gopen, int gext,
                                     int strategy, int32_t *d_offset_out, mgl_sw_score *d_score_out, char *d_cigar_out,
                                     int cigar_stride, int32_t *d_cigar_len_out, int32_t *d_status_out, int flags)
{
    if (!ctx) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (!matrix || !code) return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch_device_matrix: null matrix or code table");
    for (int k = 0; k < 256; ++k)
        if (code[k] >= MATRIX_DIM) return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch_device_matrix: code >= 32");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ctx->d_matrix.reserve(MATRIX_DIM * MATRIX_DIM + 256));
    hipStream_t st = static_cast<hipStream_t>(stream);
    int8_t *dm = static_cast<int8_t *>(ctx->d_matrix.p);
    uint8_t *dc = reinterpret_cast<uint8_t *>(dm) + MATRIX_DIM * MATRIX_DIM;
    if (!ctx->pin_matrix) {
        HIP_TRY(ctx, hipHostMalloc(&ctx->pin_matrix, MATRIX_DIM * MATRIX_DIM + 256, hipHostMallocDefault));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->matrix_copied, hipEventDisableTiming));
    } else {
        HIP_TRY(ctx, hipEventSynchronize(ctx->matrix_copied)); // the previous call's copy has left the staging buffer
    }
    if (ctx->ws_idle_set) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ws_idle, 0)); // the previous call's kernels read d_matrix
    memcpy(ctx->pin_matrix, matrix, MATRIX_DIM * MATRIX_DIM);
    memcpy(static_cast<char *>(ctx->pin_matrix) + MATRIX_DIM * MATRIX_DIM, code, 256);
    HIP_TRY(ctx, hipMemcpyAsync(dm, ctx->pin_matrix, MATRIX_DIM * MATRIX_DIM + 256, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipEventRecord(ctx->matrix_copied, st));
    // with length arrays the offsets are per-pair START positions (a database sequence may serve many pairs)
    const SeqSet ts{d_targets, d_t_off, d_t_len, max_tl, 0}, qs{d_queries, d_q_off, d_q_len, max_ql, 0};
    int cmax = matrix[0], cmin = matrix[0];
    for (int k = 0; k < MATRIX_DIM * MATRIX_DIM; ++k) {
        cmax = std::max<int>(cmax, matrix[k]);
        cmin = std::min<int>(cmin, matrix[k]);
    }
    if (flags & MGL_SW_FLAG_SHARED_TARGET) {
        int rc = kNotTaken;
        try { // (the tiles' geometries and their order are host vectors: no C++ exception crosses the C ABI)
            rc = run_shared_target(ctx, st, n, ts, qs, max_tl, max_ql, cmax, cmin, gopen, gext, strategy, d_offset_out, reinterpret_cast<Score *>(d_score_out),
                                         d_cigar_out, cigar_stride, d_cigar_len_out, d_status_out, (flags & MGL_SW_FLAG_BINARY_CIGAR) != 0, dm, dc,
                                         (flags & MGL_SW_FLAG_SCORE_ONLY) != 0);
        } catch (const std::exception &) {
            return fail(ctx, MGL_SW_ERR_NOMEM, "mgl_sw_align_batch_device_matrix: out of host memory");
        }
        if (rc != kNotTaken) return rc;
    }
    return run_device(ctx, st, n, ts, qs, max_tl, max_ql, cmax, cmin, gopen, gext, strategy, d_offset_out,
                      reinterpret_cast<Score *>(d_score_out), d_cigar_out, cigar_stride, d_cigar_len_out, d_status_out, 0,
                      geom_of(flags),
                      (flags & MGL_SW_FLAG_BINARY_CIGAR) != 0, nullptr, dm, dc, (flags & MGL_SW_FLAG_SCORE_ONLY) != 0);
}

int mgl_sw_align_batch_device_2bit(mgl_sw_ctx *ctx, void *stream, int64_t n, const uint8_t *d_target_bases,
                                   const int64_t *d_t_start, const int32_t *d_t_len, const uint8_t *d_query_bases,
                                   const int64_t *d_q_start, const int32_t *d_q_len, int max_tl, int max_ql, int match,
                                   int mismatch, int gopen, int gext, int strategy, int32_t *d_offset_out,
                                   mgl_sw_score *d_score_out, char *d_cigar_out, int cigar_stride,
                                   int32_t *d_cigar_len_out, int32_t *d_status_out, int flags)
{
    if (!ctx) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const bool uniform = (flags & MGL_SW_FLAG_UNIFORM_GEOMETRY) != 0;
    const bool grouped = (flags & MGL_SW_FLAG_GROUPED_GEOMETRY) != 0;
    // without per-pair length arrays every pair has exactly max_tl / max_ql bases
    if ((!d_t_len || !d_q_len) && !uniform)
        return fail(ctx, MGL_SW_ERR_BAD_ARG, "2-bit batch: length arrays are required unless the geometry is uniform");
    const SeqSet ts{d_target_bases, d_t_start, uniform ? nullptr : d_t_len, max_tl, 1},
        qs{d_query_bases, d_q_start, uniform ? nullptr : d_q_len, max_ql, 1};
    return run_device(ctx, static_cast<hipStream_t>(stream), n, ts, qs, max_tl, max_ql, match, mismatch, gopen, gext,
                      strategy, d_offset_out, reinterpret_cast<Score *>(d_score_out), d_cigar_out, cigar_stride,
                      d_cigar_len_out, d_status_out, 0, uniform ? GEOM_UNIFORM : grouped ? GEOM_GROUPED : GEOM_MIXED, (flags & MGL_SW_FLAG_BINARY_CIGAR) != 0, nullptr, nullptr,
                      nullptr, (flags & MGL_SW_FLAG_SCORE_ONLY) != 0);
}

static int stage_buffers_nolock(mgl_sw_ctx *ctx, size_t in_bytes, size_t out_bytes, void **in, void **out);
static int align_batch_staged_nolock(mgl_sw_ctx *ctx, int n, size_t in_bytes, size_t t_bytes_padded, int max_tl, int max_ql, int match,
                                     int mismatch, int gopen, int gext, int strategy, int cigar_stride, size_t out_bytes, int uniform,
                                     int64_t cells_hint);

namespace {

// every stream a host entry may have put work on (the error paths: nothing of the failed call may still use the context's
// buffers when the next call reuses them)
void drain_streams(mgl_sw_ctx *ctx, hipStream_t st)
{
    if (ctx->d2h) (void)hipStreamSynchronize(ctx->d2h);
    if (ctx->fill2) (void)hipStreamSynchronize(ctx->fill2);
    (void)hipStreamSynchronize(ctx->h2d);
    (void)hipStreamSynchronize(ctx->aux);
    (void)hipStreamSynchronize(st);
}

// How a host entry's results leave the device, chunk by chunk (ChunkHooks::after_traceback).  Pageable arrays: a blocking copy
// into them keeps the calling thread from the next chunk's input copies for a third of the call (10 M pairs: 17 ms out, 29 ms
// in), and a second thread inside the runtime's pageable-copy path makes both slower -- so a chunk's results go to a pinned
// buffer by an asynchronous copy on a stream of their own (the link's other direction) and a helper job moves them on into the
// caller's arrays once they have landed; three buffers in rotation.  Arrays the caller has REGISTERED
// (mgl_sw_register_host_buffer) take the asynchronous copies directly.
struct ResultPump {
    mgl_sw_ctx *ctx;
    int32_t *offset_out;
    mgl_sw_score *score_out;
    char *cigar_out;
    int cigar_stride;
    int32_t *cigar_len_out, *status_out;
    bool direct;
    std::future<int> res_job[3];
    int64_t res_seq = 0;
    ResultPump(mgl_sw_ctx *c, int32_t *off, mgl_sw_score *sc, char *cg, int stride, int32_t *len, int32_t *st, int64_t n)
        : ctx(c), offset_out(off), score_out(sc), cigar_out(cg), cigar_stride(stride), cigar_len_out(len), status_out(st)
    {
        const size_t nn = (size_t)n;
        direct = (c->is_registered(off, nn * 4) && (!sc || c->is_registered(sc, nn * sizeof(mgl_sw_score))) &&
                                 c->is_registered(cg, nn * (size_t)stride) && (!len || c->is_registered(len, nn * 4)) &&
                                 (!st || c->is_registered(st, nn * 4)));
    }
    int join()
    {
        int worst = MGL_SW_OK;
        for (auto &j : res_job)
            if (j.valid()) {
                const int r = j.get();
                if (r != MGL_SW_OK) worst = r;
            }
        return worst;
    }
    int after_traceback(int64_t first, int64_t count, hipEvent_t results_ready)
    {
        const size_t f = (size_t)first, c = (size_t)count;
        if (!ctx->d2h) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->d2h, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamWaitEvent(ctx->d2h, results_ready, 0));
        if (direct) {
            hipStream_t s = ctx->d2h;
            HIP_TRY(ctx, hipMemcpyAsync(offset_out + f, static_cast<int32_t *>(ctx->d_off.p) + f, c * 4, hipMemcpyDeviceToHost, s));
            if (score_out) HIP_TRY(ctx, hipMemcpyAsync(score_out + f, static_cast<Score *>(ctx->d_score.p) + f, c * sizeof(Score), hipMemcpyDeviceToHost, s));
            HIP_TRY(ctx, hipMemcpyAsync(cigar_out + f * cigar_stride, static_cast<char *>(ctx->d_cig.p) + f * cigar_stride, c * (size_t)cigar_stride, hipMemcpyDeviceToHost, s));
            if (cigar_len_out) HIP_TRY(ctx, hipMemcpyAsync(cigar_len_out + f, static_cast<int32_t *>(ctx->d_len.p) + f, c * 4, hipMemcpyDeviceToHost, s));
            if (status_out) HIP_TRY(ctx, hipMemcpyAsync(status_out + f, static_cast<int32_t *>(ctx->d_status.p) + f, c * 4, hipMemcpyDeviceToHost, s));
            return MGL_SW_OK;
        }
        const int r = (int)(res_seq++ % 3);
        if (res_job[r].valid() && res_job[r].get() != MGL_SW_OK) return fail(ctx, MGL_SW_ERR_DEVICE, "mgl_sw_align_batch: copying results out failed");
        // sections of the pinned buffer, 16-byte aligned: offsets | scores | cigars | lengths | statuses
        const size_t o_off = 0, o_sc = o_off + ((c * 4 + 15) & ~(size_t)15), o_cg = o_sc + ((c * sizeof(Score) + 15) & ~(size_t)15),
                     o_len = o_cg + ((c * (size_t)cigar_stride + 15) & ~(size_t)15), o_st = o_len + ((c * 4 + 15) & ~(size_t)15), bytes = o_st + c * 4 + 16;
        if (!ctx->res_copied[r]) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->res_copied[r], hipEventDisableTiming));
        if (bytes > ctx->pin_res_cap[r]) {
            if (ctx->pin_res[r]) (void)hipHostFree(ctx->pin_res[r]);
            ctx->pin_res[r] = nullptr;
            ctx->pin_res_cap[r] = 0;
            HIP_TRY(ctx, hipHostMalloc(&ctx->pin_res[r], bytes + bytes / 8, hipHostMallocDefault));
            ctx->pin_res_cap[r] = bytes + bytes / 8;
        }
        char *const pin = static_cast<char *>(ctx->pin_res[r]);
        HIP_TRY(ctx, hipMemcpyAsync(pin + o_off, static_cast<int32_t *>(ctx->d_off.p) + f, c * 4, hipMemcpyDeviceToHost, ctx->d2h));
        if (score_out) HIP_TRY(ctx, hipMemcpyAsync(pin + o_sc, static_cast<Score *>(ctx->d_score.p) + f, c * sizeof(Score), hipMemcpyDeviceToHost, ctx->d2h));
        HIP_TRY(ctx, hipMemcpyAsync(pin + o_cg, static_cast<char *>(ctx->d_cig.p) + f * cigar_stride, c * (size_t)cigar_stride, hipMemcpyDeviceToHost, ctx->d2h));
        if (cigar_len_out) HIP_TRY(ctx, hipMemcpyAsync(pin + o_len, static_cast<int32_t *>(ctx->d_len.p) + f, c * 4, hipMemcpyDeviceToHost, ctx->d2h));
        if (status_out) HIP_TRY(ctx, hipMemcpyAsync(pin + o_st, static_cast<int32_t *>(ctx->d_status.p) + f, c * 4, hipMemcpyDeviceToHost, ctx->d2h));
        HIP_TRY(ctx, hipEventRecord(ctx->res_copied[r], ctx->d2h));
        hipEvent_t landed = ctx->res_copied[r];
        const int device = ctx->device;
        int32_t *const off_ = offset_out, *const len_ = cigar_len_out, *const st_ = status_out;
        mgl_sw_score *const sc_ = score_out;
        char *const cg_ = cigar_out;
        const int stride = cigar_stride;
        res_job[r] = std::async(std::launch::async, [=]() -> int {
            if (hipSetDevice(device) != hipSuccess || hipEventSynchronize(landed) != hipSuccess) return MGL_SW_ERR_DEVICE;
            memcpy(off_ + f, pin + o_off, c * 4);
            if (sc_) memcpy(sc_ + f, pin + o_sc, c * sizeof(Score));
            memcpy(cg_ + f * stride, pin + o_cg, c * (size_t)stride);
            if (len_) memcpy(len_ + f, pin + o_len, c * 4);
            if (st_) memcpy(st_ + f, pin + o_st, c * 4);
            return MGL_SW_OK;
        });
        return MGL_SW_OK;
    }
};

} // namespace

int mgl_sw_align_batch(mgl_sw_ctx *ctx, int64_t n, const uint8_t *targets, const int64_t *t_off,
                       const uint8_t *queries, const int64_t *q_off, int match, int mismatch, int gopen, int gext,
                       int strategy, int32_t *offset_out, mgl_sw_score *score_out, char *cigar_out, int cigar_stride,
                       int32_t *cigar_len_out)
{
    return mgl_sw_align_batch_status(ctx, n, targets, t_off, queries, q_off, match, mismatch, gopen, gext, strategy,
                                     offset_out, score_out, cigar_out, cigar_stride, cigar_len_out, nullptr);
}

static int align_ascii_direct(mgl_sw_ctx *ctx, int64_t n, const uint8_t *targets, const uint8_t *queries, int tl, int ql, int match, int mismatch, int gopen, int gext,
                              int strategy, int32_t *offset_out, mgl_sw_score *score_out, char *cigar_out, int cigar_stride, int32_t *cigar_len_out, int32_t *status_out,
                              bool *taken);

int mgl_sw_align_batch_status(mgl_sw_ctx *ctx, int64_t n, const uint8_t *targets, const int64_t *t_off,
                              const uint8_t *queries, const int64_t *q_off, int match, int mismatch, int gopen,
                              int gext, int strategy, int32_t *offset_out, mgl_sw_score *score_out, char *cigar_out,
                              int cigar_stride, int32_t *cigar_len_out, int32_t *status_out)
{
    if (!ctx) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (n == 0) return MGL_SW_OK;
    if (n < 0 || !targets || !t_off || !queries || !q_off || !offset_out || !cigar_out || cigar_stride < 1 ||
        !strategy_ok(strategy))
        return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch: bad argument");
    // one pass over the offsets (16 bytes per pair): length ranges, cell count, and whether every aligned block of eight pairs
    // has one geometry (a batch sorted by read length).  Large batches split it over a few threads (160 MB at 10 M pairs:
    // 15 ms on one core, ahead of the first launch)
    struct Scan {
        int64_t lo_t = INT64_MAX, hi_t = 0, lo_q = INT64_MAX, hi_q = 0, cells = 0, ungrouped = 0;
    };
    auto scan_range = [&](int64_t a, int64_t b) {
        Scan r;
        for (int64_t k = a; k < b; ++k) {   // branch-free so that it vectorises
            const int64_t tl = t_off[k + 1] - t_off[k], ql = q_off[k + 1] - q_off[k];
            r.lo_t = std::min(r.lo_t, tl);
            r.hi_t = std::max(r.hi_t, tl);
            r.lo_q = std::min(r.lo_q, ql);
            r.hi_q = std::max(r.hi_q, ql);
            r.cells += tl * ql;
            const int64_t kp = k > 0 ? k - 1 : 0;
            const int64_t d = ((t_off[kp + 1] - t_off[kp]) ^ tl) | ((q_off[kp + 1] - q_off[kp]) ^ ql);
            r.ungrouped |= (k & 7) != 0 ? d : 0;
        }
        return r;
    };
    Scan sc;
    if (n >= (1 << 21)) {
        constexpr int kParts = 8;
        std::future<Scan> part[kParts];
        for (int p = 0; p < kParts; ++p)
            part[p] = std::async(std::launch::async, scan_range, n * p / kParts, n * (p + 1) / kParts);
        for (int p = 0; p < kParts; ++p) {
            const Scan r = part[p].get();
            sc.lo_t = std::min(sc.lo_t, r.lo_t);
            sc.hi_t = std::max(sc.hi_t, r.hi_t);
            sc.lo_q = std::min(sc.lo_q, r.lo_q);
            sc.hi_q = std::max(sc.hi_q, r.hi_q);
            sc.cells += r.cells;
            sc.ungrouped |= r.ungrouped;
        }
    } else {
        sc = scan_range(0, n);
    }
    const int64_t lo_t = sc.lo_t, hi_t = sc.hi_t, lo_q = sc.lo_q, hi_q = sc.hi_q, cells = sc.cells, ungrouped = sc.ungrouped;
    // the reference reads out of bounds for empty sequences (sw.cpp:162-163,184): rejected here
    if (lo_t < 1 || lo_q < 1 || hi_t > 0x3fffffff || hi_q > 0x3fffffff)
        return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch: sequence length < 1 or too large");
    const int max_tl = (int)hi_t, max_ql = (int)hi_q;
    const int uniform = (lo_t == hi_t && lo_q == hi_q) ? GEOM_UNIFORM : ungrouped == 0 ? GEOM_GROUPED : GEOM_MIXED; // one geometry per batch, or per block of eight
    const size_t t_bytes = (size_t)(t_off[n] - t_off[0]), q_bytes = (size_t)(q_off[n] - q_off[0]);
    if (t_off[0] != 0 || q_off[0] != 0)
        return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch: offsets must start at 0");

    // ---- small batches are latency bound: one pinned staging buffer each way instead of eleven copy commands
    {
        const size_t offs = (size_t)(n + 1) * 8, t_pad = (t_bytes + 7) & ~(size_t)7, q_pad = (q_bytes + 7) & ~(size_t)7;
        const size_t in_bytes = 2 * offs + t_pad + q_pad;
        const size_t out_bytes = (size_t)n * (12 + sizeof(mgl_sw_score)) + (size_t)n * (size_t)cigar_stride;
        if (n <= (1 << 20) && in_bytes + out_bytes <= (1u << 20)) {
            void *in = nullptr, *out = nullptr;
            int rc = stage_buffers_nolock(ctx, in_bytes, out_bytes, &in, &out);
            if (rc != MGL_SW_OK) return rc;
            uint8_t *hin = static_cast<uint8_t *>(in);
            memcpy(hin, t_off, offs);
            memcpy(hin + offs, q_off, offs);
            memcpy(hin + 2 * offs, targets, t_bytes);
            memcpy(hin + 2 * offs + t_pad, queries, q_bytes);
            rc = align_batch_staged_nolock(ctx, (int)n, in_bytes, t_pad, max_tl, max_ql, match, mismatch, gopen, gext, strategy,
                                           cigar_stride, out_bytes, uniform, cells);
            if (rc != MGL_SW_OK) return rc;
            const int32_t *off_ = static_cast<const int32_t *>(out), *len_ = off_ + n, *status_ = len_ + n;
            const mgl_sw_score *score_ = reinterpret_cast<const mgl_sw_score *>(status_ + n);
            const char *cig_ = reinterpret_cast<const char *>(score_ + n);
            memcpy(offset_out, off_, (size_t)n * 4);
            if (score_out) memcpy(score_out, score_, (size_t)n * sizeof(mgl_sw_score));
            memcpy(cigar_out, cig_, (size_t)n * (size_t)cigar_stride);
            if (cigar_len_out) memcpy(cigar_len_out, len_, (size_t)n * 4);
            int32_t any = 0;
            for (int64_t k = 0; k < n; ++k) any = std::max(any, status_[k]);
            if (status_out) {
                memcpy(status_out, status_, (size_t)n * 4);
                return MGL_SW_OK;
            }
            if (any != 0) return fail(ctx, any, any == MGL_SW_ERR_CIGAR_OVERFLOW ? "a CIGAR did not fit cigar_stride" : "device error");
            return MGL_SW_OK;
        }
    }

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // ---- the direct form (round 5: the reference's own wire format, ASCII bases in host memory, MicrosoftSmithWaterman.java:71-86): one
    // geometry, every array page-locked by the caller, the checkpointed lane kernel in one launch -- the grid waits at its gate while the
    // copy engines bring the bases in, the results are written by the waves into the caller's arrays
    if (uniform == GEOM_UNIFORM) {
        bool taken = false;
        const int drc = align_ascii_direct(ctx, n, targets, queries, max_tl, max_ql, match, mismatch, gopen, gext, strategy, offset_out, score_out, cigar_out, cigar_stride,
                                           cigar_len_out, status_out, &taken);
        if (taken) return drc;
    }
    HIP_TRY(ctx, ctx->d_t.reserve(t_bytes));
    HIP_TRY(ctx, ctx->d_q.reserve(q_bytes));
    HIP_TRY(ctx, ctx->d_toff.reserve((size_t)(n + 1) * 8));
    HIP_TRY(ctx, ctx->d_qoff.reserve((size_t)(n + 1) * 8));
    HIP_TRY(ctx, ctx->d_off.reserve((size_t)n * 4));
    HIP_TRY(ctx, ctx->d_score.reserve((size_t)n * sizeof(Score)));
    HIP_TRY(ctx, ctx->d_cig.reserve((size_t)n * cigar_stride));
    HIP_TRY(ctx, ctx->d_len.reserve((size_t)n * 4));
    HIP_TRY(ctx, ctx->d_status.reserve((size_t)n * 4));
    HIP_TRY(ctx, ctx->d_any.reserve(64));
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_any.p, 0, 4, st));
    // everything moves chunk by chunk, overlapped with the kernels: a chunk's offsets (16 bytes per pair) go with its bases

    ChunkHooks hooks;
    hooks.d_status_any = static_cast<int32_t *>(ctx->d_any.p);
    const bool host_timing = debug_knobs().host_timing;   // diagnostic: where the calling thread waits
    double t_in = 0, t_out = 0;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now();
    // (Inputs stay blocking pageable copies on this thread.  Staging them through a pinned ring by helper jobs, a chunk ahead, was
    // measured: the thread then waits 11 ms instead of 43 per 4 M mixed pairs, but the chunk sort it used to hide behind those
    // copies becomes the critical path and the helpers compete with it for the box's cores -- 64 ms per call against 54; uniform
    // batches, kernel bound, gain nothing: 3 505 against 3 477 GCUPS.)
    hooks.before_fill = [&](int64_t first, int64_t count, hipStream_t fill_stream) -> int {
        const double t0 = now();
        const int64_t ta = t_off[first], tb = t_off[first + count], qa = q_off[first], qb = q_off[first + count];
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<int64_t *>(ctx->d_toff.p) + first, t_off + first, (size_t)(count + 1) * 8, hipMemcpyHostToDevice, ctx->h2d));
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<int64_t *>(ctx->d_qoff.p) + first, q_off + first, (size_t)(count + 1) * 8, hipMemcpyHostToDevice, ctx->h2d));
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<uint8_t *>(ctx->d_t.p) + ta, targets + ta, (size_t)(tb - ta),
                                    hipMemcpyHostToDevice, ctx->h2d));
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<uint8_t *>(ctx->d_q.p) + qa, queries + qa, (size_t)(qb - qa),
                                    hipMemcpyHostToDevice, ctx->h2d));
        HIP_TRY(ctx, hipEventRecord(ctx->in_done, ctx->h2d));
        HIP_TRY(ctx, hipStreamWaitEvent(fill_stream, ctx->in_done, 0));
        t_in += now() - t0;
        return MGL_SW_OK;
    };
    ResultPump pump(ctx, offset_out, score_out, cigar_out, cigar_stride, cigar_len_out, status_out, n);
    hooks.after_traceback = [&](int64_t first, int64_t count, hipEvent_t results_ready) -> int {
        const double t0 = now();
        const int r = pump.after_traceback(first, count, results_ready);
        t_out += now() - t0;
        return r;
    };

    // ---- a batch of mixed geometries: every chunk is sorted by (tl, ql) here, on the host, while the GPU works on the
    // chunk before it -- a counting sort over the (tl, ql) grid of the batch, full blocks of eight pairs of one geometry
    // first (packed kernel), the left-over pairs behind them (int32 kernel); run_device decides whether it applies
    const int64_t range_t = hi_t - lo_t + 1, range_q = hi_q - lo_q + 1;
    const bool auto_group_on = debug_knobs().auto_group;
    std::vector<int32_t> grid_pos, grid_nfull, grid_full, grid_rest, grid_nlane, grid_lane;
    int64_t lane_total[2] = {0, 0};  // per half: pairs in the leading blocks of 128 (written by build, read after the job is joined)
    std::future<int64_t> next_job;   // the sort of the NEXT chunk runs on a helper thread while this thread is inside the
    int64_t next_first = -1;         // (blocking) pageable copies of the current one
    // (Sorting on the DEVICE instead, like a device-resident batch's chunks -- all offsets first, launch_regroup on a stream of its
    // own two chunks ahead -- was measured and is no faster: 53.5 ms per 4 M mixed pairs against 53.7.  Either way the calling
    // thread spends 42 of them inside its blocking input copies, 1.6 GB at 38 GB/s beside the result traffic.)
    if (uniform == GEOM_MIXED && auto_group_on && n >= 1024 && range_t * range_q <= (1ll << 20)) {
        const size_t cells_n = (size_t)(range_t * range_q);
        grid_pos.resize(cells_n);
        grid_nfull.resize(cells_n);
        grid_full.resize(cells_n);
        grid_rest.resize(cells_n);
        grid_nlane.resize(cells_n);
        grid_lane.resize(cells_n);
        // slot arrays of a chunk in pinned memory: int64 t_start | int64 q_start | int64 dest | int32 t_len | int32 q_len
        auto build = [&, cells_n](int64_t first, int64_t count, int h, bool lane_blocks) -> int64_t {
            if (hipSetDevice(ctx->device) != hipSuccess) return -1;
            if (hipEventSynchronize(ctx->grp_copied[h]) != hipSuccess) return -1; // the copy of chunk k-2 has left the pinned buffer
            int64_t *ts_ = static_cast<int64_t *>(ctx->pin_grp[h]), *qs_ = ts_ + count, *dest_ = qs_ + count;
            int32_t *tl_ = reinterpret_cast<int32_t *>(dest_ + count), *ql_ = tl_ + count;
            auto cell = [&](int64_t k) { return (size_t)((t_off[k + 1] - t_off[k] - lo_t) * range_q + (q_off[k + 1] - q_off[k] - lo_q)); };
            // counting sort over the (tl, ql) grid: a cell's first (count & ~7) pairs are its full blocks of eight (grouped part,
            // cells in grid order), the others its left-over pairs (behind all the full blocks, in grid order too)
            std::fill(grid_pos.begin(), grid_pos.end(), 0);
            for (int64_t k = first; k < first + count; ++k) ++grid_pos[cell(k)];
            // (lane_blocks: a cell's full blocks of 128 pairs before everything else -- whole waves of the checkpointed lane kernel)
            int64_t full_total = 0, rest_total = 0, lane_sum = 0;
            for (size_t c = cells_n; c-- > 0;) { // (in reverse grid order: the waves with the longest queries first, they take longest)
                grid_nlane[c] = lane_blocks ? grid_pos[c] & ~127 : 0;
                grid_lane[c] = (int32_t)lane_sum;
                lane_sum += grid_nlane[c];
            }
            full_total = lane_sum;
            for (size_t c = 0; c < cells_n; ++c) {
                grid_nfull[c] = grid_pos[c] & ~7;
                grid_full[c] = (int32_t)full_total;
                full_total += grid_nfull[c] - grid_nlane[c];
            }
            lane_total[h] = lane_sum;
            for (size_t c = 0; c < cells_n; ++c) {
                grid_rest[c] = (int32_t)(full_total + rest_total);
                rest_total += grid_pos[c] & 7;
                grid_pos[c] = 0;
            }
            for (int64_t k = first; k < first + count; ++k) {
                const size_t c = cell(k);
                const int32_t p = grid_pos[c]++;
                const int64_t slot = p < grid_nlane[c]   ? (int64_t)grid_lane[c] + p
                                     : p < grid_nfull[c] ? (int64_t)grid_full[c] + (p - grid_nlane[c])
                                                         : (int64_t)grid_rest[c] + (p - grid_nfull[c]);
                ts_[slot] = t_off[k];
                qs_[slot] = q_off[k];
                dest_[slot] = k;
                tl_[slot] = (int32_t)(t_off[k + 1] - t_off[k]);
                ql_[slot] = (int32_t)(q_off[k + 1] - q_off[k]);
            }
            return full_total;
        };
        hooks.regroup = [&, build](int64_t first, int64_t count, int h, hipStream_t fill_stream, Regroup *out) -> int {
            const size_t bytes = (size_t)count * 32;
            for (int hh = 0; hh < 2; ++hh) { // both halves sized for the first (= largest) chunk, events made, before any helper runs
                if (!ctx->grp_copied[hh]) {
                    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->grp_copied[hh], hipEventDisableTiming));
                    HIP_TRY(ctx, hipEventRecord(ctx->grp_copied[hh], ctx->h2d));
                }
                if (bytes > ctx->pin_grp_cap[hh]) {
                    if (next_job.valid()) next_job.wait();
                    HIP_TRY(ctx, hipEventSynchronize(ctx->grp_copied[hh]));
                    if (ctx->pin_grp[hh]) (void)hipHostFree(ctx->pin_grp[hh]);
                    ctx->pin_grp[hh] = nullptr;
                    ctx->pin_grp_cap[hh] = 0;
                    HIP_TRY(ctx, hipHostMalloc(&ctx->pin_grp[hh], bytes, hipHostMallocDefault));
                    ctx->pin_grp_cap[hh] = bytes;
                }
                HIP_TRY(ctx, ctx->d_grp[hh].reserve(bytes));
            }
            int64_t ng;
            if (next_job.valid() && next_first == first) {
                ng = next_job.get();
            } else {
                if (next_job.valid()) next_job.wait();
                ng = build(first, count, h, out->lane_blocks);
            }
            if (ng < 0) return fail(ctx, MGL_SW_ERR_DEVICE, "mgl_sw_align_batch: sorting a chunk by geometry failed");
            HIP_TRY(ctx, hipMemcpyAsync(ctx->d_grp[h].p, ctx->pin_grp[h], bytes, hipMemcpyHostToDevice, ctx->h2d));
            HIP_TRY(ctx, hipEventRecord(ctx->grp_copied[h], ctx->h2d));
            HIP_TRY(ctx, hipStreamWaitEvent(fill_stream, ctx->grp_copied[h], 0));
            // the next chunk (the other half's buffers) is sorted while this thread copies and launches
            next_first = first + count;
            if (next_first < n) next_job = std::async(std::launch::async, build, next_first, std::min(count, n - next_first), h ^ 1, out->lane_blocks);
            const int64_t *d = static_cast<const int64_t *>(ctx->d_grp[h].p);
            out->d_t_start = d;
            out->d_q_start = d + count;
            out->d_dest = d + 2 * count;
            out->d_t_len = reinterpret_cast<const int32_t *>(d + 3 * count);
            out->d_q_len = out->d_t_len + count;
            out->n_grouped = ng;
            out->n_lane = lane_total[h];
            return MGL_SW_OK;
        };
    }

    const SeqSet ts{static_cast<const uint8_t *>(ctx->d_t.p), static_cast<const int64_t *>(ctx->d_toff.p), nullptr, max_tl, 0},
        qs{static_cast<const uint8_t *>(ctx->d_q.p), static_cast<const int64_t *>(ctx->d_qoff.p), nullptr, max_ql, 0};
    int rc = run_device(ctx, st, n, ts, qs, max_tl, max_ql, match, mismatch, gopen, gext, strategy,
                        static_cast<int32_t *>(ctx->d_off.p), static_cast<Score *>(ctx->d_score.p),
                        static_cast<char *>(ctx->d_cig.p), cigar_stride, static_cast<int32_t *>(ctx->d_len.p),
                        static_cast<int32_t *>(ctx->d_status.p), cells, uniform, false, &hooks);
    if (next_job.valid()) next_job.wait();
    const int res_rc = pump.join(); // (the helper jobs hold references to nothing of this frame, but their copies must have landed)
    if (rc == MGL_SW_OK && res_rc != MGL_SW_OK) rc = fail(ctx, MGL_SW_ERR_DEVICE, "mgl_sw_align_batch: copying results out failed");
    if (rc != MGL_SW_OK) {
        drain_streams(ctx, st); // nothing of this call may still be using the workspace when the next call reuses it
        return rc;
    }
    const double t_enq = now();
    HIP_TRY(ctx, hipStreamSynchronize(ctx->h2d));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->aux));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ctx->d2h) HIP_TRY(ctx, hipStreamSynchronize(ctx->d2h)); // (results copied straight into registered arrays)
    if (const int frc = grid_fault_check(ctx)) return frc;
    if (host_timing)
        fprintf(stderr, "[mgl_sw] host entry: %.1f ms enqueue (%.1f ms in input copies, %.1f ms in result copies), %.1f ms drain\n",
                (t_enq - t_begin) * 1e3, t_in * 1e3, t_out * 1e3, (now() - t_enq) * 1e3);
    if (status_out) return MGL_SW_OK;
    int32_t any = 0;
    HIP_TRY(ctx, hipMemcpy(&any, ctx->d_any.p, 4, hipMemcpyDeviceToHost));
    if (any != 0) return fail(ctx, any, "a CIGAR did not fit cigar_stride");
    return MGL_SW_OK;
}

// ---- the caller's arrays, page-locked (hipHostRegister): copies from and into them are true asynchronous DMA
int mgl_sw_register_host_buffer(mgl_sw_ctx *ctx, void *ptr, size_t bytes)
{
    if (!ctx || !ptr || bytes == 0) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {   // already on this context's list: nothing to do.  (What the RUNTIME reports about the address is not asked here: registration is
        // this context's own bookkeeping, and what it registers it unregisters.)
        const char *c = static_cast<const char *>(ptr);
        for (const auto &r : ctx->registered)
            if (c >= r.first && c + bytes <= r.first + r.second) return MGL_SW_OK;
    }
    const hipError_t e = hipHostRegister(ptr, bytes, hipHostRegisterDefault);
    if (e == hipErrorHostMemoryAlreadyRegistered) { // page-locked by the caller's own means: usable as it is, not this context's to undo
        (void)hipGetLastError();
        return MGL_SW_OK;
    }
    if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? MGL_SW_ERR_NOMEM : MGL_SW_ERR_DEVICE, std::string("hipHostRegister: ") + hipGetErrorString(e));
    ctx->registered.emplace_back(static_cast<const char *>(ptr), bytes);
    return MGL_SW_OK;
}

int mgl_sw_unregister_host_buffer(mgl_sw_ctx *ctx, void *ptr)
{
    if (!ctx || !ptr) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    for (size_t i = 0; i < ctx->registered.size(); ++i)
        if (ctx->registered[i].first == static_cast<const char *>(ptr)) {
            HIP_TRY(ctx, hipSetDevice(ctx->device));
            drain_streams(ctx, ctx->stream); // no copy of a finished call may still be in flight
            // (a registration the runtime does NOT let go of must not be forgotten here: it would outlive the caller's array, and whatever
            // the allocator puts at that address next would look page-locked to the runtime -- round 5's intermittent fault, DESIGN.md 10,
            // is of that kind as far as it is understood)
            hipError_t e = hipHostUnregister(ptr);
            if (e != hipSuccess) {
                (void)hipGetLastError();
                (void)hipDeviceSynchronize();
                e = hipHostUnregister(ptr);
            }
            if (e != hipSuccess) return fail(ctx, MGL_SW_ERR_DEVICE, std::string("hipHostUnregister: ") + hipGetErrorString(e));
            ctx->registered.erase(ctx->registered.begin() + (long)i);
            return MGL_SW_OK;
        }
    return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_unregister_host_buffer: not a registered buffer");
}

// ---- the DIRECT form of mgl_sw_align_batch_2bit (round 4).  The chunked form below launches a kernel per chunk and copies every chunk's
// results back; traced (profiles/r04_host_timeline.txt), its result copies into page-locked arrays run as blit KERNELS, and behind a
// persistent grid that holds every wave slot of the chip they waited for the grid's end: 11 ms of copies after the last kernel.  Here
// nothing but the copy ENGINES and one grid works: the inputs go chunk by chunk into device memory, the host moves the gate word on as
// each chunk has landed, a wave waits with a tile until its pairs AND the next tile's are there (its aligned loads may touch the first
// line of the pair behind its last: no line of an input array is ever fetched before its bytes have arrived, and the caches are empty
// of them when the grid starts), and the results are written by the waves themselves, in whole lines out of LDS, into the caller's
// arrays.  (Inputs in fine-grained memory, read uncached, were measured first: 77 ms per 10 M pairs against 67 for the chunked form.)  Taken when every array is page-locked (mgl_sw_register_host_buffer), the batch has
// one geometry and plans as one launch of the checkpointed lane kernel, and the result arrays allow whole-line stores; *taken says so.
// A gate that times out (the copies did not progress while the grid was resident) switches the form off for this context and hands the
// call to the chunked form.
static int align_2bit_direct(mgl_sw_ctx *ctx, int64_t n, const uint8_t *target_bases, size_t t_bytes, const int64_t *t_start, const int32_t *t_len,
                             const uint8_t *query_bases, size_t q_bytes, const int64_t *q_start, const int32_t *q_len, int max_tl, int max_ql, int match, int mismatch,
                             int gopen, int gext, int strategy, int32_t *offset_out, mgl_sw_score *score_out, char *cigar_out, int cigar_stride, int32_t *cigar_len_out,
                             int32_t *status_out, int flags, int64_t target_base_count, int64_t query_base_count, bool *taken)
{
    *taken = false;
    const size_t nn = (size_t)n;
    const bool uniform = (flags & MGL_SW_FLAG_UNIFORM_GEOMETRY) != 0;
    const bool timing = debug_knobs().host_timing;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_enter = now();
    const char *const off_env = getenv("MGL_SW_DEBUG_HOST_DIRECT"); // (0: always the chunked form; read per call: tests compare the two)
    if (ctx->direct_broken || !uniform || (flags & MGL_SW_FLAG_BINARY_CIGAR) || (off_env && atoi(off_env) == 0)) return MGL_SW_OK;
    if (!(ctx->is_registered(target_bases, t_bytes) && ctx->is_registered(query_bases, q_bytes) && ctx->is_registered(t_start, nn * 8) && ctx->is_registered(q_start, nn * 8) &&
          ctx->is_registered(offset_out, nn * 4) && (!score_out || ctx->is_registered(score_out, nn * sizeof(mgl_sw_score))) &&
          ctx->is_registered(cigar_out, nn * (size_t)cigar_stride) && (!cigar_len_out || ctx->is_registered(cigar_len_out, nn * 4)) &&
          (!status_out || ctx->is_registered(status_out, nn * 4))))
        return MGL_SW_OK;
    // the device's view of the caller's result arrays
    void *d_off = nullptr, *d_sc = nullptr, *d_cg = nullptr, *d_len = nullptr, *d_st = nullptr;
    if (hipHostGetDevicePointer(&d_off, offset_out, 0) != hipSuccess || hipHostGetDevicePointer(&d_cg, cigar_out, 0) != hipSuccess ||
        (score_out && hipHostGetDevicePointer(&d_sc, score_out, 0) != hipSuccess) || (cigar_len_out && hipHostGetDevicePointer(&d_len, cigar_len_out, 0) != hipSuccess) ||
        (status_out && hipHostGetDevicePointer(&d_st, status_out, 0) != hipSuccess)) {
        (void)hipGetLastError();
        return MGL_SW_OK;
    }
    // does the batch plan as ONE launch of the checkpointed lane kernel whose results can leave in whole lines?
    int m_ = match, x_ = mismatch, o_ = gopen, e_ = gext;
    mgl_sw_normalize_params(&m_, &x_, &o_, &e_);
    {
        const SeqSet ts{reinterpret_cast<const uint8_t *>(8), reinterpret_cast<const int64_t *>(8), nullptr, max_tl, 1}, qs{reinterpret_cast<const uint8_t *>(8), reinterpret_cast<const int64_t *>(8), nullptr, max_ql, 1};
        BatchPlan P{};
        const std::string err = ctx->err;
        const int prc = plan_batch(ctx, n, ts, qs, max_tl, max_ql, m_, x_, o_, e_, strategy, static_cast<const Score *>(d_sc), static_cast<const char *>(d_cg), cigar_stride, GEOM_UNIFORM,
                                   false, nullptr, nullptr, false, P);
        ctx->err = err;
        TbArgs probe{};
        probe.cigar = static_cast<char *>(d_cg);
        probe.cigar_stride = cigar_stride;
        probe.offset = static_cast<int32_t *>(d_off);
        probe.score = static_cast<Score *>(d_sc);
        probe.cigar_len = static_cast<int32_t *>(d_len);
        probe.status = static_cast<int32_t *>(d_st);
        if (prc != MGL_SW_OK || !P.lane_ck || P.auto_group || P.chunk < n || !lane_ck_coalesced_ok(probe)) return MGL_SW_OK;
    }
    *taken = true;
    const double t_planned = now();
    // device memory for the inputs, the gate word, the chunk events
    HIP_TRY(ctx, ctx->d_t.reserve(t_bytes + 8));
    HIP_TRY(ctx, ctx->d_q.reserve(q_bytes + 8));
    HIP_TRY(ctx, ctx->d_toff.reserve(nn * 8));
    HIP_TRY(ctx, ctx->d_qoff.reserve(nn * 8));
    HIP_TRY(ctx, ctx->d_any.reserve(64));
    if (!ctx->pin_gate) {
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->pin_gate), 64, hipHostMallocDefault));
        memset(ctx->pin_gate, 0, 64);
    }
    void *gate_dev = nullptr;
    HIP_TRY(ctx, hipHostGetDevicePointer(&gate_dev, ctx->pin_gate, 0));
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipStreamSynchronize(st)); // (nothing of an earlier call reads the gate or the inputs any more)
    __atomic_store_n(&ctx->pin_gate[0], (int64_t)0, __ATOMIC_RELEASE);
    __atomic_store_n(reinterpret_cast<int32_t *>(&ctx->pin_gate[1]), 0, __ATOMIC_RELEASE);
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_any.p, 0, 64, st)); // (the status word and, 16 bytes on, the gate's mirror)
    // the chunks: first what gives every wave of the grid its first tile (and the tile behind it: the gate's margin) -- traced with
    // chunks doubling from 32 k pairs: the sixteen copy commands of the first four cost the waves 1.9 ms at the gate -- then a million
    // pairs at a time
    std::vector<int64_t> ends;
    const char *const dce = getenv("MGL_SW_DEBUG_DIRECT_CHUNK"); // (pairs per later chunk; tests make a batch of a few thousand pairs cross many gates with it; read per call)
    const int64_t later = dce ? std::max<int64_t>(128, atoll(dce)) : (int64_t)1 << 20;
    for (int64_t first = 0, c = ((int64_t)ctx->n_cus * LANE_CK_WAVES_PER_CU + 2) * 128; first < n; c = later) {
        first = n - (first + c) < later / 4 ? n : first + c; // (no small last chunk: small copies are blit kernels, see below)
        ends.push_back(first);
    }
    while (ctx->gate_ev.size() < ends.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->gate_ev.push_back(e);
    }
    // the grid first: its waves wait at the gate, and everything below -- the host's part of 56 copy commands takes about as long as the
    // copies themselves, 8 ms per 10 M pairs -- happens beside it (measured with the copies enqueued first: 76 ms per call against 67 for
    // the chunked form, the grid itself 63)
    const SeqSet ts{static_cast<const uint8_t *>(ctx->d_t.p), static_cast<const int64_t *>(ctx->d_toff.p), nullptr, max_tl, 1},
        qs{static_cast<const uint8_t *>(ctx->d_q.p), static_cast<const int64_t *>(ctx->d_qoff.p), nullptr, max_ql, 1};
    ctx->cur_gate = static_cast<const int64_t *>(gate_dev);
    ctx->cur_gate_dev = reinterpret_cast<int64_t *>(static_cast<char *>(ctx->d_any.p) + 16);
    ctx->direct_status_any = static_cast<int32_t *>(ctx->d_any.p);
    int rc = run_device(ctx, st, n, ts, qs, max_tl, max_ql, match, mismatch, gopen, gext, strategy, static_cast<int32_t *>(d_off), static_cast<Score *>(d_sc),
                        static_cast<char *>(d_cg), cigar_stride, static_cast<int32_t *>(d_len), static_cast<int32_t *>(d_st), n * (int64_t)max_tl * max_ql, GEOM_UNIFORM);
    ctx->cur_gate = nullptr;
    ctx->cur_gate_dev = nullptr;
    ctx->direct_status_any = nullptr;
    // (from here on a grid may be waiting at the gate: every way out that is not the good one CALLS IT OFF -- a negative gate: the waves
    // leave at their next look, those inside a tile after it; round 4 opened the gate to n here, and the grid then worked through pairs
    // whose index arrays and bases had not been copied or checked: stale or uninitialised offsets followed into device memory)
    auto bail = [&](int code) -> int {
        __atomic_store_n(&ctx->pin_gate[0], (int64_t)-1, __ATOMIC_RELEASE);
        drain_streams(ctx, st);
        return code;
    };
    if (rc != MGL_SW_OK) return bail(rc);
    const double t_launched = now();
    size_t t_done = 0, q_done = 0, landed = 0;
    // The index arrays are checked HERE, chunk by chunk beside the running grid, instead of in one pass in front of everything (2.5 ms per
    // 10 M pairs on eight threads): every pair inside its packed array, and how far into each array the chunk's pairs reach -- that much
    // of the array travels with the chunk (pairs packed back to back: a slice; windows into a genome, in any order: all of it with the
    // first chunk).  A chunk's gate opens only when its pairs have passed the check and its bytes have landed, so the grid never
    // follows an index nobody has looked at.  Eight threads per chunk (one thread: 15 ms per 10 M pairs, and the grid waited for it).
    struct Reach {
        bool bad = false;
        int64_t t_end = 0, q_end = 0;
    };
    auto check_range = [&](int64_t lo, int64_t hi) {
        Reach r;
        for (int64_t k = lo; k < hi; ++k) { // branch-free: it vectorises
            const int64_t tv = t_start[k], qv = q_start[k];
            r.bad |= tv < 0 || qv < 0 || tv + max_tl > target_base_count || qv + max_ql > query_base_count;
            r.t_end = std::max(r.t_end, tv);
            r.q_end = std::max(r.q_end, qv);
        }
        r.t_end += max_tl;
        r.q_end += max_ql;
        return r;
    };
    auto check = [&](int64_t first, int64_t end) -> Reach {
        constexpr int kParts = 8;
        if (end - first < (1 << 18)) return check_range(first, end);
        std::future<Reach> part[kParts];
        for (int p = 0; p < kParts; ++p) part[p] = std::async(std::launch::async, check_range, first + (end - first) * p / kParts, first + (end - first) * (p + 1) / kParts);
        Reach r;
        for (int p = 0; p < kParts; ++p) {
            const Reach x = part[p].get();
            r.bad |= x.bad;
            r.t_end = std::max(r.t_end, x.t_end);
            r.q_end = std::max(r.q_end, x.q_end);
        }
        return r;
    };
    // (whole 2 MB steps, and what would be left behind a step goes with it when it is less than 4 MB: a small or odd-sized copy is not the
    // copy engines' but a blit KERNEL, which behind a grid that holds every wave slot does not start -- seen: the gate stood still until
    // the waves gave up)
    auto bring = [&](const uint8_t *src, void *dst, size_t total, int64_t reach_bases, bool last, size_t &done) -> hipError_t {
        size_t upto = last ? total : std::min(total, ((size_t)((reach_bases + 3) >> 2) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1));
        if (total - upto < ((size_t)4 << 20)) upto = total;
        hipError_t e = hipSuccess;
        if (upto > done) {
            e = hipMemcpyAsync(static_cast<uint8_t *>(dst) + done, src + done, upto - done, hipMemcpyHostToDevice, ctx->h2d);
            done = upto;
        }
        return e;
    };
    for (size_t k = 0; k < ends.size(); ++k) {
        const int64_t first = k ? ends[k - 1] : 0, end = ends[k];
        const size_t f = (size_t)first, c = (size_t)(end - first);
        hipError_t e = hipMemcpyAsync(static_cast<int64_t *>(ctx->d_toff.p) + f, t_start + f, c * 8, hipMemcpyHostToDevice, ctx->h2d);
        if (e == hipSuccess) e = hipMemcpyAsync(static_cast<int64_t *>(ctx->d_qoff.p) + f, q_start + f, c * 8, hipMemcpyHostToDevice, ctx->h2d);
        Reach r;
        try {
            r = check(first, end);
        } catch (const std::exception &) {
            return bail(fail(ctx, MGL_SW_ERR_NOMEM, "mgl_sw_align_batch_2bit: out of host resources"));
        }
        if (r.bad) {
            // a pair outside its array: the grid is sent home (a negative gate), nothing of this chunk was handed to it
            __atomic_store_n(&ctx->pin_gate[0], (int64_t)-1, __ATOMIC_RELEASE);
            drain_streams(ctx, st);
            return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch_2bit: a pair lies outside its packed array or its length outside [1, max]");
        }
        if (e == hipSuccess) e = bring(target_bases, ctx->d_t.p, t_bytes, r.t_end, end == n, t_done);
        if (e == hipSuccess) e = bring(query_bases, ctx->d_q.p, q_bytes, r.q_end, end == n, q_done);
        if (e == hipSuccess) e = hipEventRecord(ctx->gate_ev[k], ctx->h2d);
        if (e != hipSuccess) return bail(hip_fail(ctx, e, "mgl_sw_align_batch_2bit: input copy"));
        // the gate moves on over every chunk that has landed meanwhile
        while (landed <= k && hipEventQuery(ctx->gate_ev[landed]) == hipSuccess) __atomic_store_n(&ctx->pin_gate[0], ends[landed++], __ATOMIC_RELEASE);
        (void)hipGetLastError(); // (hipErrorNotReady is not an error)
    }
    (void)t_len;
    (void)q_len;
    const double t_copies = now();
    for (; landed < ends.size(); ++landed) {
        if (hipEventSynchronize(ctx->gate_ev[landed]) != hipSuccess) {
            (void)hipGetLastError();
            return bail(fail(ctx, MGL_SW_ERR_DEVICE, "mgl_sw_align_batch_2bit: an input copy failed"));
        }
        __atomic_store_n(&ctx->pin_gate[0], ends[landed], __ATOMIC_RELEASE);
    }
    const double t_gate_open = now();
    if (timing)
        fprintf(stderr, "[mgl_sw] direct form: checks and plan %.2f ms, launch %.2f ms, copies enqueued %.2f ms after the launch, all landed %.2f ms after it\n",
                (t_planned - t_enter) * 1e3, (t_launched - t_planned) * 1e3, (t_copies - t_launched) * 1e3, (t_gate_open - t_launched) * 1e3);
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (timing) fprintf(stderr, "[mgl_sw] direct form: grid ended %.2f ms after the launch (%.2f ms after its last inputs landed)\n", (now() - t_launched) * 1e3, (now() - t_gate_open) * 1e3);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->h2d));
    if (const int frc = grid_fault_check(ctx)) return frc;
    if (__atomic_load_n(reinterpret_cast<int32_t *>(&ctx->pin_gate[1]), __ATOMIC_ACQUIRE) != 0) {
        // a wave gave up at the gate: the copies did not move while the grid was resident.  Not again on this context; this call goes the
        // chunked way (every result is written again)
        ctx->direct_broken = true;
        *taken = false;
        return MGL_SW_OK;
    }
    if (status_out) return MGL_SW_OK;
    int32_t any = 0;
    HIP_TRY(ctx, hipMemcpy(&any, ctx->d_any.p, 4, hipMemcpyDeviceToHost));
    if (any != 0) return fail(ctx, any, "a CIGAR did not fit cigar_stride");
    return MGL_SW_OK;
}

// ---- the DIRECT form of mgl_sw_align_batch / _status (round 5): the reference's own calling contract -- ASCII bases in host memory,
// MicrosoftSmithWaterman.java:71-86 -- for a batch of one geometry whose arrays the caller has page-locked.  As align_2bit_direct above:
// the persistent grid is launched first and waits at its gate, the copy engines bring the bases in chunk by chunk, the waves write the
// results into the caller's arrays.  What differs: 4.06 GB cross the link for 10 M pairs of 256 x 150 where the packed form sends 0.54 --
// the LINK is the bound (73-76 ms against the grid's 64), every wave stands at the gate most of the time (hence the gate's mirror in device
// memory, sw_dp16_lane_ck.hip: one wave per microsecond looks at the host's word, not 2 048), and what the call takes beyond the link's
// time is the first chunk's way in and the tiles of the last chunk -- so the chunks are 262 144 pairs (110 MB, two milliseconds) and the last
// ones halve down to 16 384.  The offset arrays do not travel at all: the scan of mgl_sw_align_batch_status has found every pair tl x ql, so pair k starts
// at k * tl, and a kernel in front of the grid writes exactly that into the device's copies (160 MB less on the link).
static int align_ascii_direct(mgl_sw_ctx *ctx, int64_t n, const uint8_t *targets, const uint8_t *queries, int tl, int ql, int match, int mismatch, int gopen, int gext,
                              int strategy, int32_t *offset_out, mgl_sw_score *score_out, char *cigar_out, int cigar_stride, int32_t *cigar_len_out, int32_t *status_out,
                              bool *taken)
{
    *taken = false;
    const size_t nn = (size_t)n, t_bytes = nn * (size_t)tl, q_bytes = nn * (size_t)ql;
    const char *const off_env = getenv("MGL_SW_DEBUG_HOST_DIRECT"); // (0: always the chunked form; read per call: tests compare the two)
    if (ctx->direct_broken || (off_env && atoi(off_env) == 0)) return MGL_SW_OK;
    if (!(ctx->is_registered(targets, t_bytes) && ctx->is_registered(queries, q_bytes) && ctx->is_registered(offset_out, nn * 4) &&
          (!score_out || ctx->is_registered(score_out, nn * sizeof(mgl_sw_score))) && ctx->is_registered(cigar_out, nn * (size_t)cigar_stride) &&
          (!cigar_len_out || ctx->is_registered(cigar_len_out, nn * 4)) && (!status_out || ctx->is_registered(status_out, nn * 4))))
        return MGL_SW_OK;
    void *d_off = nullptr, *d_sc = nullptr, *d_cg = nullptr, *d_len = nullptr, *d_st = nullptr;
    if (hipHostGetDevicePointer(&d_off, offset_out, 0) != hipSuccess || hipHostGetDevicePointer(&d_cg, cigar_out, 0) != hipSuccess ||
        (score_out && hipHostGetDevicePointer(&d_sc, score_out, 0) != hipSuccess) || (cigar_len_out && hipHostGetDevicePointer(&d_len, cigar_len_out, 0) != hipSuccess) ||
        (status_out && hipHostGetDevicePointer(&d_st, status_out, 0) != hipSuccess)) {
        (void)hipGetLastError();
        return MGL_SW_OK;
    }
    int m_ = match, x_ = mismatch, o_ = gopen, e_ = gext;
    mgl_sw_normalize_params(&m_, &x_, &o_, &e_);
    {   // does the batch plan as ONE launch of the checkpointed lane kernel whose results can leave in whole lines?
        const SeqSet ts{reinterpret_cast<const uint8_t *>(8), reinterpret_cast<const int64_t *>(8), nullptr, tl, 0}, qs{reinterpret_cast<const uint8_t *>(8), reinterpret_cast<const int64_t *>(8), nullptr, ql, 0};
        BatchPlan P{};
        const std::string err = ctx->err;
        const int prc = plan_batch(ctx, n, ts, qs, tl, ql, m_, x_, o_, e_, strategy, static_cast<const Score *>(d_sc), static_cast<const char *>(d_cg), cigar_stride, GEOM_UNIFORM, false, nullptr,
                                   nullptr, false, P);
        ctx->err = err;
        TbArgs probe{};
        probe.cigar = static_cast<char *>(d_cg);
        probe.cigar_stride = cigar_stride;
        probe.offset = static_cast<int32_t *>(d_off);
        probe.score = static_cast<Score *>(d_sc);
        probe.cigar_len = static_cast<int32_t *>(d_len);
        probe.status = static_cast<int32_t *>(d_st);
        if (prc != MGL_SW_OK || !P.lane_ck || P.auto_group || P.chunk < n || !lane_ck_coalesced_ok(probe)) return MGL_SW_OK;
    }
    *taken = true;
    HIP_TRY(ctx, ctx->d_t.reserve(t_bytes + 64));
    HIP_TRY(ctx, ctx->d_q.reserve(q_bytes + 64));
    HIP_TRY(ctx, ctx->d_toff.reserve((nn + 1) * 8));
    HIP_TRY(ctx, ctx->d_qoff.reserve((nn + 1) * 8));
    HIP_TRY(ctx, ctx->d_any.reserve(64));
    if (!ctx->pin_gate) {
        HIP_TRY(ctx, hipHostMalloc(reinterpret_cast<void **>(&ctx->pin_gate), 64, hipHostMallocDefault));
        memset(ctx->pin_gate, 0, 64);
    }
    void *gate_dev = nullptr;
    HIP_TRY(ctx, hipHostGetDevicePointer(&gate_dev, ctx->pin_gate, 0));
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipStreamSynchronize(st)); // (nothing of an earlier call reads the gate or the inputs any more)
    __atomic_store_n(&ctx->pin_gate[0], (int64_t)0, __ATOMIC_RELEASE);
    __atomic_store_n(reinterpret_cast<int32_t *>(&ctx->pin_gate[1]), 0, __ATOMIC_RELEASE);
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_any.p, 0, 64, st));
    // the offsets, made where they are used (in front of the grid, on its stream: done before the grid's first wave looks at one)
    HIP_TRY(ctx, launch_iota64(static_cast<int64_t *>(ctx->d_toff.p), n + 1, tl, st));
    HIP_TRY(ctx, launch_iota64(static_cast<int64_t *>(ctx->d_qoff.p), n + 1, ql, st));
    const char *const dce = getenv("MGL_SW_DEBUG_DIRECT_CHUNK"); // (pairs per chunk; read per call)
    // (measured, 10 M pairs of 256 x 150, chunks all alike: 65 536 pairs 53.8 GB/s and the grid ends 1.9 ms behind the last byte; 131 072: 55.3
    // and 2.4; 262 144: 56.0 and 3.3; 524 288: 56.1 and 5.1 -- large chunks for the link, small ones for the end: the last chunk and a half
    // goes in pieces that halve down to 16 384 pairs)
    const int64_t per = std::max<int64_t>(128, (dce ? atoll(dce) : (int64_t)256 * 1024) / 128 * 128), least = std::min<int64_t>(per, 16384);
    std::vector<int64_t> ends;
    for (int64_t first = 0; first < n;) {
        const int64_t left = n - first;
        int64_t c = per;
        if (left <= per + per / 2) c = left / 2 >= least ? (left / 2 + 127) / 128 * 128 : left;
        first = std::min(n, first + c);
        ends.push_back(first);
    }
    while (ctx->gate_ev.size() < ends.size()) {
        hipEvent_t e = nullptr;
        HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
        ctx->gate_ev.push_back(e);
    }
    const SeqSet ts{static_cast<const uint8_t *>(ctx->d_t.p), static_cast<const int64_t *>(ctx->d_toff.p), nullptr, tl, 0},
        qs{static_cast<const uint8_t *>(ctx->d_q.p), static_cast<const int64_t *>(ctx->d_qoff.p), nullptr, ql, 0};
    ctx->cur_gate = static_cast<const int64_t *>(gate_dev);
    ctx->cur_gate_dev = reinterpret_cast<int64_t *>(static_cast<char *>(ctx->d_any.p) + 16);
    ctx->direct_status_any = static_cast<int32_t *>(ctx->d_any.p);
    int rc = run_device(ctx, st, n, ts, qs, tl, ql, match, mismatch, gopen, gext, strategy, static_cast<int32_t *>(d_off), static_cast<Score *>(d_sc), static_cast<char *>(d_cg),
                        cigar_stride, static_cast<int32_t *>(d_len), static_cast<int32_t *>(d_st), n * (int64_t)tl * ql, GEOM_UNIFORM);
    ctx->cur_gate = nullptr;
    ctx->cur_gate_dev = nullptr;
    ctx->direct_status_any = nullptr;
    // (from here on a grid may be waiting at the gate: every way out that is not the good one calls it off)
    auto bail = [&](int code) -> int {
        __atomic_store_n(&ctx->pin_gate[0], (int64_t)-1, __ATOMIC_RELEASE);
        drain_streams(ctx, st);
        return code;
    };
    if (rc != MGL_SW_OK) return bail(rc);
    const bool timing = debug_knobs().host_timing;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_launched = now();
    size_t landed = 0;
    for (size_t k = 0; k < ends.size(); ++k) {
        const size_t f = (size_t)(k ? ends[k - 1] : 0), c = (size_t)ends[k] - f;
        hipError_t e = hipMemcpyAsync(static_cast<uint8_t *>(ctx->d_t.p) + f * (size_t)tl, targets + f * (size_t)tl, c * (size_t)tl, hipMemcpyHostToDevice, ctx->h2d);
        if (e == hipSuccess) e = hipMemcpyAsync(static_cast<uint8_t *>(ctx->d_q.p) + f * (size_t)ql, queries + f * (size_t)ql, c * (size_t)ql, hipMemcpyHostToDevice, ctx->h2d);
        if (e == hipSuccess) e = hipEventRecord(ctx->gate_ev[k], ctx->h2d);
        if (e != hipSuccess) return bail(hip_fail(ctx, e, "mgl_sw_align_batch: input copy"));
        // the gate moves on over every chunk that has landed meanwhile
        while (landed <= k && hipEventQuery(ctx->gate_ev[landed]) == hipSuccess) __atomic_store_n(&ctx->pin_gate[0], ends[landed++], __ATOMIC_RELEASE);
        (void)hipGetLastError(); // (hipErrorNotReady is not an error)
    }
    for (; landed < ends.size(); ++landed) {
        if (hipEventSynchronize(ctx->gate_ev[landed]) != hipSuccess) {
            (void)hipGetLastError();
            return bail(fail(ctx, MGL_SW_ERR_DEVICE, "mgl_sw_align_batch: an input copy failed"));
        }
        __atomic_store_n(&ctx->pin_gate[0], ends[landed], __ATOMIC_RELEASE);
    }
    const double t_gate_open = now();
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (timing)
        fprintf(stderr, "[mgl_sw] direct form, ASCII: %zu chunks of %lld pairs; all inputs landed %.2f ms after the launch (%.1f GB/s), the grid ended %.2f ms after that\n", ends.size(),
                (long long)per, (t_gate_open - t_launched) * 1e3, (double)(t_bytes + q_bytes) / (t_gate_open - t_launched) / 1e9, (now() - t_gate_open) * 1e3);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->h2d));
    if (const int frc = grid_fault_check(ctx)) return frc;
    if (__atomic_load_n(reinterpret_cast<int32_t *>(&ctx->pin_gate[1]), __ATOMIC_ACQUIRE) != 0) {
        // a wave gave up at the gate: not again on this context; this call goes the chunked way (every result is written again)
        ctx->direct_broken = true;
        *taken = false;
        return MGL_SW_OK;
    }
    if (status_out) return MGL_SW_OK;
    int32_t any = 0;
    HIP_TRY(ctx, hipMemcpy(&any, ctx->d_any.p, 4, hipMemcpyDeviceToHost));
    if (any != 0) return fail(ctx, any, "a CIGAR did not fit cigar_stride");
    return MGL_SW_OK;
}

// ---- host buffers, 2-bit packed bases (the wire format of mgl_sw_align_batch_device_2bit from host memory).  The packed arrays move
// chunk by chunk with the index arrays when the pairs' start positions ascend (reads packed back to back), or whole before the
// first chunk (windows into a genome, in any order); results leave as in mgl_sw_align_batch_status.
int mgl_sw_align_batch_2bit(mgl_sw_ctx *ctx, int64_t n, const uint8_t *target_bases, int64_t target_base_count, const int64_t *t_start,
                            const int32_t *t_len, const uint8_t *query_bases, int64_t query_base_count, const int64_t *q_start,
                            const int32_t *q_len, int max_tl, int max_ql, int match, int mismatch, int gopen, int gext, int strategy,
                            int32_t *offset_out, mgl_sw_score *score_out, char *cigar_out, int cigar_stride, int32_t *cigar_len_out,
                            int32_t *status_out, int flags)
{
    if (!ctx) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (n == 0) return MGL_SW_OK;
    // this entry's copies go over the high-priority copy streams (mgl_sw_ctx_create says why); everything it enqueues on them has
    // completed when it returns
    struct CopyStreams {
        mgl_sw_ctx *c;
        explicit CopyStreams(mgl_sw_ctx *c_) : c(c_) { swap(); }
        ~CopyStreams() { swap(); }
        void swap() const
        {
            std::swap(c->h2d, c->h2d_hi);
            std::swap(c->d2h, c->d2h_hi);
        }
    } const copy_streams(ctx);
    const bool uniform = (flags & MGL_SW_FLAG_UNIFORM_GEOMETRY) != 0, grouped = (flags & MGL_SW_FLAG_GROUPED_GEOMETRY) != 0;
    if (n < 0 || !target_bases || !t_start || !query_bases || !q_start || !offset_out || !cigar_out || cigar_stride < 1 || max_tl < 1 || max_ql < 1 ||
        target_base_count < 1 || query_base_count < 1 || !strategy_ok(strategy) || ((!t_len || !q_len) && !uniform))
        return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch_2bit: bad argument (length arrays are required unless the geometry is uniform)");
    // ---- the direct form: every array page-locked by the caller, one geometry, the checkpointed lane kernel in one launch (it checks the
    // index arrays itself, chunk by chunk beside the running grid)
    if (uniform) {
        if (hipSetDevice(ctx->device) != hipSuccess) return hip_fail(ctx, hipGetLastError(), "hipSetDevice");
        bool taken = false;
        const int drc = align_2bit_direct(ctx, n, target_bases, (size_t)((target_base_count + 3) >> 2), t_start, t_len, query_bases, (size_t)((query_base_count + 3) >> 2), q_start, q_len,
                                          max_tl, max_ql, match, mismatch, gopen, gext, strategy, offset_out, score_out, cigar_out, cigar_stride, cigar_len_out, status_out, flags,
                                          target_base_count, query_base_count, &taken);
        if (taken) return drc;
    }
    // one pass over the index arrays: every pair inside its array, lengths within the stated maxima, do the starts ascend?
    struct Scan {
        bool bad = false, t_sorted = true, q_sorted = true;
        int64_t cells = 0;
    };
    auto scan_range = [&](int64_t a, int64_t b) {
        Scan r;
        for (int64_t k = a; k < b; ++k) {
            const int64_t tl = uniform ? max_tl : t_len[k], ql = uniform ? max_ql : q_len[k];
            r.bad |= tl < 1 || ql < 1 || tl > max_tl || ql > max_ql || t_start[k] < 0 || q_start[k] < 0 || t_start[k] + tl > target_base_count ||
                     q_start[k] + ql > query_base_count;
            r.cells += tl * ql;
            if (k > 0) {
                r.t_sorted &= t_start[k] >= t_start[k - 1];
                r.q_sorted &= q_start[k] >= q_start[k - 1];
            }
        }
        return r;
    };
    // (mixed geometries, round 5: no pass over the whole batch in front of everything -- 3 ms per 4 M pairs on eight threads before the
    // first copy command --: every chunk's index arrays are looked at when the chunk is brought in, bring_chunk below)
    const bool dev_sort = !uniform && !grouped;
    Scan sc;
    try {
        constexpr int kParts = 8;
        if (dev_sort) {
        } else if (n >= (1 << 21)) {
            std::future<Scan> part[kParts];
            // (disjoint ranges: the pair at a range's start is compared with the one before it, `k > 0`, so the borders are covered)
            for (int p = 0; p < kParts; ++p) part[p] = std::async(std::launch::async, scan_range, n * p / kParts, n * (p + 1) / kParts);
            for (int p = 0; p < kParts; ++p) {
                const Scan r = part[p].get();
                sc.bad |= r.bad;
                sc.t_sorted &= r.t_sorted;
                sc.q_sorted &= r.q_sorted;
                sc.cells += r.cells;
            }
        } else {
            sc = scan_range(0, n);
        }
    } catch (const std::exception &) {
        return fail(ctx, MGL_SW_ERR_NOMEM, "mgl_sw_align_batch_2bit: out of host resources");
    }
    if (sc.bad) return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch_2bit: a pair lies outside its packed array or its length outside [1, max]");
    const size_t t_bytes = (size_t)((target_base_count + 3) >> 2), q_bytes = (size_t)((query_base_count + 3) >> 2), nn = (size_t)n;

    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ctx->d_t.reserve(t_bytes + 8));
    HIP_TRY(ctx, ctx->d_q.reserve(q_bytes + 8));
    HIP_TRY(ctx, ctx->d_toff.reserve(nn * 8));
    HIP_TRY(ctx, ctx->d_qoff.reserve(nn * 8));
    if (!uniform) {
        HIP_TRY(ctx, ctx->d_tlen.reserve(nn * 4));
        HIP_TRY(ctx, ctx->d_qlen.reserve(nn * 4));
    }
    HIP_TRY(ctx, ctx->d_off.reserve(nn * 4));
    HIP_TRY(ctx, ctx->d_score.reserve(nn * sizeof(Score)));
    HIP_TRY(ctx, ctx->d_cig.reserve(nn * (size_t)cigar_stride));
    HIP_TRY(ctx, ctx->d_len.reserve(nn * 4));
    HIP_TRY(ctx, ctx->d_status.reserve(nn * 4));
    HIP_TRY(ctx, ctx->d_any.reserve(64));
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_any.p, 0, 4, st));

    ChunkHooks hooks;
    hooks.d_status_any = static_cast<int32_t *>(ctx->d_any.p);
    // bytes [done, upto) of a packed array that the chunks so far have not brought over (ascending starts), or all of it at once
    size_t t_done = 0, q_done = 0;
    auto bring = [&](const uint8_t *src, DevBuf &dst, size_t total, bool sorted, const int64_t *start, const int32_t *len, int uni_len, int64_t first,
                     int64_t count, size_t &done) -> int {
        size_t upto = total;
        if (sorted && first + count < n) {
            // (starts ascend, ends need not: no pair of the chunk reaches beyond its last start + the longest sequence -- a bound, not a
            // scan of the chunk: a few bytes too many cross the link, and the calling thread does not walk a million pairs per chunk)
            (void)len;
            upto = std::min(total, (size_t)((start[first + count - 1] + uni_len + 3) >> 2));
        }
        if (upto > done) {
            HIP_TRY(ctx, hipMemcpyAsync(static_cast<uint8_t *>(dst.p) + done, src + done, upto - done, hipMemcpyHostToDevice, ctx->h2d));
            done = upto;
        }
        return MGL_SW_OK;
    };
    int64_t brought = 0; // pairs [0, brought) have had their inputs enqueued on ctx->h2d (the chunks come in order)
    auto bring_chunk = [&](int64_t first, int64_t count) -> int {
        if (first + count <= brought) return MGL_SW_OK;
        const size_t f = (size_t)first, c = (size_t)count;
        if (dev_sort) {
            // this chunk's index arrays, looked at now: every pair inside its array and within the stated maxima; do the starts still ascend
            // (then the packed arrays travel slice by slice with the chunks -- once they do not, the rest of the array goes at once)?
            Scan r;
            try {
                constexpr int kParts = 8;
                if (count >= (1 << 18)) {
                    std::future<Scan> part[kParts];
                    for (int p = 0; p < kParts; ++p) part[p] = std::async(std::launch::async, scan_range, first + count * p / kParts, first + count * (p + 1) / kParts);
                    for (int p = 0; p < kParts; ++p) {
                        const Scan x = part[p].get();
                        r.bad |= x.bad;
                        r.t_sorted &= x.t_sorted;
                        r.q_sorted &= x.q_sorted;
                        r.cells += x.cells;
                    }
                } else {
                    r = scan_range(first, first + count);
                }
            } catch (const std::exception &) {
                return fail(ctx, MGL_SW_ERR_NOMEM, "mgl_sw_align_batch_2bit: out of host resources");
            }
            if (r.bad) return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_align_batch_2bit: a pair lies outside its packed array or its length outside [1, max]");
            sc.t_sorted &= r.t_sorted;
            sc.q_sorted &= r.q_sorted;
            sc.cells += r.cells;
        }
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<int64_t *>(ctx->d_toff.p) + f, t_start + f, c * 8, hipMemcpyHostToDevice, ctx->h2d));
        HIP_TRY(ctx, hipMemcpyAsync(static_cast<int64_t *>(ctx->d_qoff.p) + f, q_start + f, c * 8, hipMemcpyHostToDevice, ctx->h2d));
        if (!uniform) {
            HIP_TRY(ctx, hipMemcpyAsync(static_cast<int32_t *>(ctx->d_tlen.p) + f, t_len + f, c * 4, hipMemcpyHostToDevice, ctx->h2d));
            HIP_TRY(ctx, hipMemcpyAsync(static_cast<int32_t *>(ctx->d_qlen.p) + f, q_len + f, c * 4, hipMemcpyHostToDevice, ctx->h2d));
        }
        int rc = bring(target_bases, ctx->d_t, t_bytes, sc.t_sorted, t_start, uniform ? nullptr : t_len, max_tl, first, count, t_done);
        if (rc == MGL_SW_OK) rc = bring(query_bases, ctx->d_q, q_bytes, sc.q_sorted, q_start, uniform ? nullptr : q_len, max_ql, first, count, q_done);
        if (rc != MGL_SW_OK) return rc;
        brought = first + count;
        return MGL_SW_OK;
    };
    hooks.before_fill = [&](int64_t first, int64_t count, hipStream_t fill_stream) -> int {
        // (a chunk the device sort has brought in ahead of its fill -- bring_ahead below -- is there already: run_device waits for that sort,
        // which sits behind the chunk's copies on ctx->h2d, before it launches anything of the chunk)
        if (first + count <= brought) return MGL_SW_OK;
        const int rc = bring_chunk(first, count);
        if (rc != MGL_SW_OK) return rc;
        HIP_TRY(ctx, hipEventRecord(ctx->in_done, ctx->h2d));
        HIP_TRY(ctx, hipStreamWaitEvent(fill_stream, ctx->in_done, 0));
        return MGL_SW_OK;
    };
    // Mixed geometries (round 5): the chunks are sorted by (tl, ql) ON THE DEVICE, as the device-resident entries' are -- whole waves of
    // one geometry through the checkpointed lane kernel's persistent grid, the rest through the packed and int32 kernels -- each chunk's
    // inputs brought in front of its sort, two chunks ahead of the fills.  (Before: no sort at all on this entry -- every pair through the
    // int32 kernel, 1 475 GCUPS on 4 M reads of 100-150 bases against 5 480 device resident.)
    if (!uniform && !grouped) hooks.bring_ahead = bring_chunk;
    ResultPump pump(ctx, offset_out, score_out, cigar_out, cigar_stride, cigar_len_out, status_out, n);
    hooks.after_traceback = [&](int64_t first, int64_t count, hipEvent_t results_ready) -> int { return pump.after_traceback(first, count, results_ready); };

    const SeqSet ts{static_cast<const uint8_t *>(ctx->d_t.p), static_cast<const int64_t *>(ctx->d_toff.p), uniform ? nullptr : static_cast<const int32_t *>(ctx->d_tlen.p), max_tl, 1},
        qs{static_cast<const uint8_t *>(ctx->d_q.p), static_cast<const int64_t *>(ctx->d_qoff.p), uniform ? nullptr : static_cast<const int32_t *>(ctx->d_qlen.p), max_ql, 1};
    int rc;
    try {
        rc = run_device(ctx, st, n, ts, qs, max_tl, max_ql, match, mismatch, gopen, gext, strategy, static_cast<int32_t *>(ctx->d_off.p),
                        static_cast<Score *>(ctx->d_score.p), static_cast<char *>(ctx->d_cig.p), cigar_stride, static_cast<int32_t *>(ctx->d_len.p),
                        static_cast<int32_t *>(ctx->d_status.p), sc.cells, uniform ? GEOM_UNIFORM : grouped ? GEOM_GROUPED : GEOM_MIXED,
                        (flags & MGL_SW_FLAG_BINARY_CIGAR) != 0, &hooks);
    } catch (const std::exception &) {
        rc = fail(ctx, MGL_SW_ERR_NOMEM, "mgl_sw_align_batch_2bit: out of host resources");
    }
    if (dev_sort && ctx->profiling != 3) ctx->timing.cells = sc.cells; // (counted chunk by chunk, above)
    const int res_rc = pump.join();
    if (rc == MGL_SW_OK && res_rc != MGL_SW_OK) rc = fail(ctx, MGL_SW_ERR_DEVICE, "mgl_sw_align_batch_2bit: copying results out failed");
    if (rc != MGL_SW_OK) {
        drain_streams(ctx, st);
        return rc;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->h2d));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->aux));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (ctx->d2h) HIP_TRY(ctx, hipStreamSynchronize(ctx->d2h));
    if (const int frc = grid_fault_check(ctx)) return frc;
    if (status_out) return MGL_SW_OK;
    int32_t any = 0;
    HIP_TRY(ctx, hipMemcpy(&any, ctx->d_any.p, 4, hipMemcpyDeviceToHost));
    if (any != 0) return fail(ctx, any, "a CIGAR did not fit cigar_stride");
    return MGL_SW_OK;
}

// ---- small-batch entry for the coalescing front-end (sw_batcher.cpp; library-internal, not in include/mgl_sw.h).
// One pinned host buffer in, one out, each mirrored on the device: a batch costs one copy each way, two launches
// and one synchronisation instead of the eleven copies of the general host entry.
//   in : int64 t_off[n+1] | int64 q_off[n+1] | target bytes | query bytes      (sections 8-byte aligned)
//   out: int32 offset[n] | int32 cigar_len[n] | int32 status[n] | mgl_sw_score[n] | char cigar[n][stride]
static int stage_buffers_nolock(mgl_sw_ctx *ctx, size_t in_bytes, size_t out_bytes, void **in, void **out)
{
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    auto grow = [&](void *&p, size_t &cap, size_t want) -> hipError_t {
        if (want <= cap) return hipSuccess;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        cap = 0;
        const size_t bytes = want + want / 2 + 4096;
        hipError_t e = hipHostMalloc(&p, bytes, hipHostMallocDefault);
        if (e == hipSuccess) cap = bytes;
        return e;
    };
    HIP_TRY(ctx, grow(ctx->pin_in, ctx->pin_in_cap, in_bytes));
    HIP_TRY(ctx, grow(ctx->pin_out, ctx->pin_out_cap, out_bytes));
    HIP_TRY(ctx, ctx->stage_in.reserve(ctx->pin_in_cap));
    HIP_TRY(ctx, ctx->stage_out.reserve(ctx->pin_out_cap));
    *in = ctx->pin_in;
    *out = ctx->pin_out;
    return MGL_SW_OK;
}

MGL_SW_INTERNAL int mgl_sw_stage_buffers(mgl_sw_ctx *ctx, size_t in_bytes, size_t out_bytes, void **in, void **out)
{
    if (!ctx || !in || !out) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return stage_buffers_nolock(ctx, in_bytes, out_bytes, in, out);
}

static int align_batch_staged_nolock(mgl_sw_ctx *ctx, int n, size_t in_bytes, size_t t_bytes_padded, int max_tl, int max_ql, int match,
                                     int mismatch, int gopen, int gext, int strategy, int cigar_stride, size_t out_bytes, int uniform,
                                     int64_t cells_hint)
{
    if (in_bytes > ctx->pin_in_cap || out_bytes > ctx->pin_out_cap) return fail(ctx, MGL_SW_ERR_BAD_ARG, "staged batch larger than its buffers");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    // small batches are latency bound: the kernels then read the pinned (device-mapped, coherent) staging buffer in place
    // and write their results straight into the pinned output buffer -- no copy commands at all (a copy costs more in
    // launch latency than the few hundred bytes per pair cost over PCIe); large batches are copied to HBM first
    constexpr size_t zero_copy_max = (size_t)(2u << 20);
    const bool zero_copy = in_bytes + out_bytes <= zero_copy_max;
    const uint8_t *din = static_cast<const uint8_t *>(ctx->stage_in.p);
    uint8_t *dout = static_cast<uint8_t *>(ctx->stage_out.p);
    if (zero_copy) {
        void *pi = nullptr, *po = nullptr;
        HIP_TRY(ctx, hipHostGetDevicePointer(&pi, ctx->pin_in, 0));
        HIP_TRY(ctx, hipHostGetDevicePointer(&po, ctx->pin_out, 0));
        din = static_cast<const uint8_t *>(pi);
        dout = static_cast<uint8_t *>(po);
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->stage_in.p, ctx->pin_in, in_bytes, hipMemcpyHostToDevice, st));
    }
    const size_t offs_bytes = (size_t)(n + 1) * 8;
    const SeqSet ts{din + 2 * offs_bytes, reinterpret_cast<const int64_t *>(din), nullptr, max_tl, 0},
        qs{din + 2 * offs_bytes + t_bytes_padded, reinterpret_cast<const int64_t *>(din + offs_bytes), nullptr, max_ql, 0};
    int32_t *d_off = reinterpret_cast<int32_t *>(dout), *d_len = d_off + n, *d_status = d_len + n;
    Score *d_score = reinterpret_cast<Score *>(d_status + n);
    char *d_cig = reinterpret_cast<char *>(d_score + n);
    const int rc = run_device(ctx, st, n, ts, qs, max_tl, max_ql, match, mismatch, gopen, gext, strategy, d_off, d_score, d_cig,
                              cigar_stride, d_len, d_status, cells_hint, uniform);
    if (rc != MGL_SW_OK) {
        (void)hipStreamSynchronize(st);
        return rc;
    }
    if (!zero_copy) HIP_TRY(ctx, hipMemcpyAsync(ctx->pin_out, ctx->stage_out.p, out_bytes, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return MGL_SW_OK;
}

MGL_SW_INTERNAL int mgl_sw_align_batch_staged(mgl_sw_ctx *ctx, int n, size_t in_bytes, size_t t_bytes_padded, int max_tl, int max_ql, int match,
                              int mismatch, int gopen, int gext, int strategy, int cigar_stride, size_t out_bytes)
{
    if (!ctx || n < 1) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    return align_batch_staged_nolock(ctx, n, in_bytes, t_bytes_padded, max_tl, max_ql, match, mismatch, gopen, gext, strategy,
                                     cigar_stride, out_bytes, GEOM_MIXED, 0);
}

// sw_batcher.cpp
MGL_SW_INTERNAL bool mgl_sw_coalescing_enabled();
MGL_SW_INTERNAL int mgl_sw_coalesced_align(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen, int gext,
                           int strategy, char *cigar, int cigar_cap, int *cigar_len, int *offset, mgl_sw_score *ez);
// sw_service.cpp: the calling thread's mailbox and resident wave; MGL_SW_SERVICE_DECLINED = not this call (geometry, no mailbox left, switched off)
constexpr int MGL_SW_SERVICE_DECLINED = 1 << 20;
MGL_SW_INTERNAL int mgl_sw_service_align(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen, int gext, int strategy,
                                         char *cigar, int cigar_cap, int *cigar_len, int *offset, mgl_sw_score *ez);

static mgl_sw_ctx *thread_ctx(int *rc)
{
    struct Holder {
        mgl_sw_ctx *ctx = nullptr;
        ~Holder() { mgl_sw_ctx_destroy(ctx); }
    };
    static thread_local Holder h;
    if (!h.ctx) {
        int dev = 0;
        if (const char *e = getenv("MGL_SW_DEVICE")) dev = atoi(e);
        *rc = mgl_sw_ctx_create(dev, &h.ctx);
        if (*rc != MGL_SW_OK) return nullptr;
        h.ctx->ws_limit = 8ll << 30; // a cap, not an allocation: the workspace grows with what the calls need
        h.ctx->cooperative = 4; // one pair per call: latency, not occupancy (see sw_batcher.cpp)
    }
    *rc = MGL_SW_OK;
    return h.ctx;
}

// one pair on the calling thread's own context (no coalescing)
static int align_direct(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen, int gext,
                        int strategy, char *cigar, int cigar_cap, int *cigar_len, int *offset, mgl_sw_score *ez)
{
    int rc;
    mgl_sw_ctx *ctx = thread_ctx(&rc);
    if (!ctx) return rc;
    const int64_t t_off[2] = {0, tl}, q_off[2] = {0, ql};
    int32_t off = 0, len = 0;
    mgl_sw_score sc;
    std::vector<char> buf((size_t)cigar_cap);
    rc = mgl_sw_align_batch(ctx, 1, reinterpret_cast<const uint8_t *>(t), t_off, reinterpret_cast<const uint8_t *>(q),
                            q_off, match, mismatch, gopen, gext, strategy, &off, &sc, buf.data(), cigar_cap, &len);
    *cigar_len = len;
    if (rc != MGL_SW_OK) return rc;
    memcpy(cigar, buf.data(), (size_t)len); // exactly cigar.length() bytes, no terminator (.cpp:65)
    *offset = off;
    if (ez) *ez = sc;
    return MGL_SW_OK;
}

int mgl_sw_align(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen, int gext,
                 int strategy, char *cigar, int cigar_cap, int *cigar_len, int *offset, mgl_sw_score *ez)
{
    if (!t || !q || tl < 1 || ql < 1 || !cigar || cigar_cap < 1 || !cigar_len || !offset || !strategy_ok(strategy))
        return MGL_SW_ERR_BAD_ARG;
    if (mgl_sw_coalescing_enabled()) {
        const int rc = mgl_sw_service_align(t, tl, q, ql, match, mismatch, gopen, gext, strategy, cigar, cigar_cap, cigar_len, offset, ez);
        if (rc != MGL_SW_SERVICE_DECLINED) return rc;
        return mgl_sw_coalesced_align(t, tl, q, ql, match, mismatch, gopen, gext, strategy, cigar, cigar_cap, cigar_len,
                                      offset, ez);
    }
    return align_direct(t, tl, q, ql, match, mismatch, gopen, gext, strategy, cigar, cigar_cap, cigar_len, offset, ez);
}

int mgl_sw_ctx_expand_slot(mgl_sw_ctx *ctx, int64_t slot, int tl, int ql, int32_t *btr)
{
    if (!ctx || !btr || tl < 1 || ql < 1 || (int64_t)tl * ql > (1ll << 30)) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (slot < 0 || slot >= ctx->last_chunk_count)
        return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_ctx_expand_slot: slot outside the last chunk");
    const size_t cells = (size_t)(tl + 1) * (ql + 1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ctx->d_btr.reserve(cells * 4));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_btr.p, 0, cells * 4, ctx->stream));
    if (ctx->last_packed16 == 5 || ctx->last_packed16 == 6)
        return fail(ctx, MGL_SW_ERR_UNSUPPORTED, "mgl_sw_ctx_expand_slot: the last call kept no traceback (checkpointed lane kernel); "
                                                  "switch it off with mgl_sw_ctx_set_lane_checkpoint(ctx, 1)");
    const int64_t region = ctx->last_packed16 == 2 ? slot >> 7 : ctx->last_packed16 == 1 ? slot >> 1 : slot;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->aux));
    HIP_TRY(ctx, launch_expand(static_cast<const uint32_t *>(ctx->tb[ctx->last_half].p) + region * ctx->last_stride_words,
                               static_cast<const DpRecord *>(ctx->rec[ctx->last_half].p) + slot, tl, ql, ctx->last_packed16,
                               (int)(slot & 1), ctx->last_rows, static_cast<int32_t *>(ctx->d_btr.p), ctx->stream, (int)((slot >> 1) & 63)));
    HIP_TRY(ctx, hipMemcpyAsync(btr, ctx->d_btr.p, cells * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return MGL_SW_OK;
}

int mgl_sw_ctx_slot_layout(mgl_sw_ctx *ctx, int64_t slot, int *layout)
{
    if (!ctx || !layout) return MGL_SW_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(ctx->mu);
    if (slot < 0 || slot >= ctx->last_chunk_count)
        return fail(ctx, MGL_SW_ERR_BAD_ARG, "mgl_sw_ctx_slot_layout: slot outside the last chunk");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->aux));
    mgl_sw_dev::DpRecord r;
    HIP_TRY(ctx, hipMemcpyAsync(&r, static_cast<const mgl_sw_dev::DpRecord *>(ctx->rec[ctx->last_half].p) + slot, sizeof r, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *layout = ctx->last_packed16 == 3 && r.g_tail != -16 ? 0 : ctx->last_packed16; // TbView::layout_of
    return MGL_SW_OK;
}

int mgl_sw_cigar_from_backtrack(const int32_t *btr, int tl, int ql, int strategy, const mgl_sw_score *ez, char *cigar,
                                int cigar_cap, int *cigar_len, int *offset)
{
    if (!btr || tl < 1 || ql < 1 || !ez || !cigar || cigar_cap < 1 || !cigar_len || !offset || !strategy_ok(strategy) ||
        (int64_t)tl * ql > (1ll << 30))
        return MGL_SW_ERR_BAD_ARG;
    // the walk starts at a cell named by *ez: reject anything outside the matrix (the reference would read wild)
    const bool from_ez = strategy == MGL_SW_OS_SOFTCLIP || strategy == MGL_SW_OS_IGNORE;
    if (from_ez && (ez->max_t < 1 || ez->max_t > tl || ez->max_q < 1 || ez->max_q > ql || ez->seg_length < 0))
        return MGL_SW_ERR_BAD_ARG;
    if (strategy == MGL_SW_OS_LEAD_ID && (ez->mqe_t < 1 || ez->mqe_t > tl)) return MGL_SW_ERR_BAD_ARG;
    int rc;
    mgl_sw_ctx *ctx = thread_ctx(&rc);
    if (!ctx) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    const size_t cells = (size_t)(tl + 1) * (ql + 1);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, ctx->d_btr.reserve(cells * 4));
    HIP_TRY(ctx, ctx->d_cig.reserve((size_t)cigar_cap));
    HIP_TRY(ctx, ctx->d_status.reserve(16));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_btr.p, btr, cells * 4, hipMemcpyHostToDevice, ctx->stream));
    Score sc;
    memcpy(&sc, ez, sizeof sc);
    HIP_TRY(ctx, launch_cigar_from_matrix(static_cast<const int32_t *>(ctx->d_btr.p), tl, ql, strategy, sc,
                                          static_cast<char *>(ctx->d_cig.p), cigar_cap,
                                          static_cast<int32_t *>(ctx->d_status.p), ctx->stream));
    int32_t out3[3] = {0, 0, 0};
    std::vector<char> buf((size_t)cigar_cap);
    HIP_TRY(ctx, hipMemcpyAsync(out3, ctx->d_status.p, sizeof out3, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(buf.data(), ctx->d_cig.p, (size_t)cigar_cap, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    *cigar_len = out3[1];
    if (out3[2] != 0) return fail(ctx, out3[2], "CIGAR does not fit cigar_cap");
    memcpy(cigar, buf.data(), (size_t)out3[1]);
    *offset = out3[0];
    return MGL_SW_OK;
}

int mgl_sw_backtrack_matrix(const char *t, int tl, const char *q, int ql, int match, int mismatch, int gopen,
                            int gext, int strategy, int32_t *btr, mgl_sw_score *ez)
{
    if (!t || !q || tl < 1 || ql < 1 || !btr || !strategy_ok(strategy) || (int64_t)tl * ql > (1ll << 30))
        return MGL_SW_ERR_BAD_ARG;
    int rc;
    mgl_sw_ctx *ctx = thread_ctx(&rc);
    if (!ctx) return rc;
    // fill + traceback of the single pair leaves its 4-bit cells in slot 0 of the workspace
    std::vector<char> cig((size_t)(tl + ql + 4) * 12);
    int len = 0, off = 0;
    mgl_sw_score sc;
    // always on this thread's context (never through the coalescing front-end): the expansion below reads its workspace
    rc = align_direct(t, tl, q, ql, match, mismatch, gopen, gext, strategy, cig.data(), (int)cig.size(), &len, &off, &sc);
    if (rc != MGL_SW_OK) return rc;
    rc = mgl_sw_ctx_expand_slot(ctx, 0, tl, ql, btr);
    if (rc == MGL_SW_OK && ez) *ez = sc;
    return rc;
}

int mgl_sw_band_fill(const int32_t *target, int target_length, const int32_t *query, int query_length, int32_t *bcktrack,
                     int band_count, int default_bw, int actual_bw, int32_t *score, int32_t *step, int32_t *gap, int match,
                     int mismatch, int gopen, int gext, int strategy, mgl_sw_score *ez)
{
    if (!target || !query || !bcktrack || !score || !step || !gap || !ez || target_length < 1 || query_length < 1 ||
        default_bw < 1 || default_bw > 64 || actual_bw < 1 || actual_bw > default_bw || band_count < 0 ||
        (int64_t)band_count * default_bw + actual_bw > target_length || !strategy_ok(strategy) || query_length > (1 << 24))
        return MGL_SW_ERR_BAD_ARG;
    mgl_sw_normalize_params(&match, &mismatch, &gopen, &gext);
    int rc;
    mgl_sw_ctx *ctx = thread_ctx(&rc);
    if (!ctx) return rc;
    std::lock_guard<std::mutex> lk(ctx->mu);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int bw = default_bw, ql = query_length;
    const size_t n_q = (size_t)ql + 2 * (size_t)bw, n_sc = (size_t)ql + 1, n_band = (size_t)(ql + bw - 1) * (size_t)bw;
    // one device buffer: target rows of the band | reversed query | score | step | gap | the band's backtrack cells | mqe, mqe_t
    const size_t words = (size_t)bw + n_q + 2 * n_sc + n_q + n_band + 2;
    HIP_TRY(ctx, ctx->d_btr.reserve(words * 4));
    int32_t *d = static_cast<int32_t *>(ctx->d_btr.p);
    int32_t *d_t = d, *d_q = d_t + bw, *d_score = d_q + n_q, *d_step = d_score + n_sc, *d_gap = d_step + n_sc, *d_band = d_gap + n_q,
            *d_mqe = d_band + n_band;
    hipStream_t st = ctx->stream;
    int32_t *h_band = bcktrack + (size_t)(ql + bw - 1) * (size_t)bw * (size_t)band_count; // sw_avx.cpp:173
    const int32_t mqe_in[2] = {ez->mqe, ez->mqe_t};
    HIP_TRY(ctx, hipMemcpyAsync(d_t, target + (size_t)bw * band_count, (size_t)actual_bw * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_q, query, n_q * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_score, score, n_sc * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_step, step, n_sc * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_gap, gap, n_q * 4, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d_band, h_band, n_band * 4, hipMemcpyHostToDevice, st)); // cells outside the matrix keep the caller's values
    HIP_TRY(ctx, hipMemcpyAsync(d_mqe, mqe_in, 8, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, launch_band_fill(d_t, d_q, ql, d_band, band_count, bw, actual_bw, d_score, d_step, d_gap, match, mismatch, gopen, gext,
                                  strategy, d_mqe, st));
    int32_t mqe_out[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(score, d_score, n_sc * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(step, d_step, n_sc * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(gap, d_gap, n_q * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(h_band, d_band, n_band * 4, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipMemcpyAsync(mqe_out, d_mqe, 8, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    ez->mqe = mqe_out[0];
    ez->mqe_t = mqe_out[1];
    return MGL_SW_OK;
}

int mgl_sw_group_by_geometry(int64_t n, const int32_t *t_len, const int32_t *q_len, int64_t *order_out, int64_t *n_grouped_out)
{
    if (n < 0 || (n > 0 && (!t_len || !q_len || !order_out)) || !n_grouped_out) return MGL_SW_ERR_BAD_ARG;
    *n_grouped_out = 0;
    if (n == 0) return MGL_SW_OK;
    int32_t lo_t = t_len[0], hi_t = t_len[0], lo_q = q_len[0], hi_q = q_len[0];
    for (int64_t k = 0; k < n; ++k) {
        lo_t = std::min(lo_t, t_len[k]);
        hi_t = std::max(hi_t, t_len[k]);
        lo_q = std::min(lo_q, q_len[k]);
        hi_q = std::max(hi_q, q_len[k]);
    }
    if (lo_t < 0 || lo_q < 0) return MGL_SW_ERR_BAD_ARG;
    std::vector<int64_t> sorted((size_t)n); // pair indices, sorted by (t_len, q_len), original order inside a geometry
    const int64_t range_t = (int64_t)hi_t - lo_t + 1, range_q = (int64_t)hi_q - lo_q + 1;
    if (range_t * range_q <= (1ll << 22)) {
        // counting sort over the (t_len, q_len) grid
        std::vector<int64_t> start((size_t)(range_t * range_q) + 1, 0);
        auto cell = [&](int64_t k) { return (size_t)(((int64_t)t_len[k] - lo_t) * range_q + ((int64_t)q_len[k] - lo_q)); };
        for (int64_t k = 0; k < n; ++k) ++start[cell(k) + 1];
        for (size_t c = 1; c < start.size(); ++c) start[c] += start[c - 1];
        for (int64_t k = 0; k < n; ++k) sorted[(size_t)start[cell(k)]++] = k;
    } else {
        for (int64_t k = 0; k < n; ++k) sorted[(size_t)k] = k;
        std::stable_sort(sorted.begin(), sorted.end(), [&](int64_t a, int64_t b) {
            return t_len[a] != t_len[b] ? t_len[a] < t_len[b] : q_len[a] < q_len[b];
        });
    }
    // full blocks of eight go to the front, the remainder of every geometry behind them
    std::vector<int64_t> rest;
    int64_t g = 0;
    for (int64_t s0 = 0; s0 < n;) {
        int64_t s1 = s0 + 1;
        while (s1 < n && t_len[sorted[(size_t)s1]] == t_len[sorted[(size_t)s0]] && q_len[sorted[(size_t)s1]] == q_len[sorted[(size_t)s0]]) ++s1;
        const int64_t full = (s1 - s0) & ~(int64_t)7;
        for (int64_t s = s0; s < s0 + full; ++s) order_out[g++] = sorted[(size_t)s];
        for (int64_t s = s0 + full; s < s1; ++s) rest.push_back(sorted[(size_t)s]);
        s0 = s1;
    }
    *n_grouped_out = g;
    for (int64_t r : rest) order_out[g++] = r;
    return MGL_SW_OK;
}

} // extern "C"
