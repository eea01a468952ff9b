// sw_device.h -- types shared by the kernels (sw_kernels.hip) and the C-ABI host layer (sw_capi.cpp).
#ifndef MGL_SW_DEVICE_H
#define MGL_SW_DEVICE_H

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace mgl_sw_dev {

constexpr int NEG_INF = -0x40000000; // sw_common.h:33
constexpr int OS_SOFTCLIP = 1, OS_INDEL = 2, OS_LEAD_ID = 4, OS_IGNORE = 8; // sw_common.h:22-25
constexpr int ERR_CIGAR_OVERFLOW = 2; // == MGL_SW_ERR_CIGAR_OVERFLOW
constexpr int ERR_DEVICE = 4;         // == MGL_SW_ERR_DEVICE
constexpr int MATRIX_DIM = 32;        // substitution-matrix mode: codes 0..31

// ScoreMax (sw_common.h:36-40); layout == mgl_sw_score
struct Score {
    int32_t mqe, mqe_t, max, max_t, max_q, seg_length;
};

// what the fill kernel hands to the traceback kernel, one per pair (workspace)
struct DpRecord {
    int32_t mqe, mqe_t, max, max_t, max_q, seg;
    int32_t g_tail; // sw_dp16_kernel: global step at which the stand-alone tail stripes start (0: none chained)
    int32_t sps;    // anti-diagonal steps per stripe (per chained stripe when g_tail > 0)
};

// How the sequences of a batch are addressed.  ASCII: pair p is data[off[p] .. off[p+1]) (one byte per base,
// compared raw, sw.cpp:55).  2-bit packed (packed2 = 1): four bases per byte, base k of the array in bits
// 2*(k&3) of byte k>>2 (A=0 C=1 G=2 T=3 by convention -- only equality matters); pair p starts at BASE index
// off[p] and has len[p] bases (uni_len when len is null), so windows into one packed genome may overlap.
struct SeqSet {
    const uint8_t *data;
    const int64_t *off;
    const int32_t *len; // optional
    int uni_len;        // length of every sequence when len == nullptr and packed2
    int packed2;
    __host__ __device__ inline int length(int64_t p) const
    {
        if (len) return len[p];
        return packed2 ? uni_len : (int)(off[p + 1] - off[p]);
    }
    __device__ inline int at(int64_t start, int k) const
    {
        const int64_t pos = start + k;
        if (packed2) return (data[pos >> 2] >> (2 * (int)(pos & 3))) & 3;
        return data[pos];
    }
};

struct DpArgs {
    SeqSet t, q;
    int64_t first; // pairs [first, first + count) of the batch
    int64_t count;
    int match, mismatch, gopen, gext, strategy;
    int sps_cap;             // upper bound of steps per stripe (sizes the LDS carve)
    int uni_tl, uni_ql;      // packed-int16 kernel only: the one geometry of the batch
    uint32_t *tb;            // traceback words, count * tb_stride_words
    int64_t tb_stride_words; // per pair
    DpRecord *rec;           // count records
    unsigned char *scratch;   // long queries (sw_dp_scratch_kernel): per-group carry ring + query copies in HBM, else null
    unsigned long long *diag; // profiling level 2 only: per block {shader-clock ticks, 100 MHz ticks}; else null
    // substitution-matrix scoring (sw_dp_matrix_kernel; null = the reference's match / mismatch)
    const int8_t *matrix;     // MATRIX_DIM x MATRIX_DIM scores, row = target code, column = query code
    const uint8_t *code;      // byte -> 0 .. MATRIX_DIM-1
    int matrix_lds_offset;    // set by the launcher
    int score_only;           // packed kernel: no traceback flags (MGL_SW_FLAG_SCORE_ONLY)
    int grouped;              // sw_dp16_lane_ck_kernel: every wave of 128 pairs has its own geometry (a chunk sorted by geometry)
    int strip_k;              // sw_dp16_strip_kernel: 0 = the flags of every cell are stored; K > 0 = none are: rows and checkpoints of bands of K strips (strip16_ck_*)
    int strip_passes;         // ... passes over the target: 2 * 64 * waves strips each (targets beyond 16 384 rows; 0 or 1: one)
    int strip_codes;          // ... and the LDS carve holds the query as one table dword per column (strip16_lds_bytes_codes): pairs whose targets are all ACGT run the base-code form
    int lane_slots;           // sw_dp16_lane_ck_kernel: wave slots of its persistent grid = regions at tb / scratch (lane_ck_slots)
    unsigned *tile_ctr;       // ... two words {draws, waves out} used as ONE 64-bit atomic object (8-byte aligned), zero when the launch starts (needed when the
                              //     launch holds more tiles than slots): the waves draw their next tile from the low half and count themselves out in the high
                              //     half; the last wave out zeroes the word
    int32_t *grid_fault;      // ... optional: set to 1 (pinned host memory) by a wave that draws a number no launch of this size can draw
    const int64_t *gate;      // ... optional (host entries): a word in pinned host memory holding how many pairs of the batch have arrived in device memory
                              //     so far; a wave waits with a tile until its pairs are there (null: everything is)
    int64_t *gate_dev;        // ... and its mirror in DEVICE memory, zeroed in front of the launch: [0] the word as a wave last read it over the link, [1] the 100 MHz time of
                              //     that look -- the waiting waves watch the mirror, one of them per microsecond looks at the host's word (sw_dp16_lane_ck.hip)
    int32_t *gate_failed;     // ... set to 1 (pinned host memory) by a wave that gives up waiting: gate_timeout_ticks (100 MHz) without the word moving
    unsigned gate_timeout_ticks;
    // sw_dp16_lane_matrix_kernel: the tiles in the order they are drawn (largest region first) and where the region of wave slot s starts and
    // ends (bytes off tb; slot_off[s + 1] - slot_off[s] = what the s-th largest tile needs: every later draw needs less)
    const int32_t *tile_order;
    const int64_t *slot_off;
};

struct TbArgs {
    SeqSet t, q;
    int64_t first, count;
    int strategy;
    const uint32_t *tb;
    int64_t tb_stride_words; // per pair (int32 layout) or per group of two pairs (packed16 layout)
    int packed16;            // traceback layout: 0 = int32 kernels, 1 = sw_dp16_kernel, 2 = sw_dp16_lane_kernel (stride per wave)
    int rows_per_stripe;     // 16 or 64 (int32 layout); rows per strip (lane layout)
    int uni_ql;              // lane layout: the batch's one query length
    const DpRecord *rec;
    int32_t *offset; // indexed by batch pair index
    Score *score;    // optional
    char *cigar;
    int cigar_stride;
    int binary_cigar; // BAM-style uint32 elements instead of text
    int32_t *cigar_len; // optional
    int32_t *status;    // optional
    int32_t *status_any; // optional: max of all non-zero statuses of the call
    int match, mismatch, gopen, gext; // sw_strip_ck_walk_kernel recomputes the blocks its path crosses
    int strip_rows, strip_k;         // ... rows per strip and strips per kept band of the fill
    const int64_t *dest; // optional: output index of input pair p (results of pair p go to offset[dest[p]], cigar slot dest[p], ...);
                         // null: p itself.  (Batches the host layer has reordered by geometry hand results back in the caller's order.)
    int coalesced_out;   // sw_dp16_lane_ck_kernel: a wave gathers the results of its 128 pairs in LDS and writes them out in whole lines (lane_ck_coalesced_ok)
};

// geometry helpers (host and device agree on these); rows = target rows per stripe = lanes per pair (16 or 64)
__host__ __device__ inline int sps_for_rows(int ql, int rows) { return (ql + rows + 3) & ~3; }
__host__ __device__ inline int sps_for(int ql) { return sps_for_rows(ql, 16); }
__host__ __device__ inline int stripes_for(int tl) { return (tl + 15) >> 4; }
__host__ __device__ inline int dp_ring_entries(int sps_cap, int rows) { return sps_cap + rows + 4; }
__host__ __device__ inline int dp_qcopy_bytes(int sps_cap, int rows) { return sps_cap + rows + 16; }
// int32 layout: per 32 steps, `rows` lanes x four 32-bit planes
__host__ __device__ inline int64_t tb_words_for(int tl, int sps, int rows)
{
    return ((((int64_t)(tl + rows - 1) / rows) * sps + 31) >> 5) * rows * 4;
}

// ---- schedule of sw_dp16_kernel (sw_dp16.hip): the first nc stripes run as one continuous pipeline with
// period P steps per stripe (+ one 16-step drain window), the rest stand-alone with sps_for(ql) steps each
__host__ __device__ inline int dp16_base(int tl, int ql, int match, int gext)
{
    return 32767 - match * (tl < ql ? tl : ql) - gext * (tl + ql);
}
__host__ __device__ inline int dp16_period(int ql) { return (ql + 1 + 3) & ~3; }
__host__ __device__ inline int dp16_chained_stripes(int tl, int ql)
{
    if (dp16_period(ql) < 32) return 0;          // the window needs P >= 32
    const int n = stripes_for(tl);
    return (tl & 15) == 0 ? n : n - 1;           // a partial last stripe runs stand-alone (other carry lane)
}
__host__ __device__ inline int64_t dp16_total_steps(int tl, int ql)
{
    const int n = stripes_for(tl), nc = dp16_chained_stripes(tl, ql);
    return (nc ? (int64_t)nc * dp16_period(ql) + 16 : 0) + (int64_t)(n - nc) * sps_for(ql);
}
// packed16 layout: two dwords per lane per 8 steps, per group of two pairs
__host__ __device__ inline int64_t tb_words16_for(int tl, int ql) { return ((dp16_total_steps(tl, ql) + 7) >> 3) * 32; }
// monotone upper bound of tb_words16_for over every (tl', ql') <= (tl, ql): P(ql) <= sps_for(ql), so a pair never takes
// more than stripes * sps_for(ql) + 16 steps (sizes the regions of MGL_SW_FLAG_GROUPED_GEOMETRY batches)
__host__ __device__ inline int64_t tb_words16_bound(int tl, int ql) { return (((int64_t)stripes_for(tl) * sps_for(ql) + 16 + 7) >> 3) * 32; }

// ---- sw_dp16_lane_kernel (sw_dp16_lane.hip): two pairs per lane, 128 per wave; strips of `rows` target rows
__host__ __device__ inline int lane_strips(int tl, int rows) { return (tl + rows - 1) / rows; }
// traceback dwords per WAVE: [strip][column][rows / 16][lane] uint4
__host__ __device__ inline int64_t lane_tb_words(int tl, int ql, int rows) { return (int64_t)lane_strips(tl, rows) * ql * (rows / 16) * 64 * 4; }
// carry row per wave: [column 0 .. ql][lane] uint2 {H, E}; entries of 8 bytes
__host__ __device__ inline int64_t lane_bnd_entries(int ql) { return (int64_t)(ql + 1) * 64; }
// per-wave scratch: the carry row, then both queries and both targets of every lane transposed to [4-base block][A | B][lane] dwords
__host__ __device__ inline int64_t lane_scratch_bytes(int tl, int ql, int rows)
{
    return lane_bnd_entries(ql) * 8 + ((int64_t)((ql + 3) / 4) + (int64_t)lane_strips(tl, rows) * (rows / 4)) * 2 * 64 * 4;
}

// ---- sw_dp_coop_kernel (sw_dp_coop.hip): one pair per workgroup, 64-row stripes, steps per stripe rounded to
// the 32-step traceback block so that every stripe (= every wave) owns whole blocks
__host__ __device__ inline int coop_sps_for(int ql) { return (ql + 64 + 31) & ~31; }
// 16-bit form (two 64-row half-stripes per wave): steps per 128-row double stripe; traceback dwords per pair:
// [double stripe][16 steps][lane] uint4
__host__ __device__ inline int coop16_sps_for(int ql) { return (ql + 129 + 31) & ~31; } // PE 127 reaches column ql one step before the last
__host__ __device__ inline int64_t tb_words_coop16(int tl, int ql) { return (int64_t)((tl + 127) >> 7) * (coop16_sps_for(ql) >> 4) * 64 * 4; }
__host__ __device__ inline int coop_query_bytes(int sps_cap) { return (sps_cap + 192 + 15) & ~15; } // 64 + ql + slack
__host__ __device__ inline int coop_wrap_cols(int sps_cap) { return sps_cap + 192; }                // 8 bytes each, per pair

// ---- sw_dp16_lane_ck_kernel (sw_dp16_lane_ck.hip): the same kernel without stored flags -- {H, E} of every 16th target row and
// the lanes' register state every LANE_CK_COLS columns are kept, the walk recomputes the few blocks it cannot check by score
constexpr int LANE_CK_COLS = 32;
__host__ __device__ inline int lane_ck_blocks(int ql) { return (ql + LANE_CK_COLS - 1) / LANE_CK_COLS; }
// dwords per WAVE, all [..][lane]: strips + 1 carry rows and strips middle rows of uint2 {H, E} per column, strips x (blocks - 1)
// checkpoints of 64 registers, one block of flags [column][lane] uint4
__host__ __device__ inline int64_t lane_ck_words(int tl, int ql)
{
    const int64_t strips = lane_strips(tl, 32);
    return ((strips + 1) * (ql + 1) * 2 + strips * ql * 2 + strips * (lane_ck_blocks(ql) - 1) * 64 + (int64_t)LANE_CK_COLS * 4) * 64;
}
// per-wave scratch: both queries and both targets of every lane transposed to [4-base block][A | B][lane] dwords
__host__ __device__ inline int64_t lane_ck_scratch_bytes(int tl, int ql) { return ((int64_t)((ql + 3) / 4) + (int64_t)lane_strips(tl, 32) * 8) * 2 * 64 * 4; }
// Results through LDS: the 128 pairs of a tile are neighbours in every output array (no `dest` reordering), so the wave builds their
// results -- CIGAR slots of up to 64 bytes, ScoreMax, offset, length, status: 12.5 KB -- in LDS and stores them as whole 1 KB lines instead
// of one scattered word per lane.  That is what lets the output arrays be the CALLER's page-locked host arrays (the host entries then
// copy nothing back: partial-line writes over the link cost a packet each).  Needs dword strides and 16-byte aligned bases.
constexpr int LANE_CK_OUT_STRIDE_MAX = 64;
constexpr int LANE_CK_OUT_LDS_BYTES = 128 * (LANE_CK_OUT_STRIDE_MAX + 24 + 12);
__host__ inline bool lane_ck_coalesced_ok(const TbArgs &a)
{
    auto al = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; };
    return !a.dest && !a.binary_cigar && a.cigar_stride >= 4 && a.cigar_stride <= LANE_CK_OUT_STRIDE_MAX && (a.cigar_stride & 3) == 0 && al(a.cigar) && al(a.offset) &&
           al(a.score) && al(a.cigar_len) && al(a.status) && (a.first & 127) == 0;
}
// The kernel is a PERSISTENT grid: a launch's waves take tile after tile (128 pairs each) and every wave keeps ONE region
// (lane_ck_words + lane_ck_scratch_bytes) that it reuses, so a launch needs as many regions as it has wave slots -- at most what the
// chip holds at two waves per SIMD -- whatever the number of pairs.
constexpr int LANE_CK_WAVES_PER_CU = 8;
__host__ __device__ inline int64_t lane_ck_region_bytes(int tl, int ql) { return lane_ck_words(tl, ql) * 4 + lane_ck_scratch_bytes(tl, ql); }
// wave slots of a launch of `pairs` pairs on n_cus CUs when `bytes` of workspace are left for regions
__host__ __device__ inline int64_t lane_ck_slots(int64_t pairs, int n_cus, int64_t bytes, int tl, int ql)
{
    const int64_t tiles = (pairs + 127) / 128, chip = (int64_t)n_cus * LANE_CK_WAVES_PER_CU, fit = bytes / lane_ck_region_bytes(tl, ql);
    const int64_t s = tiles < chip ? tiles : chip;
    return s < fit ? s : fit;
}

int64_t dp_group_bytes(int sps_cap, int rows); // carry ring + query copies of one pair (LDS, or HBM scratch)
int dp_lds_bytes(int sps_cap, int waves_per_block, int rows);
int dp16_lds_bytes(int sps, int waves_per_block);
bool dp16_range_ok(int tl, int ql, int match, int mismatch, int gopen, int gext, int strategy);
hipError_t launch_dp16(const DpArgs &a, int waves_per_block, hipStream_t stream);
bool lane16_supported(const SeqSet &t, const SeqSet &q);    // sw_dp16_lane_kernel (every flag stored): ASCII
bool lane16_ck_supported(const SeqSet &t, const SeqSet &q); // sw_dp16_lane_ck_kernel: ASCII, or both sequence sets 2-bit packed
struct TbArgs;
// a.scratch = per-wave scratch, a.tb_stride_words per wave; walk.cigar != null: every lane also walks the paths of its two pairs
hipError_t launch_dp16_lane(const DpArgs &a, const TbArgs &walk, int rows, hipStream_t stream);
// sw_dp16_lane_matrix.hip: substitution-matrix scoring, two pairs per lane, tiles of 128 pairs that share their target (MGL_SW_FLAG_SHARED_TARGET)
bool lane16_matrix_params_ok(int smin, int smax, int gopen, int gext); // every S + e + o a byte
int lane16_matrix_lds_bytes(int max_tl);
hipError_t launch_dp16_lane_matrix(const DpArgs &a, const TbArgs &walk, hipStream_t stream);
// ... what a tile of geometry tl x ql keeps in its wave slot's region: the flags (none when only scores are wanted), then the carry row and the queries as codes
__host__ __device__ inline int64_t lane16_matrix_tb_bytes(int tl, int ql, bool score_only) { return score_only ? 0 : lane_tb_words(tl, ql, 32) * 4; }
__host__ __device__ inline int64_t lane16_matrix_region_bytes(int tl, int ql, bool score_only) { return (lane16_matrix_tb_bytes(tl, ql, score_only) + lane_scratch_bytes(tl, ql, 32) + 255) / 256 * 256; }
// ... {target length, query length} of every tile's first pair, for the host to size the regions by
hipError_t launch_tile_geometry(const SeqSet &t, const SeqSet &q, int64_t first, int64_t count, int32_t *out, hipStream_t stream);
hipError_t launch_dp16_lane_ck(const DpArgs &a, const TbArgs &walk, hipStream_t stream); // a.tb_stride_words = lane_ck_words per wave; walk.cigar != null
hipError_t launch_dp(const DpArgs &a, int waves_per_block, int rows, hipStream_t stream);
int coop_lds_bytes(int sps_cap, int waves_per_block);
hipError_t launch_dp_coop(const DpArgs &a, int waves_per_block, hipStream_t stream);
bool coop16_possible(int match, int mismatch, int gopen, int gext);   // constants and margins fit 16 bits
bool coop16_worthwhile(int match, int mismatch, int gopen, int gext); // ... and a typical score window does too
int coop16_lds_bytes(int sps_cap, int waves_per_block);
// long reads, one strip of 32 rows per lane-half (sw_dp16_strip.hip): W waves per pair hold 128 W strips; steps of four columns
constexpr int STRIP_CPS = 4; // columns per step (2 or 4).  Two halve the pipeline ramp and the loop body but double the hand-overs: 46.3 ms per 1024 pairs of 10 kb against 45.6
__host__ __device__ inline int strip16_groups(int ql) { return (ql + STRIP_CPS - 1) / STRIP_CPS; }
__host__ __device__ inline int strip16_steps(int ql, int waves) { return strip16_groups(ql) + 2 * 64 * waves - 1; }
__host__ __device__ inline int strip16_qwords(int ql) { return ((((ql + 3) >> 2) + 4) + 3) & ~3; }
__host__ __device__ inline int strip16_table_words(int ql) { return (strip16_groups(ql) + 1) * 4; } // the CODES form: one dword per query column, whole groups of four + one
__host__ __device__ inline int64_t tb_words_strip16(int ql, int waves) { return (int64_t)waves * strip16_steps(ql, waves) * STRIP_CPS * 2 * 64 * 4; } // [wave][step][CPS][2][lane] uint4
__host__ __device__ inline int64_t strip16_scratch_bytes(int ql, int waves) { return ((int64_t)(ql + 8) * 2 + 2 * 33 * 64 * waves) * 4; } // per pair
// ... without stored flags (DpArgs::strip_k = K > 0, traceback layout 6): per pair {H, E} (int32, true scores) of the row below every band of
// K strips, all columns -- [band 1 .. NB - 1][column 0 .. ql] -- and {H, F} of every row at the band's checkpoint columns
// j = STRIP_CK_COLS * cc - CPS * K * band (a staircase: the strips of a band reach such a column K * band steps later than strip 0
// does) -- [cc][row 0 .. tl]
constexpr int STRIP_CK_COLS = 128;
__host__ __device__ inline int strip16_ck_bands(int tl, int rows, int k) { return (tl + rows * k - 1) / (rows * k); }
__host__ __device__ inline int strip16_ck_ccs(int tl, int ql, int rows, int k) { return (ql + STRIP_CPS * k * strip16_ck_bands(tl, rows, k)) / STRIP_CK_COLS + 2; }
__host__ __device__ inline int strip16_ck_row_stride(int ql) { return (ql + 4) & ~3; } // entries per kept row: column j at index j - 1, rows 32-byte aligned
// Round 4: everything the fill keeps is kept AS THE STRIPS HOLD IT -- a dword per entry, {H, gap value} as the two 16-bit values of the
// strip's registers (one v_perm_b32 in the fill; converting both to true scores and packing them took six to seven instructions per
// entry, on every lane, for the one strip in K that writes) -- beside the strip's baseline: value + baseline = score + (row + column) e,
// the walk's own representation.  The rows' baselines move every 16 columns (column j: block (j - 1) / 16), a checkpoint column has one per strip.
// Per pair, in dwords: the kept rows [band][column] | their baselines [band][16-column block] | the checkpoints [cc][row 0 .. tl] | their
// baselines [cc][strip].
__host__ __device__ inline int strip16_ck_strips(int tl, int rows) { return (tl + rows - 1) / rows + 1; }
__host__ __device__ inline int strip16_ck_row_blocks(int ql) { return strip16_ck_row_stride(ql) / 16 + 2; }
__host__ __device__ inline int64_t strip16_ck_off_rowbase(int tl, int ql, int rows, int k) { return (int64_t)strip16_ck_bands(tl, rows, k) * strip16_ck_row_stride(ql); }
__host__ __device__ inline int64_t strip16_ck_off_cols(int tl, int ql, int rows, int k)
{
    return (strip16_ck_off_rowbase(tl, ql, rows, k) + (int64_t)strip16_ck_bands(tl, rows, k) * strip16_ck_row_blocks(ql) + 3) & ~(int64_t)3;
}
__host__ __device__ inline int64_t strip16_ck_off_base(int tl, int ql, int rows, int k) { return strip16_ck_off_cols(tl, ql, rows, k) + (int64_t)strip16_ck_ccs(tl, ql, rows, k) * (tl + 1); }
__host__ __device__ inline int64_t strip16_ck_words(int tl, int ql, int rows, int k)
{
    return (strip16_ck_off_base(tl, ql, rows, k) + (int64_t)strip16_ck_ccs(tl, ql, rows, k) * strip16_ck_strips(tl, rows) + 3) & ~(int64_t)3;
}
int strip16_lds_bytes(int max_ql, int waves);
int strip16_lds_bytes_codes(int max_ql, int waves); // the query as one table dword per column (DpArgs::strip_codes)
int strip16_waves_per_simd(int rows);               // what the kernel of that many rows per strip is built for (2 or 3)
bool strip16_range_ok(int match, int mismatch, int gopen, int gext);
hipError_t launch_dp16_strip(const DpArgs &a, int waves, int rows, hipStream_t stream); // rows per strip: 17 .. 32; a.uni_ql = max_ql sizes the regions; a.scratch: strip16_scratch_bytes per pair
hipError_t launch_dp_coop16(const DpArgs &a, int waves_per_block, hipStream_t stream); // a.sps_cap = coop16_sps_for(max_ql)
hipError_t launch_traceback(const TbArgs &a, hipStream_t stream);
hipError_t launch_strip_ck_walk(const TbArgs &a, int max_tl, int max_ql, hipStream_t stream); // layout 6: one wave per pair, blocks recomputed (sw_strip_walk.hip)
// small batches: one wave per pair, H kept in LDS, fill + walk + text in ONE launch (sw_small.hip); a.match .. a.gext, a.cigar etc. as for the walks
bool small_supported(int max_tl, int max_ql, int cigar_stride, int match, int mismatch, int gopen, int gext, bool *wide); // *wide: the kept scores need 32 bits
int small_lds_bytes(int max_tl, int max_ql, int cigar_stride, bool wide);
hipError_t launch_small(const TbArgs &a, int max_tl, int max_ql, bool wide, hipStream_t stream);
bool small_fits_int16(int max_tl, int max_ql, int match, int mismatch, int gopen, int gext);
// small_pair() multiplies by `mismatch - match` and by gext with v_mul_i32_i24 / v_mad_i32_i24 (one pass where a 32-bit multiply takes
// four, and a lone wave waits for every pass): exact while both operands are 24-bit numbers.  The other factors are row and column
// numbers (tl <= 512, ql <= SERVICE_MAX_QL or the LDS bound) and a 0 / 1 byte.  Parameters beyond that -- the reference takes any int --
// go to the other kernels (normalised parameters: match > 0 > mismatch, gext > 0).
__host__ __device__ inline bool small_mul24_ok(int match, int mismatch, int gext) { return (int64_t)match - mismatch < (1 << 23) && gext < (1 << 23); }

// ---- one pair per call without a launch on the request path (sw_service.hip; host side: sw_service.cpp).  Every calling thread owns a
// MAILBOX (ServiceRequest + ServiceReply below); ONE resident grid serves them, workgroup k (one wave) mailbox k: the thread writes its pair and a
// new sequence number there, the wave -- polling over the link -- runs small_pair() on it and writes the results and the same number
// back.  The grid ends by itself -- all its waves within microseconds of each other, through the `stop` latch in device memory -- when
// no mailbox has had a request for `idle_ticks`, when it has lived `life_ticks` (100 MHz ticks), or when the host asks (`quit`);
// the host launches it again when a request finds its wave gone.
constexpr int SERVICE_MAX_TL = 512, SERVICE_MAX_QL = 2048;
constexpr int SERVICE_TEXT_BYTES = (2 * (SERVICE_MAX_TL + SERVICE_MAX_QL) + 4 + 3) & ~3;
constexpr int SERVICE_LDS_BYTES = 160 * 1024;     // the most a mailbox's wave may need: a whole CU's LDS
constexpr int SERVICE_LDS_DEFAULT = 64 * 1024;    // what a grid is launched with until a request needs more (a 256 x 150 pair: 40 KB): two mailboxes per CU, and room for others
enum : uint32_t { SERVICE_IDLE = 0, SERVICE_RUNNING = 1, SERVICE_EXITED = 3, SERVICE_LAUNCHED = 4 }; // low four bits of `state`; above them the grid's generation
// A mailbox is two pieces.  The REQUEST is written by the calling thread and read by the wave: it lives in fine-grained DEVICE memory
// where the host can store into device memory directly (large BAR: posted writes over the link, and the wave polls its own HBM instead
// of reading host memory -- 2.3 us against 3.5 per round trip of a 512-byte request, scripts/ubench/bar_probe.hip), else in pinned host
// memory.  The REPLY is written by the wave and read by the host: pinned host memory always (host reads over the BAR are slow).
struct alignas(64) ServiceRequest {
    // line 0.  The wave takes a request when seq_a and seq_b both show a number it has not served: whatever order the pieces of the
    // line arrive in, the fields were complete (fenced) before either number was written.
    uint32_t seq_a;
    int32_t tl, ql, match, mismatch, gopen, gext, strategy; // (normalised parameters)
    int32_t cigar_stride, wide;
    uint32_t quit_gen; // grids up to this generation are asked to end
    int32_t lds_need;  // bytes of LDS this pair takes (small_lds_bytes): a wave whose grid was launched with less ends, and the caller launches a larger one
    int32_t pad0[3];
    uint32_t seq_b;
    uint8_t t[SERVICE_MAX_TL];
    uint8_t q[SERVICE_MAX_QL];
};
struct alignas(64) ServiceReply {
    // the results first, done_seq last
    uint32_t done_seq;
    uint32_t state; // generation << 4 | SERVICE_*: the host writes LAUNCHED before a launch, the wave RUNNING when it starts and EXITED as its last store
    int32_t offset, cigar_len, status;
    Score score;
    int32_t pad1[5];
    char cigar[SERVICE_TEXT_BYTES];
};
static_assert(offsetof(ServiceRequest, t) == 64 && offsetof(ServiceReply, cigar) == 64, "one header line each");
struct ServiceControl { // device memory, zeroed once
    unsigned long long last_activity; // 100 MHz time of the last request any wave has served
    uint32_t stop_gen;                // the waves of grids up to this generation end at their next look (the host sets it to gen - 1 in front of every launch)
    uint32_t pad;
};
// `slots` workgroups on mailboxes 0 .. slots - 1
// Generations are 28-bit numbers (the reply's `state` keeps four bits for SERVICE_*) compared modulo 2^28: a is "at or after" b when
// the 28-bit difference a - b, read as a signed number, is not negative.
constexpr uint32_t SERVICE_GEN_MASK = 0x0fffffffu;
__host__ __device__ inline bool service_gen_reached(uint32_t a, uint32_t b) { return (int32_t)((a - b) << 4) >= 0; }
// `slots` workgroups on mailboxes 0 .. slots - 1, `lds_bytes` of dynamic LDS each
hipError_t launch_service(const ServiceRequest *requests, ServiceReply *replies, ServiceControl *ctl, int slots, uint32_t gen, uint32_t idle_ticks,
                          uint32_t life_ticks, int lds_bytes, hipStream_t stream);

// Device-side sort of a chunk by geometry (sw_kernels.hip): a counting sort over the (tl, ql) grid [1, max_tl] x [1, max_ql].
// Slots [0, total[0]) hold the full blocks of eight pairs of one geometry, cell after cell; the left-over pairs (fewer than
// eight per cell) follow.  With lane_blocks, the full blocks of 128 pairs of every cell come first (slots [0, total[1])): whole
// waves of sw_dp16_lane_ck_kernel.  Slot arrays are in the indexed form of SeqSet (start, length) plus dest = the pair's index in the batch.
struct RegroupArgs {
    SeqSet t, q;
    int64_t first, count; // pairs [first, first + count) of the batch
    int max_tl, max_ql;
    int32_t *cnt, *nfull, *full_start, *rest_start, *nlane, *lane_start; // max_tl * max_ql each
    int64_t *total;                                 // [0] = number of pairs in full blocks, [1] = of those, in the leading blocks of 128
    int lane_blocks;                                // a cell's full blocks of 128 pairs come first of all (slots [0, total[1]))
    int64_t *t_start, *q_start, *dest;              // count each
    int32_t *t_len, *q_len;
};
hipError_t launch_regroup(const RegroupArgs &a, hipStream_t stream);
hipError_t launch_scores_only(const TbArgs &a, hipStream_t stream); // DpRecord -> ScoreMax, no path walk
hipError_t launch_iota64(int64_t *dst, int64_t n, int64_t step, hipStream_t stream); // dst[k] = k * step, k = 0 .. n - 1 (the offsets of a uniform ASCII batch, made where they are used)
hipError_t launch_cigar_from_matrix(const int32_t *btr, int tl, int ql, int strategy, const Score &ez, char *cigar,
                                    int cap, int32_t *out3, hipStream_t stream);
hipError_t launch_expand(const uint32_t *tbw, const DpRecord *rec, int tl, int ql, int packed16, int half, int rows,
                         int32_t *btr, hipStream_t stream, int lane = 0);

hipError_t launch_band_fill(const int32_t *target, const int32_t *query, int ql, int32_t *band_btr, int band, int bw, int actual_bw,
                            int32_t *score, int32_t *step, int32_t *gap, int match, int mismatch, int gopen, int gext,
                            int strategy, int32_t *mqe_io, hipStream_t stream);

} // namespace mgl_sw_dev
#endif
