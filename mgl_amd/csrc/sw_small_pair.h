// sw_small_pair.h -- one pair, one wave, everything on chip: the body shared by sw_small_kernel (sw_small.hip: one launch per small
// batch) and sw_service_kernel (sw_service.hip: resident waves that serve one-pair-per-call requests out of pinned mailboxes).
//
// One wave per pair, one launch, nothing in HBM but the inputs and the results:
//
//   * fill: lane L owns target rows R*L+1 .. R*L+R (R = 2, 4, 6 or 8: targets up to 512 rows) and moves along the query one
//     column per step, one column behind the lane below it (the anti-diagonal scheme of the reference's AVX2 path,
//     sw_avx.cpp:71-80, with the 8 int32 lanes of a ymm register widened to 64 lanes x R rows); the row above a lane's first
//     row comes over `wave_shr:1` DPP, the rows inside a lane are a register chain.  Scores are 32-bit and true (no offset
//     representation): the recurrence of sw.cpp:51-96 as it stands.
//   * what is kept is H and nothing else: every cell's H goes to LDS (two rows per dword as int16 when the score range of the
//     geometry fits, else one int32 per cell) -- no traceback flags, no run lengths.
//   * walk: the same wave walks the path (calculateCigar, sw.cpp:149-255: walk_and_write in sw_traceback.h) and reads every
//     move off the H values:
//       - a cell took the diagonal  <=>  H[i][j] == H[i-1][j-1] + s(i, j)        (diag >= E and diag >= F, sw.cpp:60-71; H is
//         the maximum of the three, so equality with the diagonal candidate is exactly that);  64 cells of a diagonal are
//         checked per round, one per lane;
//       - otherwise F[i][j] = max_k H[i][j-k] - o - (k-1) e and E[i][j] = max_k H[i-k][j] - o - (k-1) e are re-evaluated from
//         the stored row / column (one candidate per lane), the move is F when F >= E (sw.cpp:64-70) and the run length the
//         reference stores is the LARGEST k that attains the maximum: an extension wins a tie against a new gap at every cell
//         (sw.cpp:73-93), so the gap that reaches (i, j) was opened at the farthest cell whose candidate is not beaten by a
//         farther one.
//   * results (offset, ScoreMax, CIGAR text, length, status) are written where the traceback kernels write them; the text is
//     built in LDS and leaves the wave as whole dwords.
//
// Geometry limits (small_supported): tl <= 512, the H matrix + sequences + text within a workgroup's LDS, text CIGARs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "sw_device.h"
#include "sw_traceback.h"

namespace mgl_sw_dev {

// -DMGL_SMALL_PHASES (scripts/build_variant.sh): every wave adds the 100 MHz ticks it spent in each part of the kernel to a device
// array; mgl_small_phases_dump() (exported by that build only) prints and clears it
#ifdef MGL_SMALL_PHASES
__device__ unsigned long long mgl_small_phase_ticks[8];
#define SMALL_PHASE(k)                                                                                 \
    do {                                                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                              \
        if (lane == 0) atomicAdd(&mgl_small_phase_ticks[k], now_ - phase_t0);                          \
        phase_t0 = __builtin_amdgcn_s_memrealtime();                                                   \
    } while (0)
#else
#define SMALL_PHASE(k) do { } while (0)
#endif

// LDS of one workgroup (= one wave = one pair), sized by the bounds of the batch; the kernel carves by the same formulas
__host__ __device__ inline int small_rows_per_lane(int tl) { return ((tl + 63) / 64 + 1) & ~1; }
// dwords per kept column: the rows the lanes own (whole lanes: up to R - 1 rows past tl), parity chosen against the lanes' stride
__host__ __device__ inline int small_column_words(int tl, bool wide)
{
    const int R = small_rows_per_lane(tl), nl = (tl + R - 1) / R;
    if (wide) return (nl * R) | 1;
    const int cs = nl * R / 2;
    return ((R / 2 - cs) & 1) ? cs : cs + 1;
}
__host__ __device__ inline int small_text_cap(int tl, int ql, int cigar_stride)
{
    const int need = 2 * (tl + ql) + 4; // an element of length n takes at most 2 n characters; the lengths add up to at most tl + ql
    return ((cigar_stride < need ? cigar_stride : need) + 3) & ~3;
}
// can the kept scores be 16-bit?  What is kept is V = H + (i + j) e - base with
//   H <= match * min(tl, ql)                          (gaps and borders cost, mismatches do not pay)
//   H[i][j] >= E[i][j] >= H[0][j] - o - (i - 1) e >= -2 o - (i + j) e
// so V + base lies in [-2 o - (tl + ql) e, match * min(tl, ql) + (tl + ql) e]; base is the middle of that range
__host__ __device__ inline int small_base(int tl, int ql, int match, int gopen, int gext)
{
    const int hi = match * (tl < ql ? tl : ql) + (tl + ql) * gext, lo = -2 * gopen - (tl + ql) * gext;
    return (hi + lo) / 2;
}

namespace small_detail {

constexpr int WAVE_SHR1 = 0x138;

// every lane takes `src` of the lane below it; lane 0 keeps `lane0_value`
__device__ __forceinline__ int from_lane_below(int lane0_value, int src)
{
    return __builtin_amdgcn_update_dpp(lane0_value, src, WAVE_SHR1, 0xf, 0xf, false);
}

// the lexicographically largest (hi, lo) of the wave -- hi signed, lo unsigned -- in every lane.  Six DPP stages (prefix maxima inside
// the rows of 16 lanes, then row to row) instead of six rounds of ds_bpermute: the reductions of a pair (best of the last column, best
// of the last row, the candidates of every gap the walk meets) sit on its latency path.
__device__ __forceinline__ void wave_max_pair(int &hi, uint32_t &lo)
{
#define MGL_SMALL_MAX_STAGE(CTRL, ROWS)                                                                          \
    {                                                                                                            \
        const int oh = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROWS, 0xf, false);                              \
        const uint32_t ol = (uint32_t)__builtin_amdgcn_update_dpp((int)lo, (int)lo, CTRL, ROWS, 0xf, false);     \
        const bool take = oh > hi || (oh == hi && ol > lo); /* (a lane the stage does not reach sees itself) */  \
        hi = take ? oh : hi;                                                                                     \
        lo = take ? ol : lo;                                                                                     \
    }
    MGL_SMALL_MAX_STAGE(0x111, 0xf) // row_shr:1
    MGL_SMALL_MAX_STAGE(0x112, 0xf) // row_shr:2
    MGL_SMALL_MAX_STAGE(0x114, 0xf) // row_shr:4
    MGL_SMALL_MAX_STAGE(0x118, 0xf) // row_shr:8: lane 15 of every row holds its row's maximum
    MGL_SMALL_MAX_STAGE(0x142, 0xa) // row_bcast:15 into rows 1 and 3
    MGL_SMALL_MAX_STAGE(0x143, 0xc) // row_bcast:31 into rows 2 and 3: lane 63 holds the wave's
#undef MGL_SMALL_MAX_STAGE
    hi = __builtin_amdgcn_readlane(hi, 63);
    lo = (uint32_t)__builtin_amdgcn_readlane((int)lo, 63);
}

// The wave's largest value, in every lane: six DPP stages folded into the maximum itself (v_max_*_dpp: a lane the stage does not reach keeps
// its own value).  The reductions of a pair sit on its latency path and a lone wave pays ~5 cycles for every instruction of any kind: the
// lexicographic (value, key) form above takes eleven instructions a stage, this one one -- so the maxima of the last column and the last
// row are taken in two passes each (the best score; then, among the lanes that hold it, the best key): round 5.
// (inline assembly: the compiler keeps the DPP move and the maximum apart -- v_mov, v_mov_dpp, s_nop, v_max: four issue slots a stage;
// a DPP operand read behind a VALU write of the same register needs two wait states: the s_nop 1 between the stages)
#define MGL_SMALL_DPP_MAX(OP)                                                            \
    asm volatile("s_nop 1\n\t" /* (whatever VALU instruction wrote v last: inside this block, where nothing can be scheduled behind it) */ \
                 OP " %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"     \
                 OP " %0, %0, %0 row_shr:2 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"     \
                 OP " %0, %0, %0 row_shr:4 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"     \
                 OP " %0, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf\n\ts_nop 1\n\t"     \
                 OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 1\n\t"  \
                 OP " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 1"       \
                 : "+v"(v))
__device__ __forceinline__ int wave_max_i32(int v)
{
    MGL_SMALL_DPP_MAX("v_max_i32_dpp");
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v)
{
    MGL_SMALL_DPP_MAX("v_max_u32_dpp");
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
#undef MGL_SMALL_DPP_MAX

// H on row 0 / column 0 (k = the other index): sw.cpp:29-40,47-49
__device__ __forceinline__ int edge_score(int k, int gopen, int gext, bool indel) { return (indel && k > 0) ? -gopen - (k - 1) * gext : 0; }

struct SmallGeom {
    int tl, ql;
    int CS;     // dwords per kept column: the column's rows (two per dword in the 16-bit form), made odd against the lanes' stride
    int R, nl;  // rows per lane, lanes that own rows
    int match, mismatch, gopen, gext;
    int base;   // what is kept of cell (i, j) is V = H + (i + j) * gext - base
    bool indel, wide;
};

// the kept scores: V[i][j] = H[i][j] + (i + j) * gext - base for 1 <= i <= tl, 1 <= j <= ql, column-major (a lane's rows of one
// column are neighbours: one address register and immediate offsets in the fill; the lanes of a step, one column apart, are
// R/2 - CS or R - CS dwords apart, an odd number: 64 different banks); at() gives H back; the borders are formulas
struct KeptScores {
    const uint32_t *hm;
    int CS, gopen, gext, base;
    bool indel, wide;
    __device__ __forceinline__ int at(int i, int j) const
    {
        if (i == 0 || j == 0) return edge_score(i + j, gopen, gext, indel);
        const int un = base - __mul24(i + j, gext); // (24-bit products: one pass each where a 32-bit multiply takes four)
        if (wide) return (int)hm[__mul24(j - 1, CS) + (i - 1)] + un;
        const uint32_t w = hm[__mul24(j - 1, CS) + ((i - 1) >> 1)];
        return (int)(int16_t)(((i - 1) & 1) ? (w >> 16) : (w & 0xffffu)) + un;
    }
};

// The fill.  Values carry the offset (i + j) * gext - base: every decision compares values of one cell, so the decisions are those of
// sw.cpp:51-96, but extending a gap needs no instruction, both new gaps start from the same H - (o - e), and the constant `base`
// (it enters through the borders and travels with every maximum) centres the range of the 16-bit form.
// CODES: the staged sequences hold base codes -- a target base its code 0 .. 3, a query base its code or 4 for a byte that is none of
// ACGT (small_pair stages them so when every target byte is one of ACGT) -- and "do the bases differ" is looked up four columns at a
// time: one v_perm_b32 per row and block of four steps (the row's table: byte c = 0 iff the target base has code c; the selector:
// the four query codes; code 4 selects the constant 1), then a multiply and a three-operand add per cell where the byte compare
// took a compare, a select and an add.  Raw bytes (CODES = false) are compared as they are (sw.cpp:55: N == N, a != A).
template <int R, bool WIDE, bool CODES>
__device__ __forceinline__ void small_fill(const SmallGeom &g, uint32_t *hm, const uint8_t *ts, const uint8_t *qs, const int lane)
{
    const int i0 = R * lane; // the row above this lane's first row
    const int gext = g.gext, o_e = g.gopen - g.gext, ql = g.ql, CS = g.CS;
    int h[R], f[R], tb[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int i = i0 + 1 + r;
        h[r] = edge_score(i, g.gopen, gext, g.indel) + i * gext - g.base; // V[i][0]
        f[r] = h[r] - o_e;                                                 // F[i][1] (sw.cpp:47-49)
        if (CODES)
            tb[r] = i <= g.tl ? (int)(0x01010101u ^ (1u << (8 * (ts[i - 1] & 3)))) : 0x01010101; // the row's table; rows past tl differ from everything
        else
            tb[r] = i <= g.tl ? (int)ts[i - 1] : 0x100; // rows past tl: never equal to a query byte; their scores go nowhere
    }
    int up_diag = edge_score(i0, g.gopen, gext, g.indel) + i0 * gext - g.base; // V[i0][j-1]: column 0 until this lane starts
    int e_bot = 0;                                                             // E leaving this lane's last row, as of the previous step
    int match2 = g.match + 2 * gext, mismatch2 = g.mismatch + 2 * gext;        // the diagonal moves the offset by 2 e
    int delta = g.mismatch - g.match;
    asm volatile("" : "+v"(match2), "+v"(mismatch2), "+v"(delta));
    const int steps = ql + g.nl - 1;
    const unsigned my_ql = lane < g.nl ? (unsigned)ql : 0u; // lanes without rows never start
    uint32_t *wp = hm + (WIDE ? i0 : (i0 >> 1)) - lane * CS;  // column j - 1 = s - lane: + s * CS
    // V[0][j] of lane 0's column j = s + 1: j e - base, or with leading / trailing gaps charged -o - (j - 1) e + j e - base.  Lane 0 has
    // no lane below it: the DPP move leaves its registers as they are, so what arrives "from below" there is what the register held --
    // the border, moved on by one column per step by an add that the other lanes' moves overwrite (one VALU instruction per register
    // and step, where a border kept in scalar registers cost a scalar add and a move each)
    int up_h = (g.indel ? -o_e : gext) - g.base - (g.indel ? 0 : gext); // (one step early: the loop adds first)
    int up_e = up_h - o_e;
    int edge_step = g.indel ? 0 : gext;
    asm volatile("" : "+v"(edge_step));
    // this lane's query bases of the next four steps: bytes s0 - lane .. + 3 of the query, out of two aligned dwords (before the
    // lane starts and behind the query's end these are addresses of other LDS data or of none -- read as whatever, never used)
    const int q_shift = (-lane) & 3;
    const uint32_t *qd = reinterpret_cast<const uint32_t *>(qs) + ((0 - lane) >> 2);
    uint32_t q_lo = qd[0], q_hi = qd[1];
    // four steps (the four query bases of one dword).  MASKED: lanes whose column lies outside 1 .. ql sit the step out (the ramps of
    // the wavefront: the lanes above have not started, the lanes below are through); without it every lane has a cell in every
    // step -- no compare, no exec juggling, no branch: a lone wave pays ~5 cycles for every instruction of any kind.
    // The diagonal candidates of all rows are taken from the previous column's H before any row overwrites it: H stays in place (no
    // copy per row) and the additions leave the chain of maxima.
    // CLAMP (with !MASKED): lanes that are through compute on (what they produce reaches nobody: a lane only ever hands values to the
    // lane above it, which is through one step later) and store into the spare column behind the matrix
    uint32_t *const wp_spare = hm + (WIDE ? i0 : (i0 >> 1)) + ql * CS;
    auto block = [&](auto masked, auto clamp, const int s0) {
        constexpr bool MASKED = decltype(masked)::value, CLAMP = decltype(clamp)::value;
        const uint32_t qw = __builtin_amdgcn_alignbyte(q_hi, q_lo, q_shift);
        qd += 1;
        q_lo = qd[0]; // (used one block later: by then four steps of stores sit behind it in the queue, nothing waits)
        q_hi = qd[1];
        uint32_t differ4[R]; // CODES: byte u = 1 where the row's base and the query base of step u differ
        if (CODES) {
#pragma unroll
            for (int r = 0; r < R; ++r) differ4[r] = __builtin_amdgcn_perm(0x01010101u, (uint32_t)tb[r], qw);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = s0 + u - lane + 1;
            const int qb = (int)((qw >> (8 * u)) & 0xffu);
            up_h = from_lane_below(up_h + edge_step, h[R - 1]); // V[i0][j]
            up_e = from_lane_below(up_e + edge_step, e_bot);    // E[i0 + 1][j] (row 0: sw.cpp:31-35)
            if (!MASKED || (unsigned)(j - 1) < my_ql) {
                int diag[R];
                if (CODES) { // sw.cpp:55 by lookup
                    // (a 24-bit multiply: v_mad_i32_i24, one pass -- the 32-bit v_mul_lo_u32 the plain product compiles to takes four, and
                    // a lone wave waits for every one of them: four of a step's forty instructions were a third of its time)
                    diag[0] = __mul24((int)((differ4[0] >> (8 * u)) & 0xffu), delta) + (up_diag + match2);
#pragma unroll
                    for (int r = 1; r < R; ++r) diag[r] = __mul24((int)((differ4[r] >> (8 * u)) & 0xffu), delta) + (h[r - 1] + match2);
                } else {
                    diag[0] = up_diag + (tb[0] == qb ? match2 : mismatch2); // sw.cpp:55
#pragma unroll
                    for (int r = 1; r < R; ++r) diag[r] = h[r - 1] + (tb[r] == qb ? match2 : mismatch2);
                }
                up_diag = up_h;
                int e_run = up_e;
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int hn = max(max(diag[r], f[r]), e_run); // sw.cpp:60-71
                    const int open = hn - o_e;
                    f[r] = max(open, f[r]);   // F[i][j + 1], sw.cpp:84-93
                    e_run = max(open, e_run); // E[i + 1][j], sw.cpp:73-82
                    h[r] = hn;
                }
                e_bot = e_run;
                uint32_t *const out = CLAMP ? (wp < wp_spare ? wp : wp_spare) : wp;
                if (WIDE) {
#pragma unroll
                    for (int r = 0; r < R; ++r) out[r] = (uint32_t)h[r];
                } else {
#pragma unroll
                    for (int r = 0; r < R; r += 2) out[r >> 1] = __builtin_amdgcn_perm((uint32_t)h[r + 1], (uint32_t)h[r], 0x05040100u); // low halves of both, one instruction
                }
            }
            wp += CS;
        }
    };
    // every lane is inside the matrix from step 63 (the last lane has started) to step ql - 1 (the first one is through) -- when all
    // 64 lanes own rows; blocks of four steps that lie within that run unmasked
    int s0 = 0;
    if (g.nl == 64) {
        for (; s0 < 64 && s0 < steps; s0 += 4) block(std::true_type{}, std::false_type{}, s0);
        for (; s0 + 3 <= ql - 1; s0 += 4) block(std::false_type{}, std::false_type{}, s0);
        if (s0 >= 64) // (the ramp down: from here on every lane has started)
            for (; s0 < steps; s0 += 4) block(std::false_type{}, std::true_type{}, s0);
    }
    for (; s0 < steps; s0 += 4) block(std::true_type{}, std::false_type{}, s0);
}

// the moves of a path, read off the kept scores by the whole wave (every lane returns the same value)
struct ScoreMoves {
    KeptScores hs;
    const uint8_t *ts, *qs;
    int match, mismatch, lane;

    // (value, k) with the largest value and, among equals, the largest k -- over the wave
    __device__ __forceinline__ static void wave_best(int &v, int &k)
    {
        uint32_t kk = (uint32_t)k; // (k >= 0)
        wave_max_pair(v, kk);
        k = (int)kk;
    }
    // F[i][j] and the length of the horizontal gap that ends at (i, j): candidates k = 1 .. j from H[i][j - k]
    __device__ __forceinline__ void row_gap(int i, int j, int &val, int &len) const
    {
        val = NEG_INF;
        len = 0;
        for (int b = 0; b < j; b += 64) {
            const int k = b + lane + 1;
            int v = k <= j ? hs.at(i, j - k) - hs.gopen - (k - 1) * hs.gext : NEG_INF, kk = k <= j ? k : 0;
            wave_best(v, kk);
            const bool take = v >= val; // a later round holds larger k
            len = take ? kk : len;
            val = take ? v : val;
        }
    }
    __device__ __forceinline__ void column_gap(int i, int j, int &val, int &len) const
    {
        val = NEG_INF;
        len = 0;
        for (int b = 0; b < i; b += 64) {
            const int k = b + lane + 1;
            int v = k <= i ? hs.at(i - k, j) - hs.gopen - (k - 1) * hs.gext : NEG_INF, kk = k <= i ? k : 0;
            wave_best(v, kk);
            const bool take = v >= val;
            len = take ? kk : len;
            val = take ? v : val;
        }
    }
    // +k (k rows up), -k (k columns left) or 0 (diagonal): the value the reference's btrack holds at (i, j)
    __device__ __forceinline__ int at(int i, int j) const
    {
        const int h = hs.at(i, j);
        const int diag = hs.at(i - 1, j - 1) + (ts[i - 1] == qs[j - 1] ? match : mismatch);
        if (h == diag) return 0;
        int fv, fk;
        row_gap(i, j, fv, fk);
        if (fv == h) return -fk; // F >= E (F is the maximum): sw.cpp:64-70
        int ev, ek;
        column_gap(i, j, ev, ek);
        return ek;
    }
    // the number of diagonal moves that start at (i, j), up to 64: lane k looks at (i - k, j - k)
    __device__ __forceinline__ int diag_run(int i, int j) const
    {
        const int ii = i - lane, jj = j - lane;
        bool is_diag = false;
        if (ii >= 1 && jj >= 1) is_diag = hs.at(ii, jj) == hs.at(ii - 1, jj - 1) + (ts[ii - 1] == qs[jj - 1] ? match : mismatch);
        const unsigned long long m = __ballot(is_diag);
        return m == ~0ull ? 64 : __builtin_ctzll(~m);
    }
};

} // namespace small_detail

// One pair: sequences a.t / a.q from element t0 / q0 on, lengths tl / ql; parameters, strategy and the result arrays from `a`, results at
// index o.  `lds`: small_lds_bytes(tl, ql, a.cigar_stride, wide) bytes at least.  The whole wave calls this together.  PAD: the CIGAR slot
// is zero-filled behind the text, as the batch entries promise (the mailboxes of sw_service.hip take the text alone over the link).
template <bool PAD>
__device__ __forceinline__ void small_pair(const TbArgs &a, const int64_t o, const int tl, const int ql, const int64_t t0, const int64_t q0,
                                           uint32_t *lds, const int wide, const int lane)
{
    using namespace small_detail;

    SmallGeom g;
    g.tl = tl;
    g.ql = ql;
    g.R = small_rows_per_lane(tl);
    g.nl = (tl + g.R - 1) / g.R;
    g.match = a.match;
    g.mismatch = a.mismatch;
    g.gopen = a.gopen;
    g.gext = a.gext;
    g.indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;
    g.wide = wide != 0;
    g.CS = small_column_words(tl, g.wide);
    g.base = g.wide ? 0 : small_base(tl, ql, g.match, g.gopen, g.gext);

    // carve (by this pair's own lengths: never more than the batch's bounds give)
    uint32_t *hm = lds;
    const int kept_words = g.CS * (ql + 1); // (+ the spare column small_fill's lanes store into once they are through)
    uint8_t *ts = reinterpret_cast<uint8_t *>(hm + kept_words);
    uint8_t *qs = ts + ((tl + 3) & ~3);
    char *text = reinterpret_cast<char *>(qs + ((ql + 3) & ~3));
    const int text_cap = small_text_cap(tl, ql, a.cigar_stride);

#ifdef MGL_SMALL_PHASES
    unsigned long long phase_t0 = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) atomicAdd(&mgl_small_phase_ticks[7], 1ull);
#endif
    // the bases come over the link when the batch sits in pinned host memory (the coalescing front-end's): every load of a
    // sequence is in flight before the first one is used -- one round trip, not one per 64 bases
    // Base codes where the target allows it: A 0, C 1, T 2, G 3 (bits 1 and 2 of the letter); a byte is "one of ACGT" when the letter of
    // its code is the byte itself.  The whole target (at most 512 bytes) is in registers before the wave decides.
    auto code_of = [](int b, bool &is_base) {
        const int c = (b >> 1) & 3;
        is_base = (int)((0x47544341u >> (8 * c)) & 0xffu) == b;
        return c;
    };
    bool codes;
    {
        int v[8];
        bool bad = false;
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 64 * k + lane < tl ? a.t.at(t0, 64 * k + lane) : 'A';
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            bool ok;
            (void)code_of(v[k], ok);
            bad |= !ok;
        }
        codes = __builtin_amdgcn_ballot_w64(bad) == 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (64 * k + lane < tl) {
                bool ok;
                const int c = code_of(v[k], ok);
                ts[64 * k + lane] = (uint8_t)(codes ? c : v[k]);
            }
    }
    for (int x0 = 0; x0 < ql; x0 += 512) {
        int v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = x0 + 64 * k + lane < ql ? a.q.at(q0, x0 + 64 * k + lane) : 0;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (x0 + 64 * k + lane < ql) {
                bool ok;
                const int c = code_of(v[k], ok);
                qs[x0 + 64 * k + lane] = (uint8_t)(codes ? (ok ? c : 4) : v[k]);
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    SMALL_PHASE(0); // staging

#define MGL_SMALL_FILL(RR)                                                                  \
    case RR:                                                                                \
        if (g.wide) {                                                                       \
            if (codes)                                                                      \
                small_fill<RR, true, true>(g, hm, ts, qs, lane);                            \
            else                                                                            \
                small_fill<RR, true, false>(g, hm, ts, qs, lane);                           \
        } else {                                                                            \
            if (codes)                                                                      \
                small_fill<RR, false, true>(g, hm, ts, qs, lane);                           \
            else                                                                            \
                small_fill<RR, false, false>(g, hm, ts, qs, lane);                          \
        }                                                                                   \
        break;
    switch (g.R) {
        MGL_SMALL_FILL(2)
        MGL_SMALL_FILL(4)
        MGL_SMALL_FILL(6)
    default:
        MGL_SMALL_FILL(8)
    }
#undef MGL_SMALL_FILL
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    SMALL_PHASE(1); // fill

    KeptScores hs{hm, g.CS, g.gopen, g.gext, g.base, g.indel, g.wide};
    // last column: best score, the later row on ties (sw.cpp:100-104).  A lane looks at the rows it OWNS (R neighbours of column ql in the
    // kept matrix: one address, immediate offsets) -- then the wave's best score, then the largest row among the lanes that hold it
    int mqe = NEG_INF, mqe_t = 0;
    {
        const int i0 = g.R * lane;
        const uint32_t *const colp = hm + (ql - 1) * g.CS + (g.wide ? i0 : i0 >> 1); // this lane's rows of column ql (whole lanes: rows past tl exist)
        const int un0 = g.base - (i0 + 1 + ql) * g.gext;
        for (int r = 0; r < g.R; ++r) { // (a scalar loop: R is the pair's)
            const int raw = g.wide ? (int)colp[r] : (int)(int16_t)((r & 1) ? colp[r >> 1] >> 16 : colp[r >> 1] & 0xffffu);
            const int sc = i0 + 1 + r <= tl ? raw + un0 - r * g.gext : NEG_INF;
            mqe_t = sc >= mqe ? i0 + 1 + r : mqe_t;
            mqe = max(sc, mqe);
        }
        const int m = wave_max_i32(mqe);
        mqe_t = (int)wave_max_u32(mqe == m ? (uint32_t)mqe_t : 0u);
        mqe = m;
    }
    // last row: best score, then closest to the diagonal, then smallest column (sw.cpp:106-127): the best score first, then the best
    // (distance, column) among the cells that hold it (both below 2^16 -- the kept scores of a longer query would not fit LDS)
    int rm = NEG_INF, rd, rj;
    {
        uint32_t key = 0u;
        for (int j = lane + 1; j <= ql; j += 64) {
            const int sc = hs.at(tl, j);
            const uint32_t k2 = (uint32_t)(0xffff - abs(tl - j)) << 16 | (uint32_t)(0xffff - j);
            key = sc > rm ? k2 : sc == rm ? max(key, k2) : key;
            rm = max(sc, rm);
        }
        const int m = wave_max_i32(rm);
        key = wave_max_u32(rm == m ? key : 0u);
        rm = m;
        rd = 0xffff - (int)(key >> 16);
        rj = 0xffff - (int)(key & 0xffffu);
    }
    const bool row_wins = rm > mqe || (rm == mqe && rd < abs(mqe_t - ql));
    Score sc;
    sc.mqe = mqe;
    sc.mqe_t = mqe_t;
    sc.max = row_wins ? rm : mqe;
    sc.max_t = row_wins ? tl : mqe_t;
    sc.max_q = row_wins ? rj : ql;
    sc.seg_length = row_wins ? ql - rj : 0;

    SMALL_PHASE(2); // last column, last row
    // ---- the walk
    ScoreMoves mv{hs, ts, qs, g.match, g.mismatch, lane};
    CigarWriter cw;
    cw.slot = text;
    cw.binary = 0;
    cw.cap = text_cap < a.cigar_stride ? text_cap : a.cigar_stride; // (text_cap is rounded up to a dword)
    cw.pos = cw.cap;
    cw.need = 0;
    cw.store = lane == 0;
    const int off = walk_and_write(mv, tl, ql, a.strategy, sc.max_t, sc.max_q, sc.mqe_t, sc.seg_length, cw);
    const int status = cw.pos < 0 ? ERR_CIGAR_OVERFLOW : 0;
    const int len = status ? 0 : cw.cap - cw.pos;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    SMALL_PHASE(3); // walk

    // ---- results: the text to the front of the slot, zeros behind it (finish_cigar; PAD = false: the text alone, up to the end of its
    // last dword), whole dwords where the slot allows
    char *slot = a.cigar + (size_t)o * a.cigar_stride;
    const char *src = text + (status ? 0 : cw.pos);
    const int out_bytes = PAD ? a.cigar_stride : ((len + 3) & ~3);
    if (((reinterpret_cast<uintptr_t>(slot) | (uintptr_t)a.cigar_stride) & 3) == 0) {
        for (int k = lane * 4; k < out_bytes; k += 256) {
            uint32_t w = 0;
#pragma unroll
            for (int x = 0; x < 4; ++x)
                if (k + x < len) w |= (uint32_t)(uint8_t)src[k + x] << (8 * x);
            *reinterpret_cast<uint32_t *>(slot + k) = w;
        }
    } else {
        for (int k = lane; k < (PAD ? a.cigar_stride : len); k += 64) slot[k] = k < len ? src[k] : (char)0;
    }
    if (lane == 0) {
        a.offset[o] = off;
        if (a.cigar_len) a.cigar_len[o] = cw.need;
        if (a.status) a.status[o] = status;
        if (a.status_any && status != 0) atomicMax(a.status_any, status);
        if (a.score) a.score[o] = sc;
    }
    SMALL_PHASE(4); // results
}

} // namespace mgl_sw_dev
