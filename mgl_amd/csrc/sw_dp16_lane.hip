// sw_dp16_lane.hip -- packed-int16 fill kernel for gfx950 with TWO PAIRS PER LANE and no cross-lane traffic at all:
// 128 pairs per wave64, every lane runs its own two matrices (low / high 16 bits of every register) from the first cell
// to the last.  Same function as sw_dp_kernel / sw_dp16_kernel (the reference's sw.cpp:5-146), same 16-bit offset
// representation and range guard as sw_dp16.hip, so every decision -- hence the traceback -- is bit-identical.
//
// Why: sw_dp16_kernel (16 lanes of a DPP row share a pair, anti-diagonal wavefront) is VALU-issue bound at ~19.5
// instructions per two-cell step, of which 2 are the DPP moves that hand H and E to the next row's lane, and it pays
// ~12 % on top for what the wavefront needs around the recurrence (the 16 window steps of every chained stripe with their
// lane-mask selects, ring reads and writes, pipeline fill / drain, stripe set-up).  Large uniform batches have enough
// pairs to give every LANE its own pair: then "up", "diagonal" and "left" are registers of the same lane, nothing moves
// between lanes, there is no LDS, no ring, no window and no fill / drain -- 16 packed + 1 plain VALU instruction per two
// cells, every cycle of them useful.
//
// Schedule of one lane: the matrix is cut into strips of R target rows.  A strip keeps, per row, H of the previous column
// and F in registers (2R VGPRs) plus the row's two target bases, and sweeps the query columns 1 .. ql; inside a column the R
// cells are computed top to bottom (E and "up" are the running values of the column, "diagonal" is the H the row above
// held before this column).  What a strip hands to the next one -- H and E of its last row, per column: the reference's
// score[] / step[] carry of sw_avx.cpp:36-47,196-197 -- goes through a per-wave row in HBM, laid out [column][lane] so
// that a wave's access is one contiguous 512-byte line pair, updated in place (column j of the row is loaded, two columns
// ahead of use, before the strip overwrites it).  Its traffic is 16 bytes per column and lane per strip = 0.25 B per cell
// at R = 32, on top of the 0.5 B per cell of traceback; it is re-read ~150 columns after it was written, by the same lane.
//
// Traceback layout ("lane16", TbArgs.packed16 == 2): per wave [strip][column][16-row group][lane] uint4; dword q of the
// uint4 holds rows 4q .. 4q+3 of the group in the byte layout of sw_dp16.hip (byte0 = pair A {E>S, F opened}, byte1 = pair
// B, byte2 = A {F>diag, E opened}, byte3 = B; row t of the four in bits 2t+1, 2t).  A wave stores 1 KB per instruction.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sw_device.h"
#include "sw_lane_cell.h"
#include "sw_traceback.h"

namespace mgl_sw_dev {

namespace {

template <int R, bool NOTB, bool LAST>
__device__ __forceinline__ void lane_strip(const int i0, const int tl, const int ql, uint2 *bnd, const unsigned *qst, const unsigned *tst,
                                           uint4 *&tbp, const LaneConsts &c, const int gopen, const int gext, const int base,
                                           const bool indel, int &bestA, int &bestA_i, int &bestB, int &bestB_i)
{
    unsigned h[R], f[R], t[R];
    // ---- the strip's target bases: t[r] = {A's base, B's base} of row i0 + r + 1, one byte per half
#pragma unroll
    for (int r4 = 0; r4 < R / 4; ++r4) {
        const unsigned ta = tst[(size_t)(2 * ((i0 >> 2) + r4)) * 64], tb = tst[(size_t)(2 * ((i0 >> 2) + r4) + 1) * 64];
        t[4 * r4 + 0] = __builtin_amdgcn_perm(tb, ta, 0x0c040c00u);
        t[4 * r4 + 1] = __builtin_amdgcn_perm(tb, ta, 0x0c050c01u);
        t[4 * r4 + 2] = __builtin_amdgcn_perm(tb, ta, 0x0c060c02u);
        t[4 * r4 + 3] = __builtin_amdgcn_perm(tb, ta, 0x0c070c03u);
    }
    // ---- column 0: H[i][0] border values, F[i][1] = H[i][0] - o (sw.cpp:24,38,47-49), in stored form
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = i0 + r + 1;
        const int hb = border(row, gopen, gext, indel) + row * gext + base;
        h[r] = pack2(hb, hb);
        f[r] = pk_sub(h[r], c.o_e);
    }
    const int hd0 = border(i0, gopen, gext, indel) + i0 * gext + base; // H[i0][0]
    unsigned hd = pack2(hd0, hd0);
    const int rl = tl - 1 - i0; // LAST: register row of target row tl (0 .. R-1)

    // ---- columns 1 .. ql, four at a time (one dword of each query).  All loads of a group are issued at its top and used
    // inside the same straight-line block, so the compiler can give every use a counted s_waitcnt (an in-flight load
    // carried around a loop edge gets vmcnt(0) -- a full drain, stores included -- at every column: measured 37 % of the
    // waves' time in s_waitcnt); the one exposed latency per group is covered by the SIMD's other waves.
    uint2 *bp = bnd + 64;                                   // column j
    auto one_column = [&](const uint2 top, const unsigned q) {
        unsigned e = top.y;
        column<R, NOTB>(h, f, t, q, hd, e, c, tbp);
        if (!NOTB) tbp += (R / 16) * 64;
        hd = top.x;
        if (!LAST) {
            bp[0] = make_uint2(h[R - 1], e);
        } else {
            unsigned bot = h[R - 1];
            if (rl != R - 1) {
#pragma unroll
                for (int r = 0; r < R - 1; ++r) bot = (r == rl) ? h[r] : bot;
            }
            bp[0] = make_uint2(bot, 0u);
        }
        bp += 64;
    };
    int j = 1;
    for (; j + 3 <= ql; j += 4) {
        const uint2 top0 = bp[0], top1 = bp[64], top2 = bp[128], top3 = bp[192];
        const unsigned qa = qst[0], qb = qst[64];
        qst += 128;
        one_column(top0, __builtin_amdgcn_perm(qb, qa, 0x0c040c00u));
        one_column(top1, __builtin_amdgcn_perm(qb, qa, 0x0c050c01u));
        one_column(top2, __builtin_amdgcn_perm(qb, qa, 0x0c060c02u));
        one_column(top3, __builtin_amdgcn_perm(qb, qa, 0x0c070c03u));
    }
    if (j <= ql) { // the last one to three columns
        const unsigned qa = qst[0], qb = qst[64];
        unsigned sel = 0x0c040c00u;
        for (; j <= ql; ++j) {
            const uint2 top = bp[0];
            one_column(top, __builtin_amdgcn_perm(qb, qa, sel));
            sel += 0x00010001u;
        }
    }
    // ---- last column of the strip's rows (sw.cpp:100-104: >= so the later row wins); rows carry different
    // offsets, so compare scores, not stored values
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = i0 + r + 1;
        if (row <= tl) {
            const int unshift = -(row + ql) * gext - base;
            const int ca = lo16(h[r]) + unshift, cb = hi16(h[r]) + unshift;
            if (ca >= bestA) {
                bestA = ca;
                bestA_i = row;
            }
            if (cb >= bestB) {
                bestB = cb;
                bestB_i = row;
            }
        }
    }
}

template <int R, int MODE>
__device__ __forceinline__ void sw_dp16_lane_body(const DpArgs &a, const TbArgs &walk)
{
    constexpr bool NOTB = (MODE & 2) != 0;
    static_assert(R % 16 == 0, "a column's traceback goes out as whole uint4 per lane");
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_ls = (a.count + 1) >> 1; // lane slots (two pairs each) of the chunk
    if (gw * 64 >= n_ls) return;
    const int64_t ls = gw * 64 + lane;
    const bool lvalid = ls < n_ls;
    const int64_t slotA = lvalid ? 2 * ls : a.count - 1;
    const bool validB = lvalid && (2 * ls + 1 < a.count);
    const int64_t slotB = validB ? 2 * ls + 1 : slotA;

    const int tl = a.uni_tl, ql = a.uni_ql; // one geometry per batch (host-checked)
    const int match = a.match, gopen = a.gopen, gext = a.gext;
    const bool indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;
    const int base = dp16_base(tl, ql, match, gext);
    LaneConsts c;
    c.delta = pack2(a.mismatch - match, a.mismatch - match);
    c.one = pack2(1, 1);
    c.o_e = pack2(gopen - gext, gopen - gext);
    c.k2 = pack2(match + 2 * gext, match + 2 * gext);
    asm volatile("" : "+v"(c.delta), "+v"(c.one), "+v"(c.o_e), "+v"(c.k2));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        c.k12[u] = 0x02020202u << (2 * u);
        c.k34[u] = 0x01010101u << (2 * u);
        asm volatile("" : "+s"(c.k12[u]), "+s"(c.k34[u]));
    }

    const int strips = lane_strips(tl, R);
    // The wave's scratch: the carry row [column 0 .. ql][lane] x {H, E} packed A|B, then both queries and both targets of
    // every lane transposed to [4-base block][A | B][lane] dwords.  A lane reading its own sequences touches 64 cache lines
    // per wave instruction (one per lane): it does that ONCE, here, and every strip then reads coalesced 256-byte rows.
    unsigned char *const wave_scratch = a.scratch + (size_t)gw * (size_t)lane_scratch_bytes(tl, ql, R);
    uint2 *const bnd = reinterpret_cast<uint2 *>(wave_scratch) + lane;
    unsigned *const qst = reinterpret_cast<unsigned *>(wave_scratch + (size_t)lane_bnd_entries(ql) * 8) + lane;
    unsigned *const tst = qst + (size_t)((ql + 3) >> 2) * 128;
    {
        const int64_t pA = a.first + slotA, pB = a.first + slotB;
        SeqWords sa, sb;
        sa.init(a.q.data + a.q.off[pA], ql);
        sb.init(a.q.data + a.q.off[pB], ql);
        unsigned loA = sa.word(0), loB = sb.word(0);
        for (int cb = 0; cb < (ql + 3) >> 2; ++cb) {
            qst[(size_t)(2 * cb) * 64] = sa.next_block(cb, loA);
            qst[(size_t)(2 * cb + 1) * 64] = sb.next_block(cb, loB);
        }
        sa.init(a.t.data + a.t.off[pA], tl);
        sb.init(a.t.data + a.t.off[pB], tl);
        loA = sa.word(0);
        loB = sb.word(0);
        for (int cb = 0; cb < strips * (R / 4); ++cb) { // (blocks past the end repeat the last dword: rows > tl, never read)
            tst[(size_t)(2 * cb) * 64] = sa.next_block(cb, loA);
            tst[(size_t)(2 * cb + 1) * 64] = sb.next_block(cb, loB);
        }
        // row 0 (the border row, sw.cpp:14-18,31-35) in stored form: H[0][j], E[1][j] = H[0][j] - o
        for (int j = 0; j <= ql; ++j) {
            const int hb0 = border(j, gopen, gext, indel) + j * gext + base;
            const unsigned hp = pack2(hb0, hb0);
            bnd[(size_t)j * 64] = make_uint2(hp, pk_sub(hp, c.o_e));
        }
    }
    // the wave's traceback: [strip][column][R/16][lane] uint4
    uint4 *tbp = reinterpret_cast<uint4 *>(a.tb + (size_t)gw * (size_t)a.tb_stride_words) + lane;

    int bestA = NEG_INF, bestA_i = -1, bestB = NEG_INF, bestB_i = -1; // last-column maxima (true scores)
    for (int k = 0; k < strips - 1; ++k)
        lane_strip<R, NOTB, false>(k * R, tl, ql, bnd, qst, tst, tbp, c, gopen, gext, base, indel, bestA, bestA_i, bestB, bestB_i);
    lane_strip<R, NOTB, true>((strips - 1) * R, tl, ql, bnd, qst, tst, tbp, c, gopen, gext, base, indel, bestA, bestA_i, bestB, bestB_i);

    // ---- last row (sw.cpp:116-127), order-free form: best score, among those the smallest |tl - j|, among those the
    // smallest j; the row wins over the last column on a greater score or an equal one closer to the diagonal
    int rmA = NEG_INF, rdA = 0x7fffffff, rjA = 0x7fffffff, rmB = NEG_INF, rdB = 0x7fffffff, rjB = 0x7fffffff;
    for (int j = 1; j <= ql; ++j) {
        const unsigned bot = bnd[(size_t)j * 64].x;
        const int unshift = -(tl + j) * gext - base;
        const int d = abs(tl - j);
        const int sa = lo16(bot) + unshift, sb = hi16(bot) + unshift;
        const bool ta_ = sa > rmA || (sa == rmA && d < rdA);
        rmA = ta_ ? sa : rmA;
        rdA = ta_ ? d : rdA;
        rjA = ta_ ? j : rjA;
        const bool tb_ = sb > rmB || (sb == rmB && d < rdB);
        rmB = tb_ ? sb : rmB;
        rdB = tb_ ? d : rdB;
        rjB = tb_ ? j : rjB;
    }

#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int mqe = half ? bestB : bestA, mqe_t = half ? bestB_i : bestA_i;
        const int rm = half ? rmB : rmA, rd = half ? rdB : rdA, rj = half ? rjB : rjA;
        const bool row_wins = rm > mqe || (rm == mqe && rd < abs(mqe_t - ql));
        const bool ok = half ? validB : lvalid;
        if (ok) {
            DpRecord r;
            r.mqe = mqe;
            r.mqe_t = mqe_t;
            r.max = row_wins ? rm : mqe;
            r.max_t = row_wins ? tl : mqe_t;
            r.max_q = row_wins ? rj : ql;
            r.seg = row_wins ? ql - rj : 0;
            r.g_tail = 0;
            r.sps = R;
            a.rec[half ? slotB : slotA] = r;
            if (!NOTB && walk.cigar) {
                // The lane walks the paths of its own two pairs right here (sw.cpp:149-255): the traceback words are its
                // own stores, and while it chases them -- a chain of dependent loads, no arithmetic to speak of -- the other
                // waves of the SIMD keep the VALU busy with their fills.  (A separate traceback kernel cannot share a
                // SIMD with this one: three waves of 168 registers leave it no room, so it only ran on drained CUs.)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TbView view;
                view.base = a.tb + (size_t)gw * (size_t)a.tb_stride_words;
                view.set_schedule(r, ql, R);
                view.packed16 = 2;
                view.half = half;
                view.lane = lane;
                view.ql = ql;
                traceback_one_pair(walk, view, r, a.first + (half ? slotB : slotA), tl, ql);
            }
        }
    }
}

} // namespace

// Register budgets: R rows per strip cost 3R VGPRs of state + ~70: four waves per SIMD (128) at 16 rows, three (168) at 32.
#define MGL_LANE_KERNEL(NAME, R, MODE, WPS, NVGPR)                                                            \
    __global__ __launch_bounds__(64, WPS) void NAME(const DpArgs a, const TbArgs walk)                        \
    {                                                                                                         \
        sw_dp16_lane_body<R, MODE>(a, walk);                                                                  \
    }
MGL_LANE_KERNEL(sw_dp16_lane_kernel_r16, 16, 0, 4, 120)
MGL_LANE_KERNEL(sw_dp16_lane_score_kernel_r16, 16, 2, 4, 120)
MGL_LANE_KERNEL(sw_dp16_lane_kernel, 32, 0, 3, 160)          // rows = 32: the default
MGL_LANE_KERNEL(sw_dp16_lane_score_kernel, 32, 2, 3, 160)
#undef MGL_LANE_KERNEL

// can this batch run on the lane kernel at all (beyond the 16-bit range guard, which is dp16_range_ok)?
bool lane16_supported(const SeqSet &t, const SeqSet &q)
{
    return !t.packed2 && !q.packed2; // ASCII wire format (bytes are read as aligned dwords)
}

hipError_t launch_dp16_lane(const DpArgs &a, const TbArgs &walk, int rows, hipStream_t stream)
{
    const int waves_per_block = 1; // (the waves share nothing: a workgroup of one releases its registers as soon as it is through -- sw_dp16_lane_ck.hip)
    const int64_t waves = ((a.count + 1) / 2 + 63) / 64;
    const dim3 grid((unsigned)((waves + waves_per_block - 1) / waves_per_block)), block(64 * waves_per_block);
#define MGL_LAUNCH_LANE(K, KS)                                                                         \
    do {                                                                                               \
        if (a.score_only)                                                                              \
            hipLaunchKernelGGL(KS, grid, block, 0, stream, a, walk);                                         \
        else                                                                                           \
            hipLaunchKernelGGL(K, grid, block, 0, stream, a, walk);                                          \
    } while (0)
    if (rows == 16)
        MGL_LAUNCH_LANE(sw_dp16_lane_kernel_r16, sw_dp16_lane_score_kernel_r16);
    else
        MGL_LAUNCH_LANE(sw_dp16_lane_kernel, sw_dp16_lane_score_kernel);
#undef MGL_LAUNCH_LANE
    return hipGetLastError();
}

} // namespace mgl_sw_dev
