// sw_dp16_lane_ck.hip -- the two-pairs-per-lane kernel (sw_dp16_lane.hip) in its CHECKPOINTED form: the fill keeps no
// traceback flags at all.  Of the 17 VALU instructions the fill spends per two cells, 8 build the four flags of
// sw_lane_cell.h -- for 38 400 cells of a 256 x 150 pair of which the path visits ~300.  Here
//
//   pass 1  runs the score-only column code (9 instructions per two cells) over the whole matrix and keeps what is needed
//           to START AGAIN anywhere on a coarse grid: the carry row entering every strip of 32 target rows (H and E per
//           column -- the reference's score[] / step[], sw_avx.cpp:36-47,196-197; the fill writes these anyway, here they
//           are kept instead of overwritten) and, every LANE_CK_COLS query columns, the lane's register state (H of the
//           previous column and the horizontal-gap value of its 32 rows);
//   pass 2  walks the path (sw.cpp:149-255) block by block: the lane recomputes the 16 x LANE_CK_COLS block its walk stands
//           in -- the SAME column code with the flags switched on, started from the two checkpoints of the block, so every
//           flag is bit for bit the one the full fill would have stored -- into a small private buffer, walks as far as the
//           block reaches, and goes on with the block the path leaves into.  Both pairs of the lane do this in lock step
//           (one packed recomputation serves pair A's block and pair B's block, which are different blocks in general).
//           Blocks are 16 rows high (half a strip: pass 1 also keeps the row in the middle of every strip) -- a path of a
//           256 x 150 pair crosses ~20 of them, 5 000 of 38 400 cells -- and a 16-row column needs 48 registers of state
//           where a 32-row one needs 96 (with 32-row blocks the compiler spilled the target bases into the column loop).
//
// Same arithmetic, same range guard (dp16_range_ok), same results as sw_dp16_lane_kernel.
//
// Per-wave region (a.tb + wave * a.tb_stride_words, lane_ck_words()):
//   rows   [band 0 .. 2 strips][column 0 .. ql][lane] uint2 {H, E} packed A|B: row s = what enters the 16-row band s (row 0: the
//          border; even rows: between strips; odd rows: the middle of a strip; row 2 strips: the last target row, read by the
//          last-row scan)
//   ckpt   [strip][block b][16][lane] uint4: h[0..31], f[0..31] at column 16 b
//   block  [column of the block][lane] uint4: the recomputed flags of 16 rows, bytes as in sw_dp16_lane.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sw_device.h"
#include "sw_lane_cell.h"
#include "sw_traceback.h"

namespace mgl_sw_dev {

namespace {

constexpr int R = 32;            // target rows per strip
constexpr int RB = 16;           // rows per recomputed block
constexpr int CK = LANE_CK_COLS; // query columns per block

// ---- pass 1: one strip, score only, keeping the carry row and the column checkpoints
template <bool LAST>
__device__ __forceinline__ void ck_strip(const int i0, const int tl, const int ql, const uint2 *rin, uint2 *rmid, uint2 *rout, uint4 *ckp,
                                         const unsigned *qst, const unsigned *tst, const LaneConsts &c, const int gopen, const int gext,
                                         const int base, const bool indel, int &bestA, int &bestA_i, int &bestB, int &bestB_i)
{
    unsigned h[R], f[R], t[R];
#pragma unroll
    for (int r4 = 0; r4 < R / 4; ++r4) {
        const unsigned ta = tst[(size_t)(2 * ((i0 >> 2) + r4)) * 64], tb = tst[(size_t)(2 * ((i0 >> 2) + r4) + 1) * 64];
        t[4 * r4 + 0] = __builtin_amdgcn_perm(tb, ta, 0x0c040c00u);
        t[4 * r4 + 1] = __builtin_amdgcn_perm(tb, ta, 0x0c050c01u);
        t[4 * r4 + 2] = __builtin_amdgcn_perm(tb, ta, 0x0c060c02u);
        t[4 * r4 + 3] = __builtin_amdgcn_perm(tb, ta, 0x0c070c03u);
    }
    // column 0 (sw.cpp:24,38,47-49), as in sw_dp16_lane.hip
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = i0 + r + 1;
        const int hb = border(row, gopen, gext, indel) + row * gext + base;
        h[r] = pack2(hb, hb);
        f[r] = pk_sub(h[r], c.o_e);
    }
    const int hd0 = border(i0, gopen, gext, indel) + i0 * gext + base; // H[i0][0]
    unsigned hd = pack2(hd0, hd0);
    rmid[0] = make_uint2(h[RB - 1], 0u); // H[i0 + 16][0], H[i0 + 32][0]: where the block recomputations below start
    rout[0] = make_uint2(h[R - 1], 0u);
    const int rl = tl - 1 - i0;

    const uint2 *ip = rin + 64;
    uint2 *op = rout + 64, *mp = rmid + 64;
    auto one_column = [&](const uint2 top, const unsigned q) {
        unsigned e = top.y;
        column<R, true, true>(h, f, t, q, hd, e, c, nullptr, mp);
        mp += 64;
        hd = top.x;
        if (!LAST) {
            op[0] = make_uint2(h[R - 1], e);
        } else {
            unsigned bot = h[R - 1];
            if (rl != R - 1) {
#pragma unroll
                for (int r = 0; r < R - 1; ++r) bot = (r == rl) ? h[r] : bot;
            }
            op[0] = make_uint2(bot, 0u);
        }
        op += 64;
    };
    auto save = [&]() { // the state BEFORE column j: H[.][j-1] and the horizontal-gap values entering column j
#pragma unroll
        for (int r4 = 0; r4 < R / 4; ++r4) {
            ckp[(size_t)r4 * 64] = make_uint4(h[4 * r4], h[4 * r4 + 1], h[4 * r4 + 2], h[4 * r4 + 3]);
            ckp[(size_t)(R / 4 + r4) * 64] = make_uint4(f[4 * r4], f[4 * r4 + 1], f[4 * r4 + 2], f[4 * r4 + 3]);
        }
        ckp += (R / 2) * 64;
    };
    int j = 1;
    for (; j + 3 <= ql; j += 4) {
        if (((j - 1) & (CK - 1)) == 0) save();
        const uint2 top0 = ip[0], top1 = ip[64], top2 = ip[128], top3 = ip[192];
        ip += 256;
        const unsigned qa = qst[0], qb = qst[64];
        qst += 128;
        one_column(top0, __builtin_amdgcn_perm(qb, qa, 0x0c040c00u));
        one_column(top1, __builtin_amdgcn_perm(qb, qa, 0x0c050c01u));
        one_column(top2, __builtin_amdgcn_perm(qb, qa, 0x0c060c02u));
        one_column(top3, __builtin_amdgcn_perm(qb, qa, 0x0c070c03u));
    }
    if (j <= ql) { // the last one to three columns (j - 1 is a multiple of four here)
        if (((j - 1) & (CK - 1)) == 0) save();
        const unsigned qa = qst[0], qb = qst[64];
        unsigned sel = 0x0c040c00u;
        for (; j <= ql; ++j) {
            const uint2 top = ip[0];
            ip += 64;
            one_column(top, __builtin_amdgcn_perm(qb, qa, sel));
            sel += 0x00010001u;
        }
    }
    // last column (sw.cpp:100-104), as in sw_dp16_lane.hip
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

// ---- pass 2: the walk of one pair (calculateCigar, sw.cpp:149-255) as a machine that can stop at a block's edge and go on in
// the next block.  (pi, pj) is the cell whose flags it needs next: the current cell, or the cell a gap run is counting through
// (TbView::vrun / hrun taken apart).
struct PathWalk {
    int I, J, pi, pj, n, seg;
    int mode;   // 0: at a cell; 1: counting a vertical run (sw.cpp:73-82); 2: a horizontal one (sw.cpp:84-93)
    char state;
    bool done;
    CigarWriter cw;

    __device__ __forceinline__ void take(char next, int step)
    {
        if (next == state) {
            seg += step;
        } else {
            cw.push_front(state, seg);
            seg = step;
            state = next;
        }
    }
    __device__ __forceinline__ void start(const TbArgs &a, const DpRecord &r, int64_t o, int tl, int ql, bool ok)
    {
        cw.slot = a.cigar + (size_t)o * a.cigar_stride;
        cw.binary = a.binary_cigar;
        cw.cap = a.binary_cigar ? (a.cigar_stride & ~3) : a.cigar_stride;
        cw.pos = cw.cap;
        cw.need = 0;
        seg = 0;
        if (a.strategy == OS_INDEL) { // sw.cpp:155-170
            I = tl;
            J = ql;
        } else if (a.strategy != OS_LEAD_ID) {
            I = r.max_t;
            J = r.max_q;
            seg = r.seg;
        } else {
            I = r.mqe_t;
            J = ql;
        }
        done = !ok;
        if (ok && seg > 0 && a.strategy == OS_SOFTCLIP) { // sw.cpp:173-176
            cw.push_front('S', seg);
            seg = 0;
        }
        state = 'M';
        mode = 0;
        n = 0;
        pi = I;
        pj = J;
    }
    // One ROUND of the walk inside block (k, b) = rows 16 k + 1 .. 16 k + 16, columns CK b + 1 .. CK b + CK: the flags of the next
    // LOOK cells along the walk's current direction -- the diagonal (mode 0), up the column (a vertical run), left along the row (a
    // horizontal one) -- are fetched at once (load()), then taken as far as they allow (apply()).  The addresses do not depend on
    // what the cells hold, so a round costs one memory latency however long the run; the caller issues the loads of both pairs of
    // the lane before it applies either.  Flag bits of row t of a dword (sw_lane_cell.h): byte h = {E > max(diag, F): bit 2t+1,
    // F opened: 2t}, byte 2 + h = {F > diag: 2t+1, E opened: 2t}.
    static constexpr int LOOK = 16;
    __device__ __forceinline__ bool can_load(int k, int b) const
    {
        return !done && pi >= 1 && pj >= 1 && ((pi - 1) >> 4) == k && (pj - 1) / CK == b;
    }
    __device__ __forceinline__ void load(const uint32_t *blk, unsigned (&w)[LOOK]) const
    {
        const int rr = (pi - 1) & (RB - 1), cl = (pj - 1) & (CK - 1);
        const int di = mode != 2, dj = mode != 1;
#pragma unroll
        for (int u = 0; u < LOOK; ++u) { // (cells past the block's edge: some valid address, masked in apply())
            const int r = max(rr - u * di, 0), cc = max(cl - u * dj, 0);
            w[u] = blk[(size_t)cc * 256 + (r >> 2)];
        }
    }
    // returns false when the walk cannot go on inside this block (finished, or the cell it needs next is in another block)
    __device__ __forceinline__ bool apply(const unsigned (&w)[LOOK], int half, int k, int b)
    {
        if (done) return false;
        if (mode == 1 && pi < 1) { // the run reached the matrix's top (TbView::vrun: r >= 1)
            take('D', n);
            I -= n;
        } else if (mode == 2 && pj < 1) {
            take('I', n);
            J -= n;
        } else {
            if (((pi - 1) >> 4) != k || (pj - 1) / CK != b) return false;
            const int rr = (pi - 1) & (RB - 1), cl = (pj - 1) & (CK - 1);
            const int di = mode != 2;
            const int room = mode == 0 ? min(rr, cl) : mode == 1 ? rr : cl; // cells beyond the first that lie inside the block
            // what stops the run: a cell that is not a diagonal move / opens the vertical gap / opens the horizontal gap
            const unsigned stop = (mode == 0 ? 0x00020002u : mode == 1 ? 0x00010000u : 0x00000001u) << (8 * half);
            int cnt = 0;
            unsigned at_stop = 0; // mode 0: is the cell that ended the diagonal run a vertical move (E > max(diag, F))?
#pragma unroll
            for (int u = 0; u < LOOK; ++u) {
                const int t2 = ((rr - u * di) & 3) * 2;
                if (cnt == u && u <= room) {
                    if ((w[u] & (stop << t2)) == 0u)
                        cnt = u + 1;
                    else
                        at_stop = w[u] & ((0x2u << (8 * half)) << t2);
                }
            }
            if (mode == 0) {
                if (cnt > 0) {
                    take('M', cnt);
                    I -= cnt;
                    J -= cnt;
                }
                if (cnt <= room) { // the run ended at a gap cell of this block (BitsMoves::at: the vertical move first)
                    const bool vertical = at_stop != 0u;
                    mode = vertical ? 1 : 2;
                    n = 1;
                    pi = vertical ? I - 1 : I;
                    pj = vertical ? J : J - 1;
                    return true; // (an edge of the matrix or of the block right behind it: the next round sees that)
                }
            } else if (mode == 1) {
                n += cnt;
                pi -= cnt;
                if (cnt > room && pi >= 1) return false; // every cell up to the block's edge extends the gap: goes on in the block above
                take('D', n);
                I -= n;
            } else {
                n += cnt;
                pj -= cnt;
                if (cnt > room && pj >= 1) return false;
                take('I', n);
                J -= n;
            }
        }
        mode = 0;
        pi = I;
        pj = J;
        done = !(I > 0 && J > 0); // sw.cpp:214
        return !done && ((pi - 1) >> 4) == k && (pj - 1) / CK == b;
    }
    // overhangs, text, per-pair results (walk_and_write's tail + traceback_one_pair)
    __device__ __forceinline__ void finish(const TbArgs &a, const DpRecord &r, int64_t o)
    {
        int off;
        if (a.strategy == OS_SOFTCLIP) { // sw.cpp:225-229
            cw.push_front(state, seg);
            if (J > 0) cw.push_front('S', J);
            off = I;
        } else if (a.strategy == OS_IGNORE) { // sw.cpp:230-233
            cw.push_front(state, seg + J);
            off = I - J;
        } else { // sw.cpp:234-248
            cw.push_front(state, seg);
            if (I > 0)
                cw.push_front('D', I);
            else if (J > 0)
                cw.push_front('I', J);
            off = 0;
        }
        const int status = finish_cigar(cw);
        for (int x = cw.cap; x < a.cigar_stride; ++x) cw.slot[x] = 0;
        a.offset[o] = off;
        if (a.cigar_len) a.cigar_len[o] = cw.need;
        if (a.status) a.status[o] = status;
        if (a.status_any && status != 0) atomicMax(a.status_any, status);
        if (a.score) {
            Score sc;
            sc.mqe = r.mqe;
            sc.mqe_t = r.mqe_t;
            sc.max = r.max;
            sc.max_t = r.max_t;
            sc.max_q = r.max_q;
            sc.seg_length = r.seg;
            a.score[o] = sc;
        }
    }
};

// low half of a, high half of b
__device__ __forceinline__ unsigned mix(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060100u); }

// ---- pass 2: recompute the flags of block (sA, bA) for the low halves and of block (sB, bB) for the high halves; s = 16-row band
// (strip s / 2, its upper or lower half), b = block of CK columns
__device__ __forceinline__ void ck_block(const int sA, const int bA, const int sB, const int bB, const int ql, const int nb, const uint2 *rows,
                                         const uint4 *ck, const unsigned *qst, const unsigned *tst, uint4 *blk, const LaneConsts &c)
{
    unsigned h[RB], f[RB], t[RB];
    const size_t row_len = (size_t)(ql + 1) * 64;
#pragma unroll
    for (int r4 = 0; r4 < RB / 4; ++r4) {
        const unsigned ta = tst[(size_t)(2 * (sA * (RB / 4) + r4)) * 64], tb = tst[(size_t)(2 * (sB * (RB / 4) + r4) + 1) * 64];
        t[4 * r4 + 0] = __builtin_amdgcn_perm(tb, ta, 0x0c040c00u);
        t[4 * r4 + 1] = __builtin_amdgcn_perm(tb, ta, 0x0c050c01u);
        t[4 * r4 + 2] = __builtin_amdgcn_perm(tb, ta, 0x0c060c02u);
        t[4 * r4 + 3] = __builtin_amdgcn_perm(tb, ta, 0x0c070c03u);
    }
    {
        // the strip's checkpoint at column CK b: quads 0 .. 7 = h[0..31], 8 .. 15 = f[0..31]; this band's rows are quads (s & 1) * 4 ..
        const uint4 *ca = ck + (((size_t)(sA >> 1) * nb + bA) * (R / 2) + (sA & 1) * (RB / 4)) * 64;
        const uint4 *cb = ck + (((size_t)(sB >> 1) * nb + bB) * (R / 2) + (sB & 1) * (RB / 4)) * 64;
#pragma unroll
        for (int r4 = 0; r4 < RB / 4; ++r4) {
            const uint4 ha = ca[(size_t)r4 * 64], hb = cb[(size_t)r4 * 64], fa = ca[(size_t)(R / 4 + r4) * 64], fb = cb[(size_t)(R / 4 + r4) * 64];
            h[4 * r4 + 0] = mix(ha.x, hb.x);
            h[4 * r4 + 1] = mix(ha.y, hb.y);
            h[4 * r4 + 2] = mix(ha.z, hb.z);
            h[4 * r4 + 3] = mix(ha.w, hb.w);
            f[4 * r4 + 0] = mix(fa.x, fb.x);
            f[4 * r4 + 1] = mix(fa.y, fb.y);
            f[4 * r4 + 2] = mix(fa.z, fb.z);
            f[4 * r4 + 3] = mix(fa.w, fb.w);
        }
    }
    const uint2 *ra = rows + (size_t)sA * row_len, *rb = rows + (size_t)sB * row_len;
    const int cA = bA * CK, cB = bB * CK;
    unsigned hd = mix(ra[(size_t)cA * 64].x, rb[(size_t)cB * 64].x);
    const int qmax = ((ql + 3) >> 2) - 1;
#pragma unroll 1
    for (int g = 0; g < CK / 4; ++g) {
        uint2 ta[4], tb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { // columns past ql: the last column again (their flags are never read)
            ta[u] = ra[(size_t)min(cA + 4 * g + u + 1, ql) * 64];
            tb[u] = rb[(size_t)min(cB + 4 * g + u + 1, ql) * 64];
        }
        const unsigned qa = qst[(size_t)(2 * min((cA >> 2) + g, qmax)) * 64], qb = qst[(size_t)(2 * min((cB >> 2) + g, qmax) + 1) * 64];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned e = mix(ta[u].y, tb[u].y);
            const unsigned q = __builtin_amdgcn_perm(qb, qa, 0x0c040c00u + 0x00010001u * u);
            column<RB, false>(h, f, t, q, hd, e, c, blk + (size_t)(4 * g + u) * 64);
            hd = mix(ta[u].x, tb[u].x);
        }
    }
}

__device__ __forceinline__ void sw_dp16_lane_ck_body(const DpArgs &a, const TbArgs &walk)
{
    const int lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int64_t n_ls = (a.count + 1) >> 1;
    if (gw * 64 >= n_ls) return;
    const int64_t ls = gw * 64 + lane;
    const bool lvalid = ls < n_ls;
    const int64_t slotA = lvalid ? 2 * ls : a.count - 1;
    const bool validB = lvalid && (2 * ls + 1 < a.count);
    const int64_t slotB = validB ? 2 * ls + 1 : slotA;

    const int tl = a.uni_tl, ql = a.uni_ql;
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

    const int strips = lane_strips(tl, R), nb = lane_ck_blocks(ql);
    // the wave's scratch, as in sw_dp16_lane.hip (its one carry row is not used here): both queries and both targets of every
    // lane transposed to [4-base block][A | B][lane] dwords
    unsigned char *const wave_scratch = a.scratch + (size_t)gw * (size_t)lane_scratch_bytes(tl, ql, R);
    unsigned *const qst = reinterpret_cast<unsigned *>(wave_scratch + (size_t)lane_bnd_entries(ql) * 8) + lane;
    unsigned *const tst = qst + (size_t)((ql + 3) >> 2) * 128;
    // the wave's region: rows, checkpoints, block buffer
    uint32_t *const region = a.tb + (size_t)gw * (size_t)a.tb_stride_words;
    const size_t row_len = (size_t)(ql + 1) * 64;
    uint2 *const rows = reinterpret_cast<uint2 *>(region) + lane;
    uint4 *const ck = reinterpret_cast<uint4 *>(region + (size_t)(2 * strips + 1) * row_len * 2) + lane;
    uint4 *const blk = reinterpret_cast<uint4 *>(region + (size_t)(2 * strips + 1) * row_len * 2 + (size_t)64 * strips * nb * (R * 2)) + lane;
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
        for (int cb = 0; cb < strips * (R / 4); ++cb) {
            tst[(size_t)(2 * cb) * 64] = sa.next_block(cb, loA);
            tst[(size_t)(2 * cb + 1) * 64] = sb.next_block(cb, loB);
        }
        // row 0 (the border row, sw.cpp:14-18,31-35) in stored form: H[0][j], E[1][j] = H[0][j] - o
        for (int j = 0; j <= ql; ++j) {
            const int hb0 = border(j, gopen, gext, indel) + j * gext + base;
            const unsigned hp = pack2(hb0, hb0);
            rows[(size_t)j * 64] = make_uint2(hp, pk_sub(hp, c.o_e));
        }
    }

    // ---- pass 1
    int bestA = NEG_INF, bestA_i = -1, bestB = NEG_INF, bestB_i = -1;
    for (int k = 0; k < strips - 1; ++k)
        ck_strip<false>(k * R, tl, ql, rows + (size_t)(2 * k) * row_len, rows + (size_t)(2 * k + 1) * row_len, rows + (size_t)(2 * k + 2) * row_len,
                        ck + (size_t)k * nb * (R / 2) * 64, qst, tst, c, gopen, gext, base, indel, bestA, bestA_i, bestB, bestB_i);
    ck_strip<true>((strips - 1) * R, tl, ql, rows + (size_t)(2 * strips - 2) * row_len, rows + (size_t)(2 * strips - 1) * row_len,
                   rows + (size_t)(2 * strips) * row_len, ck + (size_t)(strips - 1) * nb * (R / 2) * 64, qst, tst, c, gopen, gext, base, indel, bestA,
                   bestA_i, bestB, bestB_i);

    // ---- last row (sw.cpp:116-127), as in sw_dp16_lane.hip
    int rmA = NEG_INF, rdA = 0x7fffffff, rjA = 0x7fffffff, rmB = NEG_INF, rdB = 0x7fffffff, rjB = 0x7fffffff;
    {
        const uint2 *last = rows + (size_t)(2 * strips) * row_len;
        for (int j = 1; j <= ql; ++j) {
            const unsigned bot = last[(size_t)j * 64].x;
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
    }
    DpRecord rec[2];
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int mqe = half ? bestB : bestA, mqe_t = half ? bestB_i : bestA_i;
        const int rm = half ? rmB : rmA, rd = half ? rdB : rdA, rj = half ? rjB : rjA;
        const bool row_wins = rm > mqe || (rm == mqe && rd < abs(mqe_t - ql));
        DpRecord &r = rec[half];
        r.mqe = mqe;
        r.mqe_t = mqe_t;
        r.max = row_wins ? rm : mqe;
        r.max_t = row_wins ? tl : mqe_t;
        r.max_q = row_wins ? rj : ql;
        r.seg = row_wins ? ql - rj : 0;
        r.g_tail = 0;
        r.sps = R;
        if (half ? validB : lvalid) a.rec[half ? slotB : slotA] = r;
    }

    if (a.sps_cap & 1) return; // TIMING EXPERIMENT
    // ---- pass 2
    const int64_t pA = a.first + slotA, pB = a.first + slotB;
    const int64_t oA = walk.dest ? walk.dest[pA] : pA, oB = walk.dest ? walk.dest[pB] : pB;
    PathWalk wa, wb;
    wa.start(walk, rec[0], oA, tl, ql, lvalid);
    wb.start(walk, rec[1], oB, tl, ql, validB);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the lane's own rows and checkpoints are in memory
    const uint32_t *const blk_words = reinterpret_cast<const uint32_t *>(blk);
    int iters = 0, na = 0, nb_ = 0, rounds = 0;
    while (__builtin_amdgcn_ballot_w64(!wa.done || !wb.done) != 0) {
        // (a finished walk keeps recomputing some valid block: both halves run the same instructions anyway)
        int kA = max(wa.pi - 1, 0) >> 4, bA = max(wa.pj - 1, 0) / CK, kB = max(wb.pi - 1, 0) >> 4, bB = max(wb.pj - 1, 0) / CK;
        if (a.sps_cap & 64) { // EXPERIMENT: scattered blocks without the walk
            const unsigned hsh = (unsigned)(lane * 2654435761u + iters * 40503u) >> 8;
            kA = hsh % (2 * strips); bA = (hsh >> 5) % nb; kB = (hsh >> 9) % (2 * strips); bB = (hsh >> 13) % nb;
        }
        ck_block(kA, bA, kB, bB, ql, nb, rows, ck, qst, tst, blk, c);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (a.sps_cap & 2) { // TIMING EXPERIMENT: no walk, a fixed number of blocks
            if (++iters >= 20) wa.done = wb.done = true;
            continue;
        }
        if (!wa.done) ++na;
        if (!wb.done) ++nb_;
        for (;;) {
            const bool la = wa.can_load(kA, bA), lb = wb.can_load(kB, bB);
            // (a lane with nothing to fetch still runs apply(): a run that ends at the matrix's edge needs no flags)
            unsigned fa[PathWalk::LOOK], fb[PathWalk::LOOK];
            if (la) wa.load(blk_words, fa);
            if (lb) wb.load(blk_words, fb);
            const bool ga = wa.apply(fa, 0, kA, bA), gb = wb.apply(fb, 1, kB, bB);
            ++rounds;
            if (!__builtin_amdgcn_ballot_w64(ga || gb)) break;
        }
        ++iters;
    }
    if (lvalid) wa.finish(walk, rec[0], oA);
    if (validB) wb.finish(walk, rec[1], oB);
    if ((a.sps_cap & 4) && walk.status) { // EXPERIMENT: blocks the wave computed / this pair needed
        if (lvalid) walk.status[oA] = iters | (rounds << 16);
        if (validB) walk.status[oB] = iters | (rounds << 16);
    }
}

} // namespace

__global__ __launch_bounds__(256, 3) void sw_dp16_lane_ck_kernel(const DpArgs a, const TbArgs walk) { sw_dp16_lane_ck_body(a, walk); }

hipError_t launch_dp16_lane_ck(const DpArgs &a, const TbArgs &walk, hipStream_t stream)
{
    const int waves_per_block = 4;
    const int64_t waves = ((a.count + 1) / 2 + 63) / 64;
    const dim3 grid((unsigned)((waves + waves_per_block - 1) / waves_per_block)), block(64 * waves_per_block);
    static const int dbg = [] { const char *e = getenv("MGL_SW_CK_DEBUG"); return e ? atoi(e) : 0; }();
    DpArgs b = a;
    b.sps_cap = dbg;
    hipLaunchKernelGGL(sw_dp16_lane_ck_kernel, grid, block, 0, stream, b, walk);
    return hipGetLastError();
}

} // namespace mgl_sw_dev
