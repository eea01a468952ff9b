// sw_traceback.h -- the path walk shared by the traceback kernels (sw_kernels.hip) and the fill kernel that walks its
// own pairs (sw_dp16_lane.hip): traceback-bit accessor for the three layouts, CIGAR writer, calculateCigar
// (/root/reference/src/main/native/mgl_sw/sw.cpp:149-255) restated.  Device code only.
#ifndef MGL_SW_TRACEBACK_H
#define MGL_SW_TRACEBACK_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_device.h"

#ifndef MGL_SW_WALK_AHEAD
#define MGL_SW_WALK_AHEAD 8
#endif

namespace mgl_sw_dev {

// traceback bits accessor shared by the path walk and the matrix expansion
struct TbView {
    const uint32_t *base; // this pair's (int32 layout) or this group's (packed16 layout) traceback words
    int sps;
    int rows;                 // target rows per stripe (= lanes per pair) of the fill kernel: 16 or 64; lane layout: rows per strip
    int packed16, half;       // packed16: 0 int32 layout, 1 sw_dp16_kernel, 2 sw_dp16_lane_kernel (base = the WAVE's words),
                              // 3 sw_dp_coop16_kernel (per pair; a pair the kernel redid in 32 bits has layout 0: layout_of)
    int lane = 0, ql = 0;     // lane layout: this pair's lane in its wave, the batch's query length
    int waves = 0;            // strip layout (4): waves per pair
    int g_tail, nc, sps_tail; // packed16 only: stripes >= nc are stand-alone, starting at global step g_tail
    __device__ __forceinline__ void set_schedule(const DpRecord &r, int ql, int rows_per_stripe)
    {
        rows = rows_per_stripe;
        sps = r.sps;
        g_tail = r.g_tail;
        nc = r.g_tail > 0 ? (r.g_tail - 16) / r.sps : 0;
        sps_tail = sps_for_rows(ql, 16);
    }
    // the layout of ONE pair of a launch whose layout is `launch_layout`: sw_dp_coop16_kernel marks the pairs it kept in 16 bits
    __device__ __forceinline__ static int layout_of(int launch_layout, const DpRecord &r) { return launch_layout == 3 && r.g_tail != -16 ? 0 : launch_layout; }
    __device__ __forceinline__ static int waves_of(const DpRecord &r) { return r.g_tail <= -100 ? -100 - r.g_tail : 0; } // layout 4
    // nibble of cell (i, j), 1-based: bit0 F>diag, bit1 E>max(diag,F), bit2 E opened, bit3 F opened
    __device__ __forceinline__ unsigned cell(int i, int j) const
    {
        const int r = i - 1;
        if (packed16 == 4) {
            // sw_dp16_strip.hip: [wave][step][column of the group][16-row group][lane] uint4; strip g = row / rows sits in the low
            // (g < NL) or high half of lane g mod NL and works on column group cg at step cg + g; NL = 64 * waves
            const int g = r / rows, rr = r - g * rows, nl = 64 * waves; // rows per strip: 20 .. 32
            const int h = g >= nl, ell = g - h * nl;
            const size_t u4 = (((((size_t)(ell >> 6) * sps + (j - 1) / STRIP_CPS + g) * STRIP_CPS + (j - 1) % STRIP_CPS) * 2) + (rr >> 4)) * 64 + (ell & 63);
            const uint32_t w = base[u4 * 4 + ((rr >> 2) & 3)];
            const int t2 = (rr & 3) * 2;
            const unsigned be = (w >> (8 * h)) >> t2, bf = (w >> (16 + 8 * h)) >> t2;
            return ((bf >> 1) & 1u) | (((be >> 1) & 1u) << 1) | ((bf & 1u) << 2) | ((be & 1u) << 3);
        }
        if (packed16 == 3) {
            // sw_dp_coop.hip, 16-bit form: [128-row double stripe][16 steps][lane] uint4; PE = row within the double stripe,
            // lane = PE / 2, half = PE mod 2, step = column + PE; dword = 4 steps, bytes as in the packed16 layout
            const int pe = r & 127, s = j + pe, h = pe & 1;
            const uint32_t w = base[(((size_t)(r >> 7) * (sps >> 4) + (s >> 4)) * 64 + (pe >> 1)) * 4 + ((s >> 2) & 3)];
            const int t2 = (s & 3) * 2;
            const unsigned be = (w >> (8 * h)) >> t2, bf = (w >> (16 + 8 * h)) >> t2;
            return ((bf >> 1) & 1u) | (((be >> 1) & 1u) << 1) | ((bf & 1u) << 2) | ((be & 1u) << 3);
        }
        if (packed16 == 2) {
            // sw_dp16_lane.hip: [strip][column][rows/16][lane] uint4, dword = four rows, bytes as in the packed16 layout
            const int sh = rows == 32 ? 5 : 4; // rows is 16 or 32
            const int strip = r >> sh, rr = r & (rows - 1);
            const size_t u4 = ((size_t)(strip * ql + (j - 1)) * (rows >> 4) + (rr >> 4)) * 64 + lane;
            const uint32_t w = base[u4 * 4 + ((rr >> 2) & 3)];
            const int t2 = (rr & 3) * 2;
            const unsigned be = (w >> (8 * half)) >> t2, bf = (w >> (16 + 8 * half)) >> t2;
            return ((bf >> 1) & 1u) | (((be >> 1) & 1u) << 1) | ((bf & 1u) << 2) | ((be & 1u) << 3);
        }
        const int sh_rows = rows == 64 ? 6 : 4; // rows is 16 or 64: no integer division in the walk
        const int lane = r & (rows - 1);
        const int k = r >> sh_rows;
        const int g = (g_tail > 0 && k >= nc ? g_tail + (k - nc) * sps_tail : k * sps) + j + lane;
        if (packed16) {
            // sw_dp16.hip: dword per lane per 4 steps (two per 8-step block);
            // byte h = {E>S, F opened}, byte 2+h = {F>diag, E opened}
            const uint32_t w = base[(size_t)(g >> 3) * 32 + lane * 2 + ((g >> 2) & 1)];
            const int t2 = (g & 3) * 2;
            const unsigned be = (w >> (8 * half)) >> t2, bf = (w >> (16 + 8 * half)) >> t2;
            return ((bf >> 1) & 1u) | (((be >> 1) & 1u) << 1) | ((bf & 1u) << 2) | ((be & 1u) << 3);
        }
        const uint4 w = *reinterpret_cast<const uint4 *>(base + ((size_t)(g >> 5) * rows + lane) * 4);
        const int sh = 31 - (g & 31);
        return ((w.x >> sh) & 1u) | (((w.y >> sh) & 1u) << 1) | (((w.z >> sh) & 1u) << 2) |
               (((w.w >> sh) & 1u) << 3);
    }
    // run length of a vertical gap entered at (i, j): 1 + consecutive extensions above (sw.cpp:73-82)
    __device__ __forceinline__ int vrun(int i, int j) const
    {
        int n = 1;
        for (int r = i - 1; r >= 1 && !(cell(r, j) & 4u); --r) ++n;
        return n;
    }
    // run length of a horizontal gap entered at (i, j) (sw.cpp:84-93)
    __device__ __forceinline__ int hrun(int i, int j) const
    {
        int n = 1;
        for (int c = j - 1; c >= 1 && !(cell(i, c) & 8u); --c) ++n;
        return n;
    }
};

struct CigarWriter {
    char *slot;
    int cap, pos, need;
    int binary; // 1: BAM-style uint32 elements (len << 4 | op, op M=0 I=1 D=2 S=4) instead of text
    bool store = true; // wave-per-pair walk: every lane tracks pos / need, one lane stores
    // elements arrive last-first (the reference push_front()s, sw.cpp:172-248); text is built
    // right-aligned and moved to the front at the end.  Zero lengths are skipped (sw.cpp:252).
    __device__ __forceinline__ void push_front(char op, int len)
    {
        if (len <= 0) return;
        if (binary) {
            need += 4;
            if (pos - 4 < 0) {
                pos = -1;
                return;
            }
            const unsigned code = op == 'M' ? 0u : op == 'I' ? 1u : op == 'D' ? 2u : 4u;
            const unsigned v = ((unsigned)len << 4) | code;
            pos -= 4;
            if (store) {
                slot[pos] = (char)(v & 0xff);
                slot[pos + 1] = (char)((v >> 8) & 0xff);
                slot[pos + 2] = (char)((v >> 16) & 0xff);
                slot[pos + 3] = (char)(v >> 24);
            }
            return;
        }
        int digits = 1;
        for (int v = len; v >= 10; v /= 10) ++digits;
        need += digits + 1;
        if (pos - (digits + 1) < 0) {
            pos = -1;
            return;
        }
        --pos;
        if (store) slot[pos] = op;
        for (int v = len, d = 0; d < digits; ++d, v /= 10) {
            --pos;
            if (store) slot[pos] = (char)('0' + v % 10);
        }
    }
};

// Sources of one traceback move: the 4-bit device cells, or a reference-style int32 matrix.
struct BitsMoves {
    TbView tb;
    // returns +k (k rows up), -k (k columns left) or 0 (diagonal): the value the reference stores
    __device__ __forceinline__ int at(int i, int j) const
    {
        const unsigned c = tb.cell(i, j);
        if (c & 2u) return tb.vrun(i, j);
        if (c & 1u) return -tb.hrun(i, j);
        return 0;
    }
    // Lane layout only: the next cells of a diagonal run, (i - k, j - k), sit in other columns' blocks -- one cache miss per
    // path step if they are fetched when the walk gets there.  Their addresses do not depend on what the cells hold, so
    // eight are fetched at once (one miss latency for up to eight steps) and the run is taken as far as they are all
    // diagonal moves.  Other layouts: one lane per pair has nothing to look ahead with.
    __device__ __forceinline__ int diag_run(int i, int j) const
    {
        if (tb.packed16 != 2 && tb.packed16 != 4) return 0;
        constexpr int AHEAD = MGL_SW_WALK_AHEAD;
        unsigned c[AHEAD];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) c[k] = (i - k >= 1 && j - k >= 1) ? tb.cell(i - k, j - k) : 3u; // off the matrix: stop
        int n = 0;
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            if (n == k && (c[k] & 3u) == 0u) n = k + 1; // neither F > diag nor E > max(diag, F): a diagonal move (sw.cpp:60-71)
        }
        return n;
    }
};
struct MatrixMoves {
    const int32_t *btr;
    int m; // ql + 1
    __device__ __forceinline__ int at(int i, int j) const { return btr[(size_t)i * m + j]; }
    __device__ __forceinline__ int diag_run(int, int) const { return 0; }
};

// calculateCigar (sw.cpp:149-255): walk from the strategy's start cell, merge equal states,
// post-process the overhangs, emit text.  Returns the alignment offset.
// Moves::diag_run(i, j): optional shortcut, the number of consecutive diagonal moves from (i, j) that can be taken at once
// (0 = unknown, take the ordinary single step)
template <typename Moves>
__device__ __forceinline__ int walk_and_write(Moves &mv, int tl, int ql, int strategy, int max_t, int max_q,
                                              int mqe_t, int seg_length, CigarWriter &cw)
{
    // start cell, sw.cpp:155-170
    int I, J, seg = 0;
    if (strategy == OS_INDEL) {
        I = tl;
        J = ql;
    } else if (strategy != OS_LEAD_ID) {
        I = max_t;
        J = max_q;
        seg = seg_length;
    } else {
        I = mqe_t;
        J = ql;
    }
    if (seg > 0 && strategy == OS_SOFTCLIP) { // sw.cpp:173-176
        cw.push_front('S', seg);
        seg = 0;
    }
    char state = 'M';
    do { // sw.cpp:182-214
        const int run = mv.diag_run(I, J);
        if (run > 0) { // `run` times the third branch below
            if (state == 'M') {
                seg += run;
            } else {
                cw.push_front(state, seg);
                seg = run;
                state = 'M';
            }
            I -= run;
            J -= run;
            continue;
        }
        const int b = mv.at(I, J);
        char next;
        int step = 1;
        if (b > 0) {
            next = 'D';
            step = b;
            I -= step;
        } else if (b < 0) {
            next = 'I';
            step = -b;
            J -= step;
        } else {
            next = 'M';
            --I;
            --J;
        }
        if (next == state) {
            seg += step;
        } else {
            cw.push_front(state, seg);
            seg = step;
            state = next;
        }
    } while (I > 0 && J > 0);

    int off;
    if (strategy == OS_SOFTCLIP) { // sw.cpp:225-229
        cw.push_front(state, seg);
        if (J > 0) cw.push_front('S', J);
        off = I;
    } else if (strategy == OS_IGNORE) { // sw.cpp:230-233
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
    return off;
}

// move the right-aligned text to the front of the slot and zero the rest; returns the status
__device__ __forceinline__ int finish_cigar(CigarWriter &cw)
{
    if (cw.pos < 0) {
        for (int k = 0; k < cw.cap; ++k) cw.slot[k] = 0;
        return ERR_CIGAR_OVERFLOW;
    }
    const int len = cw.cap - cw.pos;
    if (cw.pos > 0)
        for (int k = 0; k < len; ++k) cw.slot[k] = cw.slot[cw.pos + k];
    for (int k = len; k < cw.cap; ++k) cw.slot[k] = 0;
    return 0;
}


// One pair, one lane: walk the path of pair `p` (slot `slot` of its chunk) and write every per-pair result the way the
// reference's calculateCigar + alignNative do (offset, CIGAR text or BAM elements zero padded to the slot, ScoreMax).
__device__ __forceinline__ void traceback_one_pair(const TbArgs &a, const TbView &view, const DpRecord &r, const int64_t p,
                                                   const int tl, const int ql)
{
    BitsMoves mv;
    mv.tb = view;
    CigarWriter cw;
    const int64_t o = a.dest ? a.dest[p] : p; // where this pair's results go
    cw.slot = a.cigar + (size_t)o * a.cigar_stride;
    cw.binary = a.binary_cigar;
    cw.cap = a.binary_cigar ? (a.cigar_stride & ~3) : a.cigar_stride;
    cw.pos = cw.cap;
    cw.need = 0;

    const int off = walk_and_write(mv, tl, ql, a.strategy, r.max_t, r.max_q, r.mqe_t, r.seg, cw);
    const int status = finish_cigar(cw);
    for (int k = cw.cap; k < a.cigar_stride; ++k) cw.slot[k] = 0;

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

} // namespace mgl_sw_dev
#endif
