// sw_dp16_lane_ck.hip -- the two-pairs-per-lane kernel (sw_dp16_lane.hip) in its CHECKPOINTED form: the fill keeps no traceback
// flags at all.  Of the 17 VALU instructions the full fill spends per two cells, 8 build the four flags of sw_lane_cell.h -- for
// 38 400 cells of a 256 x 150 pair of which the path visits ~300, nearly all of them in a way that needs no flag to be known.
//
//   pass 1  runs the score-only column code (8 instructions per two cells with base codes, sw_lane_cell.h) over the whole matrix
//           and keeps what is needed to START AGAIN anywhere on a coarse grid: {H, E} of every 16th target row per column -- the
//           carry row entering every 32-row strip (the reference's score[] / step[], sw_avx.cpp:36-47,196-197; kept for all strips
//           instead of being updated in place) and the row in a strip's middle -- and, every LANE_CK_COLS query columns, the lane's
//           register state (H of the previous column and the horizontal-gap value of its 32 rows).  All of it time-major,
//           [..][lane], stored straight from the registers in whole 256- / 512-byte rows: a lane only ever reads back its own.
//   pass 2  walks the path (sw.cpp:149-255).  The walk knows H of the cell it stands at, so a diagonal stretch up to the next
//           kept row is CHECKED instead of recomputed: if H drops by exactly the sum of the stretch's substitution scores, every
//           cell of it took the diagonal (PathWalk::verify_apply: the proof).  Only where that fails -- a gap, or a tie taken
//           elsewhere: 0.4 blocks per pair on Illumina-style reads, 3.5 rounds per wave -- the lane recomputes the 16 x LANE_CK_COLS
//           block the walk stands in: the SAME column code with the flags switched on, started from the block's kept row and
//           checkpoint, so every flag is bit for bit the one the full fill would have stored, into a small private buffer; it walks
//           as far as the block reaches and goes on.  Both pairs of a lane do this in lock step (one packed recomputation serves
//           pair A's block and pair B's, different blocks in general).
//
// Same arithmetic, same range guard (dp16_range_ok), same results as sw_dp16_lane_kernel.
//
// Per-wave region (a.tb + wave * a.tb_stride_words, lane_ck_words()): WaveMem below.  The wave's scratch (a.scratch,
// lane_ck_scratch_bytes()) holds both sequences of every lane, staged once as [4-base block][A | B][lane] dwords.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "sw_device.h"
#include "sw_lane_cell.h"
#include "sw_traceback.h"

// Timing-only experiment builds (scripts/build_variant.sh; results are WRONG with bits 1, 2, 4 or 16): parts of the kernel switched off
#ifndef MGL_CK_ABLATE
#define MGL_CK_ABLATE 0
#endif
#if (MGL_CK_ABLATE != 0 || defined(MGL_CK_PHASES) || defined(MGL_CK_TRACE)) && !defined(MGL_VARIANT_BUILD)
#error "MGL_CK_ABLATE / MGL_CK_PHASES are measurement builds (scripts/build_variant.sh defines MGL_VARIANT_BUILD): never the shipped library"
#endif

// -DMGL_CK_PHASES (scripts/build_variant.sh): every wave adds the shader-clock ticks it spent in each part of the kernel to a
// device array; mgl_ck_phases_dump() (exported by that build only) prints and clears it
#ifdef MGL_CK_PHASES
__device__ unsigned long long mgl_ck_phase_ticks[16];
#define CK_PHASE(k)                                                                                    \
    do {                                                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                  \
        if (lane == 0) atomicAdd(&mgl_ck_phase_ticks[k], now_ - phase_t0);                             \
        phase_t0 = __builtin_amdgcn_s_memtime();                                                       \
    } while (0)
#else
#define CK_PHASE(k) do { } while (0)
#endif

// -DMGL_CK_TRACE (scripts/build_variant.sh): every tile leaves {slot, hardware id, start, end (100 MHz ticks)}; mgl_ck_trace_dump(path)
// (exported by that build only) writes the array out -- scripts/ck_trace.py draws the launch's timeline from it
#ifdef MGL_CK_TRACE
__device__ unsigned long long mgl_ck_trace[4 << 17];
#endif

namespace mgl_sw_dev {

namespace {

constexpr int R = 32;            // target rows per strip
constexpr int RB = 16;           // rows per recomputed block
constexpr int CK = LANE_CK_COLS; // query columns per block

// the halves of two packed registers: {a.low, b.high}
__device__ __forceinline__ unsigned lo_hi(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060100u); }

// what pass 2 needs to know about the wave's geometry and scoring
struct BlockGeom {
    int ql, nb, match, mismatch, gopen, gext, base;
    bool indel;
    bool codes; // the staged sequences hold base codes
};

// the wave's arrays (every pointer includes the lane): what pass 1 keeps, time-major -- a lane only ever reads what it wrote itself
struct WaveMem {
    uint2 *bnd;    // [row 32 k, k = 0 .. strips][column 0 .. ql][lane] {H, E} packed A | B: the carry row entering strip k (row 0: the border)
    uint2 *mid;    // [strip][column 1 .. ql][lane]: the row in the strip's middle (row 32 k + 16)
    unsigned *ck;  // [strip][block 1 .. nb - 1][2 r | 2 r + 1][lane]: H[.][32 b] and the horizontal-gap value of the strip's row r entering column 32 b + 1
    uint4 *blk;    // [column of the block][lane]: the recomputed flags of 16 rows
    // the same arrays as pass 2 addresses them: wave-uniform bases (SGPRs) and 32-bit indices that include the lane -- one address
    // register per load instead of two, no 64-bit arithmetic in the walk
    const uint2 *rows;     // the region: bnd at 0, mid at mid_off (entries)
    const unsigned *cku;   // ck without the lane
    const uint32_t *blku;  // blk without the lane, as dwords
    const unsigned *seq;   // the wave's staged sequences: the queries' blocks at 0, the targets' at t_off (dwords)
    unsigned mid_off, t_off, lane;
    // (the BYTE offset is what is computed in 32 bits: an element index leaves its scaling to 64-bit arithmetic and the load to a
    // 64-bit address in two registers -- pass 2 holds dozens of these at once)
    template <class T>
    __device__ __forceinline__ static T at32(const void *base, unsigned byte_off) { return *reinterpret_cast<const T *>(static_cast<const char *>(base) + byte_off); }
    __device__ __forceinline__ uint2 row16(int m, int ql, int j) const // {H, E} of row 16 m (m >= 0) at column j
    {
        return at32<uint2>(rows, 8u * ((m & 1) ? mid_off + (unsigned)(((m >> 1) * ql + j - 1) * 64) + lane : (unsigned)(((m >> 1) * (ql + 1) + j) * 64) + lane));
    }
    __device__ __forceinline__ unsigned qblock(int blk4, int half) const { return at32<unsigned>(seq, 4u * ((unsigned)((2 * blk4 + half) * 64) + lane)); }
    __device__ __forceinline__ unsigned tblock(int blk4, int half) const { return at32<unsigned>(seq, 4u * (t_off + (unsigned)((2 * blk4 + half) * 64) + lane)); }
    __device__ __forceinline__ unsigned flags(int cc, int r4) const { return at32<unsigned>(blku, 4u * ((unsigned)(cc * 256 + r4) + 4u * lane)); } // the block's flags: column cc, rows 4 r4 ..
};

// ---- pass 2: the walk of one pair (calculateCigar, sw.cpp:149-255) as a machine that can stop at a block's edge and go on in
// the next block.  (pi, pj) is the cell whose flags it needs next: the current cell, or the cell a gap run is counting through
// (TbView::vrun / hrun taken apart).
struct PathWalk {
    int I, J, pi, pj, n, seg;
    int hc;     // mode 0: H[I][J], the score of the cell the walk stands at
    int mode;   // 0: at a cell; 1: counting a vertical run (sw.cpp:73-82); 2: a horizontal one (sw.cpp:84-93)
    char state;
    bool done;
    bool stuck; // the stretch above (I, J) does not add up: the walk needs this block's flags
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
    // h_corner: H[tl][ql], where the walk of SW_OS_INDEL starts
    __device__ __forceinline__ void start(const TbArgs &a, const DpRecord &r, int64_t o, int tl, int ql, bool ok, int h_corner)
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
            hc = h_corner;
        } else if (a.strategy != OS_LEAD_ID) {
            I = r.max_t;
            J = r.max_q;
            seg = r.seg;
            hc = r.max;
        } else {
            I = r.mqe_t;
            J = ql;
            hc = r.mqe;
        }
        done = !ok;
        stuck = false;
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
    __device__ __forceinline__ void load(const WaveMem &wm, unsigned (&w)[LOOK]) const
    {
        const int rr = (pi - 1) & (RB - 1), cl = (pj - 1) & (CK - 1);
        const int di = mode != 2, dj = mode != 1;
#pragma unroll
        for (int u = 0; u < LOOK; ++u) { // (cells past the block's edge: some valid address, masked in apply())
            const int r = max(rr - u * di, 0), cc = max(cl - u * dj, 0);
            w[u] = wm.flags(cc, r >> 2);
        }
    }
    // returns false when the walk cannot go on inside this block (finished, or the cell it needs next is in another block)
    // (tw, qw: win_load() of the cell the walk stands at -- mode 0 only: the scores of the diagonal steps it takes)
    __device__ __forceinline__ bool apply(const unsigned (&w)[LOOK], const unsigned (&tw)[5], const unsigned (&qw)[5], int half, int k, int b,
                                          const BlockGeom &g)
    {
        char op = 0; // the CIGAR element this round completes, if any: taken in ONE place (take() is a large piece of code)
        int len = 0;
        const bool more = moves(w, tw, qw, half, k, b, g, op, len);
        if (len > 0) take(op, len);
        return more;
    }
    __device__ __forceinline__ bool moves(const unsigned (&w)[LOOK], const unsigned (&tw)[5], const unsigned (&qw)[5], int half, int k, int b,
                                          const BlockGeom &g, char &op, int &len)
    {
        if (done) return false;
        if (mode == 1 && pi < 1) { // the run reached the matrix's top (TbView::vrun: r >= 1)
            op = 'D';
            len = n;
            I -= n;
        } else if (mode == 2 && pj < 1) {
            op = 'I';
            len = n;
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
                    hc -= cnt * g.match + mismatches(tw, qw, cnt, g.codes) * (g.mismatch - g.match);
                    op = 'M';
                    len = cnt;
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
                if (cnt == LOOK && cnt <= room) return true; // more of the run in this block than one round looks at
                if (cnt > room && pi >= 1) return false; // every cell up to the block's edge extends the gap: goes on in the block above
                hc += g.gopen + (n - 1) * g.gext; // H[I][J] = E = H[I - n][J] - o - (n - 1) e (sw.cpp:73-82)
                op = 'D';
                len = n;
                I -= n;
            } else {
                n += cnt;
                pj -= cnt;
                if (cnt == LOOK && cnt <= room) return true;
                if (cnt > room && pj >= 1) return false;
                hc += g.gopen + (n - 1) * g.gext; // sw.cpp:84-93
                op = 'I';
                len = n;
                J -= n;
            }
        }
        mode = 0;
        pi = I;
        pj = J;
        done = !(I > 0 && J > 0); // sw.cpp:214
        return !done && ((pi - 1) >> 4) == k && (pj - 1) / CK == b;
    }
    // ---- cells that need no flags: a diagonal stretch whose scores add up.  Along any diagonal stretch c0, c1 .. cL the diagonal
    // candidate of sw.cpp:51-55 gives H[ck] >= H[ck-1] + s(ck) (s = match or mismatch by the two bases); if H[cL] - H[c0] EQUALS
    // the sum of the L substitution scores, every one of these inequalities is tight, so in every cell of the stretch the diagonal
    // candidate is the maximum, sw.cpp:60-62 records btr = 0 (the diagonal wins ties), and the walk of sw.cpp:182-214 takes L
    // diagonal steps -- whatever the stretch contains (substitutions included) and whatever the penalties are.  The walk knows H of
    // the cell it stands at (hc: the start cell's score is in the record, a diagonal step subtracts s, a gap run of n adds
    // o + (n-1) e: sw.cpp:73-93), pass 1 kept H of every 16th row (WaveMem) and the borders are formulas, so one round
    // checks the stretch from the walk's cell up to the next recorded row (or the matrix's edge) with one look at that row's H and
    // at most 16 bases of both sequences (their staged copies: [4-base block][A | B][lane] dwords).  Only where the sum does NOT
    // fit -- a gap, or a tie taken elsewhere -- does the walk need the flags of the block.
    __device__ __forceinline__ bool can_verify() const { return !done && mode == 0 && !stuck; }
    // bases I - 16 .. I - 1 (0-based) of the target and J - 16 .. J - 1 of the query: byte 15 of the window belongs to cell (I, J)
    __device__ __forceinline__ void win_load(const WaveMem &wm, int half, int tblocks, int qblocks, unsigned (&tw)[5], unsigned (&qw)[5]) const
    {
        const int dt = (I - 16) >> 2, dq = (J - 16) >> 2;
#pragma unroll
        for (int k = 0; k < 5; ++k) { // (blocks before the sequence: block 0 again -- those bytes are never counted)
            tw[k] = wm.tblock(min(max(dt + k, 0), tblocks - 1), half);
            qw[k] = wm.qblock(min(max(dq + k, 0), qblocks - 1), half);
        }
    }
    // how many of the cells (I, J), (I-1, J-1) .. (I-L+1, J-L+1) hold different bases (raw byte compare, sw.cpp:55); L <= 16, I, J
    // (codes: the staged target holds codes 0 .. 3, the staged query 8 x code or 32 -- sw_lane_cell.h)
    __device__ __forceinline__ int mismatches(const unsigned (&tw)[5], const unsigned (&qw)[5], int L, bool codes) const
    {
        const unsigned st = (unsigned)(I - 16) & 3u, sq = (unsigned)(J - 16) & 3u;
        const int drop = 16 - L, dq = drop >> 2, sh = (drop & 3) * 8; // the window's low `drop` bytes are not part of the stretch
        int n = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const unsigned tt = __builtin_amdgcn_alignbyte(tw[k + 1], tw[k], st);
            const unsigned nz = nonzero_bytes((codes ? tt << 3 : tt) ^ __builtin_amdgcn_alignbyte(qw[k + 1], qw[k], sq));
            n += __popc(k > dq ? nz : k == dq ? nz >> sh : 0u);
        }
        return n;
    }
    // One round checks up to VB stretches at once -- from the walk's cell to each of the next VB kept rows: 64 bases of both
    // sequences and the VB rows' H are fetched together, their addresses depend on nothing the round computes -- and takes the
    // longest one that adds up (if the stretch to a row adds up, so does the stretch to every row before it).
    static constexpr int VB = 4;
    __device__ __forceinline__ void win64_load(const WaveMem &wm, int half, int tblocks, int qblocks, unsigned (&tw)[17], unsigned (&qw)[17]) const
    {
        const int dt = (I - 64) >> 2, dq = (J - 64) >> 2;
#pragma unroll
        for (int k = 0; k < 17; ++k) {
            tw[k] = wm.tblock(min(max(dt + k, 0), tblocks - 1), half);
            qw[k] = wm.qblock(min(max(dq + k, 0), qblocks - 1), half);
        }
    }
    // the stretch to the m-th kept row above the walk's cell (m = 0: the next one), cut where the diagonal leaves the matrix
    __device__ __forceinline__ int stretch_to(int m) const { return min(I - max((((I - 1) >> 4) - m) << 4, 0), J); }
    // H of those rows in stored form (both pairs of the lane)
    __device__ __forceinline__ void grid_load(const WaveMem &wm, int ql, unsigned (&w)[VB]) const
    {
#pragma unroll
        for (int m = 0; m < VB; ++m) {
            const int L = stretch_to(m), ie = I - L, je = J - L;
            w[m] = wm.row16(ie >> 4, ql, max(je, 1)).x; // (a stretch that ends on a border: some valid entry, not used)
        }
    }
    // returns true when all VB stretches were taken and the walk can try the next ones
    __device__ __forceinline__ bool verify_apply(const unsigned (&tw)[17], const unsigned (&qw)[17], const unsigned (&w)[VB], int half, const BlockGeom &g)
    {
        if (!can_verify()) return false;
        // bit u of `differ`: the bases of cell (I - u, J - u) differ (byte 63 - u of the window)
        const unsigned st = (unsigned)(I - 64) & 3u, sq = (unsigned)(J - 64) & 3u;
        unsigned long long differ = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const unsigned tt = __builtin_amdgcn_alignbyte(tw[k + 1], tw[k], st);
            const unsigned nz = nonzero_bytes((g.codes ? tt << 3 : tt) ^ __builtin_amdgcn_alignbyte(qw[k + 1], qw[k], sq)) >> 7;
            const unsigned nib = ((nz * 0x08040201u) >> 24) & 0xfu; // byte 3 -> bit 0 .. byte 0 -> bit 3
            differ |= (unsigned long long)nib << (4 * (15 - k));
        }
        int took = 0, h_at = hc;
        bool all = true;
#pragma unroll
        for (int m = 0; m < VB; ++m) {
            const int L = stretch_to(m), ie = I - L, je = J - L;
            const int he = ie == 0 ? border(je, g.gopen, g.gext, g.indel)
                         : je == 0 ? border(ie, g.gopen, g.gext, g.indel)
                                   : (half ? hi16(w[m]) : lo16(w[m])) - (ie + je) * g.gext - g.base;
            const unsigned long long in = L >= 64 ? ~0ull : (1ull << L) - 1ull;
            const bool ok = hc - he == L * g.match + __popcll(differ & in) * (g.mismatch - g.match);
            took = ok ? L : took; // (the stretches that add up are the first ones: each contains the one before)
            h_at = ok ? he : h_at;
            all = all && ok;
        }
        stuck = !all; // the first stretch that does not add up holds a gap (or a tie taken elsewhere): this block's flags, then
        if (took > 0) {
            take('M', took);
            I -= took;
            J -= took;
            hc = h_at;
            pi = I;
            pj = J;
            done = !(I > 0 && J > 0); // sw.cpp:214
        }
        return all && !done;
    }
    // overhangs, text, per-pair results (walk_and_write's tail + traceback_one_pair)
    // (rp: the pair's record where pass 1 left it in memory -- read back here, by the lane that wrote it, rather than carried in a dozen
    // registers through the whole of pass 2)
    __device__ __forceinline__ void finish(const TbArgs &a, const DpRecord *rp, int64_t o)
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
            const DpRecord r = *rp;
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

// ---- pass 1: strip k, score only, keeping the rows and the column checkpoints
template <bool LAST, bool CODES>
__device__ __forceinline__ void ck_strip(const int k, const int tl, const int ql, const int nb, const WaveMem &wm, const unsigned *qst, const unsigned *tst,
                                         const LaneConsts &c, const int gopen, const int gext, const int base, const bool indel, int &bestA,
                                         int &bestA_i, int &bestB, int &bestB_i)
{
    const int i0 = k * R;
    unsigned h[R], f[R], t[R];
#pragma unroll
    for (int r4 = 0; r4 < R / 4; ++r4) {
        const unsigned ta = tst[(size_t)(2 * ((i0 >> 2) + r4)) * 64], tb = tst[(size_t)(2 * ((i0 >> 2) + r4) + 1) * 64];
        t[4 * r4 + 0] = __builtin_amdgcn_perm(tb, ta, 0x0c040c00u) | (CODES ? CODE_SEL : 0u);
        t[4 * r4 + 1] = __builtin_amdgcn_perm(tb, ta, 0x0c050c01u) | (CODES ? CODE_SEL : 0u);
        t[4 * r4 + 2] = __builtin_amdgcn_perm(tb, ta, 0x0c060c02u) | (CODES ? CODE_SEL : 0u);
        t[4 * r4 + 3] = __builtin_amdgcn_perm(tb, ta, 0x0c070c03u) | (CODES ? CODE_SEL : 0u);
    }
    // column 0 (sw.cpp:24,38,47-49), as in sw_dp16_lane.hip
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = i0 + r + 1;
        const int hb = border(row, gopen, gext, indel) + row * gext + base;
        h[r] = pack2(hb, hb);
        f[r] = pk_sub(h[r], c.o_e);
    }
    const int rl = tl - 1 - i0;

    const uint2 *bp = wm.bnd + ((size_t)k * (ql + 1) + 1) * 64; // column j of the row entering this strip
    uint2 *bo = wm.bnd + ((size_t)(k + 1) * (ql + 1) + 1) * 64; // ... of the row leaving it
    uint2 *mp = wm.mid + (size_t)k * ql * 64;
    unsigned *ckp = wm.ck + (size_t)k * (nb - 1) * 64 * 64;
    unsigned hd = bp[-64].x; // H[32 k][0]
    // column u (0 .. 3) of the group whose query dwords are qa / qb
    auto one_column = [&](const uint2 top, const unsigned qa, const unsigned qb, const int u) {
        unsigned e = top.y;
        uint2 mid;
        if (CODES)
            column<R, true, true, true>(h, f, t, code_table((qa >> (8 * u)) & 0xffu), hd, e, c, nullptr, &mid, code_table((qb >> (8 * u)) & 0xffu));
        else
            column<R, true, true, false>(h, f, t, __builtin_amdgcn_perm(qb, qa, 0x0c040c00u + 0x00010001u * u), hd, e, c, nullptr, &mid);
        hd = top.x;
        if (!(MGL_CK_ABLATE & 4)) mp[0] = mid;
        mp += 64;
        if (!LAST) {
            bo[0] = make_uint2(h[R - 1], e);
        } else { // the row the last-row scan reads: H[tl][j]
            unsigned bot = h[R - 1];
            if (rl != R - 1) {
#pragma unroll
                for (int r = 0; r < R - 1; ++r) bot = (r == rl) ? h[r] : bot;
            }
            bo[0] = make_uint2(bot, 0u);
        }
        bo += 64;
    };
    auto save = [&]() { // the state BEFORE column j: H[.][j-1] and the horizontal-gap values entering column j
#pragma unroll
        for (int r = 0; r < R; ++r) {
            ckp[(size_t)(2 * r) * 64] = h[r];
            ckp[(size_t)(2 * r + 1) * 64] = f[r];
        }
        ckp += 64 * 64;
    };
    // The carry row and the query are read ONE GROUP OF FOUR COLUMNS AHEAD: what a group needs was requested at the top of the
    // group before it, so its latency -- and the acknowledgements of the stores issued in between, which s_waitcnt vmcnt counts
    // in the same queue -- hides behind a thousand instructions of arithmetic instead of stalling the wave at every group (measured
    // before: a third of the waves' lifetime in s_waitcnt at two waves per SIMD).  Reads past column ql stay inside the wave's own
    // region (the next kept row follows) and are never used.
    int j = 1;
    uint2 n0 = bp[0], n1 = bp[64], n2 = bp[128], n3 = bp[192];
    unsigned nqa = qst[0], nqb = qst[64];
    for (; j + 3 <= ql; j += 4) {
        if (!(MGL_CK_ABLATE & 2) && j > 1 && ((j - 1) & (CK - 1)) == 0) save(); // (column 0 is a formula: no checkpoint)
        const uint2 top0 = n0, top1 = n1, top2 = n2, top3 = n3;
        const unsigned qa = nqa, qb = nqb;
        qst += 128;
        bp += 256;
        n0 = bp[0];
        n1 = bp[64];
        n2 = bp[128];
        n3 = bp[192];
        nqa = qst[0];
        nqb = qst[64];
        one_column(top0, qa, qb, 0);
        one_column(top1, qa, qb, 1);
        one_column(top2, qa, qb, 2);
        one_column(top3, qa, qb, 3);
    }
    if (j <= ql) { // the last one to three columns (j - 1 is a multiple of four here)
        if (j > 1 && ((j - 1) & (CK - 1)) == 0) save();
        one_column(n0, nqa, nqb, 0);
        if (j + 1 <= ql) one_column(n1, nqa, nqb, 1);
        if (j + 2 <= ql) one_column(n2, nqa, nqb, 2);
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

// ---- pass 2: recompute the flags of block (sA, bA) for the low halves and of block (sB, bB) for the high halves; s = 16-row band
// (strip s / 2, its upper or lower half), b = block of CK columns.  needA / needB: the pair's walk waits for this block (a pair
// that does not fetches nothing; its half computes garbage nobody reads).
template <bool CODES>
__device__ __forceinline__ void ck_block(const int sA, const int bA, const int sB, const int bB, const bool needA, const bool needB, const BlockGeom &g,
                                         const WaveMem &wm, const LaneConsts &c, const int groups = CK / 8)
{
    unsigned h[RB], f[RB], t[RB];
#pragma unroll
    for (int r4 = 0; r4 < RB / 4; ++r4) {
        const unsigned ta = wm.tblock(sA * (RB / 4) + r4, 0), tb = wm.tblock(sB * (RB / 4) + r4, 1);
        t[4 * r4 + 0] = __builtin_amdgcn_perm(tb, ta, 0x0c040c00u) | (CODES ? CODE_SEL : 0u);
        t[4 * r4 + 1] = __builtin_amdgcn_perm(tb, ta, 0x0c050c01u) | (CODES ? CODE_SEL : 0u);
        t[4 * r4 + 2] = __builtin_amdgcn_perm(tb, ta, 0x0c060c02u) | (CODES ? CODE_SEL : 0u);
        t[4 * r4 + 3] = __builtin_amdgcn_perm(tb, ta, 0x0c070c03u) | (CODES ? CODE_SEL : 0u);
    }
    // the state entering the block's first column: the checkpoint of column CK b, or column 0 by its formula (sw.cpp:24,38,47-49)
    {
        const unsigned ia = (unsigned)((((sA >> 1) * (g.nb - 1) + max(bA - 1, 0)) * 64 + (sA & 1) * 2 * RB) * 64) + wm.lane;
        const unsigned ib = (unsigned)((((sB >> 1) * (g.nb - 1) + max(bB - 1, 0)) * 64 + (sB & 1) * 2 * RB) * 64) + wm.lane;
        // one half after the other (both at once: 64 registers of checkpoint values beside the 48 of the block's state -- spilled)
        unsigned v[2 * RB];
#pragma unroll
        for (int x = 0; x < 2 * RB; ++x) v[x] = 0u;
        if (needA && bA > 0) {
#pragma unroll
            for (int x = 0; x < 2 * RB; ++x) v[x] = WaveMem::at32<unsigned>(wm.cku, 4u * (ia + (unsigned)(x * 64)));
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int rowA = sA * RB + r + 1;
            const int fa_ = border(rowA, g.gopen, g.gext, g.indel) + rowA * g.gext + g.base;
            h[r] = bA > 0 ? v[2 * r] : (unsigned)fa_;
            f[r] = bA > 0 ? v[2 * r + 1] : (unsigned)(fa_ - (g.gopen - g.gext));
        }
#pragma unroll
        for (int x = 0; x < 2 * RB; ++x) v[x] = 0u;
        if (needB && bB > 0) {
#pragma unroll
            for (int x = 0; x < 2 * RB; ++x) v[x] = WaveMem::at32<unsigned>(wm.cku, 4u * (ib + (unsigned)(x * 64)));
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int rowB = sB * RB + r + 1;
            const int fb_ = border(rowB, g.gopen, g.gext, g.indel) + rowB * g.gext + g.base;
            const unsigned hb = bB > 0 ? v[2 * r] : (unsigned)fb_ << 16, gb = bB > 0 ? v[2 * r + 1] : (unsigned)(fb_ - (g.gopen - g.gext)) << 16;
            h[r] = lo_hi(h[r], hb);
            f[r] = lo_hi(f[r], gb);
        }
    }
    // the row entering the band, per column: row 16 s of the pair (WaveMem::row16; row 0 is the border row, kept like any other)
    unsigned hd;
    {
        unsigned a0 = 0u, b0 = 0u; // H[16 s][CK b]
        if (needA) a0 = wm.row16(sA, g.ql, CK * bA).x;
        if (needB) b0 = wm.row16(sB, g.ql, CK * bB).x;
        if (sA & 1) { // (the middle rows start at column 1: column 0 by its formula)
            const int row = sA * RB;
            const int v = border(row, g.gopen, g.gext, g.indel) + row * g.gext + g.base;
            a0 = bA > 0 ? a0 : (unsigned)v;
        }
        if (sB & 1) {
            const int row = sB * RB;
            const int v = border(row, g.gopen, g.gext, g.indel) + row * g.gext + g.base;
            b0 = bB > 0 ? b0 : (unsigned)v << 16;
        }
        hd = lo_hi(a0, b0);
    }
    const int qmax = ((g.ql + 3) >> 2) - 1;
#pragma unroll 1
    for (int gg = 0; gg < groups; ++gg) { // (groups of eight columns, up to the last one a waiting walk stands in: it only ever moves left and up)
        uint2 va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) va[u] = vb[u] = make_uint2(0u, 0u);
        if (needA) {
#pragma unroll
            for (int u = 0; u < 8; ++u) va[u] = wm.row16(sA, g.ql, min(CK * bA + 8 * gg + u + 1, g.ql)); // (past ql: the last column again -- flags never read)
        }
        if (needB) {
#pragma unroll
            for (int u = 0; u < 8; ++u) vb[u] = wm.row16(sB, g.ql, min(CK * bB + 8 * gg + u + 1, g.ql));
        }
        unsigned qa[2], qb[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            qa[u] = wm.qblock(min(bA * (CK / 4) + 2 * gg + u, qmax), 0);
            qb[u] = wm.qblock(min(bB * (CK / 4) + 2 * gg + u, qmax), 1);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            unsigned e = lo_hi(va[u].y, vb[u].y);
            uint4 *const out = wm.blk + (size_t)(8 * gg + u) * 64;
            if (CODES)
                column<RB, false, false, true>(h, f, t, code_table((qa[u >> 2] >> (8 * (u & 3))) & 0xffu), hd, e, c, out, nullptr,
                                               code_table((qb[u >> 2] >> (8 * (u & 3))) & 0xffu));
            else
                column<RB, false, false, false>(h, f, t, __builtin_amdgcn_perm(qb[u >> 2], qa[u >> 2], 0x0c040c00u + 0x00010001u * (u & 3)), hd, e, c, out);
            hd = lo_hi(va[u].x, vb[u].x);
        }
    }
}

// ---- staging.  Eight 4-base blocks of both pairs per round: all loads of a round in flight, then the stores (one block per round
// exposes a memory latency per block: a tenth of the wave's lifetime, measured).  dst: the wave's [block][A | B][lane] dwords + lane.
// ASCII sequences (sw.cpp:55 compares raw bytes).  CODES: store base codes -- a target as the code per byte, a QUERY as 8 x code, or 32
// for a byte that is not one of ACGT (sw_lane_cell.h) -- and return nonzero if a byte of the first `len` is not one of ACGT.
typedef uint32_t dwords4 __attribute__((ext_vector_type(4), aligned(4))); // four consecutive dwords at a dword-aligned address
// dwords k0 .. k0 + 3 of a sequence: one 16-byte load where all four hold bytes of it, else dword by dword, clamped (SeqWords::word)
__device__ __forceinline__ void seq_words4(const SeqWords &s, const int k0, unsigned (&w)[4])
{
    if (k0 + 3 <= s.kmax) {
        const dwords4 v = *reinterpret_cast<const dwords4 *>(s.base + k0);
        w[0] = v.x;
        w[1] = v.y;
        w[2] = v.z;
        w[3] = v.w;
    } else {
#pragma unroll
        for (int u = 0; u < 4; ++u) w[u] = s.word(k0 + u);
    }
}
template <bool CODES, bool QUERY>
__device__ __forceinline__ unsigned stage_ascii(const uint8_t *seqA, const uint8_t *seqB, const int len, const int nblocks, unsigned *dst)
{
    SeqWords sa, sb;
    sa.init(seqA, len);
    sb.init(seqB, len);
    unsigned bad = 0;
    for (int cb = 0; cb < nblocks; cb += 16) { // sixteen blocks of both pairs per round: 17 dwords of each sequence, in 16-byte loads
        unsigned wa[20], wb[20];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            unsigned xa[4], xb[4];
            seq_words4(sa, cb + 4 * u, xa);
            seq_words4(sb, cb + 4 * u, xb);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                wa[4 * u + x] = xa[x];
                wb[4 * u + x] = xb[x];
            }
        }
        wa[16] = sa.word(cb + 16);
        wb[16] = sb.word(cb + 16);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (cb + u < nblocks) { // (blocks past the end of a target repeat its last dword: rows > tl, never read)
                unsigned va = __builtin_amdgcn_alignbyte(wa[u + 1], wa[u], sa.shift), vb = __builtin_amdgcn_alignbyte(wb[u + 1], wb[u], sb.shift);
                if (CODES) {
                    unsigned da, db;
                    va = ascii_codes(va, da);
                    vb = ascii_codes(vb, db);
                    if (QUERY) {
                        const unsigned na = nonzero_bytes(da), nb = nonzero_bytes(db); // 0x80 in the bytes that are no ACGT: those become 32
                        va = ((va << 3) & ~((na >> 2) - (na >> 7))) | (na >> 2);
                        vb = ((vb << 3) & ~((nb >> 2) - (nb >> 7))) | (nb >> 2);
                    } else {
                        const int nv = len - 4 * (cb + u); // bytes of this block that belong to the sequence
                        const unsigned keep = nv >= 4 ? 0xffffffffu : nv <= 0 ? 0u : (1u << (8 * nv)) - 1u;
                        bad |= (da | db) & keep;
                    }
                }
                dst[(size_t)(2 * (cb + u)) * 64] = va;
                dst[(size_t)(2 * (cb + u) + 1) * 64] = vb;
            }
        }
    }
    return bad;
}

// 2-bit packed sequences (SeqSet::packed2: base k of the array in bits 2 (k & 3) of byte k >> 2; a pair's bases start at BASE index
// `start`, any alignment): read as aligned dwords of 16 bases, clamped to the last dword that holds a base of the sequence.
struct Seq2Words {
    const uint32_t *base;
    unsigned shift; // bits
    int kmax;
    __device__ __forceinline__ void init(const uint8_t *data, int64_t start, int len)
    {
        const uintptr_t a = reinterpret_cast<uintptr_t>(data) + (uintptr_t)(start >> 2), e = reinterpret_cast<uintptr_t>(data) + (uintptr_t)((start + len - 1) >> 2);
        base = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
        shift = (unsigned)(a & 3) * 8u + 2u * (unsigned)(start & 3);
        kmax = (int)(((e & ~(uintptr_t)3) - (a & ~(uintptr_t)3)) >> 2);
    }
    __device__ __forceinline__ unsigned word(int k) const { return base[k < kmax ? k : kmax]; }
};
template <bool QUERY>
__device__ __forceinline__ void stage_2bit(const uint8_t *data, const int64_t startA, const int64_t startB, const int len, const int nblocks, unsigned *dst)
{
    Seq2Words sa, sb;
    sa.init(data, startA, len);
    sb.init(data, startB, len);
    for (int cb = 0; cb < nblocks; cb += 16) { // four dwords of 16 bases = sixteen blocks of both pairs per round
        unsigned wa[5], wb[5];
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            wa[u] = sa.word((cb >> 2) + u);
            wb[u] = sb.word((cb >> 2) + u);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned va = __builtin_amdgcn_alignbit(wa[u + 1], wa[u], sa.shift), vb = __builtin_amdgcn_alignbit(wb[u + 1], wb[u], sb.shift);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                const int blk = cb + 4 * u + x;
                if (blk < nblocks) {
                    const unsigned ca = spread_2bit((va >> (8 * x)) & 0xffu), cb2 = spread_2bit((vb >> (8 * x)) & 0xffu);
                    dst[(size_t)(2 * blk) * 64] = QUERY ? ca << 3 : ca;
                    dst[(size_t)(2 * blk + 1) * 64] = QUERY ? cb2 << 3 : cb2;
                }
            }
        }
    }
}

// One TILE = 128 pairs of the launch (pairs 128 gw .. 128 gw + 127), worked on by the wave that sits in wave slot `slot` of the
// persistent grid: everything the wave keeps while it works -- WaveMem, the staged sequences -- lives in the slot's region, which
// the wave reuses tile after tile (a lane only ever reads back what it has written for the tile it is working on).
// whole lines out of LDS: `bytes` (a multiple of 4) from src (LDS, 16-byte aligned) to dst (global, 16-byte aligned)
__device__ __forceinline__ void lds_to_global(void *dst, const unsigned char *src, const int bytes, const int lane)
{
    const int whole = bytes & ~15;
    for (int k = lane * 16; k < whole; k += 1024) *reinterpret_cast<uint4 *>(static_cast<unsigned char *>(dst) + k) = *reinterpret_cast<const uint4 *>(src + k);
    for (int k = whole + lane * 4; k < bytes; k += 256) *reinterpret_cast<unsigned *>(static_cast<unsigned char *>(dst) + k) = *reinterpret_cast<const unsigned *>(src + k);
}

// VIA_LDS (= walk_in.coalesced_out, a template parameter so that every pointer the walks write through has ONE address space: with the
// choice made at run time the text went out through flat instructions, whose waits also cover the scalar loads -- pass 2 took twice as long)
// (A wave keeps its slot for all its tiles, so every address of its region is invariant in the kernel's tile loop: the compiler computes
// them once in front of it and keeps some forty of them in scratch for the kernel's whole life -- a load each where a tile needs one,
// none inside pass 1's loops.  Making slot and lane opaque per tile recomputes them instead and empties the scratch area by a third,
// but the kernel ran 2.5 ms per 10 M pairs SLOWER (65.2 against 62.6: the opaque lane number hides its range from the address
// arithmetic of every store of pass 1) -- measured, scripts/ck_regs_probe.sh, and left alone.)
template <bool VIA_LDS>
__device__ __forceinline__ void sw_dp16_lane_ck_tile(const DpArgs &a, const TbArgs &walk_in, const int64_t gw, const int64_t slot, const int lane, unsigned char *out_lds)
{
    const int64_t n_ls = (a.count + 1) >> 1;
#ifdef MGL_CK_PHASES
    unsigned long long phase_t0 = __builtin_amdgcn_s_memtime();
#endif
    const int64_t ls = gw * 64 + lane;
    const bool lvalid = ls < n_ls;
    const int64_t slotA = lvalid ? 2 * ls : a.count - 1;
    const bool validB = lvalid && (2 * ls + 1 < a.count);
    const int64_t slotB = validB ? 2 * ls + 1 : slotA;

    // one geometry per launch (a.uni_tl x a.uni_ql), or -- a.grouped: a chunk sorted by geometry, whole waves of 128 pairs --
    // one per wave, read from its first pair; a.uni_tl / a.uni_ql are then the maxima that size the wave's regions
    const bool grouped = a.grouped != 0;
    const int tl = grouped ? __builtin_amdgcn_readfirstlane(a.t.length(a.first + gw * 128)) : a.uni_tl;
    const int ql = grouped ? __builtin_amdgcn_readfirstlane(a.q.length(a.first + gw * 128)) : a.uni_ql;
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
    // the wave's scratch: both queries and both targets of every lane transposed to [4-base block][A | B][lane] dwords
    unsigned *const qst = reinterpret_cast<unsigned *>(a.scratch + (size_t)slot * (size_t)lane_ck_scratch_bytes(a.uni_tl, a.uni_ql)) + lane;
    unsigned *const tst = qst + (size_t)((ql + 3) >> 2) * 128;
    WaveMem wm;
    {
        uint2 *const region = reinterpret_cast<uint2 *>(a.tb + (size_t)slot * (size_t)a.tb_stride_words);
        wm.bnd = region + lane;
        wm.mid = wm.bnd + (size_t)(strips + 1) * (ql + 1) * 64;
        wm.ck = reinterpret_cast<unsigned *>(region + ((size_t)(strips + 1) * (ql + 1) + (size_t)strips * ql) * 64) + lane;
        wm.blk = reinterpret_cast<uint4 *>(reinterpret_cast<unsigned *>(region) + ((size_t)(strips + 1) * (ql + 1) * 2 + (size_t)strips * ql * 2 + (size_t)strips * (nb - 1) * 64) * 64) + lane;
        wm.rows = region;
        wm.mid_off = (unsigned)((strips + 1) * (ql + 1) * 64);
        wm.cku = reinterpret_cast<const unsigned *>(region + ((size_t)(strips + 1) * (ql + 1) + (size_t)strips * ql) * 64);
        wm.blku = reinterpret_cast<const uint32_t *>(region) + ((size_t)(strips + 1) * (ql + 1) * 2 + (size_t)strips * ql * 2 + (size_t)strips * (nb - 1) * 64) * 64;
        wm.seq = reinterpret_cast<const unsigned *>(a.scratch + (size_t)slot * (size_t)lane_ck_scratch_bytes(a.uni_tl, a.uni_ql));
        wm.t_off = (unsigned)(((ql + 3) >> 2) * 128);
        wm.lane = (unsigned)lane;
    }
    // ---- staging: both sequences of every lane, transposed to [4-base block][A | B][lane] dwords in the wave's scratch -- as BASE
    // CODES where the wave's targets allow it (sw_lane_cell.h: 2-bit packed inputs always do; ASCII targets when every byte is one of
    // ACGT -- the queries may hold anything), else as the raw bytes.
    bool codes;
    {
        const int64_t pA = a.first + slotA, pB = a.first + slotB;
        const int qblocks = (ql + 3) >> 2, tblocks = strips * (R / 4);
        if (a.t.packed2) {
            stage_2bit<false>(a.t.data, a.t.off[pA], a.t.off[pB], tl, tblocks, tst);
            stage_2bit<true>(a.q.data, a.q.off[pA], a.q.off[pB], ql, qblocks, qst);
            codes = true;
        } else {
            const unsigned bad = stage_ascii<true, false>(a.t.data + a.t.off[pA], a.t.data + a.t.off[pB], tl, tblocks, tst);
            codes = __builtin_amdgcn_ballot_w64(bad != 0u) == 0;
            if (codes) {
                stage_ascii<true, true>(a.q.data + a.q.off[pA], a.q.data + a.q.off[pB], ql, qblocks, qst);
            } else { // a target byte outside ACGT somewhere in the wave: raw bytes for all of it
                stage_ascii<false, false>(a.t.data + a.t.off[pA], a.t.data + a.t.off[pB], tl, tblocks, tst);
                stage_ascii<false, true>(a.q.data + a.q.off[pA], a.q.data + a.q.off[pB], ql, qblocks, qst);
            }
        }
        // row 0 (the border row, sw.cpp:14-18,31-35) in stored form: H[0][j], E[1][j] = H[0][j] - o
        for (int j = 0; j <= ql; ++j) {
            const int hb0 = border(j, gopen, gext, indel) + j * gext + base;
            const unsigned hp = pack2(hb0, hb0);
            wm.bnd[(size_t)j * 64] = make_uint2(hp, pk_sub(hp, c.o_e));
        }
        // column 0 of the rows entering the other strips: H[32 k][0] (what a strip's first diagonal starts from)
        for (int k = 1; k <= strips; ++k) {
            const int hb0 = border(k * R, gopen, gext, indel) + k * R * gext + base;
            wm.bnd[(size_t)k * (ql + 1) * 64] = make_uint2(pack2(hb0, hb0), 0u);
        }
    }

    CK_PHASE(0); // staging
    // ---- pass 1
    int bestA = NEG_INF, bestA_i = -1, bestB = NEG_INF, bestB_i = -1;
    if (codes) {
        for (int k = 0; k < strips - 1; ++k) ck_strip<false, true>(k, tl, ql, nb, wm, qst, tst, c, gopen, gext, base, indel, bestA, bestA_i, bestB, bestB_i);
        ck_strip<true, true>(strips - 1, tl, ql, nb, wm, qst, tst, c, gopen, gext, base, indel, bestA, bestA_i, bestB, bestB_i);
    } else {
        for (int k = 0; k < strips - 1; ++k) ck_strip<false, false>(k, tl, ql, nb, wm, qst, tst, c, gopen, gext, base, indel, bestA, bestA_i, bestB, bestB_i);
        ck_strip<true, false>(strips - 1, tl, ql, nb, wm, qst, tst, c, gopen, gext, base, indel, bestA, bestA_i, bestB, bestB_i);
    }

    CK_PHASE(1); // pass 1
    // ---- last row (sw.cpp:116-127), as in sw_dp16_lane.hip: the last strip left H[tl][j] in the row "entering strip `strips`"
    int rmA = NEG_INF, rdA = 0x7fffffff, rjA = 0x7fffffff, rmB = NEG_INF, rdB = 0x7fffffff, rjB = 0x7fffffff;
    int cornerA = 0, cornerB = 0; // H[tl][ql]
    {
        const uint2 *const last = wm.bnd + (size_t)strips * (ql + 1) * 64;
        for (int j0 = 1; j0 <= ql; j0 += 8) { // eight columns' loads at once
            unsigned bots[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) bots[u] = last[(size_t)min(j0 + u, ql) * 64].x;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j <= ql) {
                    const unsigned bot = bots[u];
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
                    cornerA = sa; // (the last one stays: H[tl][ql])
                    cornerB = sb;
                }
            }
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

    CK_PHASE(2); // last row, records
    // ---- pass 2
    const int64_t pA = a.first + slotA, pB = a.first + slotB;
    // where the walks write: the caller's arrays, or (walk_in.coalesced_out) this tile's 128 slots of each in LDS -- cigar [128][stride] |
    // ScoreMax [128] | offset [128] | length [128] | status [128] -- which leave as whole lines below
    constexpr bool via_lds = VIA_LDS;
    TbArgs walk = walk_in;
    unsigned char *const l_score = out_lds + 128 * LANE_CK_OUT_STRIDE_MAX, *const l_off = l_score + 128 * 24, *const l_len = l_off + 512, *const l_status = l_len + 512;
    if (via_lds) {
        walk.cigar = reinterpret_cast<char *>(out_lds);
        walk.score = walk_in.score ? reinterpret_cast<Score *>(l_score) : nullptr;
        walk.offset = reinterpret_cast<int32_t *>(l_off);
        walk.cigar_len = walk_in.cigar_len ? reinterpret_cast<int32_t *>(l_len) : nullptr;
        walk.status = walk_in.status ? reinterpret_cast<int32_t *>(l_status) : nullptr;
    }
    const int64_t oA = via_lds ? 2 * lane : walk.dest ? walk.dest[pA] : pA, oB = via_lds ? 2 * lane + 1 : walk.dest ? walk.dest[pB] : pB;
    PathWalk wa, wb;
    wa.start(walk, rec[0], oA, tl, ql, lvalid, cornerA);
    wb.start(walk, rec[1], oB, tl, ql, validB, cornerB);
    BlockGeom geom;
    geom.ql = ql;
    geom.nb = nb;
    geom.match = match;
    geom.mismatch = a.mismatch;
    geom.gopen = gopen;
    geom.gext = gext;
    geom.base = base;
    geom.indel = indel;
    geom.codes = codes;
    const bool by_score = !(MGL_CK_ABLATE & 8);
    const int tblocks = strips * (R / 4), qblocks = (ql + 3) >> 2;
    while (!(MGL_CK_ABLATE & 1) && __builtin_amdgcn_ballot_w64(!wa.done || !wb.done) != 0) {
        if (by_score) {
            for (;;) { // every walk standing at a cell takes the diagonal stretches whose scores add up (PathWalk::verify_apply)
                unsigned ta[17], qa[17], tb[17], qb[17], ga[PathWalk::VB], gb[PathWalk::VB];
                if (wa.can_verify()) {
                    wa.win64_load(wm, 0, tblocks, qblocks, ta, qa);
                    wa.grid_load(wm, ql, ga);
                }
                if (wb.can_verify()) {
                    wb.win64_load(wm, 1, tblocks, qblocks, tb, qb);
                    wb.grid_load(wm, ql, gb);
                }
                const bool ma = wa.verify_apply(ta, qa, ga, 0, geom), mb = wb.verify_apply(tb, qb, gb, 1, geom);
                if (!__builtin_amdgcn_ballot_w64(ma || mb)) break;
            }
            CK_PHASE(3); // stretches that add up
            if (__builtin_amdgcn_ballot_w64(!wa.done || !wb.done) == 0) break;
        }
        // (a finished walk's half recomputes some valid block's worth of garbage: both halves run the same instructions anyway)
        const int kA = max(wa.pi - 1, 0) >> 4, bA = max(wa.pj - 1, 0) / CK, kB = max(wb.pi - 1, 0) >> 4, bB = max(wb.pj - 1, 0) / CK;
        // the columns of its block a walk can still reach end at the one it stands in: the wave recomputes up to the farthest such column
        int reach = max(!wa.done ? (max(wa.pj - 1, 0) % CK) + 1 : 0, !wb.done ? (max(wb.pj - 1, 0) % CK) + 1 : 0);
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) reach = max(reach, __shfl_xor(reach, m));
        const int groups = __builtin_amdgcn_readfirstlane((reach + 7) >> 3); // (61.80 ms per 10 M pairs against 61.98 with every block in full)
        if (codes)
            ck_block<true>(kA, bA, kB, bB, !wa.done, !wb.done, geom, wm, c, groups);
        else
            ck_block<false>(kA, bA, kB, bB, !wa.done, !wb.done, geom, wm, c, groups);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the block's flags are in memory
        CK_PHASE(4); // a block's flags
#ifdef MGL_CK_PHASES
        if (lane == 0) atomicAdd(&mgl_ck_phase_ticks[8], 1ull);
#endif
        for (;;) {
            const bool la = wa.can_load(kA, bA), lb = wb.can_load(kB, bB);
            // (a lane with nothing to fetch still runs apply(): a run that ends at the matrix's edge needs no flags)
            unsigned fa[PathWalk::LOOK], fb[PathWalk::LOOK], ta[5], qa[5], tb[5], qb[5];
            if (la) {
                wa.load(wm, fa);
                if (wa.mode == 0) wa.win_load(wm, 0, tblocks, qblocks, ta, qa);
            }
            if (lb) {
                wb.load(wm, fb);
                if (wb.mode == 0) wb.win_load(wm, 1, tblocks, qblocks, tb, qb);
            }
            const bool ga = wa.apply(fa, ta, qa, 0, kA, bA, geom), gb = wb.apply(fb, tb, qb, 1, kB, bB, geom);
            if (!__builtin_amdgcn_ballot_w64(ga || gb)) break;
        }
        wa.stuck = wb.stuck = false; // either walk has moved on
        CK_PHASE(5); // the walk inside the block
    }
    CK_PHASE(6);
    if (lvalid) wa.finish(walk, a.rec + slotA, oA);
    if (validB) wb.finish(walk, a.rec + slotB, oB);
    if (via_lds) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const int64_t p0 = a.first + gw * 128;
        const int np = (int)min((int64_t)128, a.count - gw * 128); // pairs of this tile
        lds_to_global(walk_in.cigar + (size_t)p0 * walk_in.cigar_stride, out_lds, np * walk_in.cigar_stride, lane);
        if (walk_in.score) lds_to_global(walk_in.score + p0, l_score, np * 24, lane);
        lds_to_global(walk_in.offset + p0, l_off, np * 4, lane);
        if (walk_in.cigar_len) lds_to_global(walk_in.cigar_len + p0, l_len, np * 4, lane);
        if (walk_in.status) lds_to_global(walk_in.status + p0, l_status, np * 4, lane);
        __builtin_amdgcn_wave_barrier(); // (the next tile's walks write the same LDS)
    }
    CK_PHASE(7); // overhangs, text, results
#ifdef MGL_CK_PHASES
    if (lane == 0) atomicAdd(&mgl_ck_phase_ticks[9], 1ull);
#endif
}

} // namespace

// A PERSISTENT grid: a.lane_slots waves (at most what the chip holds at two waves per SIMD -- 256 registers: no spills in pass 1's
// loops), each with one region of the workspace, each taking tile after tile: its first one by its slot number, the following ones
// off a counter in device memory (a.tile_ctr[0]; zero when the launch starts) until the tiles are gone.  The workspace a launch needs is therefore
// lane_slots regions (2 MB each at 256 x 150: 4 GB for a whole MI355X) however many pairs it holds -- round 3 gave every TILE a region
// (15 KB per pair: 208 GiB for the bench's 10 M pairs in one launch, and chunks wherever the workspace was smaller).  Every wave
// reaches the loop's exit: the counter only grows, a tile's work is bounded, nothing waits for another wave.
// ONE wave per workgroup: the waves share nothing (no LDS, no barrier), and a workgroup's registers are released only when its LAST
// wave has ended (round 3, four per workgroup: three of four SIMD slots stood empty until the slowest was through).
// A three-wave-per-SIMD build of the same code (168 registers, spills) was slower everywhere in round 3 and is gone.
// HOW THE TWO WAVES OF A SIMD SHARE IT (traced, scripts/ck_trace.py, docs/history.md C.0): the arbiter serves the OLDER wave first -- one
// wave of every SIMD runs a tile in 1.2 ms, the other, on what the first leaves free, in 2.5 (a 10 M-pair launch: 1 024 slots with 50
// tiles each, 1 024 with 25); together 1.21 tiles per ms and SIMD, one wave alone 0.95.  The counter evens that out: whoever is through
// draws the next tile.  Rules that keep the slow waves from drawing a launch's last tiles were built and measured (the end of a
// 1.25 M-pair launch moved between 8.9 and 9.6 ms either way, box by box) and are gone.
// TWO kernels, one per way the results leave (VIA_LDS: in whole lines out of LDS, TbArgs::coalesced_out; else lane by lane through the
// caller's pointers): one kernel holding both tile functions had the registers of both to colour at once (192 spilled against 66).
template <bool VIA_LDS>
__device__ __forceinline__ void lane_ck_grid(const DpArgs &a, const TbArgs &walk, unsigned char *out_lds)
{
    const int lane = threadIdx.x & 63;
    const int64_t tiles = (((a.count + 1) >> 1) + 63) >> 6, slots = gridDim.x, slot = blockIdx.x;
    // in-kernel clock probe (profiling level 2; off in normal runs): shader-clock ticks and 100 MHz ticks of this wave's whole life
    unsigned long long diag_t0 = 0, diag_w0 = 0;
    if (a.diag) {
        diag_t0 = __builtin_amdgcn_s_memtime();
        diag_w0 = __builtin_amdgcn_s_memrealtime();
    }
    int64_t arrived = a.gate ? 0 : INT64_MAX; // pairs of the batch known to be in device memory
    unsigned long long *const ctr = reinterpret_cast<unsigned long long *>(a.tile_ctr); // {draws, waves out}: ONE object (below)
    for (int64_t tile = slot; tile < tiles;) {
        // The host entries run ONE launch over a batch whose inputs are still crossing the link: the copy engines bring them chunk by chunk
        // and the host moves a.gate on as each chunk has landed.  A wave whose tile is not there yet looks at the word (a read over the
        // link, so only then) and sleeps in between; the copies do not depend on anything this grid does, so the wait ends -- and if the
        // word stands still for gate_timeout_ticks the wave raises gate_failed and leaves, its tiles undone (the host sees the flag).
        // (a tile's aligned loads may touch the first line of the pair behind its last one: the NEXT tile's pairs must have arrived too, so
        // that no line of an input array enters a cache before its bytes are there)
        if (arrived < a.first + min(a.count, (tile + 2) * 128)) {
            const int64_t need = a.first + min(a.count, (tile + 2) * 128);
            // ONE wave at a time looks at the host's word (round 5).  Packed inputs arrive eight times faster than the grid consumes them:
            // a wave seldom waits.  ASCII inputs do not -- the link is the bound, every wave of the grid stands at the gate most of the
            // time -- and 2 048 waves each reading a word over the link every few microseconds put a gigabyte per second of read
            // completions on the very direction the inputs travel in.  So the word has a mirror in DEVICE memory (a.gate_dev[0]) that the
            // waiting waves watch, and whoever finds the mirror older than a microsecond (a.gate_dev[1], 100 MHz ticks; a compare-and-swap
            // picks one wave) reads the host's word and moves the mirror on.  A negative word (the host calls the grid off) goes into the
            // mirror as it is.
            int lo = 0, hi = 0, flags = 0; // lane 0 waits, everybody else waits for lane 0: {seen.lo, seen.hi, 1 = gave up}
            if (lane == 0) {
                unsigned long long since = __builtin_amdgcn_s_memrealtime();
                long long last = arrived, seen;
                for (;;) {
                    seen = (long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(a.gate_dev), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (seen >= need || seen < 0) break;
                    const unsigned long long now = __builtin_amdgcn_s_memrealtime();
                    const unsigned long long polled = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(a.gate_dev) + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (now - polled >= 100ull) {
                        unsigned long long expect = polled;
                        if (__hip_atomic_compare_exchange_strong(reinterpret_cast<unsigned long long *>(a.gate_dev) + 1, &expect, now, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                            const long long v = (long long)__hip_atomic_load(reinterpret_cast<const unsigned long long *>(a.gate), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            if (v < 0)
                                __hip_atomic_store(reinterpret_cast<long long *>(a.gate_dev), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            else
                                __hip_atomic_fetch_max(reinterpret_cast<long long *>(a.gate_dev), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            seen = v;
                            if (seen >= need || seen < 0) break;
                        }
                    }
                    if (seen != last) {
                        last = seen;
                        since = now;
                    } else if (now - since > (unsigned long long)a.gate_timeout_ticks) {
                        flags = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(32);
                }
                lo = (int)(unsigned)(unsigned long long)seen;
                hi = (int)(unsigned)((unsigned long long)seen >> 32);
            }
            lo = __builtin_amdgcn_readfirstlane(lo);
            hi = __builtin_amdgcn_readfirstlane(hi);
            flags = __builtin_amdgcn_readfirstlane(flags);
            arrived = (int64_t)((unsigned long long)(unsigned)lo | (unsigned long long)(unsigned)hi << 32);
            if (flags & 1) { // the word stood still for gate_timeout_ticks: the host sees the flag and does the call again the chunked way
                if (lane == 0) __hip_atomic_store(a.gate_failed, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
            if (arrived < 0) break; // the host calls the grid off (an index array failed its check, a copy failed): leave, nothing is wrong with the device
            // An acquire fence between the look that let the wave through and the loads of the pairs it announced: once per WAIT (a wave
            // waits a handful of times while the inputs cross the link, never once they are there), so it costs nothing that can be
            // measured -- and without it the order of those loads behind a relaxed load rested on a branch and on the argument that no
            // cache holds a line of pairs that have just arrived (the margin above; the grid starts with empty caches).  That argument
            // still holds; the fence is what makes the order the language's, not the hardware's.  (-DMGL_CK_NO_GATE_FENCE: round 4's form.)
#ifndef MGL_CK_NO_GATE_FENCE
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        }
#ifdef MGL_CK_TRACE
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
#endif
        sw_dp16_lane_ck_tile<VIA_LDS>(a, walk, tile, slot, lane, out_lds);
#ifdef MGL_CK_TRACE
        if (lane == 0 && tile < (1 << 17)) {
            unsigned hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            mgl_ck_trace[4 * tile + 0] = (unsigned long long)slot;
            mgl_ck_trace[4 * tile + 1] = (unsigned long long)hw | (unsigned long long)(xcc & 15u) << 32;
            mgl_ck_trace[4 * tile + 2] = t0;
            mgl_ck_trace[4 * tile + 3] = __builtin_amdgcn_s_memrealtime();
        }
#endif
        if (tiles <= slots) break; // (every tile has its wave: the counter is not even touched)
        unsigned next = 0;
        if (lane == 0) {
            next = (unsigned)__hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // The waves of one launch draw `tiles` times at most (tiles - slots draws that find a tile, one per wave that does not), so a
            // larger number means the word did not stand at zero when the launch started: somebody else's launch is on it, or memory is
            // not what it was.  The host finds the flag at its next look (mgl_sw_ctx_check, every later call): MGL_SW_ERR_DEVICE, never
            // a silent tile left undone.
            if (next >= (unsigned)tiles && a.grid_fault) __hip_atomic_store(a.grid_fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        tile = slots + (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)next);
    }
    // THE LAST WAVE OUT ZEROES THE COUNTER (round 5).  Every wave of the grid comes through here -- after its last draw, or from the gate
    // when the host has called the launch off or the gate stood still -- and counts itself out in the word's HIGH half; the one that
    // finds slots - 1 there knows that nobody will draw again and puts the word back to zero.  So a launch finds its counter at zero
    // however the launch before it on that word ended.  (Round 4 never reset the word and kept a host-side copy of where it stood, "moved
    // on by exactly `tiles` per launch" -- which a launch called off at its gate did not do: the copy and the word parted, and 64 launches
    // later the grid that came round to that word left its tiles undone with status 0.)
    // Draws (low half) and waves out (high half) are ONE 64-bit atomic object on purpose: a wave's draws come before its count-out in
    // that object's modification order (same thread, same object), the last count-out reads the value every other count-out has left,
    // and the store of zero follows it -- all with RELAXED operations.  As two words the same chain needed an acquire-release on every
    // wave's way out, which on this chip is a write-back and an invalidate of the XCD's L2 (buffer_wbl2 / buffer_inv, 2 048 of them
    // while the last tiles are still writing their rows): against round 4's library on one box, launches of 2 M pairs were
    // 1.3-1.6 % behind it in that form and are 0.7 % behind it in this one (profiles/r05_ab_r04_vs_head.txt; 10 M pairs: level).
    if (tiles > slots && lane == 0) {
        const unsigned out = (unsigned)(__hip_atomic_fetch_add(ctr, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
        if (out == (unsigned)slots - 1u) {
            __hip_atomic_store(ctr, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else if (out >= (unsigned)slots && a.grid_fault) {
            __hip_atomic_store(a.grid_fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (a.diag && lane == 0) {
        a.diag[2 * slot] = __builtin_amdgcn_s_memtime() - diag_t0;
        a.diag[2 * slot + 1] = __builtin_amdgcn_s_memrealtime() - diag_w0;
    }
}

__global__ __launch_bounds__(64, 2) void sw_dp16_lane_ck_kernel(const DpArgs a, const TbArgs walk)
{
    __shared__ __attribute__((aligned(16))) unsigned char out_lds[LANE_CK_OUT_LDS_BYTES]; // a tile's results on their way out
    lane_ck_grid<true>(a, walk, out_lds);
}

// results lane by lane through the caller's pointers (scattered destinations, strides the LDS form does not take); no LDS
__global__ __launch_bounds__(64, 2) void sw_dp16_lane_ck_scatter_kernel(const DpArgs a, const TbArgs walk) { lane_ck_grid<false>(a, walk, nullptr); }

// either wire format, the same for both sequence sets (the kernel stages base codes, sw_lane_cell.h)
bool lane16_ck_supported(const SeqSet &t, const SeqSet &q) { return (t.packed2 != 0) == (q.packed2 != 0); }

// a.lane_slots regions at a.tb / a.scratch; wherever the launch holds more tiles than slots: a.tile_ctr, one 64-bit word {draws, waves out}
// that stands at zero when the kernel starts and that no other launch in flight uses.  The grid's last wave out puts them back to zero
// (above), so no launch needs a reset in front of it -- a memset in front of every launch is a KERNEL of its own, and behind a grid that
// holds every wave slot of the chip it waited for that grid's end (traced in round 4: the two streams of the host entries stopped
// overlapping).
hipError_t launch_dp16_lane_ck(const DpArgs &a, const TbArgs &walk, hipStream_t stream)
{
    const int64_t tiles = ((a.count + 1) / 2 + 63) / 64;
    if (a.lane_slots < 1 || (tiles > a.lane_slots && !a.tile_ctr)) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(tiles < a.lane_slots ? tiles : a.lane_slots)), block(64);
    if (walk.coalesced_out)
        hipLaunchKernelGGL(sw_dp16_lane_ck_kernel, grid, block, 0, stream, a, walk);
    else
        hipLaunchKernelGGL(sw_dp16_lane_ck_scatter_kernel, grid, block, 0, stream, a, walk);
    return hipGetLastError();
}

} // namespace mgl_sw_dev

#ifdef MGL_CK_TRACE
extern "C" int mgl_ck_trace_dump(const char *path)
{
    static unsigned long long h[4 << 17];
    hipDeviceSynchronize();
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(mgl_ck_trace), sizeof h) != hipSuccess) return 1;
    FILE *f = fopen(path, "wb");
    if (!f) return 2;
    fwrite(h, 1, sizeof h, f);
    fclose(f);
    memset(h, 0, sizeof h);
    hipMemcpyToSymbol(HIP_SYMBOL(mgl_ck_trace), h, sizeof h);
    return 0;
}
#endif

#ifdef MGL_CK_PHASES
extern "C" void mgl_ck_phases_dump()
{
    unsigned long long h[16] = {0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(h, HIP_SYMBOL(mgl_ck_phase_ticks), sizeof h);
    static const char *names[8] = {"staging", "pass 1", "last row + records", "stretches", "block flags", "walk in block", "(loop exit)", "finish"};
    unsigned long long tot = 0;
    for (int k = 0; k < 8; ++k) tot += h[k];
    for (int k = 0; k < 8; ++k) fprintf(stderr, "phase %-20s %6.2f %%  %10.1f ticks per wave\n", names[k], 100.0 * h[k] / (double)tot, (double)h[k] / (double)h[9]);
    fprintf(stderr, "waves %llu, block rounds per wave %.2f\n", h[9], (double)h[8] / (double)h[9]);
    memset(h, 0, sizeof h);
    hipMemcpyToSymbol(HIP_SYMBOL(mgl_ck_phase_ticks), h, sizeof h);
}
#endif
