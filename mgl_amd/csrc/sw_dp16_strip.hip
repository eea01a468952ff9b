// sw_dp16_strip.hip -- long reads with the lane kernel's economy: one pair per WORKGROUP, every lane-half owns one STRIP of
// 32 target rows of that pair and sweeps the query columns with its strip in registers (column<32>() of sw_lane_cell.h, the
// same code sw_dp16_lane.hip runs for two whole pairs per lane).  Same function as every other fill kernel here (the
// reference's sw.cpp:5-146), same decisions, hence the same traceback bits.
//
// Schedule.  NL = 64 * W lanes per workgroup; the low halves of the registers of lane l hold strip l, the high halves strip
// NL + l; strip g works on column group cg = step - g (four columns per step), so a strip is always one step behind the
// strip above it and what that strip's last row produced in the previous step -- H and E' of four columns -- is exactly
// what it needs now.  That hand-over is a DPP move to the next lane (both halves at once), a 32-byte LDS mailbox from the
// last lane of a wave to the first lane of the next (and from the very last lane's low half to the first lane's high half:
// strip NL - 1 -> strip NL), and one workgroup barrier per step.  Everything else stays in the lane: no ring, no carry row.
// The price is the pipeline: a pair of tl x ql keeps ceil(tl / 32) of the 2 NL strip slots busy for ql / 4 of ql / 4 + 2 NL - 1
// steps (10 kb x 10 kb, W = 3: 71 %).
//
// 16 bits.  Scores of long reads leave 16 bits, but a strip only ever sees its own 32 rows of one column, the row above and
// the column before: every lane-half keeps its values relative to a baseline of its own (stored = X + (i+j)e - B, B an
// int32 per half), moved every 16 columns so that the strip's first row sits at a fixed level.  What crosses between strips
// crosses as the residue modulo 2^16 of the TRUE value; the receiver subtracts its own baseline modulo 2^16, which is exact
// because the true difference fits its window.  The window is a static property of the scoring parameters (strip16_range_ok):
// rows of one column differ by at most match + o + e (upwards) or o - e (downwards) per row, a row's value moves by at most
// match + o + e per column -- 31 rows + 16 columns of drift + the intermediates -- so there is no run-time check and no fall-back.
//
// Traceback layout ("strip16", layout 4, DpRecord.g_tail = -(100 + W), DpRecord.sps = steps): per pair
// [wave][step][column of the group][16-row group][lane] uint4, the byte layout of the lane kernel (low bytes = low half);
// a wave stores 1 KB per instruction although its lanes are at different columns.  Cell (i, j): strip g = (i-1)/32,
// lane = g mod NL, half = g / NL, step = (j-1)/4 + g.
//
// Without stored flags (DpArgs::strip_k = K > 0, round 3; layout 6).  A 10 kb x 10 kb pair's path visits 0.02 % of the cells the flags
// are written for (67 MB per pair).  In this form the fill runs the score-only column code (9 instead of 17 instructions per two
// cells) and keeps what sw_strip_ck_walk_kernel (sw_strip_walk.hip) needs to recompute the 60 x 256 blocks the path crosses: the
// true scores {H, E} of the row below every band of K strips, per column, and {H, F} of every row at the band's checkpoint columns
// (every STRIP_CK_COLS = 128 columns, at the step offset of the band so that all strips of a band save the SAME column; a dword per row as the
// strip holds it, and the strip's baseline; the rows likewise, with a baseline per 16 columns): 10 MB per 10 kb pair.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "sw_device.h"
#include "sw_lane_cell.h"

namespace mgl_sw_dev {

namespace {

// rows per strip from which the kernels are built for two waves per SIMD (see MGL_STRIP_OCC below)
#ifndef MGL_STRIP_OCC2_FROM
#define MGL_STRIP_OCC2_FROM 27
#endif
constexpr int CPS = STRIP_CPS;       // columns per step
constexpr int STRIP_LEVEL = -14000;  // where a move of the baseline puts the strip's first row

__device__ __forceinline__ unsigned dpp_wave_shr1(unsigned lane0_value, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)lane0_value, (int)src, 0x138, 0xf, 0xf, false);
}

// CODES (round 4; the kernels without stored flags, where the query's tables fit the LDS carve and every target byte of the pair is one
// of ACGT -- 2-bit packed inputs always): "do the bases differ" is ONE v_perm_b32 per two cells (sw_lane_cell.h: the row's selector picks
// its code's byte out of the column's table) where the byte compare takes a xor and a min -- 8 instead of 9 instructions per two cells.
// The tables (one dword per query column: byte c = 0 where the query base has code c, else 1; all ones for a query byte outside the
// target's alphabet) are built once per pair in LDS, where the byte form keeps the query itself: four times the bytes, and no
// instruction in the step -- each half reads its four columns' tables with one ds_read_b128.
typedef int strip_int4 __attribute__((ext_vector_type(4)));

template <int SR, bool NOTB, bool CODES>
__device__ __forceinline__ void sw_dp16_strip_body(const DpArgs &a, unsigned char *smem)
{
    const int L = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = blockDim.x >> 6, NL = 64 * W;
    const int ell = w * 64 + L;
    const int64_t slot = blockIdx.x;
    const int64_t p = a.first + slot;
    const int64_t t0 = a.t.off[p], q0 = a.q.off[p];
    const int tl = a.t.length(p), ql = a.q.length(p);
    const int NCG = strip16_groups(ql);       // column groups
    const int steps = NCG + 2 * NL - 1;
    const int steps_cap = strip16_steps(a.uni_ql, W); // what the regions are sized for
    const int match = a.match, gopen = a.gopen, gext = a.gext;
    const bool indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;

    // LDS: query as dwords (group cg = bases 4cg+1 .. 4cg+4, zero padded) | mailbox[2][W][8] | last-column key (u64) | scan results
    const int qwords = CODES ? strip16_table_words(a.uni_ql) : strip16_qwords(a.uni_ql);
    unsigned *const Q = reinterpret_cast<unsigned *>(smem);
    unsigned *const mbox = Q + qwords;
    unsigned long long *const key = reinterpret_cast<unsigned long long *>(mbox + 2 * W * 8);
    int *const wres = reinterpret_cast<int *>(key + 1);
    // per pair in HBM: row tl of every column as {packed register, baseline}, [column][2] | per lane two parked last columns of
    // 32 rows + baseline, [lane][half][33]
    int *const rowbuf = reinterpret_cast<int *>(a.scratch + (size_t)slot * (size_t)strip16_scratch_bytes(a.uni_ql, W));
    unsigned *const cap = reinterpret_cast<unsigned *>(rowbuf + (size_t)(a.uni_ql + 8) * 2) + (size_t)ell * 66;
    {
        for (int x = threadIdx.x; x < 2 * W * 8; x += blockDim.x) mbox[x] = 0u;
        if (threadIdx.x == 0) key[0] = 0ull;
        if (CODES) { // column j's table at dword j - 1; behind the query: columns that feed nothing
            for (int x = threadIdx.x; x < qwords; x += blockDim.x) {
                unsigned q8 = 32u;
                if (x < ql) {
                    const unsigned b = (unsigned)a.q.at(q0, x);
                    if (a.q.packed2) {
                        q8 = 8u * b;
                    } else {
                        unsigned diff;
                        const unsigned code = ascii_codes(b, diff);
                        q8 = (diff & 0xffu) ? 32u : 8u * (code & 3u);
                    }
                }
                Q[x] = code_table(q8);
            }
        } else {
            for (int x = threadIdx.x; x < qwords; x += blockDim.x) Q[x] = 0u;
            __syncthreads();
            unsigned char *qb = reinterpret_cast<unsigned char *>(Q);
            for (int x = threadIdx.x; x < ql; x += blockDim.x) qb[x] = (unsigned char)a.q.at(q0, x);
        }
        __syncthreads();
    }

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
    const unsigned level = pack2(STRIP_LEVEL, STRIP_LEVEL);

    // Targets beyond the 2 NL strips of 32 rows the workgroup's lanes hold (16 384 rows with four waves) take several PASSES (round 4;
    // the kernels without stored flags): pass p computes strips 2 NL p .. 2 NL (p + 1) - 1, and where pass 0 gives strip 0 the border
    // row by its formula, a later pass takes the row the pass before it kept below its last band (2 NL is a multiple of K = 2 there).
    const int passes = NOTB ? max(a.strip_passes, 1) : 1;
    unsigned h[SR], f[SR], t[SR];
    unsigned hd = 0u;
    int base_a = 0, base_b = 0;                // the halves' baselines (int32): true stored value = register + baseline
    unsigned bres = 0u;                        // the same, modulo 2^16, packed
    unsigned out_h[CPS] = {}, out_e[CPS] = {}; // what this lane hands on: TRUE residues of (H, E') of its last rows
    const int ulast = (ql - 1) % CPS;
    const int gl = (tl - 1) / SR;                     // the strip and (as a scalar) the register row of target row tl
    const int rl_s = __builtin_amdgcn_readfirstlane((tl - 1) % SR);
    int best = NEG_INF, best_i = -1;            // last-column candidates of this lane (both halves merged: later row wins)

    uint4 *const tb_wave = reinterpret_cast<uint4 *>(a.tb + (size_t)slot * (size_t)a.tb_stride_words) + (size_t)w * steps_cap * CPS * 2 * 64 + L;
    // NOTB: the rows and checkpoints kept instead of the flags (strip16_ck_words): int2 entries
    constexpr int K = NOTB ? 64 / SR : 1; // (= a.strip_k: the host sets it so, launch_dp16_strip checks; a constant here: the divisions below are multiplications)
    // what is kept instead of the flags (strip16_ck_words, sw_device.h): dwords {H, gap value} as the strips hold them, and their baselines
    int *const rows_raw = reinterpret_cast<int *>(a.tb + (size_t)slot * (size_t)a.tb_stride_words);
    const int row_stride = strip16_ck_row_stride(a.uni_ql), row_blocks = strip16_ck_row_blocks(a.uni_ql);
    int *const rows_base = rows_raw + strip16_ck_off_rowbase(a.uni_tl, a.uni_ql, SR, K);
    int *const ck_cols = rows_raw + strip16_ck_off_cols(a.uni_tl, a.uni_ql, SR, K);
    int *const ck_base = rows_raw + strip16_ck_off_base(a.uni_tl, a.uni_ql, SR, K);
    const int ck_strips = strip16_ck_strips(a.uni_tl, SR);
    // (per step, below: bandA / bandB = the band of K strips a strip belongs to; rowoffA / rowoffB = that band's kept row as a 32-bit index off
    // the pair's uniform base; rowsA / rowsB = the strip whose last row is the row below a band -- and not the matrix's last rows: it writes that row)

  for (int pass = 0; pass < passes; ++pass) {
    const int G0 = pass * 2 * NL; // the pass's first strip
    {
        const int i0A = SR * (G0 + ell), i0B = SR * (G0 + NL + ell); // rows i0 + 1 .. i0 + SR
#pragma unroll
        for (int r = 0; r < SR; ++r) {
            const int ra = i0A + r, rb = i0B + r; // 0-based row indices
            const unsigned ba = (unsigned)(ra < tl ? a.t.at(t0, ra) : 0), bb = (unsigned)(rb < tl ? a.t.at(t0, rb) : 0);
            if (CODES) // the row's selector (sw_lane_cell.h): its code for the low half's table, 4 + its code for the high half's
                t[r] = CODE_SEL | (a.t.packed2 ? ba : (ba >> 1) & 3u) | ((a.t.packed2 ? bb : (bb >> 1) & 3u) << 16);
            else
                t[r] = ba | (bb << 16);
            h[r] = f[r] = 0u;
        }
    }
    // the row that enters this pass's first strip (passes behind the first): where the pass before kept it
    const int top_band = G0 / K - 1;
    const int *const top_raw = rows_raw + (size_t)max(top_band, 0) * row_stride, *const top_base = rows_base + (size_t)max(top_band, 0) * row_blocks;
    strip_int4 top_v = {0, 0, 0, 0};
    int top_b = 0;
    if (pass > 0 && w == 0) { // (column group 0; the next group is fetched a step ahead, below)
        top_v = __builtin_nontemporal_load(reinterpret_cast<const strip_int4 *>(top_raw));
        top_b = __builtin_nontemporal_load(top_base);
    }
    for (int s = 0; s < steps; ++s) {
        // What a lane knows about its two strips is a handful of small functions of its number.  Left to itself the compiler computes
        // every one of them (and every intermediate the steps below derive from them) once in front of the loop and keeps it in a
        // register: some thirty registers the 32 x 3 + 16 of the strips' state leave no room for -- they were spilled (up to 281 per
        // lane in round 3) and loaded back step after step.  Made opaque here, they are recomputed per step: a dozen instructions
        // against the ~ 600 of a step's four columns.
        // (strips of up to 24 rows, and those of 29 and more with their 256 registers, leave room for them: there they stay loop-invariant
        // -- some 30 instructions a step less: 10 kb pairs, 20 rows, 91.2 -> 86.3 ms per 4 608 pairs)
        int el = ell;
        if (NOTB && SR >= 25 && SR < MGL_STRIP_OCC2_FROM) asm volatile("" : "+v"(el));
        const int gA = el, gB = NL + el;       // the strips' places in the pass's pipeline ...
        const int GA = G0 + gA, GB = G0 + gB;  // ... and their numbers in the pair
        const int i0A = SR * GA, i0B = SR * GB;
        const bool own_last_a = GA == gl, own_last_b = GB == gl;
        const int bandA = GA / K, bandB = GB / K;
        const int rowoffA = (int)__umul24((unsigned)bandA, (unsigned)row_stride), rowoffB = (int)__umul24((unsigned)bandB, (unsigned)row_stride); // (both below 2^24: a full-rate multiply)
        const bool rowsA = NOTB && (GA + 1) % K == 0 && (GA + 1) * SR < tl, rowsB = NOTB && (GB + 1) % K == 0 && (GB + 1) * SR < tl;
        const int cgA = s - gA, cgB = s - gB;
        const bool actA = cgA >= 0 && cgA < NCG && i0A < tl, actB = cgB >= 0 && cgB < NCG && i0B < tl;
        // ---- 1. what the strip above handed on in the previous step arrives column by column, right before it is used (below):
        // lane 0 takes it from the mailbox of the wave before -- wave 0: the border row for strip 0 (sw.cpp:14-18,31-35: H[0][j],
        // E'[1][j] = H[0][j] - o) and the last lane's low half for strip NL
        const unsigned *const mb_in = mbox + (((s + 1) & 1) * W + (w == 0 ? W - 1 : w - 1)) * 8;
        auto take = [&](const int u, unsigned &ih, unsigned &ie) {
            unsigned l0h = mb_in[u], l0e = mb_in[4 + u];
            if (w == 0) {
                const int j = CPS * s + u + 1;
                int hb0 = border(j, gopen, gext, indel) + j * gext, eb0 = hb0 - (gopen - gext);
                if (pass > 0) { // {H, E} of the row above, as the pass before kept them: 16-bit values + their baseline (residues are all that is handed on)
                    const int v = u == 0 ? top_v.x : u == 1 ? top_v.y : u == 2 ? top_v.z : top_v.w;
                    hb0 = (v & 0xffff) + top_b;
                    eb0 = (v >> 16) + top_b;
                }
                l0h = ((unsigned)hb0 & 0xffffu) | (l0h << 16);
                l0e = ((unsigned)eb0 & 0xffffu) | (l0e << 16);
            }
            ih = dpp_wave_shr1(l0h, out_h[u]);
            ie = dpp_wave_shr1(l0e, out_e[u]);
        };
        unsigned keptA[CPS] = {}, keptB[CPS] = {}; // NOTB: the row below a band, this step's columns: {H, E entering the next row} as the strip holds them
        // (wave-uniform: the DPP moves below need the lane before to be enabled; what an idle lane computes feeds nothing real)
        if (__builtin_amdgcn_ballot_w64(actA || actB)) {
            // ---- 2. a half that starts now: column 0 of its rows (sw.cpp:24,38,47-49), its baseline on its first row
            if (cgA == 0 || cgB == 0) {
                // column 0 of 32 consecutive rows in stored form, relative to the first of them: H[i][0] + i e is i e (border 0) or
                // constant (leading / trailing gaps allowed: -o - (i-1) e + i e), so row r sits at level + r * c1
                const int c1 = indel ? 0 : gext;
                const bool hb_ = cgB == 0; // (the two halves never start in the same step: they are NL steps apart)
                const int i0 = hb_ ? i0B : i0A;
                const int b0 = border(i0 + 1, gopen, gext, indel) + (i0 + 1) * gext - STRIP_LEVEL;
                base_a = hb_ ? base_a : b0;
                base_b = hb_ ? b0 : base_b;
                const unsigned keep = hb_ ? 0x0000ffffu : 0xffff0000u;
                bres = (bres & keep) | (pack2(b0, b0) & ~keep);
                int lv = STRIP_LEVEL;
#pragma unroll
                for (int r = 0; r < SR; ++r) {
                    h[r] = (h[r] & keep) | (pack2(lv, lv) & ~keep);
                    f[r] = (f[r] & keep) | (pack2(lv - (gopen - gext), lv - (gopen - gext)) & ~keep);
                    lv += c1;
                }
                const int hd0 = border(i0, gopen, gext, indel) + i0 * gext - b0; // H[i0][0]
                hd = (hd & keep) | (pack2(hd0, hd0) & ~keep);
            }
            // ---- 3. every 16 columns the baselines move: the strip's first row back to its level (both halves are in the same phase)
            if ((cgA & (16 / CPS - 1)) == 0) {
                const unsigned d = pk_sub(h[0], level);
#pragma unroll
                for (int r = 0; r < SR; ++r) {
                    h[r] = pk_sub(h[r], d);
                    f[r] = pk_sub(f[r], d);
                }
                hd = pk_sub(hd, d);
                base_a += lo16(d);
                base_b += hi16(d);
                bres = pk_add(bres, d);
                if (NOTB) { // the strip that writes the row below its band: that row's baseline for these 16 columns
                    if (rowsA && actA) rows_base[bandA * row_blocks + (cgA >> 2)] = base_a;
                    if (rowsB && actB) rows_base[bandB * row_blocks + (cgB >> 2)] = base_b;
                }
            }
            // ---- 4. four columns
            // (a query dword holds 4 / CPS groups; the two halves are 64 W groups apart, so they sit in the same place of their dwords)
            constexpr int GPD = 4 / CPS;
            static_assert(!CODES || CPS == 4, "the tables of a step are one 16-byte read");
            const unsigned qa = CODES ? 0u : Q[min(max(cgA / GPD, 0), qwords - 1)], qb = CODES ? 0u : Q[min(max(cgB / GPD, 0), qwords - 1)];
            const unsigned qsel = 0x0c040c00u + 0x00010001u * (unsigned)(CPS * (cgA & (GPD - 1)));
            uint4 tabA = make_uint4(0u, 0u, 0u, 0u), tabB = tabA; // CODES: the tables of this step's four columns, either half
            if (CODES) {
                const uint4 *const T4 = reinterpret_cast<const uint4 *>(Q);
                tabA = T4[min(max(cgA, 0), (qwords >> 2) - 1)];
                tabB = T4[min(max(cgB, 0), (qwords >> 2) - 1)];
            }
            uint4 *tbp = tb_wave + (size_t)s * CPS * 2 * 64;
#pragma unroll
            for (int u = 0; u < CPS; ++u) {
                unsigned ih, e;
                take(u, ih, e);
                e = pk_sub(e, bres);
                if (CODES) {
                    const unsigned qA = u == 0 ? tabA.x : u == 1 ? tabA.y : u == 2 ? tabA.z : tabA.w, qB = u == 0 ? tabB.x : u == 1 ? tabB.y : u == 2 ? tabB.z : tabB.w;
                    column<SR, NOTB, false, true>(h, f, t, qA, hd, e, c, tbp + (size_t)u * 2 * 64, nullptr, qB);
                } else {
                    const unsigned q = __builtin_amdgcn_perm(qb, qa, qsel + 0x00010001u * (unsigned)u);
                    column<SR, NOTB>(h, f, t, q, hd, e, c, tbp + (size_t)u * 2 * 64);
                }
                hd = pk_sub(ih, bres);
                out_h[u] = pk_add(h[SR - 1], bres);
                out_e[u] = pk_add(e, bres);
                if (NOTB) { // {H[i][j], E entering row i + 1} of the strip's last row: the low halves, the high halves
                    keptA[u] = __builtin_amdgcn_perm(e, h[SR - 1], 0x05040100u);
                    keptB[u] = __builtin_amdgcn_perm(e, h[SR - 1], 0x07060302u);
                }
                if (u == ulast && ((cgA == NCG - 1 && actA) || (cgB == NCG - 1 && actB))) {
                    // column ql of this strip's rows: parked in the lane's own scratch lines with the baseline that goes with it,
                    // looked at after the last step (comparing 32 rows here, in the unrolled column code, costs the allocator
                    // hundreds of spilled registers)
                    unsigned *cp = cap + (cgB == NCG - 1 && actB ? 33 : 0); // (one base address, 33 immediate offsets)
#pragma unroll
                    for (int r = 0; r < SR; ++r) cp[r] = h[r];
                    cp[32] = (unsigned)(cgB == NCG - 1 && actB ? base_b : base_a);
                }
                if ((own_last_a && actA) || (own_last_b && actB)) {
                    // the strip that holds row tl: H[tl][j] with its baseline, for the scan of the last row below.  The register row is the
                    // same for the whole pair: a scalar switch (a 32-way select here costs the allocator its spare registers, and
                    // parking all 32 rows made this one lane's wave -- and with it, behind the barrier, the pair -- 132 stores a step slower)
                    const int j = CPS * (own_last_b ? cgB : cgA) + u + 1;
                    if (j <= ql) {
                        unsigned v;
                        switch (rl_s) {
#define MGL_ROW(K) case K: v = h[K < SR ? K : 0]; break;
                            MGL_ROW(0) MGL_ROW(1) MGL_ROW(2) MGL_ROW(3) MGL_ROW(4) MGL_ROW(5) MGL_ROW(6) MGL_ROW(7) MGL_ROW(8) MGL_ROW(9) MGL_ROW(10)
                            MGL_ROW(11) MGL_ROW(12) MGL_ROW(13) MGL_ROW(14) MGL_ROW(15) MGL_ROW(16) MGL_ROW(17) MGL_ROW(18) MGL_ROW(19) MGL_ROW(20)
                            MGL_ROW(21) MGL_ROW(22) MGL_ROW(23) MGL_ROW(24) MGL_ROW(25) MGL_ROW(26) MGL_ROW(27) MGL_ROW(28) MGL_ROW(29) MGL_ROW(30)
#undef MGL_ROW
                        default: v = h[SR - 1]; break;
                        }
                        rowbuf[2 * j] = (int)v;
                        rowbuf[2 * j + 1] = own_last_b ? base_b : base_a;
                    }
                }
            }
        }
        // ---- 4a. the row below a band (NOTB): the group's four columns as one aligned 16-byte piece (column j at entry j - 1)
        if (NOTB && CPS == 4) {
            if (rowsA && actA) {
                int *const dst = rows_raw + (unsigned)(rowoffA + CPS * cgA);
                if (CPS * cgA + CPS <= ql) {
                    reinterpret_cast<int4 *>(dst)[0] = make_int4((int)keptA[0], (int)keptA[1], (int)keptA[2], (int)keptA[3]);
                } else {
#pragma unroll
                    for (int u = 0; u < CPS; ++u)
                        if (CPS * cgA + u + 1 <= ql) dst[u] = (int)keptA[u];
                }
            }
            if (rowsB && actB) {
                int *const dst = rows_raw + (unsigned)(rowoffB + CPS * cgB);
                if (CPS * cgB + CPS <= ql) {
                    reinterpret_cast<int4 *>(dst)[0] = make_int4((int)keptB[0], (int)keptB[1], (int)keptB[2], (int)keptB[3]);
                } else {
#pragma unroll
                    for (int u = 0; u < CPS; ++u)
                        if (CPS * cgB + u + 1 <= ql) dst[u] = (int)keptB[u];
                }
            }
        }
        // ---- 4b. checkpoints (NOTB): a band's strips save the column they have just finished when it is one of the band's checkpoint
        // columns j = STRIP_CK_COLS cc - CPS K band -- strip g reaches it (g mod K) steps after the band's first strip, so the wave
        // takes this branch on K of every 32 steps, a K-th of its lanes each time.  What is saved is the strip's registers as they are --
        // a dword per row, {H, F} -- and its baseline: the walk computes in the same representation (sw_strip_walk.hip)
        if (NOTB) {
            constexpr int PER = STRIP_CK_COLS / CPS;
            const bool ckA = actA && ((cgA + 1 + K * bandA) % PER) == 0, ckB = actB && ((cgB + 1 + K * bandB) % PER) == 0;
            if (__builtin_amdgcn_ballot_w64(ckA || ckB)) {
                if (ckA) {
                    int i0 = i0A;
                    asm volatile("" : "+v"(i0)); // (opaque: the addresses are computed here, K of 32 steps, not hoisted out of the step loop)
                    const int cc = (cgA + 1 + K * bandA) / PER;
                    int *const dst = ck_cols + (size_t)cc * (a.uni_tl + 1) + i0 + 1;
#pragma unroll
                    for (int r = 0; r < SR; ++r)
                        if (i0 + r < tl) dst[r] = (int)__builtin_amdgcn_perm(f[r], h[r], 0x05040100u); // {H, F}: the low halves
                    ck_base[(size_t)cc * ck_strips + GA] = base_a;
                }
                if (ckB) {
                    int i0 = i0B;
                    asm volatile("" : "+v"(i0));
                    const int cc = (cgB + 1 + K * bandB) / PER;
                    int *const dst = ck_cols + (size_t)cc * (a.uni_tl + 1) + i0 + 1;
#pragma unroll
                    for (int r = 0; r < SR; ++r)
                        if (i0 + r < tl) dst[r] = (int)__builtin_amdgcn_perm(f[r], h[r], 0x07060302u); // the high halves
                    ck_base[(size_t)cc * ck_strips + GB] = base_b;
                }
            }
        }
        if (pass > 0 && w == 0) { // the next group of the row above (strip 0 is at column group s + 1 then; behind the query: the last group again)
            const int cgn = min(s + 1, NCG - 1);
            top_v = __builtin_nontemporal_load(reinterpret_cast<const strip_int4 *>(top_raw + CPS * cgn));
            top_b = __builtin_nontemporal_load(top_base + (cgn >> 2));
        }
        // ---- 5. the last lane of every wave posts what it hands on; one barrier per step
        if (L == 63) {
            unsigned *mb = mbox + ((s & 1) * W + w) * 8;
#pragma unroll
            for (int u = 0; u < CPS; ++u) {
                mb[u] = out_h[u];
                mb[4 + u] = out_e[u];
            }
        }
        __syncthreads();
    }

    // ---- last column (sw.cpp:100-104: >= so the later row wins; compare scores, not stored values): every lane looks at the two
    // columns it parked in this pass, then (after the last pass) the candidates meet in one 64-bit LDS atomic (score first, then the larger row)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int i0 = SR * (G0 + (half ? NL : 0) + ell);
        asm volatile("" : "+v"(i0)); // (opaque: the rows' numbers are computed here, not kept in registers -- spilled ones -- from the prologue on)
        if (i0 < tl) {
            const unsigned *cp = cap + (half ? 33 : 0);
            const int bb = (int)__builtin_nontemporal_load(cp + 32);
            for (int r = 0; r < SR; ++r) {
                const unsigned v = __builtin_nontemporal_load(cp + r);
                const int row = i0 + r + 1;
                const int sc = (half ? hi16(v) : lo16(v)) + bb - (row + ql) * gext;
                if (row <= tl && sc >= best) {
                    best = sc;
                    best_i = row;
                }
            }
        }
    }
    __syncthreads(); // (the next pass's first steps must not overtake a wave that is still in this one: the mailboxes)
  } // passes
    if (best_i > 0) atomicMax(key, ((unsigned long long)(unsigned)(best + 0x40000000) << 32) | (unsigned)best_i);
    __threadfence();
    __syncthreads();
    // ---- last row (sw.cpp:116-127), order-free form: best score, among those the smallest |tl - j|, among those the smallest j
    if (w == 0) {
        int rm = NEG_INF, rd = 0x7fffffff, rj = 0x7fffffff;
        for (int j = 1 + L; j <= ql; j += 64) {
            const unsigned v = (unsigned)__builtin_nontemporal_load(rowbuf + 2 * j);
            const int sc = (gl % (2 * NL) >= NL ? hi16(v) : lo16(v)) + __builtin_nontemporal_load(rowbuf + 2 * j + 1) - (tl + j) * gext, d = abs(tl - j);
            const bool take = sc > rm || (sc == rm && (d < rd || (d == rd && j < rj)));
            rm = take ? sc : rm;
            rd = take ? d : rd;
            rj = take ? j : rj;
        }
#pragma unroll
        for (int m = 1; m < 64; m <<= 1) {
            const int om = __shfl_xor(rm, m), od = __shfl_xor(rd, m), oj = __shfl_xor(rj, m);
            const bool take = om > rm || (om == rm && (od < rd || (od == rd && oj < rj)));
            rm = take ? om : rm;
            rd = take ? od : rd;
            rj = take ? oj : rj;
        }
        if (L == 0) {
            const unsigned long long k = key[0];
            const int mqe = (int)(unsigned)(k >> 32) - 0x40000000, mqe_t = (int)(unsigned)k;
            const bool row_wins = rm > mqe || (rm == mqe && rd < abs(mqe_t - ql));
            DpRecord r;
            r.mqe = mqe;
            r.mqe_t = mqe_t;
            r.max = row_wins ? rm : mqe;
            r.max_t = row_wins ? tl : mqe_t;
            r.max_q = row_wins ? rj : ql;
            r.seg = row_wins ? ql - rj : 0;
            r.g_tail = NOTB ? -(200 + W) : -(100 + W);
            r.sps = steps_cap;
            a.rec[slot] = r;
        }
    }
    (void)wres;
}

} // namespace

// grid = pairs of the chunk, block = 64 * W threads (W <= 4), dynamic LDS = strip16_lds_bytes.  Rows per strip: 32, or fewer when
// that still covers the longest target with the same number of waves (10 000 rows on 384 strip slots: 27 rows each instead of
// 313 strips of 32 -- a sixth fewer instructions per column)
// Waves per SIMD.  Three (168 registers) up to 26 rows per strip; TWO (256 registers) from 27 rows on: at three, the 32-row kernel spilled
// inside the column code (16 kb pairs: 99.5 ms per 1 536 pairs against 90.7 with two waves per SIMD; 10 kb pairs, 20 rows: the same either
// way), and from 27 rows per strip on (targets of 13 313 rows and more) a pair's query tables leave LDS for two workgroups per CU anyway
// (14 kb pairs, 28 rows: 4 524 GCUPS at two against 4 340 at three; 13 kb, 26 rows, three workgroups per CU: 4 581 at three against
// 4 350 at two).  -DMGL_STRIP_OCC2_FROM=N (scripts/build_variant.sh) moves the border.
#define MGL_STRIP_OCC(ROWS) ((ROWS) >= MGL_STRIP_OCC2_FROM ? 2 : 3)

// every byte of the pair's target one of ACGT (2-bit packed inputs: by construction)?  The whole workgroup calls this together.
__device__ __forceinline__ bool strip_target_is_acgt(const DpArgs &a)
{
    if (a.t.packed2) return true;
    const int64_t p = a.first + blockIdx.x, t0 = a.t.off[p];
    const int tl = a.t.length(p);
    unsigned bad = 0u;
    for (int x = threadIdx.x; x < tl; x += blockDim.x) {
        unsigned diff;
        ascii_codes((unsigned)a.t.data[t0 + x], diff);
        bad |= diff & 0xffu;
    }
    return __syncthreads_or((int)bad) == 0;
}

#define MGL_STRIP_KERNEL(NAME, ROWS)                                                     \
    __global__ __launch_bounds__(256, MGL_STRIP_OCC(ROWS)) void NAME(const DpArgs a)     \
    {                                                                                    \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];             \
        sw_dp16_strip_body<ROWS, false, false>(a, smem);                                 \
    }                                                                                    \
    __global__ __launch_bounds__(256, MGL_STRIP_OCC(ROWS)) void NAME##_ck(const DpArgs a) \
    {                                                                                    \
        extern __shared__ __attribute__((aligned(16))) unsigned char smem[];             \
        if (a.strip_codes && strip_target_is_acgt(a))                                    \
            sw_dp16_strip_body<ROWS, true, true>(a, smem);                               \
        else                                                                             \
            sw_dp16_strip_body<ROWS, true, false>(a, smem);                              \
    }
MGL_STRIP_KERNEL(sw_dp16_strip_kernel, 32)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r31, 31)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r30, 30)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r29, 29)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r28, 28)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r27, 27)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r26, 26)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r25, 25)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r24, 24)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r23, 23)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r22, 22)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r21, 21)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r20, 20)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r19, 19) // (round 4: targets of 8 193 .. 9 728 rows on the 512 strip slots of four waves -- with 20 rows
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r18, 18) // at least, 8 300 rows filled 415 of them: 3 835 GCUPS between 4 640 at 8 150 and 4 550 at 10 000)
MGL_STRIP_KERNEL(sw_dp16_strip_kernel_r17, 17)
#undef MGL_STRIP_KERNEL

int strip16_lds_bytes(int max_ql, int waves) { return strip16_qwords(max_ql) * 4 + 2 * waves * 8 * 4 + 8 + 16 + 64; }
// ... with the query as one table per column (the CODES form of the kernels without stored flags)
int strip16_lds_bytes_codes(int max_ql, int waves) { return strip16_table_words(max_ql) * 4 + 2 * waves * 8 * 4 + 8 + 16 + 64; }
int strip16_waves_per_simd(int rows) { return MGL_STRIP_OCC(rows); }

// The static window of a strip (see the header): 31 rows of one column, 16 + 4 columns of drift, the intermediates of a cell.
bool strip16_range_ok(int match, int mismatch, int gopen, int gext)
{
    if (match <= 0 || mismatch > match || gext < 0 || gopen < gext || match > 4000 || mismatch < -4000 || gopen > 4000 || gext > 4000) return false;
    const int up = match + gopen + gext, down = gopen - gext, mis2 = mismatch + 2 * gext;
    const int above = 31 * up + 20 * up + gopen + (match + 2 * gext) + 64;                     // over the first row's level
    const int below = 31 * down + 20 * up + up + down + (mis2 < 0 ? -mis2 : mis2) + 64;        // under it
    return STRIP_LEVEL + above <= 32767 && STRIP_LEVEL - below >= -32768;
}

hipError_t launch_dp16_strip(const DpArgs &a, int waves, int rows, hipStream_t stream)
{
    const int lds = a.strip_codes ? strip16_lds_bytes_codes(a.uni_ql, waves) : strip16_lds_bytes(a.uni_ql, waves);
    if (lds > 64 * 1024 || (a.strip_codes && a.strip_k == 0)) return hipErrorInvalidValue; // (the host checks)
    static void (*const table[16])(const DpArgs) = {sw_dp16_strip_kernel_r17, sw_dp16_strip_kernel_r18, sw_dp16_strip_kernel_r19, sw_dp16_strip_kernel_r20, sw_dp16_strip_kernel_r21, sw_dp16_strip_kernel_r22, sw_dp16_strip_kernel_r23,
                                                    sw_dp16_strip_kernel_r24, sw_dp16_strip_kernel_r25, sw_dp16_strip_kernel_r26, sw_dp16_strip_kernel_r27,
                                                    sw_dp16_strip_kernel_r28, sw_dp16_strip_kernel_r29, sw_dp16_strip_kernel_r30, sw_dp16_strip_kernel_r31,
                                                    sw_dp16_strip_kernel};
    static void (*const table_ck[16])(const DpArgs) = {sw_dp16_strip_kernel_r17_ck, sw_dp16_strip_kernel_r18_ck, sw_dp16_strip_kernel_r19_ck, sw_dp16_strip_kernel_r20_ck, sw_dp16_strip_kernel_r21_ck, sw_dp16_strip_kernel_r22_ck, sw_dp16_strip_kernel_r23_ck,
                                                       sw_dp16_strip_kernel_r24_ck, sw_dp16_strip_kernel_r25_ck, sw_dp16_strip_kernel_r26_ck, sw_dp16_strip_kernel_r27_ck,
                                                       sw_dp16_strip_kernel_r28_ck, sw_dp16_strip_kernel_r29_ck, sw_dp16_strip_kernel_r30_ck, sw_dp16_strip_kernel_r31_ck,
                                                       sw_dp16_strip_kernel_ck};
    if (rows < 17 || rows > 32) return hipErrorInvalidValue;
    if (a.strip_k > 0 && a.strip_k != 64 / rows) return hipErrorInvalidValue; // (a constant in the kernels)
    void (*k)(const DpArgs) = a.strip_k > 0 ? table_ck[rows - 17] : table[rows - 17];
    hipLaunchKernelGGL(k, dim3((unsigned)a.count), dim3(64 * waves), lds, stream, a);
    return hipGetLastError();
}

} // namespace mgl_sw_dev
