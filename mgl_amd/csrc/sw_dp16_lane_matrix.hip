// sw_dp16_lane_matrix.hip -- substitution-matrix ("protein") scoring with TWO PAIRS PER LANE, for tiles of 128 pairs that share
// their TARGET (a database search: one database sequence against many queries; MGL_SW_FLAG_SHARED_TARGET).  No reference path exists
// (mgl scores by byte equality, sw.cpp:55): the function is sw_dp16_matrix_kernel's -- the recurrence, borders,
// overhang strategies and traceback of sw.cpp:5-255 with diag = H[i-1][j-1] + matrix[code[t]][code[q]] -- and the results are
// identical to that kernel's and to the CPU restatement's extension (tests/test_gpu_matrix.py).
//
// Why another mapping.  sw_dp16_matrix_kernel (16 lanes share a pair, anti-diagonal wavefront) looks its two scores up per STEP: two
// gathers out of a table in LDS, 64 lanes at random (five to a bank), 23.6 VALU instructions per two cells, and 12 bytes of LDS per
// query residue keep nine waves on a CU (profiles/r05_b_protein_pmc.txt: both pipes at two thirds, 2.08 TCUPS).  Give every lane its
// own two pairs (sw_dp16_lane.hip: a strip of 32 target rows in registers, the query's columns swept one by one, nothing crosses
// lanes) and let the 128 pairs of a wave share their target: then the 32 rows of a strip are the SAME 32 residues for every lane, and
// what a lane needs for a column -- the scores of its query residue against those 32 rows -- is one row of a 32 x 32-byte table that
// the wave builds once per strip (the STRIP PROFILE: SP[code][row] = S[t_row][code] + e + o).  A column costs a lane four 16-byte
// LDS reads (two pairs x 32 bytes) instead of 64 gathers, and a cell's score is one v_perm_b32 that picks the row's byte out of both
// pairs' dwords into the two halves.
//
// The table's bias is free.  Bytes are unsigned; a score can be negative.  The entries hold S + 2e + (o - e): the kernel keeps, per
// row, not H[i][j-1] but H[i][j-1] - (o - e) -- the value "a new gap from here" that the recurrence computes anyway -- so the
// diagonal of the next column is   (H - (o - e)) + (S + 2e + (o - e)) = H + S + 2e   in ONE add, where the byte-compare kernels spend
// a v_pk_mad and an add.  Needs 0 <= S + e + o <= 255 for every entry (BLOSUM62 11/1: 8 .. 23); the host checks, everything else
// takes sw_dp16_matrix_kernel.
//
// A PERSISTENT grid (as sw_dp16_lane_ck.hip's): a wave keeps ONE region -- the traceback flags of a tile, [strip][column][R/16][lane]
// uint4 as sw_dp16_lane.hip stores them, and the carry row between strips -- walks its 128 paths itself right behind the fill
// (sw_traceback.h, the flags are the lane's own stores) and takes the next tile off a counter.  The tiles are drawn LARGEST FIRST
// (DpArgs::tile_order: the host has looked at every tile's geometry -- launch_tile_geometry, 8 bytes per tile back over the link -- and
// sorted them by what they keep), wave slot s starts on the s-th largest tile and its region is sized for exactly that one
// (DpArgs::slot_off): whatever a slot draws later needs less.  So the workspace is the sum over the S largest tiles, not S times
// the largest (a protein database's lengths are log-normal: 41 GB instead of 117 for 3 072 slots on the bench's), and the queue ends on
// its shortest tiles.  A tile whose pairs do not share one target and one query length breaks the caller's promise: its pairs get
// MGL_SW_ERR_BAD_ARG in the status array and the call's status word, nothing is computed for them (never a wrong result).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "sw_device.h"
#include "sw_lane_cell.h"
#include "sw_traceback.h"

namespace mgl_sw_dev {

namespace {

constexpr int LM_R = 32;               // rows per strip
constexpr int LM_LDS_CODE = 0;         // byte -> code, 256 bytes
constexpr int LM_LDS_M = 256;          // the matrix, MATRIX_DIM x MATRIX_DIM int8
constexpr int LM_LDS_SP = LM_LDS_M + MATRIX_DIM * MATRIX_DIM; // the strip profile, MATRIX_DIM codes x LM_R rows, one byte each
constexpr int LM_LDS_T = LM_LDS_SP + MATRIX_DIM * LM_R;       // the tile's target as codes, uni_tl bytes rounded up to 16

// R rows of one column for both packed pairs.  ho[r]: H[row r][j-1] - (o - e) on entry, H[row r][j] - (o - e) on exit; f[r]: F of row r;
// hdo: H[row -1][j-1] - (o - e) (the strip's top row, previous column); e: E coming down from the strip above (in), E leaving the last
// row (out); hlast: H of the last row (out).  sa / sb: the column's 32 score bytes of pair A / pair B, row r in byte r & 3 of dword
// r >> 2; sel[k] picks byte k of both into the halves.  The flags are column<R, false>()'s (sw_lane_cell.h), bit for bit.
// LAST: hlast = H of register row rl (the target's last row; the rows below it are computed and thrown away).
// NOTB (MGL_SW_FLAG_SCORE_ONLY): no flags are formed or stored -- 7 of the 16 instructions of a step are left.
template <int R, bool LAST, bool NOTB>
__device__ __forceinline__ void column_sp(unsigned (&ho)[R], unsigned (&f)[R], const uint4 (&sa)[R / 16], const uint4 (&sb)[R / 16], const unsigned hdo,
                                          unsigned &e, unsigned &hlast, const LaneConsts &c, const unsigned (&sel)[4], uint4 *tbp, const int rl)
{
    unsigned w[4];
    auto word = [&](const uint4 (&s)[R / 16], const int r) {
        const uint4 &v = s[r >> 4];
        const int k = (r >> 2) & 3;
        return k == 0 ? v.x : k == 1 ? v.y : k == 2 ? v.z : v.w;
    };
    auto score = [&](const int r) { return __builtin_amdgcn_perm(word(sb, r), word(sa, r), sel[r & 3]); };
    unsigned dg = pk_add(hdo, score(0));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned diag = dg;
        if (r + 1 < R) dg = pk_add(ho[r], score(r + 1)); // (before row r overwrites ho[r]: the diagonal of row r + 1)
        const unsigned fr = f[r];
        const unsigned sm = pk_max(diag, fr);
        const unsigned hn = pk_max(sm, e);
        const unsigned open = pk_sub(hn, c.o_e);
        const unsigned eo = pk_max(open, e);
        const unsigned fo = pk_max(open, fr);
        if (!NOTB) {
            const unsigned d1 = pk_sub_sat(diag, fr);
            const unsigned d2 = pk_sub_sat(sm, e);
            const unsigned d3 = pk_sub_sat(e, open);
            const unsigned d4 = pk_sub_sat(fr, open);
            const unsigned p12 = __builtin_amdgcn_perm(d1, d2, 0x0b0a0908u);
            const unsigned p34 = __builtin_amdgcn_perm(d3, d4, 0x0b0a0908u);
            const int U = r & 3;
            const unsigned low = U == 0 ? 0u : w[(r >> 2) & 3];
            w[(r >> 2) & 3] = and_or(p34, c.k34[U], U == 0 ? (p12 & c.k12[0]) : and_or(p12, c.k12[U], low));
            if ((r & 15) == 15) tbp[(size_t)(r >> 4) * 64] = make_uint4(w[0], w[1], w[2], w[3]);
        }
        ho[r] = open;
        f[r] = fo;
        e = eo;
        // (picked here, off the value itself: a chain of selects over ho[] after the loop is turned into a load of ho[rl] -- and the whole
        // array into scratch, a load and a store per row)
        if (LAST ? r == 0 : r == R - 1) hlast = hn;
        if (LAST && r > 0) hlast = (r == rl) ? hn : hlast;
        asm volatile("" : "+v"(f[r]), "+v"(ho[r]));
    }
}

template <int R, bool LAST, bool NOTB>
__device__ __forceinline__ void lm_strip(const int i0, const int tl, const int ql, uint2 *bnd, const unsigned *qst, const unsigned char *sp, uint4 *&tbp,
                                         const LaneConsts &c, const unsigned (&sel)[4], const int gopen, const int gext, const int base, const bool indel,
                                         int &bestA, int &bestA_i, int &bestB, int &bestB_i)
{
    unsigned ho[R], f[R];
    // ---- column 0: H[i][0] border values, F[i][1] = H[i][0] - o (sw.cpp:24,38,47-49), in stored form: both are H - (o - e)
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = i0 + r + 1;
        const int hb = border(row, gopen, gext, indel) + row * gext + base;
        ho[r] = f[r] = pk_sub(pack2(hb, hb), c.o_e);
    }
    const int hd0 = border(i0, gopen, gext, indel) + i0 * gext + base; // H[i0][0]
    unsigned hdo = pk_sub(pack2(hd0, hd0), c.o_e);
    const int rl = tl - 1 - i0; // LAST: register row of target row tl (0 .. R-1)

    uint2 *bp = bnd + 64; // column j
    auto scores_of = [&](const unsigned codes_a, const unsigned codes_b, const int k, uint4 (&sa)[R / 16], uint4 (&sb)[R / 16]) {
        const uint4 *pa = reinterpret_cast<const uint4 *>(sp + ((codes_a >> (8 * k)) & 0xffu) * R);
        const uint4 *pb = reinterpret_cast<const uint4 *>(sp + ((codes_b >> (8 * k)) & 0xffu) * R);
#pragma unroll
        for (int x = 0; x < R / 16; ++x) {
            sa[x] = pa[x];
            sb[x] = pb[x];
        }
    };
    auto one_column = [&](const uint2 top, const uint4 (&sa)[R / 16], const uint4 (&sb)[R / 16]) {
        unsigned e = top.y, hlast;
        column_sp<R, LAST, NOTB>(ho, f, sa, sb, hdo, e, hlast, c, sel, tbp, rl);
        if (!NOTB) tbp += (R / 16) * 64;
        hdo = pk_sub(top.x, c.o_e);
        bp[0] = make_uint2(hlast, LAST ? 0u : e);
        bp += 64;
    };
    // columns 1 .. ql, four at a time (one dword of codes per query): the group's loads at its top (sw_dp16_lane.hip has the reasons for
    // the straight-line group)
    int j = 1;
    for (; j + 3 <= ql; j += 4) {
        const uint2 top0 = bp[0], top1 = bp[64], top2 = bp[128], top3 = bp[192];
        const unsigned qa = qst[0], qb = qst[64];
        qst += 128;
        uint4 sa0[R / 16], sb0[R / 16], sa1[R / 16], sb1[R / 16];
        // (a column's scores are read at its top: reading them a column ahead -- 16 more registers -- measured 1 % faster, within the noise)
        scores_of(qa, qb, 0, sa0, sb0);
        one_column(top0, sa0, sb0);
        scores_of(qa, qb, 1, sa1, sb1);
        one_column(top1, sa1, sb1);
        scores_of(qa, qb, 2, sa0, sb0);
        one_column(top2, sa0, sb0);
        scores_of(qa, qb, 3, sa1, sb1);
        one_column(top3, sa1, sb1);
    }
    if (j <= ql) { // the last one to three columns
        const unsigned qa = qst[0], qb = qst[64];
        for (int k = 0; j <= ql; ++j, ++k) {
            const uint2 top = bp[0];
            uint4 sa0[R / 16], sb0[R / 16];
            scores_of(qa >> (8 * k), qb >> (8 * k), 0, sa0, sb0);
            one_column(top, sa0, sb0);
        }
    }
    // ---- last column of the strip's rows (sw.cpp:100-104: >= so the later row wins)
    const int oe = gopen - gext;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int row = i0 + r + 1;
        if (row <= tl) {
            const int unshift = oe - (row + ql) * gext - base;
            const int ca = lo16(ho[r]) + unshift, cb = hi16(ho[r]) + unshift;
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

template <bool NOTB>
__device__ __forceinline__ void lm_body(const DpArgs &a, const TbArgs &walk, unsigned char *smem)
{
    constexpr int R = LM_R;
    const int lane = threadIdx.x;
    const int64_t tiles = (a.count + 127) >> 7, slots = gridDim.x, slot = blockIdx.x;
    unsigned long long *const ctr = reinterpret_cast<unsigned long long *>(a.tile_ctr); // {draws, waves out}: ONE object (sw_dp16_lane_ck.hip)

    // ---- once per wave: the code table and the matrix into LDS
    reinterpret_cast<unsigned *>(smem + LM_LDS_CODE)[lane] = reinterpret_cast<const unsigned *>(a.code)[lane];
#pragma unroll
    for (int x = 0; x < MATRIX_DIM * MATRIX_DIM / 256; ++x)
        reinterpret_cast<unsigned *>(smem + LM_LDS_M)[x * 64 + lane] = reinterpret_cast<const unsigned *>(a.matrix)[x * 64 + lane];
    const unsigned char *const code_of = smem + LM_LDS_CODE;
    const signed char *const mat = reinterpret_cast<const signed char *>(smem + LM_LDS_M);
    unsigned char *const sp = smem + LM_LDS_SP;
    unsigned char *const tcode = smem + LM_LDS_T;

    const int match = a.match, gopen = a.gopen, gext = a.gext;
    const bool indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;
    LaneConsts c;
    c.delta = c.one = c.k2 = 0u;
    c.o_e = pack2(gopen - gext, gopen - gext);
    asm volatile("" : "+v"(c.o_e));
    unsigned sel[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        c.k12[u] = 0x02020202u << (2 * u);
        c.k34[u] = 0x01010101u << (2 * u);
        sel[u] = 0x0c040c00u + 0x00010001u * (unsigned)u;
        asm volatile("" : "+s"(c.k12[u]), "+s"(c.k34[u]), "+s"(sel[u]));
    }
    const int sp_bias = gopen + gext; // an entry = S + 2e + (o - e)

    // the slot's region: the tile's traceback words, then the carry row and the two queries as codes (laid out per tile: its geometry)
    unsigned char *const region = reinterpret_cast<unsigned char *>(a.tb) + a.slot_off[slot];
    const int64_t region_cap = a.slot_off[slot + 1] - a.slot_off[slot];

    for (int64_t draw = slot; draw < tiles;) {
        const int64_t tile = a.tile_order[draw];
        const int64_t p0 = a.first + tile * 128;
        const int cnt = (int)min((int64_t)128, a.count - tile * 128);
        const bool lvalid = 2 * lane < cnt, validB = 2 * lane + 1 < cnt;
        const int64_t pA = p0 + (lvalid ? 2 * lane : cnt - 1), pB = validB ? pA + 1 : pA;
        // ---- the promise: one target, one query length
        const int tl = __builtin_amdgcn_readfirstlane(a.t.length(p0)), ql = __builtin_amdgcn_readfirstlane(a.q.length(p0));
        const int64_t t0 = a.t.off[p0];
        const unsigned t0lo = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)t0), t0hi = __builtin_amdgcn_readfirstlane((unsigned)((unsigned long long)t0 >> 32));
        const int64_t tstart = (int64_t)((unsigned long long)t0lo | (unsigned long long)t0hi << 32);
        const bool mine = a.t.off[pA] == tstart && a.t.off[pB] == tstart && a.t.length(pA) == tl && a.t.length(pB) == tl && a.q.length(pA) == ql && a.q.length(pB) == ql;
        const bool fits = tl >= 1 && ql >= 1 && tl <= a.uni_tl && ql <= a.uni_ql && lane16_matrix_region_bytes(tl, ql, NOTB) <= region_cap;
        unsigned char *const wave_scratch = region + lane16_matrix_tb_bytes(tl, ql, NOTB);
        uint2 *const bnd = reinterpret_cast<uint2 *>(wave_scratch) + lane;
        unsigned *const qst = reinterpret_cast<unsigned *>(wave_scratch + (size_t)lane_bnd_entries(ql) * 8) + lane;
        uint32_t *const tb_region = reinterpret_cast<uint32_t *>(region);
        if (__builtin_amdgcn_ballot_w64(!mine) != 0ull || !fits) {
            const int bad = 1; // MGL_SW_ERR_BAD_ARG
            if (lvalid) {
                const int64_t oA = walk.dest ? walk.dest[pA] : pA, oB = walk.dest ? walk.dest[pB] : pB;
                if (walk.status) {
                    walk.status[oA] = bad;
                    if (validB) walk.status[oB] = bad;
                }
                if (walk.cigar_len) {
                    walk.cigar_len[oA] = 0;
                    if (validB) walk.cigar_len[oB] = 0;
                }
            }
            if (lane == 0 && walk.status_any) atomicMax(walk.status_any, bad);
        } else {
            const int base = dp16_base(tl, ql, match, gext);
            const int strips = lane_strips(tl, R);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // (the previous tile's readers of tcode / sp are through; the tables of the kernel's first lines are written)
            __builtin_amdgcn_wave_barrier();
            // ---- the target as codes (every lane's target), the two queries as codes [4-residue block][A | B][lane], the border row
            for (int x = lane; x < strips * R; x += 64) tcode[x] = x < tl ? code_of[a.t.data[tstart + x]] : (unsigned char)0;
            {
                SeqWords sa, sb;
                sa.init(a.q.data + a.q.off[pA], ql);
                sb.init(a.q.data + a.q.off[pB], ql);
                unsigned loA = sa.word(0), loB = sb.word(0);
                auto codes4 = [&](const unsigned w) {
                    return (unsigned)code_of[w & 0xffu] | (unsigned)code_of[(w >> 8) & 0xffu] << 8 | (unsigned)code_of[(w >> 16) & 0xffu] << 16 | (unsigned)code_of[w >> 24] << 24;
                };
                for (int cb = 0; cb < (ql + 3) >> 2; ++cb) {
                    qst[(size_t)(2 * cb) * 64] = codes4(sa.next_block(cb, loA));
                    qst[(size_t)(2 * cb + 1) * 64] = codes4(sb.next_block(cb, loB));
                }
                for (int j = 0; j <= ql; ++j) {
                    const int hb0 = border(j, gopen, gext, indel) + j * gext + base;
                    const unsigned hp = pack2(hb0, hb0);
                    bnd[(size_t)j * 64] = make_uint2(hp, pk_sub(hp, c.o_e));
                }
            }
            uint4 *tbp = reinterpret_cast<uint4 *>(tb_region) + lane;
            int bestA = NEG_INF, bestA_i = -1, bestB = NEG_INF, bestB_i = -1;
            for (int k = 0; k < strips; ++k) {
                // ---- the strip profile: lane l writes the 16 rows (l & 1) * 16 .. + 15 of code l >> 1
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                {
                    const int code = lane >> 1, r0 = (lane & 1) * 16;
                    unsigned w[4];
#pragma unroll
                    for (int x = 0; x < 4; ++x) {
                        unsigned v = 0;
#pragma unroll
                        for (int y = 0; y < 4; ++y) {
                            const int tc = tcode[k * R + r0 + 4 * x + y];
                            v |= ((unsigned)((int)mat[tc * MATRIX_DIM + code] + sp_bias) & 0xffu) << (8 * y);
                        }
                        w[x] = v;
                    }
                    *reinterpret_cast<uint4 *>(sp + code * R + r0) = make_uint4(w[0], w[1], w[2], w[3]);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                if (k < strips - 1)
                    lm_strip<R, false, NOTB>(k * R, tl, ql, bnd, qst, sp, tbp, c, sel, gopen, gext, base, indel, bestA, bestA_i, bestB, bestB_i);
                else
                    lm_strip<R, true, NOTB>(k * R, tl, ql, bnd, qst, sp, tbp, c, sel, gopen, gext, base, indel, bestA, bestA_i, bestB, bestB_i);
            }
            // ---- last row (sw.cpp:116-127), order-free form (sw_dp16_lane.hip)
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
                if (half ? validB : lvalid) {
                    DpRecord r;
                    r.mqe = mqe;
                    r.mqe_t = mqe_t;
                    r.max = row_wins ? rm : mqe;
                    r.max_t = row_wins ? tl : mqe_t;
                    r.max_q = row_wins ? rj : ql;
                    r.seg = row_wins ? ql - rj : 0;
                    r.g_tail = 0;
                    r.sps = R;
                    if (NOTB) { // scores only: offset 0, an empty CIGAR (sw_scores_only_kernel)
                        const int64_t p = half ? pB : pA, o = walk.dest ? walk.dest[p] : p;
                        walk.offset[o] = 0;
                        if (walk.cigar_len) walk.cigar_len[o] = 0;
                        if (walk.status) walk.status[o] = 0;
                        if (walk.score) {
                            Score sc;
                            sc.mqe = r.mqe;
                            sc.mqe_t = r.mqe_t;
                            sc.max = r.max;
                            sc.max_t = r.max_t;
                            sc.max_q = r.max_q;
                            sc.seg_length = r.seg;
                            walk.score[o] = sc;
                        }
                    } else {
                        // the lane walks the paths of its own two pairs right here: the flags are its own stores
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        TbView view;
                        view.base = tb_region;
                        view.set_schedule(r, ql, R);
                        view.packed16 = 2;
                        view.half = half;
                        view.lane = lane;
                        view.ql = ql;
                        traceback_one_pair(walk, view, r, half ? pB : pA, tl, ql);
                    }
                }
            }
        }
        if (tiles <= slots) break;
        unsigned next = 0;
        if (lane == 0) {
            next = (unsigned)__hip_atomic_fetch_add(ctr, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (next >= (unsigned)tiles && a.grid_fault) __hip_atomic_store(a.grid_fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        draw = slots + (int64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)next);
    }
    if (tiles > slots && lane == 0) { // the last wave out zeroes the counter (sw_dp16_lane_ck.hip)
        const unsigned out = (unsigned)(__hip_atomic_fetch_add(ctr, 1ull << 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> 32);
        if (out == (unsigned)slots - 1u)
            __hip_atomic_store(ctr, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (out >= (unsigned)slots && a.grid_fault)
            __hip_atomic_store(a.grid_fault, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

} // namespace

// -DMGL_LM_WAVES=2 (scripts/build_variant.sh): two waves per SIMD, 256 registers -- a measurement build
#ifndef MGL_LM_WAVES
#define MGL_LM_WAVES 3
#elif !defined(MGL_VARIANT_BUILD)
#error "MGL_LM_WAVES is a measurement switch (scripts/build_variant.sh defines MGL_VARIANT_BUILD): never the shipped library"
#endif
__global__ __launch_bounds__(64, MGL_LM_WAVES) void sw_dp16_lane_matrix_kernel(const DpArgs a, const TbArgs walk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lm_body<false>(a, walk, smem);
}

// MGL_SW_FLAG_SCORE_ONLY: the same fill without flags, regions or walk (a database search's pre-filter)
__global__ __launch_bounds__(64, MGL_LM_WAVES) void sw_dp16_lane_matrix_score_kernel(const DpArgs a, const TbArgs walk)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    lm_body<true>(a, walk, smem);
}

// {tl, ql} of every tile's first pair (the kernel checks that the tile's other pairs agree)
__global__ __launch_bounds__(256) void sw_tile_geometry_kernel(const SeqSet t, const SeqSet q, const int64_t first, const int64_t count, int32_t *out)
{
    const int64_t tile = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (tile * 128 >= count) return;
    out[2 * tile] = t.length(first + tile * 128);
    out[2 * tile + 1] = q.length(first + tile * 128);
}
hipError_t launch_tile_geometry(const SeqSet &t, const SeqSet &q, int64_t first, int64_t count, int32_t *out, hipStream_t stream)
{
    const int64_t tiles = (count + 127) / 128;
    hipLaunchKernelGGL(sw_tile_geometry_kernel, dim3((unsigned)((tiles + 255) / 256)), dim3(256), 0, stream, t, q, first, count, out);
    return hipGetLastError();
}

// the table's bias is the gap's: every S + e + o must be a byte (above)
bool lane16_matrix_params_ok(int smin, int smax, int gopen, int gext) { return gext >= 0 && gopen >= gext && smin + gext + gopen >= 0 && smax + gext + gopen <= 255; }
int lane16_matrix_lds_bytes(int max_tl) { return LM_LDS_T + (lane_strips(max_tl, LM_R) * LM_R + 15) / 16 * 16; }

// a.lane_slots regions at a.tb + a.slot_off[s] (a.slot_off: lane_slots + 1 entries), the tiles drawn in the order a.tile_order;
// a.tile_ctr (zero) where the tiles outnumber the slots; walk.cigar set (or a.score_only): the waves walk their own paths
hipError_t launch_dp16_lane_matrix(const DpArgs &a, const TbArgs &walk, hipStream_t stream)
{
    const int64_t tiles = (a.count + 127) / 128;
    if (a.lane_slots < 1 || !a.matrix || !a.code || !a.tile_order || !a.slot_off || (!walk.cigar && !a.score_only) || a.t.packed2 || a.q.packed2 || (tiles > a.lane_slots && !a.tile_ctr)) return hipErrorInvalidValue;
    const int lds = lane16_matrix_lds_bytes(a.uni_tl);
    auto kernel = a.score_only ? sw_dp16_lane_matrix_score_kernel : sw_dp16_lane_matrix_kernel;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3((unsigned)std::min<int64_t>(tiles, a.lane_slots)), dim3(64), lds, stream, a, walk);
    return hipGetLastError();
}

} // namespace mgl_sw_dev
