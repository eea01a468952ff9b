// sw_dp16.hip -- packed-int16 fill kernel for gfx950: TWO pairs per lane (low / high 16 bits of every
// register), eight pairs per wave64.  Same function as sw_dp_kernel (sw_kernels.hip; the reference's
// sw.cpp:5-146), selected by the host layer when a batch has one geometry (every pair the same tl and
// ql) and the score range fits 16 bits; everything else goes through the int32 kernel.
//
// Why it is faster: sw_dp_kernel is VALU-issue bound (profiles/r01_a_first_path.txt, the ablations in
// DESIGN.md section 6); v_pk_{add,sub,max}_i16 cost the same issue slot as their 32-bit forms and do
// two cells.
//
// 16-bit representation.  A cell value is stored as
//        v = X[i][j] + (i + j)*e + BASE               (X = H, E or F;  e = gap extend)
// i.e. a per-cell offset that is linear in (i, j).  Consequences:
//   * extending a gap costs NO instruction in either direction: E moves one row down, F one column right, and
//     both become  max(H - (o-e), E or F)  with ONE shared  H - (o-e);
//   * diag = H[i-1][j-1] + s(t,q)  becomes  v_hup + (match+2e) + m*(mismatch-match)  with m = min(t^q, 1): one
//     v_pk_mad_i16 off the dependency chain plus one add -- no compare/select (there is no packed compare on CDNA4);
//   * H <= match*min(i,j), so v <= BASE + match*min(tl,ql) + e*(tl+ql), and X >= -2o - (i+j-2)*e (the all-gap path
//     from a border), so v >= BASE - 2o + 2e: the span is match*min(tl,ql) + e*(tl+ql) + 3o + |mismatch|, checked by
//     the host (dp16_range_ok()); BASE puts the top of that span at +32767;
//   * comparisons between cells of the same (i,j) are unaffected, so all decisions -- hence the
//     traceback -- are bit-identical to the int32 kernel; true scores are recovered where they are
//     read (last column, last row).
// Flags are sign bits of saturating differences (v_pk_sub_i16 clamp keeps the sign right over the whole
// 16-bit range).  Garbage lanes (columns > ql, rows > tl) may wrap; they never feed a valid cell.
//
// Traceback layout ("packed16"): per group (= 2 pairs) one dword per lane per 4 steps,
//   byte0 = pair A {E>S, F opened}, byte1 = pair B {same}, byte2 = pair A {F>diag, E opened}, byte3 = pair B,
//   step t of the 4 in bits (2t+1, 2t).  Two such dwords are stored together (8 steps), so a group
//   writes one full 128-byte line at a time (64-byte stores cost twice the HBM write traffic: PMC
//   WRITE_SIZE of the first packed build, profiles/r01_b_packed16.txt).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <algorithm>

#include "sw_device.h"

namespace mgl_sw_dev {

namespace {

constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int RING_SLACK16 = 24, QQ_SLACK16 = 48;

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned row_shr1(unsigned lane0_value, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)lane0_value, (int)src, DPP_ROW_SHR1, 0xf, 0xf, false);
}
__device__ __forceinline__ short2_t as_s2(unsigned x) { return __builtin_bit_cast(short2_t, x); }
__device__ __forceinline__ unsigned as_u(short2_t x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ ushort2_t as_us2(unsigned x) { return __builtin_bit_cast(ushort2_t, x); }
// wrapping 16-bit arithmetic on the unsigned type: the bits are those of the signed operation, and lanes that hold no
// valid cell (columns > ql, rows > tl) may wrap without that being undefined behaviour
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, as_us2(a) + as_us2(b)); }
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, as_us2(a) - as_us2(b)); }
__device__ __forceinline__ unsigned pk_sub_sat(unsigned a, unsigned b)
{
    return as_u(__builtin_elementwise_sub_sat(as_s2(a), as_s2(b)));
}
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b)
{
    return as_u(__builtin_elementwise_max(as_s2(a), as_s2(b)));
}
__device__ __forceinline__ unsigned pk_min_u(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, a),
                                                                  __builtin_bit_cast(ushort2_t, b)));
}
__device__ __forceinline__ unsigned pk_mad(unsigned a, unsigned b, unsigned c)
{
    return __builtin_bit_cast(unsigned, as_us2(a) * as_us2(b) + as_us2(c));
}
__device__ __forceinline__ unsigned pack2(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
__device__ __forceinline__ int lo16(unsigned x) { return (int)(short)(x & 0xffffu); }
__device__ __forceinline__ int hi16(unsigned x) { return (int)x >> 16; }

__device__ __forceinline__ int border(int k, int gopen, int gext, bool indel)
{
    return (indel && k > 0) ? -gopen - (k - 1) * gext : 0; // sw.cpp:29-40,47-49
}

struct Lane16 {
    unsigned h_prev, e_prev, hup, f; // packed A|B, offset representation
    unsigned acc;                    // traceback flags of the current 4-step block
    unsigned cap;                    // H of the last column (current stripe), packed
};

struct Consts16 {
    unsigned delta, one, o_e, k2; // packed constants (both halves equal): mismatch-match, 1, o-e, match+2e
};

// mask bit set ? b : a   (mask is a wave-uniform 64-bit lane mask held in SGPRs: no VALU compare)
__device__ __forceinline__ unsigned sel_mask(unsigned a, unsigned b, unsigned long long mask)
{
    unsigned r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(mask));
    return r;
}

// The arithmetic of one anti-diagonal step for both packed pairs: returns H, E', F' of the cell and
// shifts the four decision flags into st.acc.  rh / re: carry for lane 0 of each group.
// MATRIX (substitution-matrix scoring, sw_dp16_matrix_kernel): sub = {S[tA][qA] + 2e, S[tB][qB] + 2e}, looked up one
// block ahead; the range check and BASE then use max S where the DNA kernel uses `match`.
// MODE bit 0: MATRIX; bit 1: score only (no traceback flags are formed or stored: MGL_SW_FLAG_SCORE_ONLY)
template <int MODE>
__device__ __forceinline__ void cell16(const int U, Lane16 &st, const unsigned rh, const unsigned re, const unsigned q,
                                       const unsigned tt, const Consts16 &c, unsigned &h, unsigned &eo, unsigned &fo,
                                       unsigned &hup_new, const unsigned sub)
{
    constexpr bool MATRIX = (MODE & 1) != 0, NOTB = (MODE & 2) != 0;
    hup_new = row_shr1(rh, st.h_prev);
    const unsigned ein = row_shr1(re, st.e_prev);
    unsigned diag;
    if (MATRIX) {
        diag = pk_add(st.hup, sub);
    } else {
        const unsigned m = pk_min_u(q ^ tt, c.one);     // 1 where the bases differ
        const unsigned s = pk_mad(m, c.delta, c.k2);    // match + 2e, or mismatch + 2e (not on the dependency chain)
        diag = pk_add(st.hup, s);
    }
    const unsigned sm = pk_max(diag, st.f);
    h = pk_max(sm, ein);
    const unsigned open = pk_sub(h, c.o_e);             // a new gap, either direction
    eo = pk_max(open, ein);                             // extension is free in this representation
    fo = pk_max(open, st.f);
    if (!NOTB) {
        const unsigned d1 = pk_sub_sat(diag, st.f);     // < 0 <=> F > diag
        const unsigned d2 = pk_sub_sat(sm, ein);        // < 0 <=> E > max(diag, F)
        const unsigned d3 = pk_sub_sat(ein, open);      // < 0 <=> a new vertical gap wins
        const unsigned d4 = pk_sub_sat(st.f, open);     // < 0 <=> a new horizontal gap wins
        // v_perm_b32 selectors 8..11 replicate the sign bit of bytes 1 / 3 / 5 / 7: clean 0x00 / 0xff bytes
        // [d2.A, d2.B, d1.A, d1.B] and [d4.A, d4.B, d3.A, d3.B]; step U of the block owns bits 2U+1 and 2U of every byte
        const unsigned p12 = __builtin_amdgcn_perm(d1, d2, 0x0b0a0908u);
        const unsigned p34 = __builtin_amdgcn_perm(d3, d4, 0x0b0a0908u);
        const unsigned k12 = 0x02020202u << (2 * U), k34 = 0x01010101u << (2 * U);
        const unsigned low = U == 0 ? 0u : st.acc;                                  // a new block starts from zero
        st.acc = (p34 & k34) | ((p12 & k12) | low);                                 // v_and_or_b32 x 2 (v_and + v_and_or at U == 0)
    }
}

__device__ __forceinline__ void commit16(Lane16 &st, unsigned h, unsigned eo, unsigned fo, unsigned hup_new)
{
    st.h_prev = h;
    st.e_prev = eo;
    st.hup = hup_new;
    st.f = fo;
}

#define RING_U(u, A, B, x, z) ((u) == 0 ? (A).x : (u) == 1 ? (A).z : (u) == 2 ? (B).x : (B).z)

// Stand-alone stripe (own pipeline fill and drain), four steps.
//   PRO : some lane may still be at column <= 0 (forced border values)
//   EPI : some lane may be at its last column (capture H[i][ql])
template <bool PRO, bool EPI, int MODE>
__device__ __forceinline__ void step4_16(Lane16 &st, uint4 &ringA, uint4 &ringB, const uint4 *ring_next,
                                         const unsigned (&qq)[4], const unsigned tt, const int s0, const int L,
                                         const unsigned hb, const int ql, const Consts16 &c, uint2 *ring_wr,
                                         const bool writer, const unsigned (&sub)[4])
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned rh = RING_U(u, ringA, ringB, x, z), re = RING_U(u, ringA, ringB, y, w);
        if (u == 2) ringA = ring_next[0];
        unsigned h, eo, fo, hup_new;
        cell16<MODE>(u, st, rh, re, qq[u], tt, c, h, eo, fo, hup_new, sub[u]);
        if (PRO) {
            const bool at_border = (s0 + u) <= L; // column <= 0
            h = at_border ? hb : h;
            fo = at_border ? pk_sub(hb, c.o_e) : fo;
        }
        if (EPI) st.cap = ((s0 + u - L) == ql) ? h : st.cap;
        if (writer) ring_wr[u] = make_uint2(h, eo);
        commit16(st, h, eo, fo, hup_new);
    }
    ringB = ring_next[1];
}

// Chained stripes, lean part of a period: no lane is at a border; with CAP the lanes that reach the
// last column inside this block (lanes < P - ql, last block of the period) capture it.
template <bool CAP, int MODE>
__device__ __forceinline__ void lean4_16(Lane16 &st, uint4 &ringA, uint4 &ringB, const uint4 *ring_next,
                                         const unsigned (&qq)[4], const unsigned tt, const int col0, const int L,
                                         const int ql, const Consts16 &c, uint2 *ring_wr, const bool writer,
                                         const unsigned (&sub)[4])
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned rh = RING_U(u, ringA, ringB, x, z), re = RING_U(u, ringA, ringB, y, w);
        if (u == 2) ringA = ring_next[0];
        unsigned h, eo, fo, hup_new;
        cell16<MODE>(u, st, rh, re, qq[u], tt, c, h, eo, fo, hup_new, sub[u]);
        if (CAP) st.cap = ((col0 + u - L) == ql) ? h : st.cap;
        if (writer) ring_wr[u] = make_uint2(h, eo);
        commit16(st, h, eo, fo, hup_new);
    }
    ringB = ring_next[1];
}

// Chained stripes, the 16-step window that opens period k: at window step u lane u leaves stripe k-1
// and starts stripe k at column 0 (forced border), lane u + (P - ql) is at the last column of stripe
// k-1.  All lane selections use constant SGPR masks.  Block b covers window steps 4b .. 4b+3.
//   FIRST : k == 0, nothing to finish or publish yet (lane 15 has not started)
//   LAST  : k == number of chained stripes: lanes leave into nothing; lane 15 must not publish column 0
template <int B, bool FIRST, bool LAST, int MODE>
__device__ __forceinline__ void window4_16(Lane16 &st, uint4 &ringA, uint4 &ringB, const uint4 *ring_next,
                                           const unsigned (&qq)[4], unsigned &tt, const unsigned tt_nxt,
                                           const unsigned hb_nxt, const unsigned hbf_nxt, const int cgap,
                                           const Consts16 &c, uint2 *ring_wr_tail, uint2 *ring_col0, const bool lane15,
                                           const unsigned (&sub)[4])
{
    constexpr unsigned long long ROWS = 0x0001000100010001ull; // lane 0 of every 16-lane row
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int u = 4 * B + t;
        const unsigned rh = RING_U(t, ringA, ringB, x, z), re = RING_U(t, ringA, ringB, y, w);
        if (t == 2) ringA = ring_next[0];
        unsigned h, eo, fo, hup_new;
        cell16<MODE>(t, st, rh, re, qq[t], tt, c, h, eo, fo, hup_new, sub[t]);
        const unsigned long long m_u = ROWS << u;
        if (!FIRST) {
            // lane u + cgap sits on column ql of the stripe it is finishing
            const int cl = u + cgap;
            const unsigned long long m_cap = cl < 16 ? ROWS << cl : 0ull;
            st.cap = sel_mask(st.cap, h, m_cap);
        }
        // lane u: column 0 of its next row -- H[i][0] and F[i][1] are border values (sw.cpp:24,38,47-49)
        h = sel_mask(h, hb_nxt, m_u);
        fo = sel_mask(fo, hbf_nxt, m_u);
        tt = sel_mask(tt, tt_nxt, m_u);
        // lane 15 publishes the carry of the stripe it is finishing (columns P-15+u), then column 0 of the new one
        if (!FIRST && u < 15) {
            if (lane15) ring_wr_tail[u] = make_uint2(h, eo);
        } else if (!LAST && u == 15) {
            if (lane15) ring_col0[0] = make_uint2(h, eo);
        }
        commit16(st, h, eo, fo, hup_new);
    }
    ringB = ring_next[1];
}

// ---- substitution-matrix mode: the table in LDS holds int16 S[t][q] + 2e, 32 x 32; tt = row byte offsets (code * 64)
// of the two pairs packed, q = column byte offsets (code * 2) packed, so one add gives both table offsets
__device__ __forceinline__ unsigned lut2(const short *lut, const unsigned tt, const unsigned q)
{
    const unsigned s = tt + q;
    const char *base = reinterpret_cast<const char *>(lut);
    const unsigned short lo = *reinterpret_cast<const unsigned short *>(base + (s & 0xffffu));
    const unsigned short hi = *reinterpret_cast<const unsigned short *>(base + (s >> 16));
    return (unsigned)lo | ((unsigned)hi << 16);
}
// the target residue lane L uses at window step W: it adopts its next row's at step L
template <int W>
__device__ __forceinline__ unsigned tt_at(const unsigned tt, const unsigned tt_new)
{
    if (W < 0) return tt;
    if (W >= 15) return tt_new;
    constexpr unsigned long long LE = ((1ull << ((W < 0 ? 0 : W > 14 ? 14 : W) + 1)) - 1ull) * 0x0001000100010001ull; // lanes L <= W of every row
    return sel_mask(tt, tt_new, LE);
}

template <int MODE>
__device__ __forceinline__ void sw_dp16_body(const DpArgs &a, unsigned char *smem, const short *lut)
{
    constexpr bool MATRIX = (MODE & 1) != 0, NOTB = (MODE & 2) != 0;
    unsigned long long diag_t0 = 0, diag_w0 = 0;
    if (a.diag) { // in-kernel clock probe (MI355X_MICROARCH.md, DVFS give-back item 6); off in normal runs
        diag_t0 = __builtin_amdgcn_s_memtime();
        diag_w0 = __builtin_amdgcn_s_memrealtime();
    }

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = lane >> 4;
    const int L = lane & 15;
    // group slot: pairs 2*gs and 2*gs+1 of the chunk
    const int64_t wave_gs = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * 4;
    const int64_t gs = wave_gs + grp;
    const int64_t n_groups = (a.count + 1) >> 1;
    if (wave_gs >= n_groups) return;
    const bool gvalid = gs < n_groups;
    const int64_t slotA = gvalid ? 2 * gs : a.count - 1;
    const bool validB = gvalid && (2 * gs + 1 < a.count);
    const int64_t slotB = validB ? 2 * gs + 1 : slotA;
    const int64_t pA = a.first + slotA, pB = a.first + slotB;

    // one geometry per WAVE: every pair of the batch has it (MGL_SW_FLAG_UNIFORM_GEOMETRY), or at least every aligned
    // block of eight consecutive pairs (MGL_SW_FLAG_GROUPED_GEOMETRY, batches sorted by read length); a.uni_tl /
    // a.uni_ql are then the maxima that size the LDS carve and the traceback stride
    const int tl = __builtin_amdgcn_readfirstlane(a.t.length(pA)), ql = __builtin_amdgcn_readfirstlane(a.q.length(pA));
    const int64_t tA = a.t.off[pA], tB = a.t.off[pB], qA = a.q.off[pA], qB = a.q.off[pB];

    const int nstripes = stripes_for(tl);
    const int sps = sps_for(ql);              // steps of a stand-alone stripe
    const int P = dp16_period(ql);            // steps per chained stripe
    const int nc = dp16_chained_stripes(tl, ql);
    const int main_end = max(16, ql & ~3);

    // LDS carve per group: ring uint2[sps+20] (H, E' packed A|B per column) | qq uint32[sps+48], sized for the longest
    // query of the batch so that every wave of a block carves the same way
    const int sps_carve = sps_for(a.uni_ql);
    const int ring_entries = sps_carve + RING_SLACK16;
    const int qq_entries = sps_carve + QQ_SLACK16;
    const int group_bytes = ring_entries * 8 + qq_entries * 4;
    unsigned char *gbase = smem + (size_t)(wave * 4 + grp) * group_bytes;
    uint2 *ring = reinterpret_cast<uint2 *>(gbase);                          // ring[j + 16] = column j
    unsigned *qq = reinterpret_cast<unsigned *>(gbase + ring_entries * 8);   // qq[c + 16] = q[c] (0-based), A | B<<16

    const int match = a.match, gopen = a.gopen, gext = a.gext;
    const bool indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;
    Consts16 c;
    c.delta = pack2(a.mismatch - match, a.mismatch - match);
    c.one = pack2(1, 1);
    const int colw = -gext;                  // stored = X + i*gext - j*colw + BASE
    const int base = dp16_base(tl, ql, match, gext); // BASE: no stored value of a valid cell exceeds 32767
    c.o_e = pack2(gopen - gext, gopen - gext);
    c.k2 = pack2(match + 2 * gext, match + 2 * gext);
    asm volatile("" : "+v"(c.delta), "+v"(c.one), "+v"(c.o_e), "+v"(c.k2));

    // ---- stage the two queries interleaved.  qq[x] is the base of column x - 15 (x - 16 as 0-based query
    // index); beyond column P - 1 the array repeats with period P, so a chained lane keeps incrementing its
    // read pointer across the end of a row.  Then the border row (sw.cpp:14-18,31-35) in stored form.
    for (int x = L; x < qq_entries; x += 16) {
        int cidx = x - 16;
        if (nc && cidx >= P - 1) cidx -= P;
        unsigned v = 0;
        if (cidx >= 0 && cidx < ql) {
            if (MATRIX)
                v = (unsigned)a.code[a.q.at(qA, cidx) & 0xff] * 2u | ((unsigned)a.code[a.q.at(qB, cidx) & 0xff] * 2u << 16);
            else
                v = (unsigned)a.q.at(qA, cidx) | ((unsigned)a.q.at(qB, cidx) << 16);
        }
        qq[x] = v;
    }
    for (int j = L; j <= ql; j += 16) {
        const int hb0 = border(j, gopen, gext, indel) - j * colw + base;  // row 0
        const int eb1 = hb0 - gopen + gext;                                // E[1][j] = H[0][j] - o, row 1 offset
        ring[j + 16] = make_uint2(pack2(hb0, hb0), pack2(eb1, eb1));
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();

    Lane16 st;
    st.h_prev = st.e_prev = st.hup = st.f = 0;
    st.acc = 0;
    st.cap = 0;
    int bestA = NEG_INF, bestA_i = -1, bestB = NEG_INF, bestB_i = -1; // last-column maxima (scores)

    const int last_lane = (tl - 1) & 15;
    // traceback words: group region + lane; two dwords per lane per 8 steps (32 dwords per group block)
    uint32_t *tbp = a.tb + (size_t)(gvalid ? gs : n_groups - 1) * a.tb_stride_words + L * 2;
    int gsteps = 0;
    unsigned tb_hold = 0;

#define MGL_TB_ADVANCE()                                                                                  \
    if (!NOTB) {                                                                                          \
        gsteps += 4;                                                                                      \
        if (gsteps & 4) {                                                                                 \
            tb_hold = st.acc;                                                                             \
        } else {                                                                                          \
            if (gvalid) *reinterpret_cast<uint2 *>(tbp) = make_uint2(tb_hold, st.acc);                    \
            tbp += 32;                                                                                    \
        }                                                                                                 \
    }
// last column of a finished stripe row (sw.cpp:100-104: >= so the later row wins); rows carry different
// offsets, so compare scores, not stored values
#define MGL_TAKE_CAP(row)                                                                                 \
    if ((row) >= 1 && (row) <= tl) {                                                                      \
        const int unshift = ql * colw - (row) * gext - base;                                              \
        const int ca = lo16(st.cap) + unshift, cb = hi16(st.cap) + unshift;                               \
        if (ca >= bestA) {                                                                                \
            bestA = ca;                                                                                   \
            bestA_i = (row);                                                                              \
        }                                                                                                 \
        if (cb >= bestB) {                                                                                \
            bestB = cb;                                                                                   \
            bestB_i = (row);                                                                              \
        }                                                                                                 \
    }

#define MGL_TT_PACK(row)                                                                                  \
    (MATRIX ? ((unsigned)a.code[a.t.at(tA, (row)-1) & 0xff] * 64u | ((unsigned)a.code[a.t.at(tB, (row)-1) & 0xff] * 64u << 16)) \
            : ((unsigned)a.t.at(tA, (row)-1) | ((unsigned)a.t.at(tB, (row)-1) << 16)))
    int row_next = 1 + L; // row this lane takes in the next stripe
    unsigned tt_next = (row_next <= tl) ? MGL_TT_PACK(row_next) : 0u;

    // ======================= chained stripes 0 .. nc-1: one continuous systolic pipeline =======================
    if (nc > 0) {
        const bool lane15 = (L == 15);
        const int cgap = P - ql;                              // 1..4: columns between ql and the end of a period
        const unsigned *qrd = qq + 15 - L;                    // window step u of lane L reads column u - L (+P before it wraps)
        const uint4 *ring_c0 = reinterpret_cast<const uint4 *>(ring + 16);
        uint4 rA = ring_c0[0], rB = ring_c0[1];
        // a lane that has not wrapped yet is P columns further: qq is periodic, so read there
        const unsigned *qcur = qrd + P;
        unsigned qv[4] = {qcur[0], qcur[1], qcur[2], qcur[3]};
        unsigned tt = 0;
        unsigned sub[4] = {0u, 0u, 0u, 0u}; // MATRIX: scores of the current block's four steps
        if (MATRIX) { // window block 0 of stripe 0: lane L starts at step L
            sub[0] = lut2(lut, tt_at<0>(tt, tt_next), qv[0]);
            sub[1] = lut2(lut, tt_at<1>(tt, tt_next), qv[1]);
            sub[2] = lut2(lut, tt_at<2>(tt, tt_next), qv[2]);
            sub[3] = lut2(lut, tt_at<3>(tt, tt_next), qv[3]);
        }
        for (int k = 0; k <= nc; ++k) {
            // ---- what every lane adopts when it wraps inside this window
            const int row_new = row_next;
            const unsigned tt_new = tt_next;
            const int hbv = border(row_new, gopen, gext, indel) + row_new * gext + base; // column 0 of the new row
            const unsigned hb_new = pack2(hbv, hbv);
            const unsigned hbf_new = pk_sub(hb_new, c.o_e);
            if (k < nc) {
                row_next += 16;
                tt_next = (row_next <= tl) ? MGL_TT_PACK(row_next) : 0u;
            }
            const uint4 *ring_rd = ring_c0;              // lane 0 is at column 0 when the window opens
            uint2 *ring_tail = ring + 16 + (P - 15);     // lane 15: columns P-15 .. P-1 of the stripe it finishes
#define MGL_WINDOW_BLOCK(B, FIRST, LAST)                                                                  \
    {                                                                                                     \
        const unsigned n0 = qcur[4], n1 = qcur[5], n2 = qcur[6], n3 = qcur[7];                            \
        unsigned ns0 = 0, ns1 = 0, ns2 = 0, ns3 = 0;                                                      \
        if (MATRIX) { /* the next block's steps 4(B+1) .. 4(B+1)+3 */                                      \
            ns0 = lut2(lut, tt_at<4 * (B) + 4>(tt, tt_new), n0);                                          \
            ns1 = lut2(lut, tt_at<4 * (B) + 5>(tt, tt_new), n1);                                          \
            ns2 = lut2(lut, tt_at<4 * (B) + 6>(tt, tt_new), n2);                                          \
            ns3 = lut2(lut, tt_at<4 * (B) + 7>(tt, tt_new), n3);                                          \
        }                                                                                                 \
        window4_16<B, FIRST, LAST, MODE>(st, rA, rB, ring_rd + 2, qv, tt, tt_new, hb_new, hbf_new, cgap, c, \
                                           ring_tail, ring + 16, lane15, sub);                            \
        sub[0] = ns0;                                                                                     \
        sub[1] = ns1;                                                                                     \
        sub[2] = ns2;                                                                                     \
        sub[3] = ns3;                                                                                     \
        qv[0] = n0;                                                                                       \
        qv[1] = n1;                                                                                       \
        qv[2] = n2;                                                                                       \
        qv[3] = n3;                                                                                       \
        ring_rd += 2;                                                                                     \
        qcur += 4;                                                                                        \
        MGL_TB_ADVANCE()                                                                                  \
    }
            if (k == 0) {
                MGL_WINDOW_BLOCK(0, true, false)
                MGL_WINDOW_BLOCK(1, true, false)
                MGL_WINDOW_BLOCK(2, true, false)
                MGL_WINDOW_BLOCK(3, true, false)
            } else if (k == nc) {
                MGL_WINDOW_BLOCK(0, false, true)
                MGL_WINDOW_BLOCK(1, false, true)
                MGL_WINDOW_BLOCK(2, false, true)
                MGL_WINDOW_BLOCK(3, false, true)
            } else {
                MGL_WINDOW_BLOCK(0, false, false)
                MGL_WINDOW_BLOCK(1, false, false)
                MGL_WINDOW_BLOCK(2, false, false)
                MGL_WINDOW_BLOCK(3, false, false)
            }
#undef MGL_WINDOW_BLOCK
            // every lane has now finished its row of stripe k-1 ...
            if (k > 0) {
                // lanes below cgap captured in the last lean block of the previous period, the others in this window
                MGL_TAKE_CAP(row_new - 16)
            }
            if (k == nc) break;
            // ... and is inside stripe k: all read pointers are one period too far, bring them back
            qcur -= P;

            // ---- lean part of the period: columns 16 .. P-1 for lane 0
            uint2 *ring_wr = ring + 16 + 1;              // lane 15 is at column s - 15
            int s = 16;
#define MGL_LEAN_BLOCK(CAP, NEXT)                                                                         \
    {                                                                                                     \
        const unsigned n0 = qcur[4], n1 = qcur[5], n2 = qcur[6], n3 = qcur[7];                            \
        unsigned ns0 = 0, ns1 = 0, ns2 = 0, ns3 = 0;                                                      \
        if (MATRIX) {                                                                                     \
            if (CAP) { /* the next block opens the next period's window: lane L adopts tt_next at its step L */ \
                ns0 = lut2(lut, tt_at<0>(tt, tt_next), n0);                                               \
                ns1 = lut2(lut, tt_at<1>(tt, tt_next), n1);                                               \
                ns2 = lut2(lut, tt_at<2>(tt, tt_next), n2);                                               \
                ns3 = lut2(lut, tt_at<3>(tt, tt_next), n3);                                               \
            } else {                                                                                      \
                ns0 = lut2(lut, tt, n0);                                                                  \
                ns1 = lut2(lut, tt, n1);                                                                  \
                ns2 = lut2(lut, tt, n2);                                                                  \
                ns3 = lut2(lut, tt, n3);                                                                  \
            }                                                                                             \
        }                                                                                                 \
        lean4_16<CAP, MODE>(st, rA, rB, NEXT, qv, tt, s, L, ql, c, ring_wr, lane15, sub);               \
        sub[0] = ns0;                                                                                     \
        sub[1] = ns1;                                                                                     \
        sub[2] = ns2;                                                                                     \
        sub[3] = ns3;                                                                                     \
        qv[0] = n0;                                                                                       \
        qv[1] = n1;                                                                                       \
        qv[2] = n2;                                                                                       \
        qv[3] = n3;                                                                                       \
        ring_rd += 2;                                                                                     \
        ring_wr += 4;                                                                                     \
        qcur += 4;                                                                                        \
        s += 4;                                                                                           \
        MGL_TB_ADVANCE()                                                                                  \
    }
            for (; s + 12 <= P;) {
                MGL_LEAN_BLOCK(false, ring_rd + 2)
                MGL_LEAN_BLOCK(false, ring_rd + 2)
            }
            for (; s + 8 <= P;) MGL_LEAN_BLOCK(false, ring_rd + 2)
            // last block of the period: the next block is the window of period k+1 (lane 0 back at column 0)
            MGL_LEAN_BLOCK(true, ring_c0)
#undef MGL_LEAN_BLOCK
        }
    }

    // ======================= remaining stripes stand-alone (partial last stripe, or short queries) =============
    for (int k = nc; k < nstripes; ++k) {
        const int row_i = row_next;
        const unsigned tt = tt_next;
        row_next += 16;
        tt_next = (row_next <= tl) ? MGL_TT_PACK(row_next) : 0u;

        const int hbv = border(row_i, gopen, gext, indel) + row_i * gext + base; // column 0
        const unsigned hb = pack2(hbv, hbv);
        const int wl = (k == nstripes - 1) ? last_lane : 15;
        const bool writer = (L == wl);
        st.cap = 0;

        const uint4 *ring_rd = reinterpret_cast<const uint4 *>(ring + 16);
        uint2 *ring_wr = ring + 16 - wl;
        const unsigned *qrd = qq + 15 - L; // step s reads q index s - L - 1  ->  qq[s - L + 15]
        uint4 rA = ring_rd[0], rB = ring_rd[1];
        unsigned qv[4] = {qrd[0], qrd[1], qrd[2], qrd[3]};
        unsigned sub[4] = {0u, 0u, 0u, 0u};
        if (MATRIX) {
            sub[0] = lut2(lut, tt, qv[0]);
            sub[1] = lut2(lut, tt, qv[1]);
            sub[2] = lut2(lut, tt, qv[2]);
            sub[3] = lut2(lut, tt, qv[3]);
        }

        int s = 0;
#define MGL_SW_BLOCK16(PRO, EPI)                                                                          \
    {                                                                                                     \
        const unsigned n0 = qrd[4], n1 = qrd[5], n2 = qrd[6], n3 = qrd[7];                                \
        unsigned ns0 = 0, ns1 = 0, ns2 = 0, ns3 = 0;                                                      \
        if (MATRIX) {                                                                                     \
            ns0 = lut2(lut, tt, n0);                                                                      \
            ns1 = lut2(lut, tt, n1);                                                                      \
            ns2 = lut2(lut, tt, n2);                                                                      \
            ns3 = lut2(lut, tt, n3);                                                                      \
        }                                                                                                 \
        step4_16<PRO, EPI, MODE>(st, rA, rB, ring_rd + 2, qv, tt, s, L, hb, ql, c, ring_wr, writer, sub); \
        sub[0] = ns0;                                                                                     \
        sub[1] = ns1;                                                                                     \
        sub[2] = ns2;                                                                                     \
        sub[3] = ns3;                                                                                     \
        qv[0] = n0;                                                                                       \
        qv[1] = n1;                                                                                       \
        qv[2] = n2;                                                                                       \
        qv[3] = n3;                                                                                       \
        ring_rd += 2;                                                                                     \
        ring_wr += 4;                                                                                     \
        qrd += 4;                                                                                         \
        s += 4;                                                                                           \
        MGL_TB_ADVANCE()                                                                                  \
    }
        for (; s < 16;) MGL_SW_BLOCK16(true, true)
        // two blocks per trip: the one-block-ahead prefetch registers alternate instead of being copied
        for (; s + 8 <= main_end;) {
            MGL_SW_BLOCK16(false, false)
            MGL_SW_BLOCK16(false, false)
        }
        for (; s < main_end;) MGL_SW_BLOCK16(false, false)
        for (; s < sps;) MGL_SW_BLOCK16(false, true)
#undef MGL_SW_BLOCK16
        MGL_TAKE_CAP(row_i)
    }
#undef MGL_TB_ADVANCE
#undef MGL_TAKE_CAP

    if (!NOTB && (gsteps & 4) && gvalid) *reinterpret_cast<uint2 *>(tbp) = make_uint2(tb_hold, 0u);

    // global step at which the stand-alone stripes start (the traceback kernel needs it to find their cells)
    const int g_tail = nc ? nc * P + 16 : 0;

    // ---- both matrices are complete: last column max, last row scan (sw.cpp:100-127), per half
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int mqe = half ? bestB : bestA, mqe_t = half ? bestB_i : bestA_i;
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            const int ob = __shfl_xor(mqe, m), oi = __shfl_xor(mqe_t, m);
            const bool take = ob > mqe || (ob == mqe && oi > mqe_t);
            mqe = take ? ob : mqe;
            mqe_t = take ? oi : mqe_t;
        }
        int rm = NEG_INF, rd = 0x7fffffff, rj = 0x7fffffff;
        for (int j = L + 1; j <= ql; j += 16) {
            const unsigned x = ring[j + 16].x;
            const int sc = (half ? hi16(x) : lo16(x)) + j * colw - tl * gext - base;
            const int d = abs(tl - j);
            const bool take = sc > rm || (sc == rm && d < rd);
            rm = take ? sc : rm;
            rd = take ? d : rd;
            rj = take ? j : rj;
        }
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) {
            const int om = __shfl_xor(rm, m), od = __shfl_xor(rd, m), oj = __shfl_xor(rj, m);
            const bool take = om > rm || (om == rm && (od < rd || (od == rd && oj < rj)));
            rm = take ? om : rm;
            rd = take ? od : rd;
            rj = take ? oj : rj;
        }
        const bool row_wins = rm > mqe || (rm == mqe && rd < abs(mqe_t - ql));
        const bool ok = half ? validB : gvalid;
        if (L == 0 && ok) {
            DpRecord r;
            r.mqe = mqe;
            r.mqe_t = mqe_t;
            r.max = row_wins ? rm : mqe;
            r.max_t = row_wins ? tl : mqe_t;
            r.max_q = row_wins ? rj : ql;
            r.seg = row_wins ? ql - rj : 0;
            r.g_tail = g_tail;
            r.sps = nc ? P : sps;
            a.rec[half ? slotB : slotA] = r;
        }
    }
    if (a.diag && threadIdx.x == 0) {
        a.diag[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - diag_t0;
        a.diag[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - diag_w0;
    }
#undef MGL_TT_PACK
}

} // namespace

__global__ __launch_bounds__(256) void sw_dp16_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    sw_dp16_body<0>(a, smem, nullptr);
}

// MGL_SW_FLAG_SCORE_ONLY: the same fill without the four traceback flags (10 of its 22 instructions per step)
__global__ __launch_bounds__(256) void sw_dp16_score_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    sw_dp16_body<2>(a, smem, nullptr);
}

// substitution-matrix scoring (protein extension, no reference path): a.match = the largest matrix entry (range check
// and BASE); the table S + 2e (int16, 2 KB) sits in LDS behind the per-group carves
__global__ __launch_bounds__(256) void sw_dp16_matrix_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    short *lut = reinterpret_cast<short *>(smem + a.matrix_lds_offset);
    for (int x = threadIdx.x; x < MATRIX_DIM * MATRIX_DIM; x += blockDim.x) lut[x] = (short)((int)a.matrix[x] + 2 * a.gext);
    __syncthreads();
    sw_dp16_body<1>(a, smem, lut);
}

__global__ __launch_bounds__(256) void sw_dp16_matrix_score_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    short *lut = reinterpret_cast<short *>(smem + a.matrix_lds_offset);
    for (int x = threadIdx.x; x < MATRIX_DIM * MATRIX_DIM; x += blockDim.x) lut[x] = (short)((int)a.matrix[x] + 2 * a.gext);
    __syncthreads();
    sw_dp16_body<3>(a, smem, lut);
}

int dp16_lds_bytes(int sps, int waves_per_block)
{
    return waves_per_block * 4 * ((sps + RING_SLACK16) * 8 + (sps + QQ_SLACK16) * 4);
}

// Can every stored value of a tl x ql problem with these (normalised) parameters be held in 16 bits?
// stored = X + (i+j)*e + BASE: X <= match*min(i,j) bounds the top (dp16_base puts it at 32767); the all-gap path from a
// border bounds every H, E, F of a valid cell from below by -2o - (i+j-2)*e, i.e. stored >= BASE - 2o + 2e; the
// intermediates (H - (o-e), diag on a mismatch) go at most max(o, |mismatch|) lower.
bool dp16_range_ok(int tl, int ql, int match, int mismatch, int gopen, int gext, int strategy)
{
    (void)strategy; // the bound holds for every border rule (sw.cpp:29-40: INDEL borders are the lowest)
    if (match <= 0 || gopen < gext) return false;
    const int64_t top = (int64_t)match * std::min(tl, ql) + (int64_t)gext * ((int64_t)tl + ql);
    const int64_t low = -3 * (int64_t)gopen - ((int64_t)match - mismatch) - 2 * (int64_t)gext - 64;
    return 32767 - top + low >= -32768 && (int64_t)match - mismatch <= 30000 && gopen <= 10000 && gext <= 5000 &&
           (int64_t)match + 2 * gext <= 30000;
}

hipError_t launch_dp16(const DpArgs &a, int waves_per_block, hipStream_t stream)
{
    const int per_block = waves_per_block * 8; // pairs per block
    const int64_t blocks = (a.count + per_block - 1) / per_block;
    int lds = dp16_lds_bytes(sps_for(a.uni_ql), waves_per_block);
    if (lds > 64 * 1024) { // per device and rare (long queries): set every time rather than cache across devices / threads
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sw_dp16_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    if (a.matrix) {
        DpArgs b = a;
        b.matrix_lds_offset = lds;
        const int lds_m = lds + MATRIX_DIM * MATRIX_DIM * 2;
        auto kernel = a.score_only ? sw_dp16_matrix_score_kernel : sw_dp16_matrix_kernel;
        if (lds_m > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_m);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * waves_per_block), lds_m, stream, b);
        return hipGetLastError();
    }
    if (a.score_only) {
        if (lds > 64 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sw_dp16_score_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(sw_dp16_score_kernel, dim3((unsigned)blocks), dim3(64 * waves_per_block), lds, stream, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(sw_dp16_kernel, dim3((unsigned)blocks), dim3(64 * waves_per_block), lds, stream, a);
    return hipGetLastError();
}

} // namespace mgl_sw_dev
