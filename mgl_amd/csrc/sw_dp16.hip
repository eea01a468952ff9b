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
//        v = X[i][j] + i*e - j*(match+e) + BASE      (X = H, E or F;  e = gap extend)
// i.e. a per-cell offset that is linear in (i, j).  Consequences:
//   * diag = H[i-1][j-1] + s(t,q)  becomes  v_diag = v_hup + (t==q ? 0 : mismatch-match): the constant of
//     the match case vanishes, so the substitution score is  m*delta  with m = min(t^q, 1) -- one
//     v_pk_mad_i16, no compare/select (there is no packed compare on CDNA4);
//   * E moves one row down: E' = max(H - (o-e), E) -- extending a vertical gap costs NO instruction;
//     F moves one column right: F' = max(H - (o+match+e), F - (match+2e));
//   * H <= match*min(i,j), so v <= BASE + e*(i-j) <= BASE + e*tl, and v >= BASE - (match+2e)*ql - (gap
//     terms): the span is about match*ql + e*(tl+2ql), checked by the host (dp16_range_ok()); BASE
//     puts the top of that span at +32767;
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

#include "sw_device.h"

namespace mgl_sw_dev {

namespace {

constexpr int DPP_ROW_SHR1 = 0x111;
constexpr int RING_SLACK16 = 20, QQ_SLACK16 = 36;

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned row_shr1(unsigned lane0_value, unsigned src)
{
    return (unsigned)__builtin_amdgcn_update_dpp((int)lane0_value, (int)src, DPP_ROW_SHR1, 0xf, 0xf, false);
}
__device__ __forceinline__ short2_t as_s2(unsigned x) { return __builtin_bit_cast(short2_t, x); }
__device__ __forceinline__ unsigned as_u(short2_t x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) { return as_u(as_s2(a) - as_s2(b)); }
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
    return as_u(as_s2(a) * as_s2(b) + as_s2(c));
}
__device__ __forceinline__ unsigned pack2(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
__device__ __forceinline__ int lo16(unsigned x) { return (int)(short)(x & 0xffffu); }
__device__ __forceinline__ int hi16(unsigned x) { return (int)x >> 16; }

__device__ __forceinline__ int border(int k, int gopen, int gext, bool indel)
{
    return (indel && k > 0) ? -gopen - (k - 1) * gext : 0; // sw.cpp:29-40,47-49
}

__host__ __device__ inline int dp16_base(int tl, int gext) { return 32767 - gext * tl; }

struct Lane16 {
    unsigned h_prev, e_prev, hup, f; // packed A|B, column-shifted + biased
    unsigned acc;                    // traceback flags of the current 4-step block
    unsigned cap;                    // H of the last column (this stripe), packed
};

struct Consts16 {
    unsigned delta, one, o_e, o_f, e_f; // packed constants (both halves equal)
};

template <bool PRO, bool EPI>
__device__ __forceinline__ void step4_16(Lane16 &st, uint4 &ringA, uint4 &ringB, const uint4 *ring_next,
                                         const unsigned (&qq)[4], const unsigned tt, const int s0, const int L,
                                         const unsigned hb, const int ql, const Consts16 &c, uint2 *ring_wr,
                                         const bool writer)
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const unsigned rh = u == 0 ? ringA.x : u == 1 ? ringA.z : u == 2 ? ringB.x : ringB.z;
        const unsigned re = u == 0 ? ringA.y : u == 1 ? ringA.w : u == 2 ? ringB.y : ringB.w;
        if (u == 2) ringA = ring_next[0];
        const unsigned hup_new = row_shr1(rh, st.h_prev);
        const unsigned ein = row_shr1(re, st.e_prev);
        const unsigned m = pk_min_u(qq[u] ^ tt, c.one);     // 1 where the bases differ
        const unsigned diag = pk_mad(m, c.delta, st.hup);   // + (mismatch - match) on a mismatch
        const unsigned d1 = pk_sub_sat(diag, st.f);         // < 0 <=> F > diag
        const unsigned sm = pk_max(diag, st.f);
        const unsigned d2 = pk_sub_sat(sm, ein);            // < 0 <=> E > max(diag, F)
        unsigned h = pk_max(sm, ein);
        const unsigned open_e = pk_sub(h, c.o_e);
        const unsigned open_f = pk_sub(h, c.o_f);
        const unsigned d3 = pk_sub_sat(ein, open_e);        // < 0 <=> a new vertical gap wins
        const unsigned eo = pk_max(open_e, ein);            // extension is free in this representation
        const unsigned fe = pk_sub(st.f, c.e_f);
        const unsigned d4 = pk_sub_sat(fe, open_f);         // < 0 <=> a new horizontal gap wins
        unsigned fo = pk_max(open_f, fe);
        if (PRO) {
            const bool at_border = (s0 + u) <= L; // column <= 0
            h = at_border ? hb : h;
            fo = at_border ? pk_sub(hb, c.o_f) : fo;
        }
        // sign bytes: [d2.A, d2.B, d1.A, d1.B] and [d4.A, d4.B, d3.A, d3.B]
        const unsigned p12 = __builtin_amdgcn_perm(d1, d2, 0x07050301u);
        const unsigned p34 = __builtin_amdgcn_perm(d3, d4, 0x07050301u);
        const unsigned y = (p12 & 0x80808080u) | ((p34 >> 1) & ~0x80808080u);      // v_bfi_b32
        st.acc = (y & 0xC0C0C0C0u) | ((st.acc >> 2) & ~0xC0C0C0C0u);                // v_bfi_b32
        if (EPI) st.cap = ((s0 + u - L) == ql) ? h : st.cap;
        if (writer) ring_wr[u] = make_uint2(h, eo);
        st.h_prev = h;
        st.e_prev = eo;
        st.hup = hup_new;
        st.f = fo;
    }
    ringB = ring_next[1];
}

} // namespace

__global__ __launch_bounds__(256) void sw_dp16_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
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

    const int tl = a.uni_tl, ql = a.uni_ql; // one geometry for the whole batch
    const uint8_t *tA = a.targets + a.t_off[pA], *tB = a.targets + a.t_off[pB];
    const uint8_t *qA = a.queries + a.q_off[pA], *qB = a.queries + a.q_off[pB];

    const int nstripes = stripes_for(tl);
    const int sps = sps_for(ql);
    const int main_end = max(16, ql & ~3);

    // LDS carve per group: ring uint2[sps+20] (H, E' packed A|B per column) | qq uint32[sps+36]
    const int ring_entries = sps + RING_SLACK16;
    const int qq_entries = sps + QQ_SLACK16;
    const int group_bytes = ring_entries * 8 + qq_entries * 4;
    unsigned char *gbase = smem + (size_t)(wave * 4 + grp) * group_bytes;
    uint2 *ring = reinterpret_cast<uint2 *>(gbase);                          // ring[j + 16] = column j
    unsigned *qq = reinterpret_cast<unsigned *>(gbase + ring_entries * 8);   // qq[c + 16] = q[c] (0-based), A | B<<16

    const int match = a.match, gopen = a.gopen, gext = a.gext;
    const bool indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;
    Consts16 c;
    c.delta = pack2(a.mismatch - match, a.mismatch - match);
    c.one = pack2(1, 1);
    const int colw = match + gext;          // column offset per j
    const int base = dp16_base(tl, gext);    // BASE: the largest stored value is exactly 32767
    c.o_e = pack2(gopen - gext, gopen - gext);
    c.o_f = pack2(gopen + colw, gopen + colw);
    c.e_f = pack2(gext + colw, gext + colw);
    asm volatile("" : "+v"(c.delta), "+v"(c.one), "+v"(c.o_e), "+v"(c.o_f), "+v"(c.e_f));

    // ---- stage the two queries interleaved, and the border row (sw.cpp:14-18,31-35) in stored form
    for (int x = L; x < qq_entries; x += 16) {
        const int cidx = x - 16;
        unsigned v = 0;
        if (cidx >= 0 && cidx < ql) v = (unsigned)qA[cidx] | ((unsigned)qB[cidx] << 16);
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
    int bestA = NEG_INF, bestA_i = -1, bestB = NEG_INF, bestB_i = -1; // last-column maxima (scores)

    const int last_lane = (tl - 1) & 15;
    // traceback words: group region + lane; two dwords per lane per 8 steps (32 dwords per group block)
    uint32_t *tbp = a.tb + (size_t)(gvalid ? gs : n_groups - 1) * a.tb_stride_words + L * 2;
    int gsteps = 0;
    unsigned tb_hold = 0;

    int row_next = 1 + L;
    unsigned tt_next = (row_next <= tl) ? ((unsigned)tA[row_next - 1] | ((unsigned)tB[row_next - 1] << 16)) : 0u;

    for (int k = 0; k < nstripes; ++k) {
        const int row_i = row_next;
        const unsigned tt = tt_next;
        row_next += 16;
        tt_next = (row_next <= tl) ? ((unsigned)tA[row_next - 1] | ((unsigned)tB[row_next - 1] << 16)) : 0u;

        const int hbv = border(row_i, gopen, gext, indel) + row_i * gext + base; // column 0
        const unsigned hb = pack2(hbv, hbv);
        const int wl = (k == nstripes - 1) ? last_lane : 15;
        const bool writer = (L == wl);
        const bool cap_valid = row_i <= tl;
        st.cap = 0;

        const uint4 *ring_rd = reinterpret_cast<const uint4 *>(ring + 16);
        uint2 *ring_wr = ring + 16 - wl;
        const unsigned *qrd = qq + 15 - L; // step s reads q index s - L - 1  ->  qq[s - L + 15]
        uint4 rA = ring_rd[0], rB = ring_rd[1];
        unsigned qv[4] = {qrd[0], qrd[1], qrd[2], qrd[3]};

        int s = 0;
#define MGL_SW_BLOCK16(PRO, EPI)                                                                          \
    {                                                                                                     \
        const unsigned n0 = qrd[4], n1 = qrd[5], n2 = qrd[6], n3 = qrd[7];                                \
        step4_16<PRO, EPI>(st, rA, rB, ring_rd + 2, qv, tt, s, L, hb, ql, c, ring_wr, writer);            \
        qv[0] = n0;                                                                                       \
        qv[1] = n1;                                                                                       \
        qv[2] = n2;                                                                                       \
        qv[3] = n3;                                                                                       \
        ring_rd += 2;                                                                                     \
        ring_wr += 4;                                                                                     \
        qrd += 4;                                                                                         \
        s += 4;                                                                                           \
        gsteps += 4;                                                                                      \
        if (gsteps & 4) {                                                                                 \
            tb_hold = st.acc;                                                                             \
        } else {                                                                                          \
            if (gvalid) *reinterpret_cast<uint2 *>(tbp) = make_uint2(tb_hold, st.acc);                    \
            tbp += 32;                                                                                    \
        }                                                                                                 \
    }
        for (; s < 16;) MGL_SW_BLOCK16(true, true)
#ifndef MGL_NO_UNROLL2
        // two blocks per trip: the one-block-ahead prefetch registers alternate instead of being copied
        for (; s + 8 <= main_end;) {
            MGL_SW_BLOCK16(false, false)
            MGL_SW_BLOCK16(false, false)
        }
#endif
        for (; s < main_end;) MGL_SW_BLOCK16(false, false)
        for (; s < sps;) MGL_SW_BLOCK16(false, true)
#undef MGL_SW_BLOCK16

        // last column of this stripe's rows (sw.cpp:100-104: >= so the later row wins); rows carry
        // different offsets, so compare scores, not stored values
        if (cap_valid) {
            const int unshift = ql * colw - row_i * gext - base;
            const int ca = lo16(st.cap) + unshift, cb = hi16(st.cap) + unshift;
            if (ca >= bestA) {
                bestA = ca;
                bestA_i = row_i;
            }
            if (cb >= bestB) {
                bestB = cb;
                bestB_i = row_i;
            }
        }
    }

    if ((gsteps & 4) && gvalid) *reinterpret_cast<uint2 *>(tbp) = make_uint2(tb_hold, 0u);

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
            const unsigned xe = ring[ql + 16].x;
            r.h_end = (half ? hi16(xe) : lo16(xe)) + ql * colw - tl * gext - base;
            r.sps = sps;
            a.rec[half ? slotB : slotA] = r;
        }
    }
    if (a.diag && threadIdx.x == 0) {
        a.diag[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - diag_t0;
        a.diag[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - diag_w0;
    }
}

int dp16_lds_bytes(int sps, int waves_per_block)
{
    return waves_per_block * 4 * ((sps + RING_SLACK16) * 8 + (sps + QQ_SLACK16) * 4);
}

// Can every stored value of a tl x ql problem with these (normalised) parameters be held in 16 bits?
// stored = X + i*e - j*(match+e) + BASE with BASE = 32767 - e*tl (dp16_base): X <= match*min(i,j) gives
// stored <= BASE + e*(i-j) <= 32767.
bool dp16_range_ok(int tl, int ql, int match, int mismatch, int gopen, int gext, int strategy)
{
    const bool indel = (strategy & (OS_INDEL | OS_LEAD_ID)) != 0;
    const int64_t colw = (int64_t)match + gext;
    // lowest H: the cheapest gap path from a border (sw.cpp:29-40), at i >= 0, j = ql
    int64_t low = -(gopen + (int64_t)(ql - 1) * gext) - colw * ql;
    if (indel) low -= gopen + (int64_t)(tl - 1) * gext;
    // E / F / open / extend / diag intermediates below H
    low -= 2 * (gopen + colw) + (gext + colw) + ((int64_t)match - mismatch);
    const int64_t base = 32767 - (int64_t)gext * tl;
    return low + base >= -32768 + 16 && match > 0 && (int64_t)match - mismatch <= 30000 && gopen + colw <= 30000 &&
           gext + colw <= 30000 && (int64_t)gext * tl <= 30000;
}

hipError_t launch_dp16(const DpArgs &a, int waves_per_block, hipStream_t stream)
{
    const int per_block = waves_per_block * 8; // pairs per block
    const int64_t blocks = (a.count + per_block - 1) / per_block;
    int lds = dp16_lds_bytes(sps_for(a.uni_ql), waves_per_block);
    if (const char *e = getenv("MGL_SW_EXTRA_LDS")) lds += atoi(e); // occupancy experiments only
    static int configured_lds = 0;
    if (lds > 64 * 1024 && lds > configured_lds) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sw_dp16_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured_lds = lds;
    }
    hipLaunchKernelGGL(sw_dp16_kernel, dim3((unsigned)blocks), dim3(64 * waves_per_block), lds, stream, a);
    return hipGetLastError();
}

} // namespace mgl_sw_dev
