// sw_dp_coop.hip -- long-read fill kernel for gfx950: ONE pair per WORKGROUP, the waves of the workgroup
// running consecutive 64-row stripes of that pair as a systolic pipeline.  Same function as sw_dp_kernel
// (sw_kernels.hip; the reference's sw.cpp:5-146), same int32 arithmetic, same traceback layout
// (rows = 64), so the traceback kernel does not know which fill kernel ran.
//
// Why: with one pair per wave (sw_dp64_kernel) a 10 kb x 10 kb batch needs one 50 MB traceback area per
// resident wave, so the workspace -- not the chip -- bounds occupancy, and queries whose carry ring does not
// fit LDS (> ~3.3 kb) paid an L2 round trip per 4-step block for the ring in HBM scratch (282 ms for 256 pairs;
// profiles/r01_e_long_reads.txt).  Here W waves share one pair:
//   * wave w runs stripes w, w + W, w + 2W, ...; stripe k+1 trails stripe k by ~100 columns;
//   * the stripe carry (H and E of a stripe's last row: the reference's score[]/step[], sw_avx.cpp:36-47,
//     196-197) goes from the wave of stripe k to the wave of stripe k+1 through a small circular LDS ring
//     (RING_COLS columns) with a produced / consumed counter pair per ring -- whatever the query length, LDS
//     holds the query once plus W * 2 KB;
//   * sequence numbers: column c of the stripe consumed as stripe k has number (k / W) * S + c on ring
//     (k-1) mod W (S = a multiple of RING_COLS >= steps per stripe + slack), so `produced >= n` / `consumed >= n`
//     are single integer comparisons and ring slot = c mod RING_COLS in every round;
//   * the LAST wave's carry cannot go through a small ring: wave 0 only comes back for stripe W after it has
//     finished stripe 0, so that hop must hold a whole row or the back-pressure closes a cycle (deadlock).  It
//     goes through a per-pair row in HBM instead, with NO back-pressure and no fence: every column is one
//     64-bit agent-scope atomic {H : 32, H - E' : 16, tag : 16} (0 <= H - E' <= gap open - gap extend in the stored form for
//     every valid cell, sw.cpp:73-82; tag = round number, never 0, the row is zeroed when the kernel starts).  Wave 0 loads its
//     columns two 32-step groups ahead, checks the tags when it needs them (re-loading a column that is not
//     there yet) and drops them into its private LDS ring, from where the step code reads them like any other
//     carry;
//   * row 0 (the border row, sw.cpp:14-18,31-35) is written into that private ring by the wave of stripe 0
//     itself, 32 columns ahead of where it reads;
//   * the last stripe has no consumer: instead of publishing its last row it keeps the running best of
//     sw.cpp:116-127 in registers (the lane that owns row tl sees H[tl][j] for every j in order).
// Every spin is bounded (SPIN_LIMIT) so the grid drains even if a counter were wrong; the pair is then
// marked failed (DpRecord.sps = -1, reported as MGL_SW_ERR_DEVICE by the traceback kernel's status).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "sw_device.h"

namespace mgl_sw_dev {

namespace {

constexpr int DPP_WAVE_SHR1 = 0x138;
constexpr int RING_COLS = 256;     // columns per ring (8 bytes each)
constexpr int RING_MASK = RING_COLS - 1;
constexpr int AHEAD = 40;          // a 32-step group loads carry columns up to s + 39 (one block of prefetch)
constexpr int SPIN_LIMIT = 1 << 24;

__device__ __forceinline__ int wave_shr1(int lane0_value, int src)
{
    return __builtin_amdgcn_update_dpp(lane0_value, src, DPP_WAVE_SHR1, 0xf, 0xf, false);
}
__device__ __forceinline__ unsigned shift_in_sign(unsigned acc, int d)
{
    return __builtin_amdgcn_alignbit(acc, (unsigned)d, 31); // (acc << 1) | (d < 0)
}
__device__ __forceinline__ int border(int k, int gopen, int gext, bool indel)
{
    return (indel && k > 0) ? -gopen - (k - 1) * gext : 0; // sw.cpp:29-40,47-49
}

struct CoopLane {
    int h_prev, e_prev, hup, f;
    unsigned a0, a1, a2, a3; // traceback bit planes of the current 32-step block
    int best, best_i;        // last-column maximum over this lane's rows (ties: later row)
    int rm, rd, rj;          // last stripe only: running best of the last row (score, |tl-j|, j)
};

// Stored values are X[i][j] + (i + j) * gext, as in sw_dp_body (sw_kernels.hip): match / mismatch include the 2 * gext
// of a diagonal step, o_e = gap open - gap extend starts a gap in either direction, extensions are free.
struct CoopConsts {
    int match, mismatch, o_e, gopen, gext, tl, ql;
};

__device__ __forceinline__ bool wait_at_least(const int *ctr, int need)
{
    for (int spin = 0; spin < SPIN_LIMIT; ++spin) {
        const int v = __hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (__builtin_amdgcn_readfirstlane(v) >= need) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false; // (uniform: every lane saw the same counter values)
}

__device__ __forceinline__ unsigned long long wrap_load(const unsigned long long *row, int col, int cols)
{
    return __hip_atomic_load(row + min(col, cols - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Wave 0, stripes W, 2W, ...: put columns col_base + (0..31) of the row above (written by the last wave one
// round earlier, into the pair's HBM row) into the private LDS ring.  v was loaded two groups ago; a column
// whose tag is not this round's has not been written yet and is loaded again.
// ring entry formats: int2 {H, E'} in true values (int32 kernel); unsigned {H mod 2^16 | E' mod 2^16 << 16} (16-bit kernel: the
// consumer subtracts its baseline modulo 2^16, and what it gets is exact because the true difference fits its window)
__device__ __forceinline__ void ring_put(int2 *ring, int col, int h, int e) { ring[col & RING_MASK] = make_int2(h, e); }
__device__ __forceinline__ void ring_put(unsigned *ring, int col, int h, int e) { ring[col & RING_MASK] = ((unsigned)h & 0xffffu) | ((unsigned)e << 16); }
template <typename RingT>
__device__ __forceinline__ bool stage_wrap(unsigned long long v, const int col, const bool active, const int ql,
                                           const unsigned tag, const unsigned long long *row, const int cols, RingT *ring)
{
    bool ok = true;
    for (int spin = 0;; ++spin) {
        const bool missing = active && col <= ql && (unsigned)(v >> 48) != tag;
        if (!__builtin_amdgcn_ballot_w64(missing)) break;
        if (spin >= SPIN_LIMIT) {
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        if (missing) v = wrap_load(row, col, cols);
    }
    if (active) {
        const int h = (int)(unsigned)v;
        ring_put(ring, col, h, h - (int)((unsigned)(v >> 32) & 0xffffu));
    }
    return ok;
}

enum { OUT_RING = 0, OUT_WRAP = 1, OUT_LAST = 2 };

// 32 anti-diagonal steps of one 64-row stripe.
//   EDGE : some lane may be at a column <= 0 (forced border) or at its last column (capture H[i][ql])
//   OUT  : where the carry of this stripe's last row goes: the LDS ring of the next wave, the pair's HBM row
//          (last wave of the workgroup), or nowhere (the stripe that holds row tl: running last-row best instead)
template <bool EDGE, int OUT>
__device__ __forceinline__ void coop_group32(CoopLane &st, int4 &rA, int4 &rB, const int2 *ring_in, int2 *ring_out,
                                             unsigned long long *wrap_out, const unsigned tag_out, const unsigned *qrd,
                                             unsigned &q_lo, const int q_shift, const int tb, const int s_begin,
                                             const int L, const int hb, const int qcap, const int row_i,
                                             const int cap_unshift, const CoopConsts &c, const bool writer)
{
#pragma unroll 1
    for (int b = 0; b < 8; ++b) {
        const int s0 = s_begin + 4 * b;
        // bases of the four columns of this block: bytes (s0 - L + 63) .. +3 of the padded query
        const unsigned q_hi = qrd[(s0 >> 2) + 1];
        const unsigned qw = __builtin_amdgcn_alignbyte(q_hi, q_lo, (unsigned)q_shift);
        q_lo = q_hi;
        // carry of the NEXT block (all lanes read the same two addresses), reloaded as soon as the registers are free
        const int4 *nxt = reinterpret_cast<const int4 *>(ring_in + ((s0 + 4) & RING_MASK));
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int rh = u == 0 ? rA.x : u == 1 ? rA.z : u == 2 ? rB.x : rB.z;
            const int re = u == 0 ? rA.y : u == 1 ? rA.w : u == 2 ? rB.y : rB.w;
            if (u == 2) rA = nxt[0];
            const int hup_new = wave_shr1(rh, st.h_prev);
            const int ein = wave_shr1(re, st.e_prev);
            const int qb = (int)((qw >> (8 * u)) & 0xffu);
            const int diag = st.hup + (qb == tb ? c.match : c.mismatch);
            const int d1 = diag - st.f; // < 0 <=> F > diag
            const int sm = max(diag, st.f);
            const int d2 = sm - ein; // < 0 <=> E > max(diag, F)
            int h = max(sm, ein);
            const int open_from = h - c.o_e;
            const int d3 = ein - open_from; // < 0 <=> a new vertical gap beats extending
            const int eo = max(open_from, ein);
            const int d4 = st.f - open_from; // < 0 <=> a new horizontal gap beats extending
            int fo = max(open_from, st.f);
            const int j = s0 + u - L; // this lane's column
            if (EDGE) {
                const bool at_border = j <= 0;
                h = at_border ? hb : h;
                fo = at_border ? hb - c.o_e : fo;
                const int score = h - cap_unshift;            // rows carry different offsets: compare scores
                const bool take = j == qcap && score >= st.best; // sw.cpp:100-104 (>=: later row wins)
                st.best = take ? score : st.best;
                st.best_i = take ? row_i : st.best_i;
            }
            st.a0 = shift_in_sign(st.a0, d1);
            st.a1 = shift_in_sign(st.a1, d2);
            st.a2 = shift_in_sign(st.a2, d3);
            st.a3 = shift_in_sign(st.a3, d4);
            if (OUT == OUT_LAST) {
                // sw.cpp:116-127 in column order: better score, or same score closer to the diagonal
                const int d = abs(c.tl - j);
                const int score = h - (c.tl + j) * c.gext;
                const bool take = j >= 1 && j <= c.ql && (score > st.rm || (score == st.rm && d < st.rd));
                st.rm = take ? score : st.rm;
                st.rd = take ? d : st.rd;
                st.rj = take ? j : st.rj;
            } else if (!EDGE || s0 + u >= 63) {
                // lane 63 finishes column s - 63 of this stripe's last row
                const int col = s0 + u - 63;
                if (OUT == OUT_RING) {
                    if (writer) ring_out[col & RING_MASK] = make_int2(h, eo);
                } else {
                    // {H, H - E' (16 bits), tag}: in the lean part lane 63 is on a valid cell, 0 <= H - E' <= gap open - gap extend
                    const unsigned hi = EDGE ? (((unsigned)(h - eo) & 0xffffu) | tag_out) : (unsigned)(h - eo) + tag_out;
                    if (writer)
                        __hip_atomic_store(wrap_out + col, (unsigned long long)(unsigned)h | ((unsigned long long)hi << 32),
                                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            st.h_prev = h;
            st.e_prev = eo;
            st.hup = hup_new;
            st.f = fo;
        }
        rB = nxt[1];
    }
}

__device__ __forceinline__ unsigned wrap_tag(int consumer_stripe, int W) { return (unsigned)((consumer_stripe / W) & 0x7fff) + 1u; }

// the int32 kernel's body (also the fall-back of sw_dp_coop16_kernel below)
__device__ __forceinline__ void coop32_body(const DpArgs &a, unsigned char *smem)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); // wave-uniform, and the compiler knows it
    const int W = blockDim.x >> 6;
    const int L = lane;
    const int64_t slot = blockIdx.x;
    const int64_t p = a.first + slot;

    const int64_t t0 = a.t.off[p], q0 = a.q.off[p];
    const int tl = a.t.length(p);
    const int ql = a.q.length(p);
    const int nstripes = (tl + 63) >> 6;
    const int sps = coop_sps_for(ql);                  // steps per stripe, a multiple of 32
    const int S = (sps + 64 + RING_MASK) & ~RING_MASK; // sequence numbers per stripe
    const int main_lo = 64, main_hi = ql & ~31;        // groups inside [main_lo, main_hi) touch no edge

    // LDS: query bytes (64 zero bytes, q, zeros) | W rings | produced[W] | consumed[W] | per-wave results
    const int qbytes = coop_query_bytes(a.sps_cap);
    unsigned char *qbuf = smem;
    int2 *rings = reinterpret_cast<int2 *>(smem + qbytes);
    int *produced = reinterpret_cast<int *>(rings + (size_t)W * RING_COLS);
    int *consumed = produced + W;
    int *wres = consumed + W; // [W][2] last-column best, then [4] last row {rm, rd, rj, failed}

    // the pair's row in HBM for the hop last wave -> wave 0
    const int wrap_cols = coop_wrap_cols(a.sps_cap);
    unsigned long long *wrap = reinterpret_cast<unsigned long long *>(a.scratch) + (size_t)slot * wrap_cols;

    {
        unsigned *qz = reinterpret_cast<unsigned *>(qbuf);
        for (int w = threadIdx.x; w < (qbytes >> 2); w += blockDim.x) qz[w] = 0u;
        if ((int)threadIdx.x < 2 * W) produced[threadIdx.x] = 0; // produced[] and consumed[] are contiguous
        if (threadIdx.x < 4) wres[2 * W + threadIdx.x] = threadIdx.x == 0 ? NEG_INF : threadIdx.x == 3 ? 0 : 0x7fffffff;
        if (nstripes > W)
            for (int x = threadIdx.x; x < wrap_cols; x += blockDim.x) wrap[x] = 0ull; // tag 0 = not written
        __threadfence();
        __syncthreads();
        for (int x = threadIdx.x; x < ql; x += blockDim.x) qbuf[64 + x] = (unsigned char)a.q.at(q0, x);
        __syncthreads();
    }

    CoopConsts c;
    c.match = a.match + 2 * a.gext;
    c.mismatch = a.mismatch + 2 * a.gext;
    c.o_e = a.gopen - a.gext;
    asm volatile("" : "+v"(c.match), "+v"(c.mismatch)); // both feed a v_cndmask every step
    c.gopen = a.gopen;
    c.gext = a.gext;
    c.tl = tl;
    c.ql = ql;
    const bool indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;

    CoopLane st;
    st.h_prev = st.e_prev = st.hup = st.f = 0;
    st.a0 = st.a1 = st.a2 = st.a3 = 0u;
    st.best = NEG_INF;
    st.best_i = -1;
    st.rm = NEG_INF;
    st.rd = 0x7fffffff;
    st.rj = 0x7fffffff;
    int failed = 0; // wave-uniform

    const int q_shift = (63 - L) & 3;
    const unsigned *qrd = reinterpret_cast<const unsigned *>(qbuf) + ((63 - L) >> 2);
    const bool writer = (L == 63);
    const int last_lane = (tl - 1) & 63;
    // wave 0 feeds itself (border row or HBM row) through ring W-1; the others read the ring of the wave before
    const int b_in = (wave + W - 1) % W, b_out = wave;
    int2 *ring_in = rings + (size_t)b_in * RING_COLS;
    int2 *ring_out = rings + (size_t)b_out * RING_COLS;

    for (int k = wave; k < nstripes; k += W) {
        const bool first = (k == 0), last = (k == nstripes - 1);
        const int out = last ? OUT_LAST : (wave == W - 1 ? OUT_WRAP : OUT_RING);
        const int row_i = k * 64 + 1 + L;
        const int tb = row_i <= tl ? a.t.at(t0, row_i - 1) : 0;
        const int hb = border(row_i, c.gopen, c.gext, indel) + row_i * c.gext; // column 0
        const int cap_unshift = (row_i + ql) * c.gext;                          // offset of this row's last column
        const int qcap = row_i <= tl ? ql : NEG_INF;
        const int base_in = (k / W) * S, base_out = ((k + 1) / W) * S;
        const unsigned tag_in = wrap_tag(k, W), tag_out = wrap_tag(k + 1, W) << 16;
        uint32_t *tbp = a.tb + (size_t)slot * a.tb_stride_words + ((size_t)k * (sps >> 5) * 64 + L) * 4;

        int4 rA = make_int4(0, 0, 0, 0), rB = rA;
        unsigned q_lo = qrd[0];
        unsigned long long pend_a = 0, pend_b = 0; // wave 0: HBM-row columns of the next two groups
        for (int s = 0; s < sps; s += 32) {
            // ---- carry in: columns < s + AHEAD of the row above this stripe
            if (wave == 0) {
                if (first) {
                    // row 0: H[0][j], E[1][j] = H[0][j] - o
                    for (int col = (s == 0 ? 0 : s + AHEAD - 32) + L; col < s + AHEAD; col += 64) {
                        const int hb0 = border(col, c.gopen, c.gext, indel) + col * c.gext;
                        ring_in[col & RING_MASK] = make_int2(hb0, hb0 - c.o_e);
                    }
                } else if (s == 0) {
                    const unsigned long long v = wrap_load(wrap, L, wrap_cols);
                    pend_a = wrap_load(wrap, AHEAD + L, wrap_cols);
                    pend_b = wrap_load(wrap, AHEAD + 32 + L, wrap_cols);
                    if (!failed) failed = !stage_wrap(v, L, true, ql, tag_in, wrap, wrap_cols, ring_in);
                } else {
                    const int col = s + AHEAD - 32 + L;
                    if (!failed) failed = !stage_wrap(pend_a, col, L < 32, ql, tag_in, wrap, wrap_cols, ring_in);
                    pend_a = pend_b;
                    pend_b = wrap_load(wrap, col + 64, wrap_cols);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else if (!failed) {
                failed = !wait_at_least(produced + b_in, base_in + s + AHEAD);
            }
            // ---- carry out through a ring: the slots of columns (s - 63) .. (s - 32) must have been read one lap ago
            if (out == OUT_RING && !failed && s + 31 - 63 >= 0) {
                const int need = base_out + s + 31 - 63 - RING_MASK;
                if (need > 0) failed = !wait_at_least(consumed + b_out, need);
            }
            if (s == 0) {
                const int4 *r0 = reinterpret_cast<const int4 *>(ring_in);
                rA = r0[0];
                rB = r0[1];
            }
            const bool lean = s >= main_lo && s + 32 <= main_hi;
#define MGL_COOP_GROUP(EDGE, OUT)                                                                                      \
    coop_group32<EDGE, OUT>(st, rA, rB, ring_in, ring_out, wrap, tag_out, qrd, q_lo, q_shift, tb, s, L, hb, qcap, row_i, cap_unshift, c, \
                            writer)
            if (out == OUT_RING) {
                if (lean)
                    MGL_COOP_GROUP(false, OUT_RING);
                else
                    MGL_COOP_GROUP(true, OUT_RING);
            } else if (out == OUT_WRAP) {
                if (lean)
                    MGL_COOP_GROUP(false, OUT_WRAP);
                else
                    MGL_COOP_GROUP(true, OUT_WRAP);
            } else {
                if (lean)
                    MGL_COOP_GROUP(false, OUT_LAST);
                else
                    MGL_COOP_GROUP(true, OUT_LAST);
            }
#undef MGL_COOP_GROUP
            *reinterpret_cast<uint4 *>(tbp) = make_uint4(st.a0, st.a1, st.a2, st.a3);
            tbp += 256;
            // ---- publish progress (the release orders lane 63's ring stores before the counter)
            if (L == 0) {
                if (out == OUT_RING && s + 32 - 63 > 0)
                    __hip_atomic_store(produced + b_out, base_out + s + 32 - 63, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (wave != 0)
                    __hip_atomic_store(consumed + b_in, base_in + s + 32, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (L == 0) {
            if (out == OUT_RING)
                __hip_atomic_store(produced + b_out, base_out + S, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave != 0) __hip_atomic_store(consumed + b_in, base_in + S, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // the lane that owns row tl holds the last-row best
        if (last && L == last_lane) {
            wres[2 * W + 0] = st.rm;
            wres[2 * W + 1] = st.rd;
            wres[2 * W + 2] = st.rj;
        }
    }

    // ---- last column: reduce over the lanes of this wave, then over the waves (ties: larger row)
    int mqe = st.best, mqe_t = st.best_i;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int ob = __shfl_xor(mqe, m), oi = __shfl_xor(mqe_t, m);
        const bool take = ob > mqe || (ob == mqe && oi > mqe_t);
        mqe = take ? ob : mqe;
        mqe_t = take ? oi : mqe_t;
    }
    if (L == 0) {
        wres[2 * wave] = mqe;
        wres[2 * wave + 1] = mqe_t;
        if (failed) wres[2 * W + 3] = 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < W; ++w) {
            const int ob = wres[2 * w], oi = wres[2 * w + 1];
            const bool take = ob > mqe || (ob == mqe && oi > mqe_t);
            mqe = take ? ob : mqe;
            mqe_t = take ? oi : mqe_t;
        }
        const int rm = wres[2 * W], rd = wres[2 * W + 1], rj = wres[2 * W + 2];
        // sequential rule of sw.cpp:116-127 starting from (mqe, mqe_t, ql)
        const bool row_wins = rm > mqe || (rm == mqe && rd < abs(mqe_t - ql));
        DpRecord r;
        r.mqe = mqe;
        r.mqe_t = mqe_t;
        r.max = row_wins ? rm : mqe;
        r.max_t = row_wins ? tl : mqe_t;
        r.max_q = row_wins ? rj : ql;
        r.seg = row_wins ? ql - rj : 0;
        r.g_tail = 0;
        r.sps = wres[2 * W + 3] ? -1 : sps;
        a.rec[slot] = r;
    }
}

// =====================================================================================================================
// 16-bit form of the long-read kernel: 128 target rows per wave, two per lane, packed in the low / high half of every register.
//
// A wave runs a "double stripe" of 128 target rows as 128 processing elements (PE p = row k*128 + p + 1, at column s - p in
// step s); lane l holds PE 2l in the low halves of its registers and PE 2l+1 in the high halves.  Handing H and E' to the
// next PE is then one DPP move plus one v_alignbit: the new low half is the high half of the lane before, the new high
// half is the lane's own low half; PE 0 takes the row above from the carry ring, PE 127 (lane 63, high) feeds the next
// wave's.  Scores reach 2 * 10^6 on 10 kb reads, so a wave keeps its values relative to a BASELINE B (wave-uniform int32):
//     stored16 = X[i][j] + (i + j) * e - B                           (X = H, E or F; the int32 kernel's form minus B)
// and moves B every 32 steps.  Exactness does not rest on an a-priori range claim: at every move the wave reduces the exact
// minimum and maximum of H over its 128 PEs and checks that the window they span, widened by what 34 more steps can add
// (coop16_below / coop16_above), fits 16 bits; the widening uses only one-step facts of the recurrence (sw.cpp:60-93), in
// stored units and for ANY sequences -- the padded rows > tl and columns > ql are cells of a larger matrix and obey them too:
//     E'[i+1][j], F'[i][j+1] in [H - (o-e), H];   H[i-1][j] (the next step's diagonal) in [H - (match+o+e), H + o];
//     per step the maximum rises by at most match + o + e (match + 2e inside the wave; the larger figure covers PE 0, whose
//     inputs come from outside: H[i][j+1] <= H[i][j] + match + o + e for any cell), the minimum falls by at most o - e.
// If the check ever fails the pair is marked and the WORKGROUP redoes it with the int32 body (coop32_body) -- never a wrong
// answer.  For GATK parameters the 128 PEs span <= 127 * (match + 2e) + 2 * (o - e) ~ 28.7 k against a limit of ~39 k, so the
// fall-back is for other parameter sets; the host does not try when that estimate cannot fit (coop16_worthwhile).
// Hand-over between waves, the HBM hop, tags and counters are those of the int32 kernel, in true (int32) values: the
// consumer subtracts its own baseline.  Traceback layout ("coop16", DpRecord.g_tail == -16): per double stripe and 16
// steps one uint4 per lane, dword = 4 steps in the byte layout of sw_dp16.hip (byte0 = low half {E>S, F opened}, byte1 =
// high half, byte2 = low {F>diag, E opened}, byte3 = high; step t in bits 2t+1, 2t): 4 bits per cell, 1 KB per store.

typedef short short2c_t __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2c_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned c_pk_add(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(ushort2c_t, a) + __builtin_bit_cast(ushort2c_t, b)); }
__device__ __forceinline__ unsigned c_pk_sub(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, __builtin_bit_cast(ushort2c_t, a) - __builtin_bit_cast(ushort2c_t, b)); }
__device__ __forceinline__ unsigned c_pk_sub_sat(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_sub_sat(__builtin_bit_cast(short2c_t, a), __builtin_bit_cast(short2c_t, b)));
}
__device__ __forceinline__ unsigned c_pk_max(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(short2c_t, a), __builtin_bit_cast(short2c_t, b)));
}
__device__ __forceinline__ unsigned c_pk_min(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(short2c_t, a), __builtin_bit_cast(short2c_t, b)));
}
__device__ __forceinline__ unsigned c_pk_min_u(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(__builtin_bit_cast(ushort2c_t, a), __builtin_bit_cast(ushort2c_t, b)));
}
__device__ __forceinline__ unsigned c_pk_mad(unsigned a, unsigned b, unsigned c)
{
    return __builtin_bit_cast(unsigned, __builtin_bit_cast(ushort2c_t, a) * __builtin_bit_cast(ushort2c_t, b) + __builtin_bit_cast(ushort2c_t, c));
}
__device__ __forceinline__ unsigned c_pack2(int lo, int hi) { return ((unsigned)lo & 0xffffu) | ((unsigned)hi << 16); }
__device__ __forceinline__ int c_lo16(unsigned x) { return (int)(short)(x & 0xffffu); }
__device__ __forceinline__ int c_hi16(unsigned x) { return (int)x >> 16; }
__device__ __forceinline__ unsigned c_and_or(unsigned a, unsigned k, unsigned b)
{
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(b));
    return r;
}

// What the window check adds below the minimum and above the maximum of H found when the baseline moves.  Between two
// moves lie 32 steps; the registers also hold values up to two steps old, hence 34:
//   below: 34 steps of falling (o-e each), the diagonal input (match+o+e) and the E' input of PE 0 another
//          (match+o+e) + (o-e) under the H they feed, and the intermediate diag = hup + (mismatch + 2e)
//   above: 34 steps of rising (match+o+e, see above), the diagonal input (o) and diag = hup + match + 2e
__host__ __device__ inline int coop16_below(int match, int mismatch, int gopen, int gext)
{
    const int mis2 = mismatch + 2 * gext;
    return 34 * (gopen - gext) + 2 * (match + gopen + gext) + (gopen - gext) + (mis2 < 0 ? -mis2 : mis2) + 64;
}
__host__ __device__ inline int coop16_above(int match, int gopen, int gext) { return 34 * (match + gopen + gext) + gopen + (match + 2 * gext) + 64; }

struct Coop16Lane {
    unsigned h_prev, e_prev, hup, f; // PE 2l | PE 2l+1, stored form
    unsigned w[3];                   // traceback dwords of the last three 4-step blocks
    unsigned keep;                   // PE 127's column of the previous block's last step (ring stores go four aligned columns at a time)
    int best_lo, best_lo_i, best_hi, best_hi_i; // last-column maxima of this lane's two rows (true scores)
    int rm, rd, rj;                  // the double stripe that holds row tl: running best of the last row
};

struct Coop16Consts {
    unsigned delta, one, o_e, k2; // packed (both halves equal): mismatch - match, 1, o - e, match + 2e
    unsigned k12[4], k34[4];      // SGPR bit masks of the four steps of a traceback dword
    unsigned qsel[4];             // v_perm selectors: the query bytes of step u for the low (byte u+1) and high (byte u) PE
    int o_e32, gopen, gext, tl, ql;
    int floor16, check_margin;    // where a move of the baseline puts the minimum; what the window check adds to the spread
};

// wave-wide minimum and maximum over both halves of x in all lanes (wave-uniform)
__device__ __forceinline__ void wave_minmax_pk(unsigned x, int &mn_out, int &mx_out)
{
    unsigned mn = x, mx = x;
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const unsigned a = (unsigned)__shfl_xor((int)mn, m), b = (unsigned)__shfl_xor((int)mx, m);
        mn = c_pk_min(mn, a);
        mx = c_pk_max(mx, b);
    }
    const unsigned smn = (unsigned)__builtin_amdgcn_readfirstlane((int)mn), smx = (unsigned)__builtin_amdgcn_readfirstlane((int)mx);
    mn_out = min(c_lo16(smn), c_hi16(smn));
    mx_out = max(c_lo16(smx), c_hi16(smx));
}

// 32 anti-diagonal steps of one 128-row double stripe.  EDGE / OUT as in coop_group32.  B: the baseline.
template <bool EDGE, int OUT>
__device__ __forceinline__ void coop16_group32(Coop16Lane &st, uint4 &rA, const unsigned *ring_in, unsigned *ring_out,
                                               unsigned long long *wrap_out, const unsigned tag_out, const unsigned *qrd, unsigned &q_lo,
                                               const unsigned tt, const int s_begin, const int L, const unsigned hb, const int qcap_lo,
                                               const int qcap_hi, const int row_lo, const int B, const Coop16Consts &c, const bool writer,
                                               const int last_half, uint4 *&tbp, const int vzero)
{
    const unsigned negB16 = 0u - ((unsigned)B << 16);
#pragma unroll 1
    for (int b = 0; b < 8; ++b) {
        const int s0 = s_begin + 4 * b;
        // the five query bytes of this block: columns s0 - 2L - 1 (high PE, first step) .. s0 - 2L + 3 (low PE, last step) start a dword
        // of this lane's copy of the query (even lanes read the copy shifted by two bytes)
        const unsigned q_hi = qrd[(s0 >> 2) + 1];
        unsigned wnew = 0u, po[4];
        // this block's four carry columns, one dword each, and the NEXT block's (every lane reads the same address; `vzero` keeps
        // the values in vector registers), loaded here and used one block later
        const uint4 cur = rA;
        rA = *(reinterpret_cast<const uint4 *>(ring_in + ((s0 + 4) & RING_MASK)) + vzero);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned rp = u == 0 ? cur.x : u == 1 ? cur.y : u == 2 ? cur.z : cur.w;
            // ---- shift H and E' by one PE; PE 0 takes the carry of the row above minus the baseline, modulo 2^16 (only the high
            // half of either operand is used)
            unsigned in_h = (rp << 16) + negB16, in_e = rp + negB16;
            if (EDGE && s0 + u > c.ql) {
                // PE 0 is past the last column: the carry row ends there; the cells it keeps computing feed nothing, but they must
                // stay inside the window, so they see their own last values as the row above
                in_h = st.h_prev << 16;
                in_e = st.e_prev << 16;
            }
            const unsigned hup_new = __builtin_amdgcn_alignbit(st.h_prev, (unsigned)wave_shr1((int)in_h, (int)st.h_prev), 16);
            const unsigned ein = __builtin_amdgcn_alignbit(st.e_prev, (unsigned)wave_shr1((int)in_e, (int)st.e_prev), 16);
            // ---- the cell, both halves (sw_dp16.hip's cell16)
            const unsigned q = __builtin_amdgcn_perm(q_hi, q_lo, c.qsel[u]);
            const unsigned m = c_pk_min_u(q ^ tt, c.one);
            const unsigned sc = c_pk_mad(m, c.delta, c.k2);
            const unsigned diag = c_pk_add(st.hup, sc);
            const unsigned sm = c_pk_max(diag, st.f);
            unsigned h = c_pk_max(sm, ein);
            const unsigned open = c_pk_sub(h, c.o_e);
            const unsigned eo = c_pk_max(open, ein);
            unsigned fo = c_pk_max(open, st.f);
            const unsigned d1 = c_pk_sub_sat(diag, st.f), d2 = c_pk_sub_sat(sm, ein), d3 = c_pk_sub_sat(ein, open), d4 = c_pk_sub_sat(st.f, open);
            const unsigned p12 = __builtin_amdgcn_perm(d1, d2, 0x0b0a0908u), p34 = __builtin_amdgcn_perm(d3, d4, 0x0b0a0908u);
            wnew = c_and_or(p34, c.k34[u], u == 0 ? (p12 & c.k12[0]) : c_and_or(p12, c.k12[u], wnew));
            const int j_lo = s0 + u - 2 * L, j_hi = j_lo - 1; // this lane's columns
            if (EDGE) {
                // column <= 0: the border values H[i][0], F[i][1] (sw.cpp:24,38,47-49), relative to the baseline
                const unsigned hbr = c_pk_sub(hb, c_pack2(B, B));
                const unsigned hbf = c_pk_sub(hbr, c.o_e);
                const unsigned msk = (j_lo <= 0 ? 0x0000ffffu : 0u) | (j_hi <= 0 ? 0xffff0000u : 0u);
                h = (hbr & msk) | (h & ~msk);
                fo = (hbf & msk) | (fo & ~msk);
                // last column of this lane's rows (sw.cpp:100-104: >= so the later row wins): true scores
                const int sc_lo = c_lo16(h) + B - (row_lo + c.ql) * c.gext, sc_hi = c_hi16(h) + B - (row_lo + 1 + c.ql) * c.gext;
                const bool t_lo = j_lo == qcap_lo && sc_lo >= st.best_lo, t_hi = j_hi == qcap_hi && sc_hi >= st.best_hi;
                st.best_lo = t_lo ? sc_lo : st.best_lo;
                st.best_lo_i = t_lo ? row_lo : st.best_lo_i;
                st.best_hi = t_hi ? sc_hi : st.best_hi;
                st.best_hi_i = t_hi ? row_lo + 1 : st.best_hi_i;
            }
            if (OUT == OUT_LAST) {
                // the half that holds row tl: sw.cpp:116-127 in column order -- better score, or same score closer to the diagonal
                const int j = last_half ? j_hi : j_lo;
                const int d = abs(c.tl - j);
                const int score = (last_half ? c_hi16(h) : c_lo16(h)) + B - (c.tl + j) * c.gext;
                const bool take = j >= 1 && j <= c.ql && (score > st.rm || (score == st.rm && d < st.rd));
                st.rm = take ? score : st.rm;
                st.rd = take ? d : st.rd;
                st.rj = take ? j : st.rj;
            } else if (OUT == OUT_WRAP) {
                // PE 127 (lane 63, high half) finishes column s - 127 of the double stripe's last row: hand it on in true values
                const int col = s0 + u - 127;
                const unsigned diff = (unsigned)(c_hi16(h) - c_hi16(eo)) & 0xffffu; // 0 <= H - E' <= o - e
                if (writer && (!EDGE || col >= 0))
                    __hip_atomic_store(wrap_out + col, (unsigned long long)(unsigned)(c_hi16(h) + B) | ((unsigned long long)(diff | tag_out) << 32),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                // PE 127's column, {H | E'} of the high halves as residues of the true values
                po[u] = c_pk_add(__builtin_amdgcn_perm(eo, h, 0x07060302u), c_pack2(B, B));
            }
            st.h_prev = h;
            st.e_prev = eo;
            st.hup = hup_new;
            st.f = fo;
        }
        q_lo = q_hi;
        if (OUT == OUT_RING) {
            // columns s0 - 128 .. s0 - 125 of the last row: the previous block's last step and three of this block's, one aligned 16 bytes
            if (writer && (!EDGE || s0 >= 128))
                *reinterpret_cast<uint4 *>(ring_out + ((s0 - 128) & RING_MASK)) = make_uint4(st.keep, po[0], po[1], po[2]);
            st.keep = po[3];
        }
        // the dwords of 16 steps rotate through three registers (a dynamic register index would go through scratch, and
        // sixteen steps unrolled spill: the compiler hoists the ring loads of all four blocks)
        if ((b & 3) == 3) {
            *tbp = make_uint4(st.w[0], st.w[1], st.w[2], wnew); // 1 KB per wave
            tbp += 64;
        }
        st.w[0] = st.w[1];
        st.w[1] = st.w[2];
        st.w[2] = wnew;
    }
}

// returns false (workgroup-uniform) when some window check failed: the pair must be redone in 32 bits
__device__ __forceinline__ bool coop16_body(const DpArgs &a, unsigned char *smem)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int W = blockDim.x >> 6;
    const int L = lane;
    const int64_t slot = blockIdx.x;
    const int64_t p = a.first + slot;

    const int64_t t0 = a.t.off[p], q0 = a.q.off[p];
    const int tl = a.t.length(p);
    const int ql = a.q.length(p);
    const int nds = (tl + 127) >> 7;                    // double stripes
    const int sps = coop16_sps_for(ql);                 // steps per double stripe, a multiple of 32
    const int S = (sps + 64 + RING_MASK) & ~RING_MASK;  // sequence numbers per double stripe
    const int main_lo = 128, main_hi = ql & ~31;        // groups inside [main_lo, main_hi) touch no edge

    // LDS: two copies of the query bytes (128 zero bytes, q, zeros; the second shifted by two bytes) | W rings | produced[W] |
    // consumed[W] | per-wave results
    const int qbytes = coop_query_bytes(a.sps_cap);
    unsigned char *qbuf = smem;
    int2 *rings = reinterpret_cast<int2 *>(smem + 2 * qbytes); // (sized for the int32 body; this one uses 4 bytes per column)
    int *produced = reinterpret_cast<int *>(rings + (size_t)W * RING_COLS);
    int *consumed = produced + W;
    int *wres = consumed + W; // [W][2] last-column best, then [5] last row {rm, rd, rj, failed, needs32}

    const int wrap_cols = coop_wrap_cols(a.sps_cap);
    unsigned long long *wrap = reinterpret_cast<unsigned long long *>(a.scratch) + (size_t)slot * wrap_cols;
    {
        unsigned *qz = reinterpret_cast<unsigned *>(qbuf);
        for (int w = threadIdx.x; w < (qbytes >> 1); w += blockDim.x) qz[w] = 0u;
        if ((int)threadIdx.x < 2 * W) produced[threadIdx.x] = 0;
        if (threadIdx.x < 5) wres[2 * W + threadIdx.x] = threadIdx.x == 0 ? NEG_INF : threadIdx.x >= 3 ? 0 : 0x7fffffff;
        if (nds > W)
            for (int x = threadIdx.x; x < wrap_cols; x += blockDim.x) wrap[x] = 0ull; // tag 0 = not written
        __threadfence();
        __syncthreads();
        for (int x = threadIdx.x; x < ql; x += blockDim.x) {
            const unsigned char ch = (unsigned char)a.q.at(q0, x);
            qbuf[128 + x] = ch;          // column j at byte 127 + j
            qbuf[qbytes + 126 + x] = ch; // the same, two bytes earlier
        }
        __syncthreads();
    }

    const int match = a.match, gopen = a.gopen, gext = a.gext;
    Coop16Consts c;
    c.delta = c_pack2(a.mismatch - match, a.mismatch - match);
    c.one = c_pack2(1, 1);
    c.o_e = c_pack2(gopen - gext, gopen - gext);
    c.k2 = c_pack2(match + 2 * gext, match + 2 * gext);
    asm volatile("" : "+v"(c.delta), "+v"(c.one), "+v"(c.o_e), "+v"(c.k2));
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        c.k12[u] = 0x02020202u << (2 * u);
        c.k34[u] = 0x01010101u << (2 * u);
        c.qsel[u] = 0x0c000c00u | (unsigned)(u + 1) | ((unsigned)u << 16);
        asm volatile("" : "+s"(c.k12[u]), "+s"(c.k34[u]), "+s"(c.qsel[u]));
    }
    c.o_e32 = gopen - gext;
    c.gopen = gopen;
    c.gext = gext;
    c.tl = tl;
    c.ql = ql;
    const int below = coop16_below(match, a.mismatch, gopen, gext), above = coop16_above(match, gopen, gext);
    c.floor16 = -32768 + below;
    c.check_margin = below + above;
    const bool indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;

    Coop16Lane st;
    st.h_prev = st.e_prev = st.hup = st.f = 0u;
    st.w[0] = st.w[1] = st.w[2] = 0u;
    st.keep = 0u;
    st.best_lo = st.best_hi = NEG_INF;
    st.best_lo_i = st.best_hi_i = -1;
    st.rm = NEG_INF;
    st.rd = 0x7fffffff;
    st.rj = 0x7fffffff;
    int failed = 0, needs32 = 0; // wave-uniform
    int vzero = 0;
    asm volatile("" : "+v"(vzero));

    // this lane's query dwords: block s0 starts at byte 128 + s0 - 2L - 2 of the first copy, a multiple of four for odd L; even
    // lanes read the same bytes two positions earlier in the shifted copy
    const unsigned *qrd = reinterpret_cast<const unsigned *>(qbuf + ((L & 1) ? 0 : qbytes)) + ((128 - 2 * L - 2 - ((L & 1) ? 0 : 2)) >> 2);
    const bool writer = (L == 63);
    const int last_lane = ((tl - 1) & 127) >> 1, last_half = (tl - 1) & 1;
    const int b_in = (wave + W - 1) % W, b_out = wave;
    unsigned *ring_in = reinterpret_cast<unsigned *>(rings) + (size_t)b_in * RING_COLS;
    unsigned *ring_out = reinterpret_cast<unsigned *>(rings) + (size_t)b_out * RING_COLS;

    for (int k = wave; k < nds; k += W) {
        const bool first = (k == 0), last = (k == nds - 1);
        const int out = last ? OUT_LAST : (wave == W - 1 ? OUT_WRAP : OUT_RING);
        const int row_lo = k * 128 + 1 + 2 * L, row_hi = row_lo + 1;
        const unsigned tt = (unsigned)(row_lo <= tl ? a.t.at(t0, row_lo - 1) : 0) | ((unsigned)(row_hi <= tl ? a.t.at(t0, row_hi - 1) : 0) << 16);
        const int hb_lo = border(row_lo, gopen, gext, indel) + row_lo * gext, hb_hi = border(row_hi, gopen, gext, indel) + row_hi * gext; // column 0
        const int qcap_lo = row_lo <= tl ? ql : NEG_INF, qcap_hi = row_hi <= tl ? ql : NEG_INF;
        const int base_in = (k / W) * S, base_out = ((k + 1) / W) * S;
        const unsigned tag_in = wrap_tag(k, W), tag_out = wrap_tag(k + 1, W) << 16;
        uint4 *tbp = reinterpret_cast<uint4 *>(a.tb + (size_t)slot * a.tb_stride_words) + (size_t)k * (sps >> 4) * 64 + L;

        // every PE starts on its column-0 border value; the baseline starts on PE 0's
        int B = __builtin_amdgcn_readfirstlane(hb_lo);
        const unsigned hb_res = c_pack2(hb_lo, hb_hi); // the 16-bit residues of the border values: minus B (mod 2^16) they are exact
        st.h_prev = c_pack2(hb_lo - B, hb_hi - B);
        st.hup = st.h_prev;
        st.e_prev = st.f = c_pk_sub(st.h_prev, c.o_e);

        uint4 rA = make_uint4(0u, 0u, 0u, 0u);
        unsigned q_lo = qrd[0];
        unsigned long long pend_a = 0, pend_b = 0;
        for (int s = 0; s < sps; s += 32) {
            // ---- move the baseline: exact minimum / maximum of H over the 128 PEs, window check, re-base the four state registers
            {
                int lo, hi;
                wave_minmax_pk(st.h_prev, lo, hi);
                if ((hi - lo) + c.check_margin > 65535) needs32 = 1;
                const int d = lo - c.floor16;
                const unsigned dd = c_pack2(d, d);
                B += d;
                st.h_prev = c_pk_sub(st.h_prev, dd);
                st.e_prev = c_pk_sub(st.e_prev, dd);
                st.hup = c_pk_sub(st.hup, dd);
                st.f = c_pk_sub(st.f, dd);
            }
            // ---- carry in: columns < s + AHEAD of the row above this double stripe
            if (wave == 0) {
                if (first) {
                    for (int col = (s == 0 ? 0 : s + AHEAD - 32) + L; col < s + AHEAD; col += 64) {
                        const int hb0 = border(col, gopen, gext, indel) + col * gext;
                        ring_put(ring_in, col, hb0, hb0 - c.o_e32);
                    }
                } else if (s == 0) {
                    const unsigned long long v = wrap_load(wrap, L, wrap_cols);
                    pend_a = wrap_load(wrap, AHEAD + L, wrap_cols);
                    pend_b = wrap_load(wrap, AHEAD + 32 + L, wrap_cols);
                    if (!failed) failed = !stage_wrap(v, L, true, ql, tag_in, wrap, wrap_cols, ring_in);
                } else {
                    const int col = s + AHEAD - 32 + L;
                    if (!failed) failed = !stage_wrap(pend_a, col, L < 32, ql, tag_in, wrap, wrap_cols, ring_in);
                    pend_a = pend_b;
                    pend_b = wrap_load(wrap, col + 64, wrap_cols);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            } else if (!failed) {
                failed = !wait_at_least(produced + b_in, base_in + s + AHEAD);
            }
            // ---- carry out through a ring: this group stores columns up to s - 97; their slots must have been read one lap ago
            if (out == OUT_RING && !failed && s - 97 >= 0) {
                const int need = base_out + s - 97 - RING_MASK;
                if (need > 0) failed = !wait_at_least(consumed + b_out, need);
            }
            if (s == 0) {
                rA = *(reinterpret_cast<const uint4 *>(ring_in) + vzero);
            }
            const bool lean = s >= main_lo && s + 32 <= main_hi;
#define MGL_COOP16_GROUP(EDGE, OUT)                                                                                                \
    coop16_group32<EDGE, OUT>(st, rA, ring_in, ring_out, wrap, tag_out, qrd, q_lo, tt, s, L, hb_res, qcap_lo, qcap_hi, row_lo, B, c, \
                              writer, last_half, tbp, vzero)
            if (out == OUT_RING) {
                if (lean)
                    MGL_COOP16_GROUP(false, OUT_RING);
                else
                    MGL_COOP16_GROUP(true, OUT_RING);
            } else if (out == OUT_WRAP) {
                if (lean)
                    MGL_COOP16_GROUP(false, OUT_WRAP);
                else
                    MGL_COOP16_GROUP(true, OUT_WRAP);
            } else {
                if (lean)
                    MGL_COOP16_GROUP(false, OUT_LAST);
                else
                    MGL_COOP16_GROUP(true, OUT_LAST);
            }
#undef MGL_COOP16_GROUP
            // ---- publish progress (the release orders lane 63's ring stores before the counter)
            if (L == 0) {
                if (out == OUT_RING && s + 32 - 128 > 0)
                    __hip_atomic_store(produced + b_out, base_out + s + 32 - 128, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (wave != 0)
                    __hip_atomic_store(consumed + b_in, base_in + s + 32, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
        }
        if (L == 0) {
            if (out == OUT_RING)
                __hip_atomic_store(produced + b_out, base_out + S, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (wave != 0) __hip_atomic_store(consumed + b_in, base_in + S, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        if (last && L == last_lane) {
            wres[2 * W + 0] = st.rm;
            wres[2 * W + 1] = st.rd;
            wres[2 * W + 2] = st.rj;
        }
    }

    // ---- last column: this lane's two candidates, then the lanes of the wave, then the waves (ties: larger row)
    int mqe = st.best_lo, mqe_t = st.best_lo_i;
    {
        const bool take = st.best_hi > mqe || (st.best_hi == mqe && st.best_hi_i > mqe_t);
        mqe = take ? st.best_hi : mqe;
        mqe_t = take ? st.best_hi_i : mqe_t;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) {
        const int ob = __shfl_xor(mqe, m), oi = __shfl_xor(mqe_t, m);
        const bool take = ob > mqe || (ob == mqe && oi > mqe_t);
        mqe = take ? ob : mqe;
        mqe_t = take ? oi : mqe_t;
    }
    if (L == 0) {
        wres[2 * wave] = mqe;
        wres[2 * wave + 1] = mqe_t;
        if (failed) wres[2 * W + 3] = 1;
        if (needs32) wres[2 * W + 4] = 1;
    }
    __syncthreads();
    const bool ok16 = wres[2 * W + 4] == 0;
    if (threadIdx.x == 0 && ok16) {
        for (int w = 1; w < W; ++w) {
            const int ob = wres[2 * w], oi = wres[2 * w + 1];
            const bool take = ob > mqe || (ob == mqe && oi > mqe_t);
            mqe = take ? ob : mqe;
            mqe_t = take ? oi : mqe_t;
        }
        const int rm = wres[2 * W], rd = wres[2 * W + 1], rj = wres[2 * W + 2];
        const bool row_wins = rm > mqe || (rm == mqe && rd < abs(mqe_t - ql));
        DpRecord r;
        r.mqe = mqe;
        r.mqe_t = mqe_t;
        r.max = row_wins ? rm : mqe;
        r.max_t = row_wins ? tl : mqe_t;
        r.max_q = row_wins ? rj : ql;
        r.seg = row_wins ? ql - rj : 0;
        r.g_tail = -16; // the coop16 traceback layout
        r.sps = wres[2 * W + 3] ? -1 : sps;
        a.rec[slot] = r;
    }
    __syncthreads();
    return ok16;
}

} // namespace

// grid = pairs of the chunk, block = 64 * W threads: the 16-bit form, and the int32 body for a pair whose window check failed
__global__ __launch_bounds__(1024) void sw_dp_coop16_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!coop16_body(a, smem)) coop32_body(a, smem);
}

// grid = pairs of the chunk, block = 64 * W threads
__global__ __launch_bounds__(1024) void sw_dp_coop_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    coop32_body(a, smem);
}

int coop_lds_bytes(int sps_cap, int waves_per_block)
{
    return coop_query_bytes(sps_cap) + waves_per_block * (RING_COLS * 8 + 8 + 8) + 32;
}

// Is the 16-bit form worth trying for these (normalised) parameters?  The 64 PEs of a half-stripe span about
// 64 * (match + 2e) (+ 2(o - e)); with the window check's margins that must stay inside 16 bits, else every pair would only
// fall back.  (Exactness never depends on this estimate: the kernel checks the real window.)
int coop16_lds_bytes(int sps_cap, int waves_per_block) { return coop_lds_bytes(sps_cap, waves_per_block) + coop_query_bytes(sps_cap); } // two query copies

// Can the kernel run at all: its packed constants and the margins of the window check must fit 16 bits
bool coop16_possible(int match, int mismatch, int gopen, int gext)
{
    if (match <= 0 || mismatch > match || gext < 0 || gopen < gext || match > 4000 || mismatch < -4000 || gopen > 4000 || gext > 4000) return false;
    return coop16_below(match, mismatch, gopen, gext) + coop16_above(match, gopen, gext) <= 60000;
}
bool coop16_worthwhile(int match, int mismatch, int gopen, int gext)
{
    if (!coop16_possible(match, mismatch, gopen, gext)) return false;
    const int margin = coop16_below(match, mismatch, gopen, gext) + coop16_above(match, gopen, gext);
    return 128 * (match + 2 * gext) + 2 * (gopen - gext) + margin <= 64000;
}

hipError_t launch_dp_coop16(const DpArgs &a, int waves_per_block, hipStream_t stream)
{
    const int lds = coop16_lds_bytes(a.sps_cap, waves_per_block);
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sw_dp_coop16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sw_dp_coop16_kernel, dim3((unsigned)a.count), dim3(64 * waves_per_block), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_dp_coop(const DpArgs &a, int waves_per_block, hipStream_t stream)
{
    const int lds = coop_lds_bytes(a.sps_cap, waves_per_block);
    if (lds > 64 * 1024) { // per device and rare: set every time rather than cache across devices / threads
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(sw_dp_coop_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(sw_dp_coop_kernel, dim3((unsigned)a.count), dim3(64 * waves_per_block), lds, stream, a);
    return hipGetLastError();
}

} // namespace mgl_sw_dev
