// sw_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for mgl's Smith-Waterman affine-gap core.
//
// Written for wave64 CDNA4 only; there is no other backend.
//
// What is computed (the function defined by the reference's sw.cpp:5-255, which its AVX2
// path sw_avx.cpp:110-322 reproduces cell for cell; paths relative to
// /root/reference/src/main/native/mgl_sw/):
//
//   diag = H[i-1][j-1] + (t[i-1]==q[j-1] ? match : mismatch)                 sw.cpp:55
//   H[i][j] = diag if diag>=E && diag>=F, else F if F>=E, else E             sw.cpp:60-71
//   E[i+1][j] = max(H-o, E-e)  (extension wins ties)                         sw.cpp:73-82
//   F[i][j+1] = max(H-o, F-e)  (extension wins ties)                         sw.cpp:84-93
//   no zero floor; maxima over last column / last row only                   sw.cpp:100-127
//
// Mapping onto the machine
// ------------------------
// * One pair per 16-lane DPP row ("group"), four pairs per wave64.  Lane L of a group owns
//   one target row of a 16-row stripe; a step advances every lane one query column along an
//   anti-diagonal (lane L is at column j = s - L), the AVX2 path's scheme (sw_avx.cpp:11,
//   71-80) widened from 8 to 16 rows.  16 rather than 64 rows per stripe keeps the
//   pipeline fill/drain at 15 steps per ql+16 (91 % lane use at ql = 150; 64 rows would
//   give 70 %), and `row_shr:1` DPP moves H and E to the next row with no LDS traffic.
// * The stripe carry (H and E of a stripe's last row, per column: the reference's
//   score[]/step[] arrays, sw_avx.cpp:36-47) lives in an LDS ring per group; the lane that
//   owns the stripe's last row (lane 15, or lane (tl-1)%16 in a partial final stripe -- the
//   reference's actual_bw, sw_avx.cpp:74,196) writes column j when it gets there, lane 0 of
//   the next stripe reads it at step j.  Rows past tl in the final stripe compute garbage
//   that only ever flows to higher (also padding) lanes, as in sw_avx.cpp:154-156.
// * The query is staged in LDS as four byte-shifted copies so that every lane fetches the
//   four bases of a 4-step block with one aligned ds_read_b32.
// * Traceback: 4 bits per cell {F>diag, E>max(diag,F), E opened, F opened} instead of the
//   reference's int32 run lengths (equivalent: a run length is 1 + the number of
//   consecutive "extended" decisions behind the cell, sw.cpp:73-93).  Each lane shifts the
//   sign bits of four differences into four 32-step accumulators (v_alignbit) and stores
//   16 bytes every 32 steps, so a group's stores are 256 contiguous bytes.
// * A second kernel walks the path (one pair per lane) and writes offset, ScoreMax and the
//   CIGAR text.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_device.h"
#include "sw_traceback.h"

namespace mgl_sw_dev {

constexpr int DPP_ROW_SHR1 = 0x111;

constexpr int DPP_WAVE_SHR1 = 0x138;

// every lane takes src from the lane below it inside its group of G lanes (G = 16: one DPP row, row_shr:1;
// G = 64: the whole wave, wave_shr:1); lane 0 of the group keeps lane0_value (the stripe carry)
template <int G>
__device__ __forceinline__ int lane_shr1(int lane0_value, int src)
{
    return __builtin_amdgcn_update_dpp(lane0_value, src, G == 16 ? DPP_ROW_SHR1 : DPP_WAVE_SHR1, 0xf, 0xf, false);
}

__device__ __forceinline__ unsigned shift_in_sign(unsigned acc, int d)
{
    // (acc << 1) | (d < 0)
    return __builtin_amdgcn_alignbit(acc, (unsigned)d, 31);
}

__device__ __forceinline__ int border(int k, int gopen, int gext, bool indel)
{
    // H on row 0 / column 0: sw.cpp:29-40,47-49
    return (indel && k > 0) ? -gopen - (k - 1) * gext : 0;
}

struct LaneState {
    int h_prev, e_prev; // this lane's H and E' of the previous step (shifted to lane+1 next step)
    int hup;            // H[i-1][j] as of the previous step == H[i-1][j-1] for this step
    int f;              // F entering this step
    unsigned a0, a1, a2, a3; // traceback bit planes
    int best, best_i;   // running last-column maximum of this lane's rows (ties: later row)
};

// One block of four anti-diagonal steps.
//   PRO   : some lane may still be at column <= 0 (forced border values)
//   EPI   : some lane may be at its last column (capture H[i][ql])
// ringA / ringB hold the carry of columns s0, s0+1 / s0+2, s0+3; they are reloaded for the
// next block as soon as their last use is behind (no register rotation at the loop edge).
#ifdef MGL_ABLATE_TBSTORE
#define MGL_ABLATE_TBSTORE_V 1
#else
#define MGL_ABLATE_TBSTORE_V 0
#endif
#ifdef MGL_ABLATE_QREAD
#define MGL_QREAD(p) (qw * 1664525u + 1013904223u)
#else
#define MGL_QREAD(p) (Carry<SCRATCH>::loadq((p) + 1))
#endif

// Where the per-group carry ring and query copies live: LDS (normal), or -- when a query is too long for the
// LDS carve -- a scratch area in HBM that the same wave writes and reads back.  The scratch accesses are
// agent-scope relaxed atomics (global_load/store ... sc1): they are served by L2, so a ring entry written by
// lane 15 is what lane 0 reads a stripe later whatever the CU's L1 still holds.
template <bool SCRATCH>
struct Carry {
    static __device__ __forceinline__ int4 load2cols(const int4 *p)
    {
        if (!SCRATCH) return *p;
        const unsigned long long *q = reinterpret_cast<const unsigned long long *>(p);
        const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_int4((int)a, (int)(a >> 32), (int)b, (int)(b >> 32));
    }
    static __device__ __forceinline__ int2 load1col(const int2 *p)
    {
        if (!SCRATCH) return *p;
        const unsigned long long a =
            __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return make_int2((int)a, (int)(a >> 32));
    }
    static __device__ __forceinline__ void store1col(int2 *p, int h, int e)
    {
        if (!SCRATCH) {
            *p = make_int2(h, e);
            return;
        }
        const unsigned long long v = (unsigned long long)(unsigned)h | ((unsigned long long)(unsigned)e << 32);
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    static __device__ __forceinline__ unsigned loadq(const unsigned *p)
    {
        if (!SCRATCH) return *p;
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
};

template <bool PRO, bool EPI, bool SCRATCH, int G, bool MATRIX = false>
__device__ __forceinline__ void step4(LaneState &st, int4 &ringA, int4 &ringB, const int4 *ring_next,
                                      const unsigned qw, const int tb, const int s0, const int L, const int hb,
                                      const int qcap, const int row_i, const int match2, const int mismatch2,
                                      const int o_e, const int cap_unshift, int2 *ring_wr, const bool writer,
                                      const int (&sub)[4] = {0, 0, 0, 0})
{
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int rh = u == 0 ? ringA.x : u == 1 ? ringA.z : u == 2 ? ringB.x : ringB.z;
        const int re = u == 0 ? ringA.y : u == 1 ? ringA.w : u == 2 ? ringB.y : ringB.w;
#ifndef MGL_ABLATE_RINGREAD
        if (u == 2) ringA = Carry<SCRATCH>::load2cols(ring_next);
#endif
        const int hup_new = lane_shr1<G>(rh, st.h_prev);
        const int ein = lane_shr1<G>(re, st.e_prev);
        const int qb = (int)((qw >> (8 * u)) & 0xffu);
        // stored values carry the offset (i + j) * gext (see sw_dp_body): the diagonal adds 2 * gext (folded into match2 /
        // mismatch2 / sub), extending a gap costs nothing, and both new gaps start from the same H - (o - e)
        // substitution-matrix mode (protein extension): the score was looked up one block ahead
        const int diag = st.hup + (MATRIX ? sub[u] : (qb == tb ? match2 : mismatch2));
        const int d1 = diag - st.f; // < 0 <=> F > diag
        const int sm = max(diag, st.f);
        const int d2 = sm - ein; // < 0 <=> E > max(diag, F)
        int h = max(sm, ein);
        const int open_from = h - o_e;
        const int d3 = ein - open_from; // < 0 <=> a new vertical gap beats extending
        int eo = max(open_from, ein);
        const int d4 = st.f - open_from; // < 0 <=> a new horizontal gap beats extending
        int fo = max(open_from, st.f);
        if (PRO) {
            const bool at_border = (s0 + u) <= L; // column j = s - L <= 0
            h = at_border ? hb : h;
            fo = at_border ? hb - o_e : fo;
        }
#ifndef MGL_ABLATE_TB
        st.a0 = shift_in_sign(st.a0, d1);
        st.a1 = shift_in_sign(st.a1, d2);
        st.a2 = shift_in_sign(st.a2, d3);
        st.a3 = shift_in_sign(st.a3, d4);
#endif
        if (EPI) {
            const bool last_col = (s0 + u - L) == qcap;
            const int score = h - cap_unshift;           // rows carry different offsets: compare scores
            const bool take = last_col && score >= st.best; // sw.cpp:100-104 (>=: later row wins)
            st.best = take ? score : st.best;
            st.best_i = take ? row_i : st.best_i;
        }
#ifndef MGL_ABLATE_RINGWRITE
        if (writer) Carry<SCRATCH>::store1col(ring_wr + u, h, eo);
#endif
        st.h_prev = h;
        st.e_prev = eo;
        st.hup = hup_new;
        st.f = fo;
    }
#ifndef MGL_ABLATE_RINGREAD
    ringB = Carry<SCRATCH>::load2cols(ring_next + 1);
#endif
}

// G = target rows per stripe = lanes per pair: 16 (four pairs per wave; short reads: little fill/drain per
// stripe) or 64 (one pair per wave, the classic 64-row wavefront; long reads: 4x the waves for the same
// traceback memory, fill/drain 63/(ql+64)).
template <int G, bool SCRATCH, bool MATRIX = false>
__device__ __forceinline__ void sw_dp_body(const DpArgs &a, unsigned char *smem, const signed char *mat_lds = nullptr)
{
    constexpr int PW = 64 / G; // pairs per wave
    unsigned long long diag_t0 = 0, diag_w0 = 0;
    if (a.diag) { // in-kernel clock probe (MI355X_MICROARCH.md, DVFS give-back item 6); off in normal runs
        diag_t0 = __builtin_amdgcn_s_memtime();
        diag_w0 = __builtin_amdgcn_s_memrealtime();
    }

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int grp = lane / G;
    const int L = lane % G;
    const int64_t slot = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * PW + grp;
    // a wave with no pair at all leaves; in a partly filled wave the idle groups recompute the
    // last pair (same wave => same geometry) but never store anything
    if (slot - grp >= a.count) return;
    const bool valid = slot < a.count;
    const int64_t p = a.first + (valid ? slot : a.count - 1);

    const int64_t t0 = a.t.off[p], q0 = a.q.off[p];
    const int tl = a.t.length(p);
    const int ql = a.q.length(p);

    const int nstripes = (tl + G - 1) / G;

    // wave-uniform loop bounds over the groups of the wave
    int ql_max = ql, ql_min = ql, ns_max = nstripes;
#pragma unroll
    for (int m = G; m < 64; m <<= 1) {
        ql_max = max(ql_max, __shfl_xor(ql_max, m));
        ql_min = min(ql_min, __shfl_xor(ql_min, m));
        ns_max = max(ns_max, __shfl_xor(ns_max, m));
    }
    ql_max = __builtin_amdgcn_readfirstlane(ql_max);
    ql_min = __builtin_amdgcn_readfirstlane(ql_min);
    ns_max = __builtin_amdgcn_readfirstlane(ns_max);
    const int sps = sps_for_rows(ql_max, G);                     // steps per stripe
    const int main_end = max(G, ql_min & ~3);                    // [G, main_end): no border, no last column

    // carve per group: ring[sps_cap + G + 4] int2 | 4 x qcopy[sps_cap + G + 16] bytes
    // (the slack covers columns -G..-1 and the one-block-ahead prefetches)
    const int ring_entries = dp_ring_entries(a.sps_cap, G);
    const int qcopy_bytes = dp_qcopy_bytes(a.sps_cap, G);
    const int group_bytes = ring_entries * 8 + 4 * qcopy_bytes;
    unsigned char *gbase =
        SCRATCH ? a.scratch + ((size_t)blockIdx.x * (blockDim.x >> 6) * PW + (size_t)(wave * PW + grp)) * group_bytes
                : smem + (size_t)(wave * PW + grp) * group_bytes;
    int2 *ring = reinterpret_cast<int2 *>(gbase); // ring[j + G] holds column j
    unsigned char *qcopy = gbase + ring_entries * 8;

    // match / mismatch feed a v_cndmask every step: pin them in VGPRs (two SGPR sources
    // would be re-materialised into VGPRs each step)
    // Stored values are X[i][j] + (i + j) * gext (X = H, E, F): every comparison is between values of one cell, so the
    // decisions are those of sw.cpp:51-96, but E - e and F - e need no instruction and the two gap-open candidates
    // are one value.  Scores are restored where they are read (last column, last row).
    const int gopen = a.gopen, gext = a.gext;
    const int o_e = gopen - gext;
    int match = a.match + 2 * gext, mismatch = a.mismatch + 2 * gext;
    asm volatile("" : "+v"(match), "+v"(mismatch));
    const bool indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;

    // ---- stage the query: copy k holds q shifted right by G + k bytes, zero elsewhere
    {
        unsigned *qz = reinterpret_cast<unsigned *>(qcopy);
        for (int w = L; w < qcopy_bytes; w += G) qz[w] = 0u; // 4 * qcopy_bytes bytes == qcopy_bytes dwords
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int x = L; x < ql; x += G) {
            unsigned char c = (unsigned char)a.q.at(q0, x);
            if (MATRIX) c = a.code[c]; // the copies hold matrix column codes
#pragma unroll
            for (int k = 0; k < 4; ++k) qcopy[k * qcopy_bytes + x + G + k] = c;
        }
        // ---- border row into the ring: H[0][j], E[1][j] = H[0][j] - o   (sw.cpp:14-18,31-35)
        for (int j = L; j <= ql_max; j += G) {
            const int hb0 = border(j, gopen, gext, indel) + j * gext;   // row 0; E[1][j] = H[0][j] - o sits one row lower
            Carry<SCRATCH>::store1col(ring + j + G, hb0, hb0 - o_e);
        }
        if (SCRATCH)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // the plain byte stores above reach L2 before any read
        else
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    LaneState st;
    st.h_prev = st.e_prev = st.hup = st.f = 0;
    st.a0 = st.a1 = st.a2 = st.a3 = 0u;
    st.best = NEG_INF;
    st.best_i = -1;

    const int last_lane = (tl - 1) % G; // owner of row tl in the final stripe
    const int qk = (L + 1) & 3, qm = (L + 1) >> 2;
    const unsigned *qrd0 = reinterpret_cast<const unsigned *>(qcopy + qk * qcopy_bytes) + (G / 4 - qm);

    uint32_t *tbp = a.tb + (size_t)(valid ? slot : a.count - 1) * a.tb_stride_words + L * 4;
    int gsteps = 0; // wave-uniform global step counter (traceback bit position)

    int row_next = 1 + L; // row of this lane in stripe 0
    // matrix mode: tb is the byte offset of the target base's matrix row
    int tb_next = (row_next >= 1 && row_next <= tl) ? a.t.at(t0, row_next - 1) : 0;
    if (MATRIX) tb_next = a.code[tb_next & 0xff] * MATRIX_DIM;

    for (int k = 0; k < ns_max; ++k) {
        const int row_i = row_next;
        const int tb = tb_next;
        row_next += G;
        tb_next = (row_next >= 1 && row_next <= tl) ? a.t.at(t0, row_next - 1) : 0;
        if (MATRIX) tb_next = a.code[tb_next & 0xff] * MATRIX_DIM;

        const int hb = border(row_i, gopen, gext, indel) + row_i * gext; // column 0
        const int cap_unshift = (row_i + ql) * gext;                      // offset of this row's last column
        const int qcap = row_i <= tl ? ql : NEG_INF; // s - L > -G never equals NEG_INF
        // the lane owning this stripe's last row publishes the carry (sw_avx.cpp:196-197)
        const int wl = (k == nstripes - 1) ? last_lane : G - 1;
        const bool writer = (L == wl);

        const int4 *ring_rd = reinterpret_cast<const int4 *>(ring + G);
        int2 *ring_wr = ring + G - wl; // at step s the writer is at column s - wl
        const unsigned *qrd = qrd0;
        int4 rA = Carry<SCRATCH>::load2cols(ring_rd), rB = Carry<SCRATCH>::load2cols(ring_rd + 1);
        unsigned qw = Carry<SCRATCH>::loadq(qrd);
        int sub[4] = {0, 0, 0, 0}; // matrix mode: scores of this block's four columns, looked up one block ahead
        if (MATRIX) {
#pragma unroll
            for (int u = 0; u < 4; ++u) sub[u] = mat_lds[tb + (int)((qw >> (8 * u)) & 0xffu)] + 2 * gext;
        }

        int s = 0;
#define MGL_SW_BLOCK(PRO, EPI)                                                                             \
    {                                                                                                      \
        const unsigned nq = MGL_QREAD(qrd);                                                                \
        int nsub[4] = {0, 0, 0, 0};                                                                        \
        if (MATRIX) {                                                                                      \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) nsub[u] = mat_lds[tb + (int)((nq >> (8 * u)) & 0xffu)] + 2 * gext; \
        }                                                                                                  \
        step4<PRO, EPI, SCRATCH, G, MATRIX>(st, rA, rB, ring_rd + 2, qw, tb, s, L, hb, qcap, row_i, match, \
                                            mismatch, o_e, cap_unshift, ring_wr, writer, sub);             \
        if (MATRIX) {                                                                                      \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) sub[u] = nsub[u];                               \
        }                                                                                                  \
        qw = nq;                                                                                           \
        ring_rd += 2;                                                                                      \
        ring_wr += 4;                                                                                      \
        qrd += 1;                                                                                          \
        s += 4;                                                                                            \
        gsteps += 4;                                                                                       \
        if ((gsteps & 31) == 0) {                                                                          \
            if (valid && !MGL_ABLATE_TBSTORE_V) *reinterpret_cast<uint4 *>(tbp) = make_uint4(st.a0, st.a1, st.a2, st.a3); \
            tbp += G * 4;                                                                                  \
        }                                                                                                  \
    }
        for (; s < G;) MGL_SW_BLOCK(true, true)
        for (; s < main_end;) MGL_SW_BLOCK(false, false)
        for (; s < sps;) MGL_SW_BLOCK(false, true)
#undef MGL_SW_BLOCK

        if (k == nstripes - 1) {
            // ---- this group's matrix is complete: last column max, last row scan (sw.cpp:100-127)
            if (!SCRATCH) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            int mqe = st.best, mqe_t = st.best_i;
#pragma unroll
            for (int m = 1; m < G; m <<= 1) {
                const int ob = __shfl_xor(mqe, m), oi = __shfl_xor(mqe_t, m);
                const bool take = ob > mqe || (ob == mqe && oi > mqe_t);
                mqe = take ? ob : mqe;
                mqe_t = take ? oi : mqe_t;
            }
            // last row: best score, then closest to the diagonal, then smallest column
            int rm = NEG_INF, rd = 0x7fffffff, rj = 0x7fffffff;
            for (int j = L + 1; j <= ql; j += G) {
                const int sc = Carry<SCRATCH>::load1col(ring + j + G).x - (tl + j) * gext;
                const int d = abs(tl - j);
                const bool take = sc > rm || (sc == rm && d < rd);
                rm = take ? sc : rm;
                rd = take ? d : rd;
                rj = take ? j : rj;
            }
#pragma unroll
            for (int m = 1; m < G; m <<= 1) {
                const int om = __shfl_xor(rm, m), od = __shfl_xor(rd, m), oj = __shfl_xor(rj, m);
                const bool take = om > rm || (om == rm && (od < rd || (od == rd && oj < rj)));
                rm = take ? om : rm;
                rd = take ? od : rd;
                rj = take ? oj : rj;
            }
            // sequential rule of sw.cpp:116-127 starting from (mqe, mqe_t, ql)
            const bool row_wins = rm > mqe || (rm == mqe && rd < abs(mqe_t - ql));
            if (L == 0 && valid) {
                DpRecord r;
                r.mqe = mqe;
                r.mqe_t = mqe_t;
                r.max = row_wins ? rm : mqe;
                r.max_t = row_wins ? tl : mqe_t;
                r.max_q = row_wins ? rj : ql;
                r.seg = row_wins ? ql - rj : 0;
                r.g_tail = 0;
                r.sps = sps;
                a.rec[slot] = r;
            }
        }
    }
    // flush the partial traceback block, left-aligned like the full ones
    const int rem = gsteps & 31;
    if (rem && valid) {
        const int sh = 32 - rem;
        *reinterpret_cast<uint4 *>(tbp) = make_uint4(st.a0 << sh, st.a1 << sh, st.a2 << sh, st.a3 << sh);
    }
    if (a.diag && threadIdx.x == 0) {
        a.diag[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - diag_t0;
        a.diag[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - diag_w0;
    }
}

__global__ __launch_bounds__(256) void sw_dp_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    sw_dp_body<16, false>(a, smem);
}

// substitution-matrix scoring (protein extension, SURVEY.md 8f rank 4 -- no reference path): the 32 x 32 int8 matrix
// sits in LDS behind the per-group carves; a lane looks its four scores of a block up one block ahead
__global__ __launch_bounds__(256) void sw_dp_matrix_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    signed char *mat = reinterpret_cast<signed char *>(smem) + a.matrix_lds_offset;
    for (int x = threadIdx.x; x < MATRIX_DIM * MATRIX_DIM; x += blockDim.x) mat[x] = a.matrix[x];
    __syncthreads();
    sw_dp_body<16, false, true>(a, smem, mat);
}

__global__ __launch_bounds__(256) void sw_dp64_matrix_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    signed char *mat = reinterpret_cast<signed char *>(smem) + a.matrix_lds_offset;
    for (int x = threadIdx.x; x < MATRIX_DIM * MATRIX_DIM; x += blockDim.x) mat[x] = a.matrix[x];
    __syncthreads();
    sw_dp_body<64, false, true>(a, smem, mat);
}

// long queries: the carry ring and the query copies do not fit LDS and live in an HBM scratch area
__global__ __launch_bounds__(256) void sw_dp_scratch_kernel(const DpArgs a) { sw_dp_body<16, true>(a, nullptr); }

// one pair per wave (64-row stripes): long reads
__global__ __launch_bounds__(256) void sw_dp64_kernel(const DpArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    sw_dp_body<64, false>(a, smem);
}
__global__ __launch_bounds__(256) void sw_dp64_scratch_kernel(const DpArgs a) { sw_dp_body<64, true>(a, nullptr); }

__global__ __launch_bounds__(256) void sw_traceback_kernel(const TbArgs a)
{
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.count) return;
    const int64_t p = a.first + slot;
    const int tl = a.t.length(p);
    const int ql = a.q.length(p);
    const DpRecord r = a.rec[slot];
    if (r.sps < 0) {
        // the fill kernel gave up on this pair (sw_dp_coop_kernel: a bounded wait ran out)
        const int64_t o = a.dest ? a.dest[p] : p;
        a.offset[o] = 0;
        if (a.cigar_len) a.cigar_len[o] = 0;
        if (a.status) a.status[o] = ERR_DEVICE;
        if (a.status_any) atomicMax(a.status_any, ERR_DEVICE);
        for (int k = 0; k < a.cigar_stride; ++k) a.cigar[(size_t)o * a.cigar_stride + k] = 0;
        return;
    }

    TbView view;
    view.base = a.tb + (size_t)(a.packed16 == 2 ? slot >> 7 : a.packed16 == 1 ? slot >> 1 : slot) * a.tb_stride_words;
    view.set_schedule(r, ql, a.rows_per_stripe);
    view.packed16 = TbView::layout_of(a.packed16, r);
    view.waves = TbView::waves_of(r);
    view.half = (int)(slot & 1);
    view.lane = (int)((slot >> 1) & 63);
    view.ql = a.uni_ql;
    traceback_one_pair(a, view, r, p, tl, ql);
}

// ---------------------------------------------------------------------------------------------
// Long reads: ONE WAVE per pair.  With one lane per pair every path step of a 10 kb x 10 kb pair is a
// dependent 16-byte load from a 50 MB area (~1 us each, 20 ms per batch whatever its size).  Here all 64
// lanes run the same walk (wave-uniform state, scalar branches); a lane keeps ITS 16 bytes of the current
// 32-step traceback block in registers (one coalesced 1 KB load per block), the cell of lane l is fetched
// with v_readlane, and the block below (where an up-left path goes next inside a stripe) is already on its
// way; a run of diagonal moves inside those two blocks is taken in one go (diag_run).  int32 layout only (rows = 16 or 64).
struct WaveMoves {
    const uint4 *base; // this pair's traceback, as [block][rows] x 16 bytes
    int sps, rows, L;
    int cur_blk, nxt_blk;
    uint4 cur, nxt;
    __device__ __forceinline__ uint4 load(int blk) const
    {
        return L < rows ? base[(size_t)blk * rows + L] : make_uint4(0u, 0u, 0u, 0u);
    }
    __device__ __forceinline__ void init(const uint32_t *words, int sps_, int rows_, int lane)
    {
        base = reinterpret_cast<const uint4 *>(words);
        sps = sps_;
        rows = rows_;
        L = lane;
        cur_blk = nxt_blk = -1;
        cur = nxt = make_uint4(0u, 0u, 0u, 0u);
    }
    __device__ __forceinline__ unsigned cell(int i, int j)
    {
        const int r = i - 1;
        const int lane = r & (rows - 1); // rows is 16 or 64
        const int g = (r >> (rows == 64 ? 6 : 4)) * sps + j + lane;
        const int blk = g >> 5;
        if (blk != cur_blk) {
            cur = blk == nxt_blk ? nxt : load(blk);
            cur_blk = blk;
            nxt_blk = blk - 1;
            if (nxt_blk >= 0) nxt = load(nxt_blk);
        }
        const int sh = 31 - (g & 31);
        const unsigned x = (unsigned)__builtin_amdgcn_readlane((int)cur.x, lane), y = (unsigned)__builtin_amdgcn_readlane((int)cur.y, lane),
                       z = (unsigned)__builtin_amdgcn_readlane((int)cur.z, lane), w = (unsigned)__builtin_amdgcn_readlane((int)cur.w, lane);
        return ((x >> sh) & 1u) | (((y >> sh) & 1u) << 1) | (((z >> sh) & 1u) << 2) | (((w >> sh) & 1u) << 3);
    }
    __device__ __forceinline__ int at(int i, int j)
    {
        const unsigned c = cell(i, j);
        if (c & 2u) { // vertical gap: 1 + consecutive extensions above (sw.cpp:73-82)
            int n = 1;
            for (int r = i - 1; r >= 1 && !(cell(r, j) & 4u); --r) ++n;
            return n;
        }
        if (c & 1u) { // horizontal gap (sw.cpp:84-93)
            int n = 1;
            for (int q = j - 1; q >= 1 && !(cell(i, q) & 8u); --q) ++n;
            return -n;
        }
        return 0;
    }
    // Length of the run of diagonal moves that starts at (i, j), as far as it can be seen without loading anything:
    // cell k of the run, (i - k, j - k), belongs to lane (lane0 - k) of the same stripe and to global step g0 - 2k, i.e. to
    // the block in `cur` or the one below it in `nxt` -- every lane tests ITS cell of the run in its own registers and one
    // ballot finds the first cell that is not a diagonal move.  Returns 0 when (i, j) itself is not known to be one.
    // (Measured: 256 x 150 pairs 40 -> 16 us per launch, 10 kb x 10 kb 2.7 -> 1.1 ms per 256 pairs; a four-block window
    // fetched at once was no faster: 18 us / 1.6 ms.)
    __device__ __forceinline__ int diag_run(int i, int j)
    {
        const int r = i - 1;
        const int lane0 = r & (rows - 1);
        const int g0 = (r >> (rows == 64 ? 6 : 4)) * sps + j + lane0;
        if ((g0 >> 5) != cur_blk) return 0;
        const int k = lane0 - L;                 // this lane's position in the run (valid for 0 <= k)
        const int g = g0 - 2 * k;
        const int blk = g >> 5;
        const bool in_cur = blk == cur_blk, in_nxt = blk == nxt_blk && nxt_blk >= 0;
        const uint4 v = in_cur ? cur : nxt;
        const int sh = 31 - (g & 31);
        const bool is_diag = (((v.x | v.y) >> sh) & 1u) == 0u; // neither F > diag nor E > S (sw.cpp:60-71)
        // the run also ends at the matrix border: rows i - k >= 1 is k <= i - 1 (and k <= lane0 inside the stripe), columns k <= j - 1
        const bool ok = k >= 0 && k <= j - 1 && k <= i - 1 && (in_cur || in_nxt) && is_diag;
        const unsigned long long stop = __builtin_amdgcn_ballot_w64(!ok) & ((2ull << lane0) - 1ull); // lanes lane0, lane0-1, ..
        if (stop == 0ull) return lane0 + 1;      // every lane down to lane 0 continues the run
        return lane0 - (63 - __builtin_clzll(stop));
    }
};

// The same walk over the layout of sw_dp_coop16_kernel: a block is 16 steps of a 128-row double stripe, lane l holds the
// flags of PE l (low bytes) and PE 64 + l (high bytes); the block below is the same double stripe's previous 16 steps.
struct WaveMoves16 {
    const uint4 *base; // this pair's traceback, as [block][lane] x 16 bytes
    int bps, L;        // blocks per double stripe
    int cur_blk;       // win[d] = block cur_blk - d: the block of the current cell and the three below it, where an up-left path goes next
    uint4 win[4];
    __device__ __forceinline__ uint4 load(int blk) const { return blk >= 0 ? base[(size_t)blk * 64 + L] : make_uint4(0u, 0u, 0u, 0u); }
    __device__ __forceinline__ void init(const uint32_t *words, int sps, int lane)
    {
        base = reinterpret_cast<const uint4 *>(words);
        bps = sps >> 4;
        L = lane;
        cur_blk = -8;
        win[0] = win[1] = win[2] = win[3] = make_uint4(0u, 0u, 0u, 0u);
    }
    __device__ __forceinline__ static unsigned dword_of(const uint4 &v, int d) { return d == 0 ? v.x : d == 1 ? v.y : d == 2 ? v.z : v.w; }
    __device__ __forceinline__ static unsigned nibble(unsigned w, int h, int s)
    {
        const int t2 = (s & 3) * 2;
        const unsigned be = (w >> (8 * h)) >> t2, bf = (w >> (16 + 8 * h)) >> t2;
        return ((bf >> 1) & 1u) | (((be >> 1) & 1u) << 1) | ((bf & 1u) << 2) | ((be & 1u) << 3);
    }
    __device__ __forceinline__ void move_to(int blk) // wave-uniform
    {
        const int d = cur_blk - blk;
        if (d >= 1 && d <= 3) { // the window slides down: one new block per step, three blocks ahead of where the path is
            for (int t = 0; t < d; ++t) {
                win[0] = win[1];
                win[1] = win[2];
                win[2] = win[3];
                --cur_blk;
                win[3] = load(cur_blk - 3);
            }
        } else {
            cur_blk = blk;
#pragma unroll
            for (int t = 0; t < 4; ++t) win[t] = load(blk - t);
        }
    }
    __device__ __forceinline__ unsigned cell(int i, int j)
    {
        const int r = i - 1;
        const int pe = r & 127, s = j + pe;
        const int blk = (r >> 7) * bps + (s >> 4);
        if (blk != cur_blk) move_to(blk);
        const unsigned w = (unsigned)__builtin_amdgcn_readlane((int)dword_of(win[0], (s >> 2) & 3), pe >> 1);
        return nibble(w, pe & 1, s);
    }
    __device__ __forceinline__ int at(int i, int j)
    {
        const unsigned c = cell(i, j);
        if (c & 2u) {
            int n = 1;
            for (int r = i - 1; r >= 1 && !(cell(r, j) & 4u); --r) ++n;
            return n;
        }
        if (c & 1u) {
            int n = 1;
            for (int q = j - 1; q >= 1 && !(cell(i, q) & 8u); --q) ++n;
            return -n;
        }
        return 0;
    }
    // cell k of a diagonal run from (i, j) is PE pe0 - k at step s0 - 2k: lane l tests the two that are its own, PE 2l+1 and
    // PE 2l, in its registers (each block of the window holds 8 cells of the run); the run ends at the largest PE that fails
    __device__ __forceinline__ bool run_cell_ok(int i, int j, int r, int pe0, int s0, int pe) const
    {
        const int k = pe0 - pe, s = s0 - 2 * k;
        const bool inside = k >= 0 && k <= j - 1 && k <= i - 1; // (then s >= 1: the same double stripe)
        const int d = cur_blk - ((r >> 7) * bps + (s >> 4));
        const uint4 v = d == 0 ? win[0] : d == 1 ? win[1] : d == 2 ? win[2] : win[3];
        const bool is_diag = (nibble(dword_of(v, (s >> 2) & 3), pe & 1, s) & 3u) == 0u; // neither F > diag nor E > S (sw.cpp:60-71)
        return inside && d >= 0 && d <= 3 && is_diag;
    }
    __device__ __forceinline__ int diag_run(int i, int j)
    {
        const int r = i - 1;
        const int pe0 = r & 127, s0 = j + pe0;
        if ((r >> 7) * bps + (s0 >> 4) != cur_blk) return 0;
        // PEs above pe0 are not part of the run: they do not stop it
        const bool stop_hi = 2 * L + 1 <= pe0 && !run_cell_ok(i, j, r, pe0, s0, 2 * L + 1);
        const bool stop_lo = 2 * L <= pe0 && !run_cell_ok(i, j, r, pe0, s0, 2 * L);
        const unsigned long long mh = __builtin_amdgcn_ballot_w64(stop_hi), ml = __builtin_amdgcn_ballot_w64(stop_lo);
        const int top_hi = mh ? 2 * (63 - __builtin_clzll(mh)) + 1 : -1, top_lo = ml ? 2 * (63 - __builtin_clzll(ml)) : -1;
        return pe0 - max(top_hi, top_lo); // (no stop at all: the run reaches PE 0, pe0 + 1 cells)
    }
};

// The same walk over ANY layout TbView knows, one wave per pair, nothing kept in registers: lane k fetches cell k of what may come
// next -- the diagonal run from (i, j), or the gap run above / left of it -- and one ballot finds where it ends: up to 64 path
// steps per memory round trip (layout 4, sw_dp16_strip.hip: one lane per pair took 23 ms per 1 024 pairs of 10 kb).
struct WaveMovesView {
    TbView tb;
    int L;
    __device__ __forceinline__ int diag_run(int i, int j)
    {
        const bool inside = L <= i - 1 && L <= j - 1;
        const bool ok = inside && (tb.cell(inside ? i - L : 1, inside ? j - L : 1) & 3u) == 0u; // neither F > diag nor E > S (sw.cpp:60-71)
        const unsigned long long stop = __builtin_amdgcn_ballot_w64(!ok);
        return stop == 0ull ? 64 : __builtin_ctzll(stop);
    }
    __device__ __forceinline__ int at(int i, int j)
    {
        const unsigned c = (unsigned)__builtin_amdgcn_readfirstlane((int)tb.cell(i, j));
        if (c & 2u) { // vertical gap: 1 + consecutive extensions above (sw.cpp:73-82): rows i-1, i-2, .. until one that opened
            int n = 1;
            for (int top = i - 1; top >= 1; top -= 64) {
                const int r = top - L;
                const bool stop = r < 1 || (tb.cell(r < 1 ? 1 : r, j) & 4u) != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(stop);
                if (m != 0ull) return n + __builtin_ctzll(m);
                n += 64;
            }
            return n;
        }
        if (c & 1u) { // horizontal gap (sw.cpp:84-93)
            int n = 1;
            for (int right = j - 1; right >= 1; right -= 64) {
                const int q = right - L;
                const bool stop = q < 1 || (tb.cell(i, q < 1 ? 1 : q) & 8u) != 0u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(stop);
                if (m != 0ull) return -(n + __builtin_ctzll(m));
                n += 64;
            }
            return -n;
        }
        return 0;
    }
};

__global__ __launch_bounds__(256) void sw_traceback_wave_kernel(const TbArgs a)
{
    const int lane = threadIdx.x & 63;
    const int64_t slot = (int64_t)blockIdx.x * (blockDim.x >> 6) + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    if (slot >= a.count) return;
    const int64_t p = a.first + slot;
    const int tl = a.t.length(p);
    const int ql = a.q.length(p);
    const DpRecord r = a.rec[slot];
    const int64_t o = a.dest ? a.dest[p] : p;
    char *const slot_out = a.cigar + (size_t)o * a.cigar_stride;
    if (r.sps < 0) {
        // the fill kernel gave up on this pair (sw_dp_coop_kernel: a bounded wait ran out)
        for (int k = lane; k < a.cigar_stride; k += 64) slot_out[k] = 0;
        if (lane == 0) {
            a.offset[o] = 0;
            if (a.cigar_len) a.cigar_len[o] = 0;
            if (a.status) a.status[o] = ERR_DEVICE;
            if (a.status_any) atomicMax(a.status_any, ERR_DEVICE);
        }
        return;
    }

    CigarWriter cw;
    cw.slot = slot_out;
    cw.binary = a.binary_cigar;
    cw.cap = a.binary_cigar ? (a.cigar_stride & ~3) : a.cigar_stride;
    cw.pos = cw.cap;
    cw.need = 0;
    cw.store = (lane == 0);

    int off;
    if (a.packed16 == 4) {
        WaveMovesView mv;
        mv.tb.base = a.tb + (size_t)slot * a.tb_stride_words;
        mv.tb.set_schedule(r, ql, a.rows_per_stripe);
        mv.tb.packed16 = 4;
        mv.tb.waves = TbView::waves_of(r);
        mv.tb.half = 0;
        mv.tb.lane = 0;
        mv.tb.ql = ql;
        mv.L = lane;
        off = walk_and_write(mv, tl, ql, a.strategy, r.max_t, r.max_q, r.mqe_t, r.seg, cw);
    } else if (TbView::layout_of(a.packed16, r) == 3) {
        WaveMoves16 mv;
        mv.init(a.tb + (size_t)slot * a.tb_stride_words, r.sps, lane);
        off = walk_and_write(mv, tl, ql, a.strategy, r.max_t, r.max_q, r.mqe_t, r.seg, cw);
    } else {
        WaveMoves mv;
        mv.init(a.tb + (size_t)slot * a.tb_stride_words, r.sps, a.rows_per_stripe, lane);
        off = walk_and_write(mv, tl, ql, a.strategy, r.max_t, r.max_q, r.mqe_t, r.seg, cw);
    }

    // the text was built right-aligned by lane 0: move it to the front and zero the rest, 64 bytes at a time
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_wave_barrier();
    int status = 0;
    if (cw.pos < 0) {
        for (int k = lane; k < a.cigar_stride; k += 64) slot_out[k] = 0;
        status = ERR_CIGAR_OVERFLOW;
    } else {
        const int len = cw.cap - cw.pos;
        if (cw.pos > 0)
            for (int k0 = 0; k0 < len; k0 += 64) {
                const int k = k0 + lane;
                const char ch = k < len ? __builtin_nontemporal_load(slot_out + cw.pos + k) : 0;
                if (k < len) slot_out[k] = ch;
            }
        for (int k = len + lane; k < a.cigar_stride; k += 64) slot_out[k] = 0;
    }
    if (lane == 0) {
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
}

// calculateCigar on a caller-supplied int32 backtrack matrix (sw_scalar.h:8): one thread.
// out[0] = offset, out[1] = text length needed, out[2] = status
__global__ void sw_cigar_from_matrix_kernel(const int32_t *btr, int tl, int ql, int strategy, Score ez, char *cigar,
                                            int cap, int32_t *out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    MatrixMoves mv;
    mv.btr = btr;
    mv.m = ql + 1;
    CigarWriter cw;
    cw.slot = cigar;
    cw.binary = 0;
    cw.cap = cap;
    cw.pos = cap;
    cw.need = 0;
    out[0] = walk_and_write(mv, tl, ql, strategy, ez.max_t, ez.max_q, ez.mqe_t, ez.seg_length, cw);
    out[2] = finish_cigar(cw);
    out[1] = cw.need;
}

// Logical backtrack matrix of ONE pair (slot 0 of the workspace): the int32 run lengths the
// reference stores (sw.cpp:62,66,70), rebuilt from the 4-bit cells.
__global__ __launch_bounds__(256) void sw_expand_kernel(const uint32_t *tbw, const DpRecord *rec, int tl, int ql,
                                                        int packed16, int half, int rows, int32_t *btr, int lane)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= tl * ql) return;
    const int i = idx / ql + 1, j = idx % ql + 1;
    TbView tb;
    tb.base = tbw;
    tb.set_schedule(rec[0], ql, rows);
    tb.packed16 = TbView::layout_of(packed16, rec[0]);
    tb.waves = TbView::waves_of(rec[0]);
    tb.half = half;
    tb.lane = lane;
    tb.ql = ql;
    const unsigned c = tb.cell(i, j);
    int v = 0;
    if (c & 2u)
        v = tb.vrun(i, j);
    else if (c & 1u)
        v = -tb.hrun(i, j);
    btr[(size_t)i * (ql + 1) + j] = v;
}

// ---------------------------------------------------------------------------------------------
// calculateMatrix_avx (sw_avx.h:7, sw_avx.cpp:110-322) as a stage: ONE band of `actual_bw` target rows over the caller's
// arrays, in the reference's layouts: `query` reversed with `bw` ints of padding on each side, `gap`
// indexed like it, `score` / `step` = H and E of the row above the band per column (0 .. ql), the band's backtrack cells at
// bcktrack[bw * band * (ql + bw - 1) + (j - 1 + J) * bw + J] for row J of the band, column j.  A compatibility entry for
// callers that drive the band loop themselves (sw_avx.cpp:71-80), not a fast path: one lane walks the band row by row,
// which yields every cell the anti-diagonal order of the reference yields (same recurrence, same comparisons).
__global__ void sw_band_fill_kernel(const int32_t *target, const int32_t *query, int ql, int32_t *band_btr, int band, int bw,
                                    int actual_bw, int32_t *score, int32_t *step, int32_t *gap, int match, int mismatch,
                                    int gopen, int gext, int strategy, int32_t *mqe_io)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    const bool indel = (strategy & (OS_INDEL | OS_LEAD_ID)) != 0;
    const int n_col = ql + bw - 1;
    int mqe = mqe_io[0], mqe_t = mqe_io[1];
    int e0 = step[0]; // E running down column 0 (never feeds a cell; kept so that step[0] leaves as the reference leaves it)
    const int score0_in = score[0];
    for (int J = 0; J < actual_bw; ++J) {
        const int row = bw * band + J + 1;                                   // 1-based target row
        const int hb = indel ? -gopen - (row - 1) * gext : 0;                // H[row][0]: _s_col, sw_avx.cpp:117-124
        const int tbase = target[J];                                         // (the caller hands over the band's rows)
        int diag = score[0];                                                 // H[row-1][0]
        score[0] = hb;
        e0 = max(hb - gopen, e0 - gext);
        int F = hb - gopen, gap_h = -1;                                      // _f_col, sw_avx.cpp:114,123
        for (int j = 1; j <= ql; ++j) {
            const int gi = bw + ql - 1 - (j - 1);                            // reversed index of column j (sw_avx.cpp:161-162)
            const int up = score[j];
            const int E = step[j];
            int gv = gap[gi];
            int S = diag + (query[gi] == tbase ? match : mismatch);          // sw_avx.cpp:163-165
            int dir = F > S ? gap_h : 0;                                     // :167-169
            S = max(S, F);
            dir = E > S ? gv : dir;                                          // :171-174
            S = max(S, E);
            band_btr[(size_t)(j - 1 + J) * bw + J] = dir;
            const int tmp = S - gopen;                                       // :179-191 (extension wins ties)
            const int Ee = E - gext;
            gv = tmp > Ee ? 1 : gv + 1;
            const int Fe = F - gext;
            gap_h = tmp > Fe ? -1 : gap_h - 1;
            F = max(tmp, Fe);
            diag = up;
            score[j] = S;
            step[j] = max(tmp, Ee);
            gap[gi] = gv;
        }
        if (score[ql] >= mqe) { // last column, later rows win ties (sw_avx.cpp:313-321)
            mqe = score[ql];
            mqe_t = row;
        }
    }
    if (actual_bw > 1) {
        step[0] = e0;
    } else {
        score[0] = score0_in; // a band of one row: the reference's first-triangle loop (sw_avx.cpp:159-206) does not run and
    }                         // column 0 of score[] / step[] keeps the previous band's values (nothing reads them any more)
    (void)n_col;
    mqe_io[0] = mqe;
    mqe_io[1] = mqe_t;
}

hipError_t launch_band_fill(const int32_t *target, const int32_t *query, int ql, int32_t *band_btr, int band, int bw, int actual_bw,
                            int32_t *score, int32_t *step, int32_t *gap, int match, int mismatch, int gopen, int gext,
                            int strategy, int32_t *mqe_io, hipStream_t stream)
{
    hipLaunchKernelGGL(sw_band_fill_kernel, dim3(1), dim3(64), 0, stream, target, query, ql, band_btr, band, bw, actual_bw, score,
                       step, gap, match, mismatch, gopen, gext, strategy, mqe_io);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// launch wrappers (called from sw_capi.cpp)

int64_t dp_group_bytes(int sps_cap, int rows)
{
    return (int64_t)dp_ring_entries(sps_cap, rows) * 8 + 4ll * dp_qcopy_bytes(sps_cap, rows);
}

int dp_lds_bytes(int sps_cap, int waves_per_block, int rows)
{
    const int64_t b = waves_per_block * (64 / rows) * dp_group_bytes(sps_cap, rows);
    return b > (1 << 30) ? (1 << 30) : (int)b;
}

hipError_t launch_dp(const DpArgs &a, int waves_per_block, int rows, hipStream_t stream)
{
    const int per_block = waves_per_block * (64 / rows);
    const int64_t blocks = (a.count + per_block - 1) / per_block;
    const dim3 grid((unsigned)blocks), block(64 * waves_per_block);
    if (a.matrix) {
        // (the host layer only takes this path with an LDS carve below 64 KB)
        DpArgs b = a;
        b.matrix_lds_offset = dp_lds_bytes(a.sps_cap, waves_per_block, rows);
        if (rows == 64)
            hipLaunchKernelGGL(sw_dp64_matrix_kernel, grid, block, b.matrix_lds_offset + MATRIX_DIM * MATRIX_DIM, stream, b);
        else
            hipLaunchKernelGGL(sw_dp_matrix_kernel, grid, block, b.matrix_lds_offset + MATRIX_DIM * MATRIX_DIM, stream, b);
        return hipGetLastError();
    }
    if (a.scratch) {
        if (rows == 64)
            hipLaunchKernelGGL(sw_dp64_scratch_kernel, grid, block, 0, stream, a);
        else
            hipLaunchKernelGGL(sw_dp_scratch_kernel, grid, block, 0, stream, a);
        return hipGetLastError();
    }
    const int lds = dp_lds_bytes(a.sps_cap, waves_per_block, rows);
    if (lds > 64 * 1024) { // per device and rare: set every time rather than cache across devices / threads
        hipError_t e = hipFuncSetAttribute(rows == 64 ? reinterpret_cast<const void *>(sw_dp64_kernel)
                                                      : reinterpret_cast<const void *>(sw_dp_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    if (rows == 64)
        hipLaunchKernelGGL(sw_dp64_kernel, grid, block, lds, stream, a);
    else
        hipLaunchKernelGGL(sw_dp_kernel, grid, block, lds, stream, a);
    return hipGetLastError();
}

// MGL_SW_FLAG_SCORE_ONLY: hand the fill kernel's ScoreMax to the caller, no path walk (offset 0, empty CIGAR)
__global__ __launch_bounds__(256) void sw_scores_only_kernel(const TbArgs a)
{
    const int64_t slot = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= a.count) return;
    const int64_t p = a.first + slot;
    const DpRecord r = a.rec[slot];
    const int64_t o = a.dest ? a.dest[p] : p;
    a.offset[o] = 0;
    if (a.cigar_len) a.cigar_len[o] = 0;
    if (a.status) a.status[o] = 0;
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

// ---------------------------------------------------------------------------------------------
// Sort of a chunk by geometry on the device (RegroupArgs, sw_device.h): what mgl_sw_align_batch does on the host for a
// host batch of mixed geometries, for batches that are already resident.  The order of the pairs inside a cell is whatever
// the atomics make it; every pair's results land at its own index (dest), so the outputs do not depend on it.
__device__ __forceinline__ int regroup_cell(const RegroupArgs &a, int64_t p)
{
    const int tl = min(max(a.t.length(p), 1), a.max_tl), ql = min(max(a.q.length(p), 1), a.max_ql); // (lengths outside the promise: stay inside the grid)
    return (tl - 1) * a.max_ql + (ql - 1);
}
// One global atomic per WORKGROUP and distinct cell instead of one per pair.  Reads of one length arrive together (a batch
// that is sorted already has one cell per block) and returning atomics on one address are serialised (measured: 2 ms per
// 190 k pairs for each of the two passes, as long as the alignment itself); and even aggregated per wave, the atomics of an
// unsorted chunk all land on the two cache lines that hold the 51 counters in use, and the fill kernel running beside the
// sort lost a third of its speed behind that L2 channel.  So a block of 1024 pairs first counts its cells in an LDS hash
// table (LDS atomics), then adds each distinct cell's count to the global counter once; a pair's rank is the block's base
// for the cell plus its rank inside the block.
// A workgroup takes RG_TILES tiles of 1024 pairs through ONE table: 4 M unsorted reads of 51 lengths in blocks of 1 024 pairs were
// still 200 000 global adds on 51 addresses -- 0.93 ms for the count and 0.95 for the scatter where nothing else ran; eight tiles per
// workgroup leave an eighth of them.  A pair whose cell finds no place in the table (more distinct cells in 8 192 pairs than the
// table holds: not with read lengths, possible with arbitrary batches) adds to the global counter itself.
constexpr int RG_BLOCK = 1024, RG_TABLE = 2048, RG_TILES = 8;
struct RegroupTable {
    int key[RG_TABLE], num[RG_TABLE], base[RG_TABLE];
};
// the pairs of this thread: k0 + u * RG_BLOCK, u = 0 .. RG_TILES - 1; c[u] their cells, valid[u].  want_rank: rank[u] = the pair's
// rank among the pairs of its cell (whatever order the atomics give).
__device__ __forceinline__ void regroup_take(int32_t *cnt, const int (&c)[RG_TILES], const bool (&valid)[RG_TILES], RegroupTable &tab, bool want_rank, int (&rank)[RG_TILES])
{
    for (int x = threadIdx.x; x < RG_TABLE; x += blockDim.x) {
        tab.key[x] = -1;
        tab.num[x] = 0;
    }
    __syncthreads();
    int slot[RG_TILES], local[RG_TILES];
#pragma unroll
    for (int u = 0; u < RG_TILES; ++u) {
        slot[u] = -1;
        local[u] = 0;
        if (valid[u]) {
            int h = (int)(((unsigned)c[u] * 2654435761u) >> 21) & (RG_TABLE - 1);
            for (int probes = 0; probes < RG_TABLE; ++probes) {
                const int old = atomicCAS(&tab.key[h], -1, c[u]);
                if (old == -1 || old == c[u]) {
                    local[u] = atomicAdd(&tab.num[h], 1);
                    slot[u] = h;
                    break;
                }
                h = (h + 1) & (RG_TABLE - 1);
            }
            if (slot[u] < 0) local[u] = atomicAdd(cnt + c[u], 1); // (the table is full of other cells: this pair counts for itself)
        }
    }
    __syncthreads();
    for (int x = threadIdx.x; x < RG_TABLE; x += blockDim.x)
        if (tab.key[x] >= 0) {
            if (want_rank)
                tab.base[x] = atomicAdd(cnt + tab.key[x], tab.num[x]);
            else
                atomicAdd(cnt + tab.key[x], tab.num[x]);
        }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < RG_TILES; ++u) rank[u] = !valid[u] || !want_rank ? 0 : slot[u] >= 0 ? tab.base[slot[u]] + local[u] : local[u];
}
__global__ __launch_bounds__(RG_BLOCK) void sw_regroup_count_kernel(const RegroupArgs a)
{
    __shared__ RegroupTable tab;
    const int64_t k0 = (int64_t)blockIdx.x * (RG_BLOCK * RG_TILES) + threadIdx.x;
    int c[RG_TILES], rank[RG_TILES];
    bool valid[RG_TILES];
#pragma unroll
    for (int u = 0; u < RG_TILES; ++u) {
        const int64_t k = k0 + (int64_t)u * RG_BLOCK;
        valid[u] = k < a.count;
        c[u] = valid[u] ? regroup_cell(a, a.first + k) : 0;
    }
    regroup_take(a.cnt, c, valid, tab, false, rank);
}
// one workgroup: exclusive prefix sums over the grid, 1024 cells at a time, of the pairs in blocks of 128 (lane_blocks only), of
// the pairs in the remaining full blocks of eight, and of the left-over pairs; full_start is relative to the end of the blocks
// of 128 (total[1]), rest_start to the end of all full blocks (total[0]): both added by the scatter
__global__ __launch_bounds__(1024) void sw_regroup_scan_kernel(const RegroupArgs a)
{
    __shared__ unsigned long long s_wave[16];
    __shared__ unsigned s_rest[16];
    const int cells = a.max_tl * a.max_ql, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long run = 0; // totals of the tiles before this one: blocks of 128 (high word), of eight (low word)
    unsigned run_rest = 0;
    for (int c0 = 0; c0 < cells; c0 += 1024) {
        const int c = c0 + (int)threadIdx.x;
        const int n = c < cells ? a.cnt[c] : 0;
        const int nl = a.lane_blocks ? n & ~127 : 0, n8 = (n - nl) & ~7;
        const unsigned long long mine = ((unsigned long long)(unsigned)nl << 32) | (unsigned)n8;
        const unsigned mine_rest = (unsigned)(n & 7);
        unsigned long long incl = mine;
        unsigned incl_rest = mine_rest;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned long long up = __shfl_up(incl, d);
            const unsigned up_rest = __shfl_up(incl_rest, d);
            if (lane >= d) {
                incl += up;
                incl_rest += up_rest;
            }
        }
        if (lane == 63) {
            s_wave[wave] = incl;
            s_rest[wave] = incl_rest;
        }
        __syncthreads();
        unsigned long long before = 0, tile = 0;
        unsigned before_rest = 0, tile_rest = 0;
        for (int w = 0; w < 16; ++w) {
            const unsigned long long v = s_wave[w];
            const unsigned vr = s_rest[w];
            if (w < wave) {
                before += v;
                before_rest += vr;
            }
            tile += v;
            tile_rest += vr;
        }
        const unsigned long long excl = run + before + incl - mine;
        if (c < cells) {
            a.nlane[c] = nl;
            a.nfull[c] = nl + n8;
            a.lane_start[c] = -(int)(excl >> 32) - nl; // from the END of the lane part (total[1], added by the scatter): the cells with
                                                       // the longest queries first -- the waves that take longest should not start last
            a.full_start[c] = (int)(excl & 0xffffffffu);
            a.rest_start[c] = (int)(run_rest + before_rest + incl_rest - mine_rest);
            a.cnt[c] = 0; // the scatter counts again
        }
        run += tile;
        run_rest += tile_rest;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        a.total[1] = (int64_t)(run >> 32);
        a.total[0] = (int64_t)(run >> 32) + (int64_t)(run & 0xffffffffu);
    }
}
__global__ __launch_bounds__(RG_BLOCK) void sw_regroup_scatter_kernel(const RegroupArgs a)
{
    __shared__ RegroupTable tab;
    const int64_t k0 = (int64_t)blockIdx.x * (RG_BLOCK * RG_TILES) + threadIdx.x;
    int c[RG_TILES], rank[RG_TILES];
    bool valid[RG_TILES];
#pragma unroll
    for (int u = 0; u < RG_TILES; ++u) {
        const int64_t k = k0 + (int64_t)u * RG_BLOCK;
        valid[u] = k < a.count;
        c[u] = regroup_cell(a, a.first + (valid[u] ? k : 0));
    }
    regroup_take(a.cnt, c, valid, tab, true, rank);
#pragma unroll
    for (int u = 0; u < RG_TILES; ++u) {
        if (!valid[u]) continue;
        const int64_t p = a.first + k0 + (int64_t)u * RG_BLOCK;
        const int cc = c[u], pos = rank[u];
        const int nl = a.nlane[cc], nf = a.nfull[cc];
        const int64_t slot = pos < nl ? a.total[1] + a.lane_start[cc] + pos
                             : pos < nf ? a.total[1] + a.full_start[cc] + (pos - nl) : a.total[0] + a.rest_start[cc] + (pos - nf);
        a.t_start[slot] = a.t.off[p];
        a.q_start[slot] = a.q.off[p];
        a.dest[slot] = p;
        a.t_len[slot] = a.t.length(p);
        a.q_len[slot] = a.q.length(p);
    }
}

__global__ __launch_bounds__(256) void sw_iota64_kernel(int64_t *dst, int64_t n, int64_t step)
{
    for (int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x; k < n; k += (int64_t)gridDim.x * 256) dst[k] = k * step;
}
hipError_t launch_iota64(int64_t *dst, int64_t n, int64_t step, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(sw_iota64_kernel, dim3((unsigned)std::min<int64_t>((n + 255) / 256, 4096)), dim3(256), 0, stream, dst, n, step);
    return hipGetLastError();
}

hipError_t launch_regroup(const RegroupArgs &a, hipStream_t stream)
{
    hipError_t e = hipMemsetAsync(a.cnt, 0, (size_t)a.max_tl * a.max_ql * 4, stream);
    if (e != hipSuccess) return e;
    const unsigned blocks = (unsigned)((a.count + RG_BLOCK * RG_TILES - 1) / (RG_BLOCK * RG_TILES));
    hipLaunchKernelGGL(sw_regroup_count_kernel, dim3(blocks), dim3(RG_BLOCK), 0, stream, a);
    hipLaunchKernelGGL(sw_regroup_scan_kernel, dim3(1), dim3(1024), 0, stream, a);
    hipLaunchKernelGGL(sw_regroup_scatter_kernel, dim3(blocks), dim3(RG_BLOCK), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_scores_only(const TbArgs &a, hipStream_t stream)
{
    hipLaunchKernelGGL(sw_scores_only_kernel, dim3((unsigned)((a.count + 255) / 256)), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_traceback(const TbArgs &a, hipStream_t stream)
{
    if ((!a.packed16 && a.rows_per_stripe == 64) || a.packed16 == 3 || a.packed16 == 4) {
        // long reads: one wave per pair (the path walk is the latency, not the lane count)
        hipLaunchKernelGGL(sw_traceback_wave_kernel, dim3((unsigned)a.count), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    const int64_t blocks = (a.count + 255) / 256;
    hipLaunchKernelGGL(sw_traceback_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_cigar_from_matrix(const int32_t *btr, int tl, int ql, int strategy, const Score &ez, char *cigar,
                                    int cap, int32_t *out3, hipStream_t stream)
{
    hipLaunchKernelGGL(sw_cigar_from_matrix_kernel, dim3(1), dim3(64), 0, stream, btr, tl, ql, strategy, ez, cigar, cap,
                       out3);
    return hipGetLastError();
}

hipError_t launch_expand(const uint32_t *tbw, const DpRecord *rec, int tl, int ql, int packed16, int half, int rows,
                         int32_t *btr, hipStream_t stream, int lane)
{
    const int64_t n = (int64_t)tl * ql;
    hipLaunchKernelGGL(sw_expand_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, tbw, rec, tl, ql,
                       packed16, half, rows, btr, lane);
    return hipGetLastError();
}

} // namespace mgl_sw_dev
