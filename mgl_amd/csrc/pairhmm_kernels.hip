// pairhmm_kernels.hip -- gfx950 (MI355X / CDNA4) kernels for mgl's PairHMM forward algorithm
// (the function defined by the reference's compute_prob_scalar.cc:5-45,47-342,
// paths relative to /root/reference/src/main/native/mgl_pairhmm/).
//
//   prior   = (rs == hap || rs == 'N' || hap == 'N') ? 1 - distm[r] : distm[r] / 3              :27-36
//   M[r][c] = prior * (M[r-1][c-1] * pMM[r] + (X[r-1][c-1] + Y[r-1][c-1]) * pGapM[r])           :39
//   Y[r][c] = M[r][c-1] * pMY[r] + Y[r][c-1] * pZZ[r]                                           :41
//   X[r][c] = M[r-1][c] * pMX[r] + X[r-1][c] * pZZ[r]                                           :43
//   row 0: M = X = 0, Y = INITIAL_CONSTANT / haplen; column 0 of rows >= 1: all 0               :122-152
//   result  = sum over c = 1..haplen of M[R][c] + X[R][c], ascending c                          :214,:318
//
// Mapping: the Smith-Waterman fill kernel's (sw_kernels.hip).  One (read, haplotype) pair per 16-lane DPP row,
// four pairs per wave64; lane L owns read row 16k + L + 1 of stripe k; a step advances every lane one
// haplotype column along an anti-diagonal (lane L is at column s - L).  "Up" values M, X, X+Y of row r-1 arrive
// by `row_shr:1` DPP from the lane below; the diagonal values are last step's up values; Y stays in the lane.
// The carry of a stripe's last row (M, X, X+Y per column) lives in an LDS ring per pair, written by the lane that
// owns the stripe's last row and fed to lane 0 of the next stripe as the DPP's lane-0 operand; row 0 is the ring's
// initial content.  The haplotype sits in LDS once; a lane fetches its four bases of a 4-step block with one
// ds_read_b32 + v_alignbyte.  The lane that owns row R accumulates M + X in column order, which is the
// reference's summation order.  Float first; pairs whose float sum is below 1e-28 are redone by the same
// template in double (…PairHmm.cc:149-216).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "pairhmm_device.h"

namespace mgl_ph_dev {

namespace {

constexpr int DPP_ROW_SHR1 = 0x111, DPP_WAVE_SHR1 = 0x138;
constexpr double MIN_ACCEPTED = (double)1e-28f; // pairhmm_common.h:32, compared as double (…PairHmm.cc:149,185)

template <int G>
__device__ __forceinline__ int dpp_shr1(int lane0_value, int src)
{
    const int r = __builtin_amdgcn_update_dpp(lane0_value, src, G == 16 ? DPP_ROW_SHR1 : DPP_WAVE_SHR1, 0xf, 0xf, false);
    if (G == 16 || G == 64) return r;
    // two groups of 32 lanes or three of 21: the first lane of a group must not see the last lane of the group before
    // (one v_cndmask on a wave-constant lane mask)
    return (threadIdx.x & 63) % G == 0 ? lane0_value : r;
}
// every lane takes src from the lane below it inside its group of G lanes (16: one DPP row; 21 / 32: a third / half of the wave; 64: the wave);
// lane 0 of the group takes lane0_value
template <int G>
__device__ __forceinline__ float shift_in(float lane0_value, float src)
{
    return __int_as_float(dpp_shr1<G>(__float_as_int(lane0_value), __float_as_int(src)));
}
template <int G>
__device__ __forceinline__ double shift_in(double lane0_value, double src)
{
    const int lo = dpp_shr1<G>(__double2loint(lane0_value), __double2loint(src));
    const int hi = dpp_shr1<G>(__double2hiint(lane0_value), __double2hiint(src));
    return __hiloint2double(hi, lo);
}

template <typename T>
struct Num;
template <>
struct Num<float> {
    static __device__ __forceinline__ float initial() { return __builtin_ldexpf(1.f, 120); } // Context.h:147
    static __device__ __forceinline__ double log10_initial(const PhArgs &a) { return a.log10_initial_f; }
    static __device__ __forceinline__ const float *ph2pr(const PhArgs &a) { return a.ph2pr_f; }
    static __device__ __forceinline__ const float *m2m(const PhArgs &a) { return a.m2m_f; }
};
template <>
struct Num<double> {
    static __device__ __forceinline__ double initial() { return __builtin_ldexp(1.0, 1020); } // Context.h:108
    static __device__ __forceinline__ double log10_initial(const PhArgs &a) { return a.log10_initial_d; }
    static __device__ __forceinline__ const double *ph2pr(const PhArgs &a) { return a.ph2pr_d; }
    static __device__ __forceinline__ const double *m2m(const PhArgs &a) { return a.m2m_d; }
};

template <typename T>
struct Carry { // one column of a stripe's last row
    T m, x, xy;
};

template <typename T>
struct Lane {
    T m, x, y, xy;       // this lane's cell of the previous step: (r, c-1)
    T dm, dxy;           // up values of the previous step = diagonal values of this step: (r-1, c-1)
    T acc;               // running sum of M + X along this lane's row (this stripe)
};

template <typename T>
struct RowConst {
    T pMM, pGapM, pMX, pMY, pZZ, prior_match, prior_mismatch;
    int rs;
};

// Four anti-diagonal steps.  PRO: some lane may still be at a column <= 0 (forced zeros, :150-152);
// EPI: some lane may be past its last column (stop accumulating); HAP_N: some haplotype of this wave holds an 'N';
// ACC: accumulate M + X along the row (the stripe that holds a pair's last row).
// cin holds the carry of the four columns lane 0 visits in this block (only lane 0 of a group needs it: it is the
// DPP's lane-0 operand); nxt receives the next block's, loaded here by lane 0 alone -- one masked region, a sixteenth
// (or a sixty-fourth) of the LDS traffic of a whole-wave read -- so no step waits for LDS.
template <typename T, int G, bool PRO, bool EPI, bool HAP_N, bool ACC>
__device__ __forceinline__ void ph_step4(Lane<T> &st, const Carry<T> (&cin)[4], Carry<T> (&nxt)[4], const Carry<T> *ring_next,
                                         Carry<T> *ring_wr, const unsigned hw, const RowConst<T> &rc, const int s0, const int L,
                                         const int hap_len, const bool writer)
{
    if (L == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) nxt[u] = ring_next[u]; // columns s0 + 4 .. s0 + 7
    }
    Carry<T> o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const T um = shift_in<G>(cin[u].m, st.m);     // M[r-1][c]
        const T ux = shift_in<G>(cin[u].x, st.x);     // X[r-1][c]
        const T uxy = shift_in<G>(cin[u].xy, st.xy);  // (X + Y)[r-1][c]
        const int hb = (int)((hw >> (8 * u)) & 0xffu);
        const bool match = HAP_N ? ((hb == rc.rs) | (hb == 'N')) : (hb == rc.rs); // rs == 'N' is folded into prior_mismatch
        const T prior = match ? rc.prior_match : rc.prior_mismatch;
        T mn = prior * (st.dm * rc.pMM + st.dxy * rc.pGapM);
        T yn = st.m * rc.pMY + st.y * rc.pZZ;
        T xn = um * rc.pMX + ux * rc.pZZ;
        const int c = s0 + u - L; // this lane's column
        if (PRO) {
            const bool border = c <= 0;
            mn = border ? (T)0 : mn;
            yn = border ? (T)0 : yn;
            xn = border ? (T)0 : xn;
        }
        const T xyn = xn + yn;
        if (ACC) { // only the stripe that holds row R needs the sum
            if (EPI)
                st.acc = c <= hap_len ? st.acc + (mn + xn) : st.acc;
            else
                st.acc = st.acc + (mn + xn);
        }
        o[u].m = mn;
        o[u].x = xn;
        o[u].xy = xyn;
        st.m = mn;
        st.x = xn;
        st.y = yn;
        st.xy = xyn;
        st.dm = um;
        st.dxy = uxy;
    }
    // the lane that owns the stripe's last row publishes its four columns (one masked region per block)
    if (writer) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ring_wr[u] = o[u];
    }
}

template <typename T, int G, bool RESCUE>
__device__ __forceinline__ void pairhmm_body(const PhArgs &a, unsigned char *smem)
{
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    constexpr int PW = 64 / G; // pairs per wave
    const int grp = lane / G;
    const int L = lane % G;
    const int64_t slot = ((int64_t)blockIdx.x * (blockDim.x >> 6) + wave) * PW + grp;
    if (slot - grp >= a.n_pairs) return;
    const bool valid = slot < a.n_pairs;
    const int64_t p = valid ? slot : a.n_pairs - 1;
    if (RESCUE && a.rescue_only) {
        // only pairs whose float sum fell below MIN_ACCEPTED (…PairHmm.cc:185-189)
        const bool need = valid && a.need_double[p] != 0;
        if (!__builtin_amdgcn_ballot_w64(need)) return;
    }
    const int32_t ri = a.pair_read[p], hi = a.pair_hap[p];
    const int64_t r0 = a.read_off[ri], h0 = a.hap_off[hi];
    const int R = (int)(a.read_off[ri + 1] - r0), H = (int)(a.hap_off[hi + 1] - h0);
    const uint8_t *rbase = a.reads + 5 * r0;
    const int nstripes = (R + G - 1) / G;

    int h_max = H, h_min = H, ns_max = nstripes;
#pragma unroll
    for (int m = G; m < 64; m <<= 1) {
        h_max = max(h_max, __shfl_xor(h_max, m));
        h_min = min(h_min, __shfl_xor(h_min, m));
        ns_max = max(ns_max, __shfl_xor(ns_max, m));
    }
    h_max = __builtin_amdgcn_readfirstlane(h_max);
    h_min = __builtin_amdgcn_readfirstlane(h_min);
    ns_max = __builtin_amdgcn_readfirstlane(ns_max);
    const int sps8 = (h_max + G + 7) & ~7;     // steps per stripe: the last lane reaches column h_max at step h_max + G - 1
    const int lean_end8 = max(G, h_min & ~7);  // [G, lean_end8): no border, nobody past the last column

    // LDS carve per group: ring[hap_cap + 2G + 16] carries of 12 (float) / 24 (double) bytes (ring[j + G] = column j) | hap bytes (G zeros, hap, zeros)
    const int ring_entries = a.hap_cap + 2 * G + 16;
    const int hap_bytes = (a.hap_cap + 2 * G + 28 + 7) & ~7; // keeps the next group's ring 8-byte aligned
    const int group_bytes = ring_entries * (int)sizeof(Carry<T>) + hap_bytes;
    unsigned char *gbase = smem + (size_t)(wave * PW + grp) * group_bytes;
    Carry<T> *ring = reinterpret_cast<Carry<T> *>(gbase);
    unsigned char *hbuf = gbase + ring_entries * sizeof(Carry<T>);

    const T y_initial = Num<T>::initial() / (T)H; // :101
    bool hap_has_n = false;
    {
        unsigned *hz = reinterpret_cast<unsigned *>(hbuf);
        for (int w = L; w < (hap_bytes >> 2); w += G) hz[w] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int x = L; x < H; x += G) {
            const uint8_t ch = a.haps[h0 + x];
            hap_has_n |= ch == 'N';
            hbuf[G + x] = ch;
        }
        // row 0: M = X = 0, Y = INITIAL_CONSTANT / haplen (:126-136)
        for (int j = L; j < ring_entries - G; j += G) {
            Carry<T> o;
            o.m = (T)0;
            o.x = (T)0;
            o.xy = y_initial;
            ring[j + G] = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    const bool any_n = __builtin_amdgcn_ballot_w64(hap_has_n) != 0; // wave-uniform: which block variant runs
    const T *ph2pr = Num<T>::ph2pr(a);
    const T *m2m = Num<T>::m2m(a);
    const T three_over = (T)1.0 / (T)3.0; // :18
    const int last_lane = (R - 1) % G;     // owner of row R in this pair's last stripe
    const int q_shift = (G - 1 - L) & 3;
    const unsigned *hrd0 = reinterpret_cast<const unsigned *>(hbuf) + ((G - 1 - L) >> 2);

    Lane<T> st;
    T result = (T)0;
    for (int k = 0; k < ns_max; ++k) {
        const int row = k * G + L + 1;
        RowConst<T> rc;
        if (row <= R) {
            // per-row tables, :68-86 (qualities masked with 127 as there)
            const int q = rbase[row - 1 + R] & 127, qi = rbase[row - 1 + 2 * R] & 127, qd = rbase[row - 1 + 3 * R] & 127,
                      qc = rbase[row - 1 + 4 * R] & 127;
            const int mx = max(qi, qd), mn = min(qi, qd);
            rc.pMM = m2m[((mx * (mx + 1)) >> 1) + mn]; // Context.h:121-133
            rc.pGapM = (T)1.0 - ph2pr[qc];
            rc.pMX = ph2pr[qi];
            rc.pMY = ph2pr[qd];
            rc.pZZ = ph2pr[qc];
            const T distm = ph2pr[q];
            rc.rs = rbase[row - 1];
            rc.prior_match = (T)1.0 - distm;
            rc.prior_mismatch = rc.rs == 'N' ? rc.prior_match : distm * three_over;
        } else {
            rc.pMM = rc.pGapM = rc.pMX = rc.pMY = rc.pZZ = rc.prior_match = rc.prior_mismatch = (T)0;
            rc.rs = 0;
        }
        st.m = st.x = st.y = st.xy = st.dm = st.dxy = (T)0;
        st.acc = (T)0;
        const int wl = (k == nstripes - 1) ? last_lane : G - 1;
        const bool writer = (L == wl);
        const Carry<T> *ring_rd = ring + G;  // lane 0 is at column s
        Carry<T> *ring_wr = ring + G - wl;   // the writer is at column s - wl
        Carry<T> ca[4] = {ring_rd[0], ring_rd[1], ring_rd[2], ring_rd[3]}, cb[4] = {ca[0], ca[1], ca[2], ca[3]};
        const unsigned *hrd = hrd0;
        unsigned h_lo = hrd[0], h_hi = hrd[1]; // the haplotype dwords are fetched one block ahead as well
        int s = 0;
#define MGL_PH_BLOCK(PRO, EPI, HAP_N, ACC, CUR, NXT)                                                  \
    {                                                                                                 \
        const unsigned hw = __builtin_amdgcn_alignbyte(h_hi, h_lo, (unsigned)q_shift);                \
        h_lo = h_hi;                                                                                  \
        h_hi = hrd[2];                                                                                \
        ph_step4<T, G, PRO, EPI, HAP_N, ACC>(st, CUR, NXT, ring_rd + 4, ring_wr, hw, rc, s, L, H, writer); \
        ring_rd += 4;                                                                                 \
        ring_wr += 4;                                                                                 \
        hrd += 1;                                                                                     \
        s += 4;                                                                                       \
    }
// two blocks per trip: the two carry register sets alternate instead of being copied (every phase boundary is
// a multiple of 8 steps: G, lean_end8, sps8)
#define MGL_PH_PAIR(PRO, EPI, HAP_N, ACC) { MGL_PH_BLOCK(PRO, EPI, HAP_N, ACC, ca, cb) MGL_PH_BLOCK(PRO, EPI, HAP_N, ACC, cb, ca) }
#define MGL_PH_STRIPE(HAP_N, ACC)                                                                     \
    {                                                                                                 \
        for (; s < G;) MGL_PH_PAIR(true, true, HAP_N, ACC)                                            \
        for (; s < lean_end8;) MGL_PH_PAIR(false, false, HAP_N, ACC)                                  \
        for (; s < sps8;) MGL_PH_PAIR(false, true, HAP_N, ACC)                                        \
    }
        // with four pairs per wave some pair may end in any stripe; with one pair per wave only the last stripe sums
        const bool acc_here = G == 16 || k == __builtin_amdgcn_readfirstlane(nstripes) - 1; // wave-uniform
        if (any_n) {
            if (acc_here) MGL_PH_STRIPE(true, true) else MGL_PH_STRIPE(true, false)
        } else {
            if (acc_here) MGL_PH_STRIPE(false, true) else MGL_PH_STRIPE(false, false)
        }
#undef MGL_PH_STRIPE
#undef MGL_PH_PAIR
#undef MGL_PH_BLOCK
        if (k == nstripes - 1) result = st.acc; // meaningful on the lane that owns row R
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }

    if (valid && L == last_lane) {
        const double rd = (double)result;
        if (!RESCUE) {
            // …PairHmm.cc:179-195
            const bool small = rd < MIN_ACCEPTED;
            a.need_double[p] = small ? 1 : 0;
            if (!small) a.out[p] = log10(rd) - Num<T>::log10_initial(a);
        } else if (!a.rescue_only || a.need_double[p] != 0) {
            a.out[p] = log10(rd) - Num<T>::log10_initial(a); // :203-209
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// One pair per wave, K consecutive read rows per lane (K = ceil(rows / 64), up to 4).  A lane computes its K cells of
// the current column one after the other -- row j+1 takes "up" from the value row j has just produced and "diagonal"
// from the value row j held before this step -- so only the lane's LAST row is shifted to the next lane: three DPP
// moves per K cells instead of per cell, a K times shorter carry ring, and a read of up to 256 bases in a single
// stripe.  The rows are numbered so that the pair's last row is always the last row of its lane: p = (K - R mod K) mod K
// virtual rows are put in front, set up to reproduce row 0 (M = X = 0, Y = INITIAL / haplen, prior 0).
template <typename T, int K>
struct LaneK {
    T m[K], x[K], y[K], xy[K]; // this lane's cells of the previous step: (row j, c-1)
    T dm, dxy;                 // (row above the lane's first row, c-1)
    T acc;
};

template <typename T, int K>
struct RowConstK {
    T pMM[K], pGapM[K], pMX[K], pMY[K], pZZ[K], prior_match[K], prior_mismatch[K], border_y[K];
    int rs[K];
};

template <typename T, int G, int K, bool PRO, bool EPI, bool HAP_N, bool ACC>
__device__ __forceinline__ void phk_step4(LaneK<T, K> &st, const Carry<T> (&cin)[4], Carry<T> (&nxt)[4], const Carry<T> *ring_next,
                                          Carry<T> *ring_wr, const unsigned hw, const RowConstK<T, K> &rc, const int s0, const int L,
                                          const int hap_len, const bool writer)
{
    if (L == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) nxt[u] = ring_next[u];
    }
    Carry<T> o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const T um = shift_in<G>(cin[u].m, st.m[K - 1]);
        const T ux = shift_in<G>(cin[u].x, st.x[K - 1]);
        const T uxy = shift_in<G>(cin[u].xy, st.xy[K - 1]);
        const int hb = (int)((hw >> (8 * u)) & 0xffu);
        const int c = s0 + u - L; // this lane's column
        const bool border = PRO && c <= 0;
        T up_m = um, up_x = ux, dg_m = st.dm, dg_xy = st.dxy;
        T nm[K], nx[K], ny[K], nxy[K];
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const bool match = HAP_N ? ((hb == rc.rs[j]) | (hb == 'N')) : (hb == rc.rs[j]);
            const T prior = match ? rc.prior_match[j] : rc.prior_mismatch[j];
            T mn = prior * (dg_m * rc.pMM[j] + dg_xy * rc.pGapM[j]);
            T yn = st.m[j] * rc.pMY[j] + st.y[j] * rc.pZZ[j];
            T xn = up_m * rc.pMX[j] + up_x * rc.pZZ[j];
            if (PRO) {
                mn = border ? (T)0 : mn;
                yn = border ? rc.border_y[j] : yn; // 0, or row 0's Y for the virtual rows in front
                xn = border ? (T)0 : xn;
            }
            dg_m = st.m[j]; // what row j held before this step is the diagonal of row j + 1
            dg_xy = st.xy[j];
            up_m = mn;
            up_x = xn;
            nm[j] = mn;
            nx[j] = xn;
            ny[j] = yn;
            nxy[j] = xn + yn;
        }
#pragma unroll
        for (int j = 0; j < K; ++j) {
            st.m[j] = nm[j];
            st.x[j] = nx[j];
            st.y[j] = ny[j];
            st.xy[j] = nxy[j];
        }
        st.dm = um;
        st.dxy = uxy;
        if (ACC) {
            if (EPI)
                st.acc = c <= hap_len ? st.acc + (nm[K - 1] + nx[K - 1]) : st.acc;
            else
                st.acc = st.acc + (nm[K - 1] + nx[K - 1]);
        }
        o[u].m = nm[K - 1];
        o[u].x = nx[K - 1];
        o[u].xy = nxy[K - 1];
    }
    if (writer) {
#pragma unroll
        for (int u = 0; u < 4; ++u) ring_wr[u] = o[u];
    }
}

// G = lanes per pair: 64 (one pair per wave; R, H wave-uniform), 32 or 21 (two / three pairs per wave, each with its own R, H, LDS
// carve and result lane; reads up to 32 * K rows, a single stripe: 150-base reads fill 150 of 160 row slots instead of
// 150 of 192, and the pipeline is 32 steps deep instead of 64).  `store`: this group holds a real pair.
template <typename T, int G, int K, bool RESCUE>
__device__ __forceinline__ void pairhmm_body_k(const PhArgs &a, unsigned char *smem, const int64_t p, const bool store, const int R,
                                               const int H, const uint8_t *rbase, const int64_t h0)
{
    constexpr int PW = 64 / G;               // pairs per wave (G = 21: lane 63 rides along as a 22nd lane of the last group)
    const int grp = min((int)(threadIdx.x & 63) / G, PW - 1);
    const int L = (int)(threadIdx.x & 63) - grp * G;
    const int pad = (K - R % K) % K;         // virtual rows in front
    const int rows_v = R + pad;              // a multiple of K
    // loop bounds are wave-uniform: the largest / smallest of the groups'
    int ns_w = (rows_v + G * K - 1) / (G * K), h_max = H, h_min = H;
#pragma unroll
    for (int g = 1; g < PW; ++g) {
        ns_w = max(ns_w, max(__shfl(ns_w, 0), __shfl(ns_w, g * G)));
        h_max = max(h_max, max(__shfl(h_max, 0), __shfl(h_max, g * G)));
        h_min = min(h_min, min(__shfl(h_min, 0), __shfl(h_min, g * G)));
    }
    const int nstripes = __builtin_amdgcn_readfirstlane(ns_w);
    const int sps8 = (__builtin_amdgcn_readfirstlane(h_max) + G + 7) & ~7;
    const int lean_end8 = max(G, __builtin_amdgcn_readfirstlane(h_min) & ~7);

    const int ring_entries = a.hap_cap + 2 * G + 16;
    const int hap_bytes = (a.hap_cap + 2 * G + 28 + 7) & ~7; // keeps the next group's ring 8-byte aligned
    unsigned char *gbase = smem + (size_t)grp * ((size_t)ring_entries * sizeof(Carry<T>) + hap_bytes);
    Carry<T> *ring = reinterpret_cast<Carry<T> *>(gbase);
    unsigned char *hbuf = gbase + ring_entries * sizeof(Carry<T>);

    const T y_initial = Num<T>::initial() / (T)H; // :101
    bool hap_has_n = false;
    {
        unsigned *hz = reinterpret_cast<unsigned *>(hbuf);
        for (int w = L; w < (hap_bytes >> 2); w += G) hz[w] = 0u;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int x = L; x < H; x += G) {
            const uint8_t ch = a.haps[h0 + x];
            hap_has_n |= ch == 'N';
            hbuf[G + x] = ch;
        }
        for (int j = L; j < ring_entries - G; j += G) { // row 0 (:126-136)
            Carry<T> o;
            o.m = (T)0;
            o.x = (T)0;
            o.xy = y_initial;
            ring[j + G] = o;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    const bool any_n = __builtin_amdgcn_ballot_w64(hap_has_n) != 0;
    const T *ph2pr = Num<T>::ph2pr(a);
    const T *m2m = Num<T>::m2m(a);
    const T three_over = (T)1.0 / (T)3.0; // :18
    const int last_lane = ((rows_v / K) - 1) % G; // the lane whose last row is read row R (in the last stripe)
    const int q_shift = (G - 1 - L) & 3;
    const unsigned *hrd0 = reinterpret_cast<const unsigned *>(hbuf) + ((G - 1 - L) >> 2);

    LaneK<T, K> st;
    T result = (T)0;
    for (int k = 0; k < nstripes; ++k) {
        RowConstK<T, K> rc;
#pragma unroll
        for (int j = 0; j < K; ++j) {
            const int row = (k * G + L) * K + j + 1 - pad; // read row, 1-based; <= 0: virtual row in front
            if (row >= 1 && row <= R) {
                const int q = rbase[row - 1 + R] & 127, qi = rbase[row - 1 + 2 * R] & 127, qd = rbase[row - 1 + 3 * R] & 127,
                          qc = rbase[row - 1 + 4 * R] & 127;
                const int mx = max(qi, qd), mn = min(qi, qd);
                rc.pMM[j] = m2m[((mx * (mx + 1)) >> 1) + mn];
                rc.pGapM[j] = (T)1.0 - ph2pr[qc];
                rc.pMX[j] = ph2pr[qi];
                rc.pMY[j] = ph2pr[qd];
                rc.pZZ[j] = ph2pr[qc];
                const T distm = ph2pr[q];
                rc.rs[j] = rbase[row - 1];
                rc.prior_match[j] = (T)1.0 - distm;
                rc.prior_mismatch[j] = rc.rs[j] == 'N' ? rc.prior_match[j] : distm * three_over;
                rc.border_y[j] = (T)0;
            } else {
                // rows past R compute nothing that is used; virtual rows in front repeat row 0: M = X = 0, Y constant
                rc.pMM[j] = rc.pGapM[j] = rc.pMX[j] = rc.pMY[j] = rc.prior_match[j] = rc.prior_mismatch[j] = (T)0;
                rc.pZZ[j] = row <= 0 ? (T)1 : (T)0;
                rc.border_y[j] = row <= 0 ? y_initial : (T)0;
                rc.rs[j] = 0;
            }
        }
#pragma unroll
        for (int j = 0; j < K; ++j) st.m[j] = st.x[j] = st.y[j] = st.xy[j] = (T)0;
        st.dm = st.dxy = (T)0;
        st.acc = (T)0;
        const bool last = (k == nstripes - 1);
        const int wl = last ? last_lane : G - 1;
        const bool writer = (L == wl);
        const Carry<T> *ring_rd = ring + G;
        Carry<T> *ring_wr = ring + G - wl;
        Carry<T> ca[4] = {ring_rd[0], ring_rd[1], ring_rd[2], ring_rd[3]}, cb[4] = {ca[0], ca[1], ca[2], ca[3]};
        const unsigned *hrd = hrd0;
        unsigned h_lo = hrd[0], h_hi = hrd[1];
        int s = 0;
#define MGL_PHK_BLOCK(PRO, EPI, HAP_N, ACC, CUR, NXT)                                                    \
    {                                                                                                    \
        const unsigned hw = __builtin_amdgcn_alignbyte(h_hi, h_lo, (unsigned)q_shift);                   \
        h_lo = h_hi;                                                                                     \
        h_hi = hrd[2];                                                                                   \
        phk_step4<T, G, K, PRO, EPI, HAP_N, ACC>(st, CUR, NXT, ring_rd + 4, ring_wr, hw, rc, s, L, H, writer); \
        ring_rd += 4;                                                                                    \
        ring_wr += 4;                                                                                    \
        hrd += 1;                                                                                        \
        s += 4;                                                                                          \
    }
#define MGL_PHK_PAIR(PRO, EPI, HAP_N, ACC) { MGL_PHK_BLOCK(PRO, EPI, HAP_N, ACC, ca, cb) MGL_PHK_BLOCK(PRO, EPI, HAP_N, ACC, cb, ca) }
#define MGL_PHK_STRIPE(HAP_N, ACC)                                                                       \
    {                                                                                                    \
        for (; s < G;) MGL_PHK_PAIR(true, true, HAP_N, ACC)                                              \
        for (; s < lean_end8;) MGL_PHK_PAIR(false, false, HAP_N, ACC)                                    \
        for (; s < sps8;) MGL_PHK_PAIR(false, true, HAP_N, ACC)                                          \
    }
        if (any_n) {
            if (last) MGL_PHK_STRIPE(true, true) else MGL_PHK_STRIPE(true, false)
        } else {
            if (last) MGL_PHK_STRIPE(false, true) else MGL_PHK_STRIPE(false, false)
        }
#undef MGL_PHK_STRIPE
#undef MGL_PHK_PAIR
#undef MGL_PHK_BLOCK
        if (last) result = st.acc;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
    }
    if (L == last_lane && store) {
        const double rd = (double)result;
        if (!RESCUE) {
            const bool small = rd < MIN_ACCEPTED; // …PairHmm.cc:179-195
            a.need_double[p] = small ? 1 : 0;
            if (!small) a.out[p] = log10(rd) - Num<T>::log10_initial(a);
        } else if (!a.rescue_only || a.need_double[p] != 0) {
            a.out[p] = log10(rd) - Num<T>::log10_initial(a); // :203-209
        }
    }
}

// one pair per wave: picks the rows-per-lane variant for this pair's read length
template <typename T, bool RESCUE, int KMAX>
__device__ __forceinline__ void pairhmm_wave(const PhArgs &a, unsigned char *smem, const int64_t slot)
{
    if (slot >= a.n_pairs) return;
    if (RESCUE && a.rescue_only && a.need_double[slot] == 0) return;
    const int32_t ri = a.pair_read[slot], hi = a.pair_hap[slot];
    const int64_t r0 = a.read_off[ri], h0 = a.hap_off[hi];
    const int R = __builtin_amdgcn_readfirstlane((int)(a.read_off[ri + 1] - r0));
    const int H = __builtin_amdgcn_readfirstlane((int)(a.hap_off[hi + 1] - h0));
    const uint8_t *rbase = a.reads + 5 * r0;
    // (the kernel is compiled per KMAX so that batches of short reads do not pay the registers of the 4-row variant)
    const int k = min(KMAX, (R + 63) >> 6);
    if (KMAX == 1 || k <= 1)
        pairhmm_body_k<T, 64, 1, RESCUE>(a, smem, slot, true, R, H, rbase, h0);
    else if (KMAX == 2 || k == 2)
        pairhmm_body_k<T, 64, 2, RESCUE>(a, smem, slot, true, R, H, rbase, h0);
    else if (KMAX == 3 || k == 3)
        pairhmm_body_k<T, 64, 3, RESCUE>(a, smem, slot, true, R, H, rbase, h0);
    else
        pairhmm_body_k<T, 64, 4, RESCUE>(a, smem, slot, true, R, H, rbase, h0);
}

// several pairs per wave, G = 32 or 21 lanes x K rows each (reads up to G * KMAX bases: one stripe); K is the wave's:
// ceil(longest read / G)
template <typename T, int G, bool RESCUE, int KMAX>
__device__ __forceinline__ void pairhmm_wave_multi(const PhArgs &a, unsigned char *smem)
{
    constexpr int PW = 64 / G;
    const int grp = min((int)(threadIdx.x & 63) / G, PW - 1);
    const int64_t slot = (int64_t)blockIdx.x * PW + grp;
    if ((int64_t)blockIdx.x * PW >= a.n_pairs) return;
    const bool store = slot < a.n_pairs;
    const int64_t p = store ? slot : a.n_pairs - 1; // a partly filled last wave: the idle groups repeat the last pair, store nothing
    if (RESCUE && a.rescue_only) {
        const bool need = store && a.need_double[p] != 0;
        if (!__builtin_amdgcn_ballot_w64(need)) return;
    }
    const int32_t ri = a.pair_read[p], hi = a.pair_hap[p];
    const int64_t r0 = a.read_off[ri], h0 = a.hap_off[hi];
    const int R = (int)(a.read_off[ri + 1] - r0), H = (int)(a.hap_off[hi + 1] - h0);
    const uint8_t *rbase = a.reads + 5 * r0;
    int r_w = R;
#pragma unroll
    for (int g = 1; g < PW; ++g) r_w = max(r_w, max(__shfl(r_w, 0), __shfl(r_w, g * G)));
    const int r_max = __builtin_amdgcn_readfirstlane(r_w);
    const int k = min(KMAX, (r_max + G - 1) / G);
    if (k <= 2)
        pairhmm_body_k<T, G, 2, RESCUE>(a, smem, p, store, R, H, rbase, h0);
    else if (k == 3)
        pairhmm_body_k<T, G, 3, RESCUE>(a, smem, p, store, R, H, rbase, h0);
    else if (KMAX == 4 || k == 4)
        pairhmm_body_k<T, G, 4, RESCUE>(a, smem, p, store, R, H, rbase, h0);
    else
        pairhmm_body_k<T, G, 5, RESCUE>(a, smem, p, store, R, H, rbase, h0);
}

} // namespace

template <int KMAX>
__global__ __launch_bounds__(64) void pairhmm_float32_kernel(const PhArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pairhmm_wave_multi<float, 32, false, KMAX>(a, smem);
}
template <int KMAX>
__global__ __launch_bounds__(64) void pairhmm_float21_kernel(const PhArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pairhmm_wave_multi<float, 21, false, KMAX>(a, smem);
}
__global__ __launch_bounds__(64) void pairhmm_float_kernel(const PhArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pairhmm_body<float, 16, false>(a, smem);
}
template <int KMAX>
__global__ __launch_bounds__(64) void pairhmm_float64_kernel(const PhArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pairhmm_wave<float, false, KMAX>(a, smem, (int64_t)blockIdx.x);
}
__global__ __launch_bounds__(64) void pairhmm_double_kernel(const PhArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    pairhmm_body<double, 16, true>(a, smem);
}
template <int KMAX>
__global__ __launch_bounds__(64) void pairhmm_double64_kernel(const PhArgs a)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (!a.rescue_only) {
        pairhmm_wave<double, true, KMAX>(a, smem, (int64_t)blockIdx.x);
        return;
    }
    // rescue pass after the float pass: almost no pair needs it.  A few thousand waves sweep the flags 64 at a time (one
    // coalesced load + a ballot) instead of one wave per pair launching only to return: 0.33 -> 0.02 ms per 1.6 M pairs
    const int lane = threadIdx.x & 63;
    for (int64_t base = (int64_t)blockIdx.x * 64; base < a.n_pairs; base += (int64_t)gridDim.x * 64) {
        const bool flagged = base + lane < a.n_pairs && a.need_double[base + lane] != 0;
        unsigned long long todo = __builtin_amdgcn_ballot_w64(flagged);
        while (todo) {
            const int l = __builtin_ctzll(todo);
            todo &= todo - 1;
            pairhmm_wave<double, true, KMAX>(a, smem, base + l);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); // the next pair reuses the LDS carve
            __builtin_amdgcn_wave_barrier();
        }
    }
}

int ph_lds_bytes(int hap_cap, int rows, int elem_bytes)
{
    const int ring_entries = hap_cap + 2 * rows + 16;
    const int hap_bytes = (hap_cap + 2 * rows + 28 + 7) & ~7;
    const int64_t b = (int64_t)(64 / rows) * ((int64_t)ring_entries * 3 * elem_bytes + hap_bytes);
    return b > (1 << 30) ? (1 << 30) : (int)b;
}

template <typename K>
static hipError_t launch(K kernel, const PhArgs &a, int rows, int elem_bytes, hipStream_t stream, bool sweep = false)
{
    const int lds = ph_lds_bytes(a.hap_cap, rows, elem_bytes);
    if (lds > 64 * 1024) { // per device and rare (haplotypes beyond ~1 300 bases): set every time, no cache to go stale
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
    }
    const int per_wave = 64 / rows;
    int64_t blocks = (a.n_pairs + per_wave - 1) / per_wave; // one wave per block: the LDS carve, not the wave count, limits a CU
    if (sweep) blocks = std::min<int64_t>((a.n_pairs + 63) / 64, 4096); // rescue pass of the 64-lane double kernels: see there
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64), lds, stream, a);
    return hipGetLastError();
}

// rows_per_lane: read rows per lane of the one-pair-per-wave kernels = min(4, ceil(longest read / 64))
hipError_t launch_pairhmm_float(const PhArgs &a, int rows, int rows_per_lane, hipStream_t stream)
{
    if (rows == 32) // two pairs per wave; rows_per_lane = ceil(longest read / 32), 3 .. 5 (the host layer picks the range)
        return rows_per_lane <= 4 ? launch(pairhmm_float32_kernel<4>, a, 32, 4, stream) : launch(pairhmm_float32_kernel<5>, a, 32, 4, stream);
    if (rows == 21) // three pairs per wave; rows_per_lane = ceil(longest read / 21), up to 5 (reads up to 105 bases)
        return rows_per_lane <= 4 ? launch(pairhmm_float21_kernel<4>, a, 21, 4, stream) : launch(pairhmm_float21_kernel<5>, a, 21, 4, stream);
    if (rows != 64) return launch(pairhmm_float_kernel, a, 16, 4, stream);
    switch (rows_per_lane) {
    case 1: return launch(pairhmm_float64_kernel<1>, a, 64, 4, stream);
    case 2: return launch(pairhmm_float64_kernel<2>, a, 64, 4, stream);
    case 3: return launch(pairhmm_float64_kernel<3>, a, 64, 4, stream);
    default: return launch(pairhmm_float64_kernel<4>, a, 64, 4, stream);
    }
}

hipError_t launch_pairhmm_double(const PhArgs &a, int rows, int rows_per_lane, hipStream_t stream)
{
    if (rows != 64) return launch(pairhmm_double_kernel, a, 16, 8, stream);
    switch (rows_per_lane) {
    case 1: return launch(pairhmm_double64_kernel<1>, a, 64, 8, stream, a.rescue_only != 0);
    case 2: return launch(pairhmm_double64_kernel<2>, a, 64, 8, stream, a.rescue_only != 0);
    case 3: return launch(pairhmm_double64_kernel<3>, a, 64, 8, stream, a.rescue_only != 0);
    default: return launch(pairhmm_double64_kernel<4>, a, 64, 8, stream, a.rescue_only != 0);
    }
}

} // namespace mgl_ph_dev
