// sw_strip_walk.hip -- the path walk for long reads whose fill kept NO traceback flags (sw_dp16_strip.hip with DpArgs::strip_k > 0,
// traceback layout 6): one wave per pair walks the path (sw.cpp:149-255) and RECOMPUTES the flags of the block it stands in -- the rows
// of one band of K strips (at most 64) x at most STRIP_CK_COLS columns -- from what the fill kept: the true scores {H, E} of the row
// above the band and {H, F} of every row at the band's checkpoint column to the left (strip16_ck_words in sw_device.h).  A 10 kb x
// 10 kb path crosses ~200 such blocks, 3 M of the matrix's 100 M cells.
//
// The block is an anti-diagonal wavefront in plain int32 (no offsets, no range question): lane l owns row r0 + l of the block and is
// at column c = step - l; H and the vertical-gap value of the row above arrive by DPP from lane l - 1, lane 0 takes them from the kept
// row; what a lane needs from its own row (H of the column before, the horizontal-gap value) it keeps.  The recurrence and its four
// decisions are those of every fill kernel here (sw.cpp:51-93): diag = H[i-1][j-1] + (t == q ? match : mismatch); F > diag; E >
// max(diag, F); H = max; E' = max(H - o, E - e) and F' likewise, extension winning ties -- so the nibble of a cell {F > diag, E > max,
// E' opened, F' opened} is bit for bit what the fill would have stored.  Flags go to LDS, one dword per lane per eight STEPS (cell
// (row l, column c) of the block sits at step c + l: the store is the same instruction for every lane, no lane-dependent branch), and
// the walk reads them there: every lane runs the same walk (wave-uniform state), lane 0 stores the text.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_device.h"
#include "sw_traceback.h"

namespace mgl_sw_dev {

namespace {

constexpr int BC = STRIP_CK_COLS;     // columns per block
constexpr int FW4 = (BC + 64 + 31) / 32 + 1; // uint4 entries per row of the flag array: one per 32 STEPS of the block's wavefront (columns + rows - 1 steps), + 1 against bank conflicts

__device__ __forceinline__ int dpp_shr1(int lane0_value, int src) { return __builtin_amdgcn_update_dpp(lane0_value, src, 0x138, 0xf, 0xf, false); } // wave_shr:1, lane 0 keeps lane0_value
__device__ __forceinline__ int dpp_rol1(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x134, 0xf, 0xf, false); }                                  // wave_rol:1: lane k takes lane k + 1's, lane 63 lane 0's
__device__ __forceinline__ int border_of(int k, int gopen, int gext, bool indel) { return (indel && k > 0) ? -gopen - (k - 1) * gext : 0; } // sw.cpp:29-40,47-49

// Round 4: the block's wavefront in the representation the fill kernels use -- every value carries + (i + j) e, so extending a gap costs
// nothing and both new gaps start from the one H - (o - e) -- with its four decisions taken as the SIGN BITS of four differences, shifted
// into four bit-plane accumulators by one v_alignbit_b32 each (a cell: 4 subtractions + 4 shifts where compare, select, shift and or took
// 12), the kept row handed to lane 0 by rotating its 64 columns one lane per step (a readlane, a move and a DPP move per value before), the
// query read as a dword per four steps, and no lane masked once the last row's lane has started: 22 instructions per step where there
// were 38.  With the checkpoint columns every 128 columns instead of 256 (a block: 7 KB of flags in LDS instead of 11, so that 4 608 pairs'
// waves are resident at once; its wavefront a third shorter) the walk kernel takes 9.4 ms per 4 608 pairs of 10 kb where it took 14.3.  The decisions compare values of ONE cell, all shifted alike: bit for
// bit what the fill would have stored (sw.cpp:60-93).
struct BlockMoves {
    // the pair
    const SeqSet *t, *q;
    int64_t t0, q0;
    int tl, ql, match, mismatch, gopen, gext;
    bool indel;
    int rbk, kcols;            // rows per band (K strips), CPS * K: how far a band's checkpoint columns lie before the band above's
    // what the fill kept (strip16_ck_words): dwords {H, gap value} as 16-bit values of the strips' registers, beside the strips' baselines --
    // value + baseline = score + (row + column) e, the representation compute() works in
    const int *rows, *rows_base;  // the row below every band [band][column j at j - 1] and its baselines [band][(j - 1) / 16]
    const int *ck_cols, *ck_base; // the checkpoint columns [cc][row] and their baselines [cc][strip]
    int row_stride, row_blocks, tl_cap; // entries per kept row, baselines per kept row, rows per checkpoint column
    int strip_rows, ck_strips;    // rows per strip, baselines per checkpoint column
    uint4 *flags;              // LDS: [64 rows][FW4]: the bit planes {F > diag, E > max(diag, F), E' opened, F' opened} of 32 steps each, step t of its 32 at bit 31 - t
    unsigned char *qb;         // LDS: the block's query bases (BC + 8 bytes)
    int lane;
    // the block whose flags are in LDS: rows r0 + 1 .. imax, columns jl + 1 .. jr.  The walk only ever moves up and to the left, so a
    // block is computed up to the cell the walk enters it at and no further (on average half its columns and half its rows: the
    // wavefront takes columns + rows steps)
    int cur_b = -1, cur_cc = -1, r0 = 0, jl = 0, jr = 0, imax = 0;

    __device__ __forceinline__ int cc_of(int b, int j) const { return (j - 1 + kcols * b) / BC; }
    // {H[r0][j], E entering row r0 + 1 at column j}, each + (its row + column) e
    __device__ __forceinline__ void top(int b, int j, int &h_, int &e_) const
    {
        if (b == 0) {
            const int hb = border_of(j, gopen, gext, indel);
            h_ = hb + j * gext;
            e_ = hb - gopen + (j + 1) * gext;
        } else {
            const int v = rows[(size_t)(b - 1) * row_stride + (j - 1)], base = rows_base[(b - 1) * row_blocks + ((j - 1) >> 4)];
            h_ = (int)(short)(v & 0xffff) + base;
            e_ = (v >> 16) + base;
        }
    }
    // four steps of the wavefront.  MASKED: some lane has not reached its first column yet (it keeps what it holds); PLANES: the accumulators
    template <bool MASKED>
    __device__ __forceinline__ void steps4(const int s, const unsigned qw, const int tb_, const int match2, const int mismatch2, const int o_e, int &tops_h, int &tops_e,
                                           int &hdiag, int &h_out, int &e_out, int &f, unsigned (&acc)[4]) const
    {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // what comes down from the row above: lane 0 from the kept row (its lane 0 holds this step's column), the others from the lane above's last step
            const int hup = dpp_shr1(tops_h, h_out), e = dpp_shr1(tops_e, e_out);
            tops_h = dpp_rol1(tops_h);
            tops_e = dpp_rol1(tops_e);
            const int diag = hdiag + ((int)((qw >> (8 * u)) & 0xffu) == tb_ ? match2 : mismatch2);
            const int sm = max(diag, f);
            const int hn = max(sm, e);
            const int open = hn - o_e;
            // the four decisions (sw.cpp:60-93) as sign bits: F > diag, E > max(diag, F), a new vertical gap wins (strictly), a new horizontal one
            acc[0] = __builtin_amdgcn_alignbit(acc[0], (unsigned)(diag - f), 31);
            acc[1] = __builtin_amdgcn_alignbit(acc[1], (unsigned)(sm - e), 31);
            acc[2] = __builtin_amdgcn_alignbit(acc[2], (unsigned)(e - open), 31);
            acc[3] = __builtin_amdgcn_alignbit(acc[3], (unsigned)(f - open), 31);
            const int en = max(open, e), fn = max(open, f);
            if (MASKED) {
                const bool started = s + u >= lane;
                h_out = started ? hn : h_out;
                f = started ? fn : f;
            } else {
                h_out = hn;
                f = fn;
            }
            e_out = en; // (before a lane's first column: read by nobody -- the lane below starts a step later)
            hdiag = hup;
        }
    }
    // flags of rows rbk b + 1 .. and columns jl + 1 .. jr into LDS
    __device__ void compute(int b, int cc, int i_in, int j_in)
    {
        cur_b = b;
        cur_cc = cc;
        r0 = rbk * b;
        jl = max(BC * cc - kcols * b, 0);
        jr = j_in;
        imax = i_in;
        const int nr = imax - r0, nc = jr - jl;
        const int i = r0 + lane + 1;                     // this lane's row
        const bool row_ok = lane < nr;
        const int ge = gext;
        __builtin_amdgcn_wave_barrier();
        for (int x = lane; x < nc; x += 64) qb[x] = (unsigned char)q->at(q0, jl + x);
        const int tb_ = row_ok ? t->at(t0, i - 1) : -1;
        // left border: H[i][jl], F entering column jl + 1 (rows past the block's last: anything -- they feed nothing that is read)
        int hleft, f;
        // (every value + (row + column) e from here on: the representation the checkpoints are kept in)
        if (jl == 0) {
            hleft = border_of(i, gopen, ge, indel);
            f = hleft - gopen + (i + 1) * ge;
            hleft += i * ge;
        } else {
            const int v = row_ok ? ck_cols[(size_t)cc * (tl_cap + 1) + i] : 0;
            const int base = row_ok ? ck_base[(size_t)cc * ck_strips + (i - 1) / strip_rows] : 0;
            hleft = (int)(short)(v & 0xffff) + base;
            f = (v >> 16) + base;
        }
        // H[r0][jl]: the first diagonal of lane 0
        int corner = border_of(r0, gopen, ge, indel) + r0 * ge, dummy = 0;
        if (jl > 0) top(b, jl, corner, dummy);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // hdiag: H of the row above at the column before this lane's current one (the diagonal input)
        int hdiag = dpp_shr1(corner, hleft); // lane l: H[i-1][jl] = the lane above's left border; lane 0: the corner
        int h_out = hleft, e_out = 0;        // what this lane handed down in its last step
        const int steps = (nc + nr - 1 + 3) & ~3; // (whole groups of four: the steps past the last compute cells nobody reads)
        // the kept row above the block, 64 columns per fetch (lane k: column jl + 1 + s0 + k), one fetch ahead
        auto fetch = [&](int s0, int &h_, int &e_) {
            const int j = jl + 1 + s0 + lane;
            h_ = e_ = 0;
            if (j <= jr) top(b, j, h_, e_);
        };
        int tops_h, tops_e, next_h, next_e;
        fetch(0, tops_h, tops_e);
        const int match2 = match + 2 * ge, mismatch2 = mismatch + 2 * ge, o_e = gopen - ge;
        // this lane's query bases of four steps: bytes s - lane .. + 3 of the block's, out of two aligned dwords
        const unsigned *const qd = reinterpret_cast<const unsigned *>(qb);
        const unsigned qsh = (unsigned)(-lane) & 3u;
        auto qword = [&](int s) { return qd[min(max(((s - lane) >> 2) + 1, 0), BC / 4 + 1)]; }; // the dword behind the one that holds byte s - lane
        unsigned q_lo = qd[min(max((0 - lane) >> 2, 0), BC / 4 + 1)], q_hi = qword(0);
        unsigned acc[4] = {0u, 0u, 0u, 0u};
        const int ramp = min(steps, (min(nr, 64) + 3) & ~3); // from here on every lane that owns a row has started
        for (int s0 = 0; s0 < steps; s0 += 64) {
            fetch(s0 + 64, next_h, next_e);
            const int s_end = min(s0 + 64, steps);
            for (int s = s0; s < s_end; s += 4) {
                const unsigned qw = __builtin_amdgcn_alignbyte(q_hi, q_lo, qsh);
                q_lo = q_hi;
                q_hi = qword(s + 4);
                if (s < ramp)
                    steps4<true>(s, qw, tb_, match2, mismatch2, o_e, tops_h, tops_e, hdiag, h_out, e_out, f, acc);
                else
                    steps4<false>(s, qw, tb_, match2, mismatch2, o_e, tops_h, tops_e, hdiag, h_out, e_out, f, acc);
                if (((s + 4) & 31) == 0) flags[lane * FW4 + (s >> 5)] = make_uint4(acc[0], acc[1], acc[2], acc[3]);
            }
            tops_h = next_h;
            tops_e = next_e;
        }
        if (steps & 31) { // the last, partial group of 32 steps: its first step to bit 31 like everywhere
            const int sh = 32 - (steps & 31);
            flags[lane * FW4 + (steps >> 5)] = make_uint4(acc[0] << sh, acc[1] << sh, acc[2] << sh, acc[3] << sh);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    // one decision of cell (i, j): plane 0 F > diag, 1 E > max(diag, F), 2 E' opened, 3 F' opened
    __device__ __forceinline__ unsigned bit(int i, int j, int plane)
    {
        const int b = (i - 1) / rbk, cc = cc_of(b, j);
        if (b != cur_b || cc != cur_cc || i > imax || j > jr) compute(b, cc, i, j);
        const int l = i - r0 - 1, st = (j - jl - 1) + l; // row of the block, step of its wavefront
        return (reinterpret_cast<const unsigned *>(flags)[(l * FW4 + (st >> 5)) * 4 + plane] >> (31 - (st & 31))) & 1u;
    }
    // the number of consecutive diagonal moves from (i, j) inside the block that holds it, up to 64: lane k looks at the flags of cell
    // (i - k, j - k) -- its own entry of the LDS array -- and one ballot finds where the run ends.  An ONT-style path is a diagonal broken
    // by an indel every ten cells or so: the walk takes a run per look instead of a cell per look (every look is a dependent LDS read).
    __device__ __forceinline__ int diag_run(int i, int j)
    {
        const int b = (i - 1) / rbk, cc = cc_of(b, j);
        if (b != cur_b || cc != cur_cc || i > imax || j > jr) compute(b, cc, i, j);
        const int ii = i - lane, jj = j - lane;
        bool is_diag = false;
        if (ii > r0 && jj > jl) {
            const int l = ii - r0 - 1, st = (jj - jl - 1) + l;
            const uint2 w = reinterpret_cast<const uint2 *>(flags)[(l * FW4 + (st >> 5)) * 2];
            is_diag = (((w.x | w.y) >> (31 - (st & 31))) & 1u) == 0u; // neither F > diag nor E > max(diag, F): sw.cpp:60-62
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(is_diag);
        return m == ~0ull ? 64 : (int)__builtin_ctzll(~m);
    }
    // +k rows up, -k columns left, 0 diagonal: what the reference stores (sw.cpp:60-71); run lengths as TbView::vrun / hrun
    __device__ __forceinline__ int at(int i, int j)
    {
        if (bit(i, j, 1)) { // E > max(diag, F): up, through the cells that extended the gap
            int n = 1;
            for (int r = i - 1; r >= 1 && !bit(r, j, 2); --r) ++n;
            return n;
        }
        if (bit(i, j, 0)) { // F > diag: left
            int n = 1;
            for (int k = j - 1; k >= 1 && !bit(i, k, 3); --k) ++n;
            return -n;
        }
        return 0;
    }
};

} // namespace

__global__ __launch_bounds__(64) void sw_strip_ck_walk_kernel(const TbArgs a, const int tl_cap, const int ql_cap)
{
    __shared__ uint4 flags[64 * FW4];
    __shared__ __attribute__((aligned(16))) unsigned char qb[BC + 16];
    const int lane = threadIdx.x;
    const int64_t slot = blockIdx.x;
    const int64_t p = a.first + slot;
    const int tl = a.t.length(p), ql = a.q.length(p);
    const DpRecord r = a.rec[slot];
    const int64_t o = a.dest ? a.dest[p] : p;
    char *const slot_out = a.cigar + (size_t)o * a.cigar_stride;

    CigarWriter cw;
    cw.slot = slot_out;
    cw.binary = a.binary_cigar;
    cw.cap = a.binary_cigar ? (a.cigar_stride & ~3) : a.cigar_stride;
    cw.pos = cw.cap;
    cw.need = 0;
    cw.store = (lane == 0);

    BlockMoves mv;
    mv.t = &a.t;
    mv.q = &a.q;
    mv.t0 = a.t.off[p];
    mv.q0 = a.q.off[p];
    mv.tl = tl;
    mv.ql = ql;
    mv.match = a.match;
    mv.mismatch = a.mismatch;
    mv.gopen = a.gopen;
    mv.gext = a.gext;
    mv.indel = (a.strategy & (OS_INDEL | OS_LEAD_ID)) != 0;
    mv.rbk = a.strip_rows * a.strip_k;
    mv.kcols = STRIP_CPS * a.strip_k;
    mv.rows = reinterpret_cast<const int *>(a.tb + (size_t)slot * (size_t)a.tb_stride_words);
    mv.row_stride = strip16_ck_row_stride(ql_cap);
    mv.row_blocks = strip16_ck_row_blocks(ql_cap);
    mv.rows_base = mv.rows + strip16_ck_off_rowbase(tl_cap, ql_cap, a.strip_rows, a.strip_k);
    mv.ck_cols = mv.rows + strip16_ck_off_cols(tl_cap, ql_cap, a.strip_rows, a.strip_k);
    mv.ck_base = mv.rows + strip16_ck_off_base(tl_cap, ql_cap, a.strip_rows, a.strip_k);
    mv.ck_strips = strip16_ck_strips(tl_cap, a.strip_rows);
    mv.strip_rows = a.strip_rows;
    mv.tl_cap = tl_cap;
    mv.flags = flags;
    mv.qb = qb;
    mv.lane = lane;
    const int off = walk_and_write(mv, tl, ql, a.strategy, r.max_t, r.max_q, r.mqe_t, r.seg, cw);

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

hipError_t launch_strip_ck_walk(const TbArgs &a, int max_tl, int max_ql, hipStream_t stream)
{
    if (a.strip_rows * a.strip_k > 64 || a.strip_k < 1) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sw_strip_ck_walk_kernel, dim3((unsigned)a.count), dim3(64), 0, stream, a, max_tl, max_ql);
    return hipGetLastError();
}

} // namespace mgl_sw_dev
