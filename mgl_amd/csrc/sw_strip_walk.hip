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
constexpr int FW = (BC + 64) / 8;     // flag dwords per row: one nibble per STEP of the block's wavefront (columns + rows - 1 steps)

__device__ __forceinline__ int dpp_shr1(int lane0_value, int src) { return __builtin_amdgcn_update_dpp(lane0_value, src, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int border_of(int k, int gopen, int gext, bool indel) { return (indel && k > 0) ? -gopen - (k - 1) * gext : 0; } // sw.cpp:29-40,47-49

struct BlockMoves {
    // the pair
    const SeqSet *t, *q;
    int64_t t0, q0;
    int tl, ql, match, mismatch, gopen, gext;
    bool indel;
    int rbk, kcols;            // rows per band (K strips), CPS * K: how far a band's checkpoint columns lie before the band above's
    const int2 *rows, *ck;     // what the fill kept (strip16_ck_words)
    int pack;                  // ... as {H, gap value} pairs (0) or packed into one int32 each (strip16_pack_bits)
    int row_stride, tl_cap;    // entries per kept row (column j at j - 1), rows per checkpoint column
    unsigned *flags;           // LDS: [64 rows][FW] dwords
    unsigned char *qb;         // LDS: the block's query bases
    int lane;
    // the block whose flags are in LDS: rows r0 + 1 .. imax, columns jl + 1 .. jr.  The walk only ever moves up and to the left, so a
    // block is computed up to the cell the walk enters it at and no further (on average half its columns and half its rows: the
    // wavefront takes columns + rows steps)
    int cur_b = -1, cur_cc = -1, r0 = 0, jl = 0, jr = 0, imax = 0;

    __device__ __forceinline__ int cc_of(int b, int j) const { return (j - 1 + kcols * b) / BC; }
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
        __builtin_amdgcn_wave_barrier();
        for (int x = lane; x < nc; x += 64) qb[x] = (unsigned char)q->at(q0, jl + x);
        const int tb_ = row_ok ? t->at(t0, i - 1) : -1;
        // left border: H[i][jl], F entering column jl + 1
        int hleft, f;
        if (jl == 0) {
            hleft = border_of(i, gopen, gext, indel);
            f = hleft - gopen;
        } else {
            if (pack) {
                strip16_unpack(row_ok ? reinterpret_cast<const int *>(ck)[(size_t)cc * (tl_cap + 1) + i] : 0, pack, hleft, f);
            } else {
                const int2 v = row_ok ? ck[(size_t)cc * (tl_cap + 1) + i] : make_int2(0, 0);
                hleft = v.x;
                f = v.y;
            }
        }
        // H[r0][jl]: the first diagonal of lane 0
        auto top = [&](int j, int &h_, int &e_) { // {H[r0][j], E entering row r0 + 1 at column j}
            if (b == 0) {
                h_ = border_of(j, gopen, gext, indel);
                e_ = h_ - gopen;
            } else {
                if (pack) {
                    strip16_unpack(reinterpret_cast<const int *>(rows)[(size_t)(b - 1) * row_stride + (j - 1)], pack, h_, e_);
                } else {
                    const int2 v = rows[(size_t)(b - 1) * row_stride + (j - 1)];
                    h_ = v.x;
                    e_ = v.y;
                }
            }
        };
        int corner = jl == 0 ? border_of(r0, gopen, gext, indel) : 0, dummy = 0;
        if (jl > 0) top(jl, corner, dummy);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        // hdiag: H of the row above at the column before this lane's current one (the diagonal input)
        int hdiag = dpp_shr1(corner, hleft); // lane l: H[i-1][jl] = the lane above's left border; lane 0: the corner
        int h_out = hleft, e_out = 0;        // what this lane handed down in its last step
        const int steps = nc + nr - 1;
        int tops_h[2] = {0, 0}, tops_e[2] = {0, 0}; // the kept row, 64 columns per fetch, one fetch ahead
        auto fetch = [&](int s0, int &h_, int &e_) {
            const int j = jl + 1 + s0 + lane;
            if (j <= jr) top(j, h_, e_);
        };
        fetch(0, tops_h[0], tops_e[0]);
        const int gopen_ = gopen, gext_ = gext, match_ = match, mismatch_ = mismatch;
        for (int s0 = 0; s0 < steps; s0 += 64) {
            fetch(s0 + 64, tops_h[1], tops_e[1]);
            for (int s8 = s0; s8 < min(s0 + 64, steps); s8 += 8) {
                unsigned acc = 0;
#pragma unroll
                for (int ds = 0; ds < 8; ++ds) {
                    const int s = s8 + ds, c = s - lane; // this lane's column within the block, 0-based (steps past the last: idle for every lane)
                    // what comes down from the row above: lane 0 from the kept row (column s), the others from the lane above's last step
                    const int l0h = __builtin_amdgcn_readlane(tops_h[0], s & 63), l0e = __builtin_amdgcn_readlane(tops_e[0], s & 63);
                    const int hup = dpp_shr1(l0h, h_out), e = dpp_shr1(l0e, e_out);
                    const bool active = row_ok && (unsigned)c < (unsigned)nc;
                    const int sub = (int)qb[min(max(c, 0), BC - 1)] == tb_ ? match_ : mismatch_;
                    const int diag = hdiag + sub;
                    const unsigned dF = f > diag;
                    const int sm = max(diag, f);
                    const unsigned dE = e > sm;
                    const int hn = max(sm, e);
                    const int open = hn - gopen_, ee = e - gext_, fe = f - gext_;
                    const unsigned eo = open > ee, fo = open > fe; // a new gap wins only strictly (sw.cpp:73-93)
                    e_out = active ? max(open, ee) : e_out;
                    f = active ? max(open, fe) : f;
                    h_out = active ? hn : h_out;
                    acc |= (dF | (dE << 1) | (eo << 2) | (fo << 3)) << (4 * ds);
                    hdiag = hup;
                }
                flags[lane * FW + (s8 >> 3)] = acc; // (the nibbles of idle steps are never read)
            }
            tops_h[0] = tops_h[1];
            tops_e[0] = tops_e[1];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
    }
    __device__ __forceinline__ unsigned cell(int i, int j)
    {
        const int b = (i - 1) / rbk, cc = cc_of(b, j);
        if (b != cur_b || cc != cur_cc || i > imax || j > jr) compute(b, cc, i, j);
        const int l = i - r0 - 1, st = (j - jl - 1) + l; // row of the block, step of its wavefront
        return (flags[l * FW + (st >> 3)] >> (4 * (st & 7))) & 15u;
    }
    // the number of consecutive diagonal moves from (i, j) inside the block that holds it, up to 64: lane k looks at the flags of cell
    // (i - k, j - k) -- its own dword of the LDS array -- and one ballot finds where the run ends.  An ONT-style path is a diagonal broken
    // by an indel every ten cells or so: the walk takes a run per look instead of a cell per look (every look is a dependent LDS read).
    __device__ __forceinline__ int diag_run(int i, int j)
    {
        const int b = (i - 1) / rbk, cc = cc_of(b, j);
        if (b != cur_b || cc != cur_cc || i > imax || j > jr) compute(b, cc, i, j);
        const int ii = i - lane, jj = j - lane;
        bool is_diag = false;
        if (ii > r0 && jj > jl) {
            const int l = ii - r0 - 1, st = (jj - jl - 1) + l;
            is_diag = ((flags[l * FW + (st >> 3)] >> (4 * (st & 7))) & 3u) == 0u; // neither F > diag nor E > max(diag, F): sw.cpp:60-62
        }
        const unsigned long long m = __builtin_amdgcn_ballot_w64(is_diag);
        return m == ~0ull ? 64 : (int)__builtin_ctzll(~m);
    }
    // +k rows up, -k columns left, 0 diagonal: what the reference stores (sw.cpp:60-71); run lengths as TbView::vrun / hrun
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
            for (int k = j - 1; k >= 1 && !(cell(i, k) & 8u); --k) ++n;
            return -n;
        }
        return 0;
    }
};

} // namespace

__global__ __launch_bounds__(64) void sw_strip_ck_walk_kernel(const TbArgs a, const int tl_cap, const int ql_cap)
{
    __shared__ unsigned flags[64 * FW];
    __shared__ unsigned char qb[BC];
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
    mv.rows = reinterpret_cast<const int2 *>(a.tb + (size_t)slot * (size_t)a.tb_stride_words);
    mv.row_stride = strip16_ck_row_stride(ql_cap);
    mv.pack = a.strip_pack;
    // (packed entries are one int32 each: the checkpoints start half as far in)
    mv.ck = a.strip_pack ? reinterpret_cast<const int2 *>(reinterpret_cast<const int *>(mv.rows) + (size_t)strip16_ck_bands(tl_cap, a.strip_rows, a.strip_k) * mv.row_stride)
                         : mv.rows + (size_t)strip16_ck_bands(tl_cap, a.strip_rows, a.strip_k) * mv.row_stride;
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
