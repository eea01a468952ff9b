// sw_lane_cell.h -- the cell arithmetic shared by the kernels that keep a strip of R target rows in a lane's registers and sweep
// the query columns (sw_dp16_lane.hip: two PAIRS per lane; sw_dp16_strip.hip: two STRIPS of one long pair per lane): packed
// 16-bit helpers, the per-launch constants and column<R>(), R rows of one column for both halves of every register.  Device
// code only; every function is inline in an unnamed namespace (one copy per translation unit).
#ifndef MGL_SW_LANE_CELL_H
#define MGL_SW_LANE_CELL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_device.h"

namespace mgl_sw_dev {

namespace {

typedef short short2_t __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ short2_t as_s2(unsigned x) { return __builtin_bit_cast(short2_t, x); }
__device__ __forceinline__ unsigned as_u(short2_t x) { return __builtin_bit_cast(unsigned, x); }
__device__ __forceinline__ ushort2_t as_us2(unsigned x) { return __builtin_bit_cast(ushort2_t, x); }
// wrapping 16-bit arithmetic on the unsigned type (cells outside the matrix may wrap; they never feed a valid cell)
__device__ __forceinline__ unsigned pk_add(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, as_us2(a) + as_us2(b)); }
__device__ __forceinline__ unsigned pk_sub(unsigned a, unsigned b) { return __builtin_bit_cast(unsigned, as_us2(a) - as_us2(b)); }
__device__ __forceinline__ unsigned pk_sub_sat(unsigned a, unsigned b) { return as_u(__builtin_elementwise_sub_sat(as_s2(a), as_s2(b))); }
__device__ __forceinline__ unsigned pk_max(unsigned a, unsigned b) { return as_u(__builtin_elementwise_max(as_s2(a), as_s2(b))); }
__device__ __forceinline__ unsigned pk_min_u(unsigned a, unsigned b)
{
    return __builtin_bit_cast(unsigned, __builtin_elementwise_min(as_us2(a), as_us2(b)));
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

// One sequence of one lane, read as ALIGNED dwords (a dword that holds at least one byte of the sequence never leaves
// the page the sequence ends in, so nothing beyond the caller's array is touched whatever its alignment); bytes
// 4c .. 4c+3 of the sequence = alignbyte(dword c+1, dword c, start & 3), loads past the last dword are clamped.
struct SeqWords {
    const uint32_t *base;
    unsigned shift;
    int kmax;
    __device__ __forceinline__ void init(const uint8_t *p, int len)
    {
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        base = reinterpret_cast<const uint32_t *>(a & ~(uintptr_t)3);
        shift = (unsigned)(a & 3);
        kmax = (int)((((a + (uintptr_t)len - 1) & ~(uintptr_t)3) - (a & ~(uintptr_t)3)) >> 2);
    }
    __device__ __forceinline__ unsigned word(int k) const { return base[k < kmax ? k : kmax]; }
    __device__ __forceinline__ unsigned block(int c) const { return __builtin_amdgcn_alignbyte(word(c + 1), word(c), shift); }
    // consecutive blocks: `lo` carries dword c in and dword c + 1 out (one load per block)
    __device__ __forceinline__ unsigned next_block(int c, unsigned &lo) const
    {
        const unsigned hi = word(c + 1);
        const unsigned v = __builtin_amdgcn_alignbyte(hi, lo, shift);
        lo = hi;
        return v;
    }
};

// ---- base codes (sw_dp16_lane_ck.hip).  Staged targets hold one code 0 .. 3 per byte, staged queries 8 x code, or 32 for a base
// outside the target alphabet.
constexpr unsigned CODE_SEL = 0x0c040c00u; // a row's selector: or-ed onto {0, code B, 0, code A}
// the column's table of one pair from its staged query byte
__device__ __forceinline__ unsigned code_table(unsigned q8) { return 0x01010101u ^ (unsigned)(1ull << q8); }
// four ASCII bases -> four codes (A 0, C 1, T 2, G 3: bits 1 and 2 of the byte); `bad` collects the bytes that are not one of
// these four upper-case letters (nonzero = such a byte exists among the dword's first `valid` bytes)
__device__ __forceinline__ unsigned ascii_codes(unsigned w, unsigned &diff)
{
    const unsigned code = (w >> 1) & 0x03030303u;
    diff = __builtin_amdgcn_perm(0u, 0x47544341u, code) ^ w; // the letter each code stands for, against the byte itself
    return code;
}
__device__ __forceinline__ unsigned nonzero_bytes(unsigned x) { return (((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u; } // bit 7 of each
// four 2-bit bases (the low byte of x) -> four codes
__device__ __forceinline__ unsigned spread_2bit(unsigned x)
{
    const unsigned y = (x | (x << 12)) & 0x000f000fu;
    return (y | (y << 6)) & 0x03030303u;
}

// (a & k) | b in one VOP3 instruction, k in an SGPR (left to the compiler this becomes v_and_b32 + v_or_b32)
__device__ __forceinline__ unsigned and_or(unsigned a, unsigned k, unsigned b)
{
    unsigned r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(b));
    return r;
}

struct LaneConsts {
    unsigned delta, one, o_e, k2; // packed constants (both halves equal): mismatch-match, 1, o-e, match+2e
    // bit masks of the four rows of a traceback dword, held in SGPRs: gfx9 VOP3 takes no literal, and with the masks as
    // literals the compiler splits every v_and_or_b32 into v_and_b32 + v_or_b32 (two instructions more per cell)
    unsigned k12[4], k34[4];
};

// R rows of one column for both packed pairs.  h[r]: H[row r][j-1] on entry, H[row r][j] on exit; f[r]: F of row r (in);
// hd: H[row -1][j-1] (the strip's top row, previous column); hup / e: H and E coming down from the strip above.
// On exit e = E leaving the strip's last row, h[R-1] = H of its last row.  w: the R/4 traceback dwords of the column.
// MID: {H, E} leaving row 15 also go to *mid (sw_dp16_lane_ck.hip: the carry row of the strip's lower half).
// How a cell learns whether its two bases differ (m = 1 where they do, per half; raw byte compare, sw.cpp:55):
//   bytes (CODES = false)  t[r] = {target byte A, target byte B} in the low bytes of the halves, q likewise: xor, then min(., 1)
//   codes (CODES = true)   one v_perm_b32: q / q2 are the COLUMN's tables for pair A / B -- byte c = 0 if the query base has code c,
//                          else 1 -- and t[r] is the ROW's selector 0x0c, 4 + code B, 0x0c, code A (0x0c selects a zero byte), so
//                          the lookup lands m in both halves at once.  Codes 0 .. 3 are the target's alphabet; a query base outside
//                          it has the all-ones table.  One instruction less per two cells.
template <int R, bool NOTB, bool MID = false, bool CODES = false>
__device__ __forceinline__ void column(unsigned (&h)[R], unsigned (&f)[R], const unsigned (&t)[R], const unsigned q, unsigned hd,
                                       unsigned &e, const LaneConsts &c, uint4 *tbp, uint2 *mid = nullptr, const unsigned q2 = 0u)
{
    unsigned w[4];
    auto differ = [&](const int r) { return CODES ? __builtin_amdgcn_perm(q2, q, t[r]) : pk_min_u(q ^ t[r], c.one); };
    // the diagonal of row r + 1 is taken from H[r][j-1] BEFORE row r overwrites it with H[r][j] (so that H stays in place,
    // no copy per row), one row ahead of the recurrence
    unsigned dg = pk_add(hd, pk_mad(differ(0), c.delta, c.k2));
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const unsigned diag = dg;
        if (r + 1 < R) {
            const unsigned s = pk_mad(differ(r + 1), c.delta, c.k2); // match + 2e or mismatch + 2e
            dg = pk_add(h[r], s);
        }
        const unsigned fr = f[r];
        const unsigned sm = pk_max(diag, fr);
        const unsigned hn = pk_max(sm, e);              // sw.cpp:60-71: diag >= F >= E priority via the two strict flags below
        const unsigned open = pk_sub(hn, c.o_e);        // a new gap, either direction
        const unsigned eo = pk_max(open, e);            // extension is free in this representation (sw.cpp:73-93)
        const unsigned fo = pk_max(open, fr);
        if (!NOTB) {
            const unsigned d1 = pk_sub_sat(diag, fr);   // < 0 <=> F > diag
            const unsigned d2 = pk_sub_sat(sm, e);      // < 0 <=> E > max(diag, F)
            const unsigned d3 = pk_sub_sat(e, open);    // < 0 <=> a new vertical gap wins
            const unsigned d4 = pk_sub_sat(fr, open);   // < 0 <=> a new horizontal gap wins
            const unsigned p12 = __builtin_amdgcn_perm(d1, d2, 0x0b0a0908u); // sign bytes [d2.A, d2.B, d1.A, d1.B]
            const unsigned p34 = __builtin_amdgcn_perm(d3, d4, 0x0b0a0908u); //            [d4.A, d4.B, d3.A, d3.B]
            const int U = r & 3;
            const unsigned low = U == 0 ? 0u : w[(r >> 2) & 3];
            w[(r >> 2) & 3] = and_or(p34, c.k34[U], U == 0 ? (p12 & c.k12[0]) : and_or(p12, c.k12[U], low));
            // the 16 rows' flags leave right here (1 KB per wave), in the block that computed them: the differences
            // they are made of must not stay live until the end of the column
            if ((r & 15) == 15 || r == R - 1) tbp[(size_t)(r >> 4) * 64] = make_uint4(w[0], w[1], w[2], w[3]); // (a last group of fewer than 16 rows: R = 20, 24, 28)
        }
        h[r] = hn;
        f[r] = fo;
        e = eo;
        if (MID && r == 15) *mid = make_uint2(hn, eo);
        // F' is only needed in the next column: left alone, the compiler sinks its max to the end of the loop body and keeps
        // `open` and the old F of all R rows alive until there
        asm volatile("" : "+v"(f[r]), "+v"(h[r]));
    }
}

} // namespace

} // namespace mgl_sw_dev
#endif
