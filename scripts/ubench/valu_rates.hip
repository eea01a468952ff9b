// valu_rates.hip -- issue cost of the integer VALU instructions the SW kernels are built from, gfx950.
// For each instruction: a loop of UNROLL copies on NREG independent registers (or one dependent chain),
// 16 waves per CU (4 per SIMD), all 256 CUs.  Reports SIMD cycles per wave-instruction, using the in-kernel
// clock (s_memtime / s_memrealtime).   Build: hipcc --offload-arch=gfx950 -O3 valu_rates.hip -o valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#include <cstring>

#define CHECK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); return 1; } } while (0)

static int ITERS = 2048;
constexpr int UNROLL = 32;

// 8 independent registers r0..r7 and two sources a,b
#define REP8(I) I(0) I(1) I(2) I(3) I(4) I(5) I(6) I(7)
#define REP32(I) REP8(I) REP8(I) REP8(I) REP8(I)

#define KERNEL(NAME, ASM_INDEP, ASM_DEP)                                                              \
    __global__ __launch_bounds__(256) void k_##NAME(unsigned *out, unsigned long long *clk, int dep, int ITERS)  \
    {                                                                                                 \
        unsigned r0 = threadIdx.x, r1 = r0 * 3, r2 = r0 * 5, r3 = r0 * 7, r4 = r0 * 11, r5 = r0 * 13, \
                 r6 = r0 * 17, r7 = r0 * 19;                                                          \
        unsigned a = threadIdx.x * 2654435761u + blockIdx.x, b = a ^ 0x5bd1e995u;                     \
        unsigned long long t0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();  \
        if (dep) {                                                                                    \
            for (int i = 0; i < ITERS; ++i) {                                                         \
                asm volatile(ASM_DEP : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),    \
                             "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");                            \
            }                                                                                         \
        } else {                                                                                      \
            for (int i = 0; i < ITERS; ++i) {                                                         \
                asm volatile(ASM_INDEP : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5),  \
                             "+v"(r6), "+v"(r7) : "v"(a), "v"(b) : "vcc");                            \
            }                                                                                         \
        }                                                                                             \
        unsigned long long t1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();  \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 ^ r1 ^ r2 ^ r3 ^ r4 ^ r5 ^ r6 ^ r7;           \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }   \
    }

// helper to write one instruction applied to register %N (independent) or always %0 (dependent chain)
#define I_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define D_ADD(n) "v_add_u32 %0, %0, %8\n"
#define I_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define I_MAX(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define D_MAX(n) "v_max_i32 %0, %0, %8\n"
#define I_ALIGN(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 31\n"
#define D_ALIGN(n) "v_alignbit_b32 %0, %0, %8, 31\n"
#define I_DPP(n) "v_mov_b32_dpp %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define D_DPP(n) "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_ADDDPP(n) "v_add_u32_dpp %" #n ", %8, %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_CMPSDWA(n) "v_cmp_eq_u32_sdwa vcc, %8, %9 src0_sel:BYTE_1 src1_sel:DWORD\nv_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define I_CMP(n) "v_cmp_eq_u32 vcc, %8, %" #n "\nv_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define I_CMPADDC(n) "v_cmp_gt_i32 vcc, %8, %" #n "\nv_addc_co_u32 %" #n ", vcc, %" #n ", %" #n ", vcc\n"
#define I_MAX3(n) "v_max3_i32 %" #n ", %" #n ", %8, %9\n"
#define I_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define I_PKADD(n) "v_pk_add_i16 %" #n ", %" #n ", %8\n"
#define D_PKADD(n) "v_pk_add_i16 %0, %0, %8\n"
#define I_PKSUB(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n"
#define I_PKMAX(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define D_PKMAX(n) "v_pk_max_i16 %0, %0, %8\n"
#define I_PKMINU(n) "v_pk_min_u16 %" #n ", %" #n ", %8\n"
#define I_PKMAD(n) "v_pk_mad_i16 %" #n ", %" #n ", %8, %9\n"
#define I_PKASHR(n) "v_pk_ashrrev_i16 %" #n ", 15, %" #n "\n"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define I_BFI(n) "v_bfi_b32 %" #n ", %8, %" #n ", %9\n"
#define I_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define I_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 1, %8\n"
#define I_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define I_LSHR(n) "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define I_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define I_SADU8(n) "v_sad_u8 %" #n ", %8, %9, %" #n "\n"
#define I_MINMAX(n) "v_max_i16 %" #n ", %" #n ", %8\n"
#define I_ADDI16SAT(n) "v_pk_add_i16 %" #n ", %" #n ", %8 clamp\n"
#define I_CNDMASK(n) "v_cndmask_b32 %" #n ", %8, %9, vcc\n"
#define I_SUBB(n) "v_subrev_u32 %" #n ", %8, %" #n "\n"
#define I_MBCNT(n) "v_max_u16 %" #n ", %" #n ", %8\n"

#define I_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define I_OR(n) "v_or_b32 %" #n ", %" #n ", %8\n"
#define I_MIN(n) "v_min_i32 %" #n ", %" #n ", %8\n"
#define I_MAXU(n) "v_max_u32 %" #n ", %" #n ", %8\n"
#define I_PKMUL(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define I_MAD24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
#define I_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define I_PKLSHR(n) "v_pk_lshrrev_b16 %" #n ", 15, %" #n "\n"
#define I_CMPONLY(n) "v_cmp_gt_i32 vcc, %8, %" #n "\n"
#define I_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define I_ADDCO(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n"
#define I_SUBREV(n) "v_subrev_u32 %" #n ", %8, %" #n "\n"
#define I_MAXI16(n) "v_max_i16 %" #n ", %" #n ", %8\n"
#define I_ADDU16(n) "v_add_u16 %" #n ", %" #n ", %8\n"
#define I_MAXF(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define I_ADDF(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define I_SUBF(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define I_MAXF16(n) "v_max_f16 %" #n ", %" #n ", %8\n"
#define I_PKMAXF16(n) "v_pk_max_f16 %" #n ", %" #n ", %8\n"
#define I_PKADDF16(n) "v_pk_add_f16 %" #n ", %" #n ", %8\n"
#define I_MINI16(n) "v_min_i16 %" #n ", %" #n ", %8\n"
#define I_SUBU16(n) "v_sub_u16 %" #n ", %" #n ", %8\n"
#define I_ASHR(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n"
#define I_MULU24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define I_FMAC(n) "v_fmac_f32 %" #n ", %8, %9\n"
KERNEL(maxf, REP32(I_MAXF), REP32(I_MAXF))
KERNEL(addf, REP32(I_ADDF), REP32(I_ADDF))
KERNEL(subf, REP32(I_SUBF), REP32(I_SUBF))
KERNEL(maxf16, REP32(I_MAXF16), REP32(I_MAXF16))
KERNEL(pkmaxf16, REP32(I_PKMAXF16), REP32(I_PKMAXF16))
KERNEL(pkaddf16, REP32(I_PKADDF16), REP32(I_PKADDF16))
KERNEL(mini16, REP32(I_MINI16), REP32(I_MINI16))
KERNEL(subu16, REP32(I_SUBU16), REP32(I_SUBU16))
KERNEL(ashr, REP32(I_ASHR), REP32(I_ASHR))
KERNEL(mulu24, REP32(I_MULU24), REP32(I_MULU24))
KERNEL(fma, REP32(I_FMA), REP32(I_FMA))
KERNEL(fmac, REP32(I_FMAC), REP32(I_FMAC))
KERNEL(and_, REP32(I_AND), REP32(I_AND))
KERNEL(or_, REP32(I_OR), REP32(I_OR))
KERNEL(min_, REP32(I_MIN), REP32(I_MIN))
KERNEL(maxu, REP32(I_MAXU), REP32(I_MAXU))
KERNEL(pkmul, REP32(I_PKMUL), REP32(I_PKMUL))
KERNEL(mad24, REP32(I_MAD24), REP32(I_MAD24))
KERNEL(bfe, REP32(I_BFE), REP32(I_BFE))
KERNEL(pklshr, REP32(I_PKLSHR), REP32(I_PKLSHR))
KERNEL(cmponly, REP32(I_CMPONLY), REP32(I_CMPONLY))
KERNEL(lshl, REP32(I_LSHL), REP32(I_LSHL))
KERNEL(addco, REP32(I_ADDCO), REP32(I_ADDCO))
KERNEL(subrev, REP32(I_SUBREV), REP32(I_SUBREV))
KERNEL(maxi16, REP32(I_MAXI16), REP32(I_MAXI16))
KERNEL(addu16, REP32(I_ADDU16), REP32(I_ADDU16))
KERNEL(add, REP32(I_ADD), REP32(D_ADD))
KERNEL(sub, REP32(I_SUB), REP32(D_ADD))
KERNEL(max, REP32(I_MAX), REP32(D_MAX))
KERNEL(alignbit, REP32(I_ALIGN), REP32(D_ALIGN))
KERNEL(mov_dpp, REP32(I_DPP), REP32(D_DPP))
KERNEL(add_dpp, REP32(I_ADDDPP), REP32(I_ADDDPP))
KERNEL(cmp_sdwa_cnd, REP32(I_CMPSDWA), REP32(I_CMPSDWA))
KERNEL(cmp_cnd, REP32(I_CMP), REP32(I_CMP))
KERNEL(cmp_addc, REP32(I_CMPADDC), REP32(I_CMPADDC))
KERNEL(max3, REP32(I_MAX3), REP32(I_MAX3))
KERNEL(add3, REP32(I_ADD3), REP32(I_ADD3))
KERNEL(pk_add_i16, REP32(I_PKADD), REP32(D_PKADD))
KERNEL(pk_sub_i16, REP32(I_PKSUB), REP32(D_PKADD))
KERNEL(pk_max_i16, REP32(I_PKMAX), REP32(D_PKMAX))
KERNEL(pk_min_u16, REP32(I_PKMINU), REP32(I_PKMINU))
KERNEL(pk_mad_i16, REP32(I_PKMAD), REP32(I_PKMAD))
KERNEL(pk_ashr_i16, REP32(I_PKASHR), REP32(I_PKASHR))
KERNEL(pk_add_clamp, REP32(I_ADDI16SAT), REP32(I_ADDI16SAT))
KERNEL(perm, REP32(I_PERM), REP32(I_PERM))
KERNEL(bfi, REP32(I_BFI), REP32(I_BFI))
KERNEL(and_or, REP32(I_ANDOR), REP32(I_ANDOR))
KERNEL(lshl_or, REP32(I_LSHLOR), REP32(I_LSHLOR))
KERNEL(xor_, REP32(I_XOR), REP32(I_XOR))
KERNEL(lshr, REP32(I_LSHR), REP32(I_LSHR))
KERNEL(mov, REP32(I_MOV), REP32(I_MOV))
KERNEL(sad_u8, REP32(I_SADU8), REP32(I_SADU8))
KERNEL(cndmask, REP32(I_CNDMASK), REP32(I_CNDMASK))

struct Entry { const char *name; void (*fn)(unsigned *, unsigned long long *, int, int); int instr_per_rep; bool has_dep; };

int main(int argc, char **argv)
{
    int blocks_per_cu = argc > 1 ? atoi(argv[1]) : 4; // x 4 waves = waves per CU
    if (argc > 2) ITERS = atoi(argv[2]);
    const char *only = argc > 3 ? argv[3] : nullptr;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    const int blocks = cus * blocks_per_cu;
    unsigned *out;
    unsigned long long *clk;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMalloc(&clk, (size_t)blocks * 16));
    std::vector<unsigned long long> h(2 * blocks);
    Entry es[] = {
        {"v_add_u32", k_add, 1, true}, {"v_sub_u32", k_sub, 1, false}, {"v_max_i32", k_max, 1, true},
        {"v_alignbit_b32", k_alignbit, 1, true}, {"v_mov_b32_dpp row_shr", k_mov_dpp, 1, true},
        {"v_add_u32_dpp row_shr", k_add_dpp, 1, false}, {"v_cmp_eq_sdwa+v_cndmask", k_cmp_sdwa_cnd, 2, false},
        {"v_cmp_eq+v_cndmask", k_cmp_cnd, 2, false}, {"v_cmp_gt+v_addc_co", k_cmp_addc, 2, false},
        {"v_max3_i32", k_max3, 1, false}, {"v_add3_u32", k_add3, 1, false}, {"v_pk_add_i16", k_pk_add_i16, 1, true},
        {"v_pk_sub_i16", k_pk_sub_i16, 1, false}, {"v_pk_max_i16", k_pk_max_i16, 1, true},
        {"v_pk_min_u16", k_pk_min_u16, 1, false}, {"v_pk_mad_i16", k_pk_mad_i16, 1, false},
        {"v_pk_ashrrev_i16", k_pk_ashr_i16, 1, false}, {"v_pk_add_i16 clamp", k_pk_add_clamp, 1, false},
        {"v_perm_b32", k_perm, 1, false}, {"v_bfi_b32", k_bfi, 1, false}, {"v_and_or_b32", k_and_or, 1, false},
        {"v_lshl_or_b32", k_lshl_or, 1, false}, {"v_xor_b32", k_xor_, 1, false}, {"v_lshrrev_b32", k_lshr, 1, false},
        {"v_max_f32", k_maxf, 1, false}, {"v_add_f32", k_addf, 1, false}, {"v_sub_f32", k_subf, 1, false},
        {"v_max_f16", k_maxf16, 1, false}, {"v_pk_max_f16", k_pkmaxf16, 1, false}, {"v_pk_add_f16", k_pkaddf16, 1, false},
        {"v_min_i16", k_mini16, 1, false}, {"v_sub_u16", k_subu16, 1, false}, {"v_ashrrev_i32", k_ashr, 1, false},
        {"v_mul_u32_u24", k_mulu24, 1, false}, {"v_fma_f32", k_fma, 1, false}, {"v_fmac_f32", k_fmac, 1, false},
        {"v_mov_b32", k_mov, 1, false}, {"v_and_b32", k_and_, 1, false}, {"v_or_b32", k_or_, 1, false},
        {"v_min_i32", k_min_, 1, false}, {"v_max_u32", k_maxu, 1, false}, {"v_pk_mul_lo_u16", k_pkmul, 1, false},
        {"v_mad_i32_i24", k_mad24, 1, false}, {"v_bfe_u32", k_bfe, 1, false}, {"v_pk_lshrrev_b16", k_pklshr, 1, false},
        {"v_cmp_gt_i32 (vcc)", k_cmponly, 1, false}, {"v_lshlrev_b32", k_lshl, 1, false},
        {"v_add_co_u32", k_addco, 1, false}, {"v_subrev_u32", k_subrev, 1, false}, {"v_max_i16", k_maxi16, 1, false},
        {"v_add_u16", k_addu16, 1, false}, {"v_sad_u8", k_sad_u8, 1, false}, {"v_cndmask_b32", k_cndmask, 1, false},
    };
    // one "round" of blocks: exactly blocks_per_cu resident per CU, enforced by the LDS each block claims
    const int lds = (160 * 1024 / blocks_per_cu) & ~255;
    printf("CUs=%d waves/CU=%d (per SIMD %d), LDS/block=%d\n", cus, blocks_per_cu * 4, blocks_per_cu, lds);
    printf("%-28s %10s %10s %12s %10s\n", "instruction", "indep cyc", "dep cyc", "clock GHz", "ms");
    for (auto &e : es) {
        if (only && !strstr(e.name, only)) continue;
        double res[2] = {0, 0}, ghz = 0, ms_ = 0;
        CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(e.fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        for (int dep = 0; dep < (e.has_dep ? 2 : 1); ++dep) {
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), lds, 0, out, clk, dep, ITERS); // warm
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(e.fn, dim3(blocks), dim3(256), lds, 0, out, clk, dep, ITERS);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms; CHECK(hipEventElapsedTime(&ms, a, b));
            CHECK(hipMemcpy(h.data(), clk, (size_t)blocks * 16, hipMemcpyDeviceToHost));
            double cyc = 0, real = 0;
            for (int i = 0; i < blocks; ++i) { cyc += h[2 * i]; real += h[2 * i + 1]; }
            cyc /= blocks; real /= blocks;
            const double instr_per_wave = (double)ITERS * UNROLL * e.instr_per_rep;
            // waves per SIMD share the SIMD: cycles per wave-instruction on the SIMD = cyc / (instr_per_wave * waves_per_simd)
            ghz = cyc / real * 0.1; ms_ = ms;
            // wall-clock based: SIMD cycles per wave-instruction (kernel time x in-kernel clock / instr per SIMD)
            res[dep] = (double)ms * 1e-3 * ghz * 1e9 / (instr_per_wave * blocks_per_cu);
        }
        printf("%-28s %10.3f %10.3f %12.3f %10.3f\n", e.name, res[0], res[1], ghz, ms_);
    }
    return 0;
}
