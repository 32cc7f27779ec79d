// valu_peak.hip -- developer microbenchmark (GPU box): how many wave64 VALU instructions per second does the MI355X issue,
// per instruction class and per occupancy?  Decides the denominator of bench.py's VALU roofline (VERDICT round 3, item 1:
// the data sheet's 157 TFLOP/s of fp32 vector is either 32 lanes x fma x 2 cycles or 16 lanes x fma x packed x 4 cycles).
//
//   hipcc --offload-arch=gfx950 -O3 tools_dev/valu_peak.hip -o /tmp/valu_peak && /tmp/valu_peak
//
// Every kernel is a loop of 64 instructions on eight INDEPENDENT accumulators (inline assembly: the compiler neither
// removes nor fuses them), run by W wavefronts per SIMD on all 256 CUs (grid = 256 x blocks-per-CU workgroups, 4 x W'
// wavefronts each).  Printed: wave-instructions per second for the whole chip, cycles per instruction and SIMD at the clock the
// run really had (s_memtime ticks of one wavefront / its wall time), and the same with a DEPENDENT chain (one accumulator).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

enum Op { FMA_F32, MUL_F32, ADD_U32, MIN_F32, CMP_CNDMASK, PK_FMA_F32, PK_MUL_F32, FMA_F64, ADD_F64, MUL_F64, RCP_F32, SQRT_F32,
          DPP_MOV, DPP_ADD, READLANE, MAD_U64_U32, MUL_LO_U32, MUL_HI_U32, LSHL_ADD, MED3_F32, CVT_F64_F32, MBCNT, PERM_B32,
          XOR3, MIX_SALU,
          X_V_ADD_F32, X_V_SUB_F32, X_V_MAX_F32, X_V_AND_B32, X_V_OR_B32, X_V_XOR_B32, X_V_LSHLREV_B32, X_V_LSHRREV_B32, X_V_ASHRREV_I32, X_V_SUB_U32, X_V_MIN_U32, X_V_MAX_I32, X_V_MUL_U32_U24, X_V_LDEXP_F32, X_V_PK_ADD_F32, X_V_FMAC_F32, X_V_CVT_PKRTZ_F16_F32, X_V_ADD3_U32, X_V_AND_OR_B32, X_V_OR3_B32, X_V_BFE_U32, X_V_MAD_U32_U24, X_V_MIN3_F32, X_V_MAX3_F32, X_V_ALIGNBIT_B32, X_V_LSHL_OR_B32, X_V_BFI_B32, X_V_ADD_LSHL_U32, X_V_MAD_I32_I24, X_V_MOV_B32, X_V_CVT_F32_U32, X_V_CVT_U32_F32, X_V_CVT_F32_I32, X_V_CVT_I32_F32, X_V_RNDNE_F32, X_V_FLOOR_F32, X_V_FRACT_F32, X_V_TRUNC_F32, X_V_FREXP_MANT_F32, X_V_NOT_B32, X_V_BFREV_B32, X_V_FFBH_U32, X_V_RCP_F64, X_V_SQRT_F64, X_V_RSQ_F32, X_V_EXP_F32, X_V_LOG_F32, X_CMP_ONLY, X_CNDMASK_ONLY, X_CMP_SGPR, X_SALU_ONLY, X_ADDC_PAIR, X_READFIRSTLANE, X_WRITELANE, N_OPS };
static const char *op_name[N_OPS] = {"v_fma_f32", "v_mul_f32", "v_add_u32", "v_min_f32", "v_cmp_lt_f32 + v_cndmask_b32 (2 instr)",
    "v_pk_fma_f32", "v_pk_mul_f32", "v_fma_f64", "v_add_f64", "v_mul_f64", "v_rcp_f32", "v_sqrt_f32", "v_mov_b32 dpp row_shr:1",
    "v_add_f32 dpp row_shr:1", "v_readlane_b32 (-> sgpr)", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_lshl_add_u32", "v_med3_f32",
    "v_cvt_f64_f32", "v_mbcnt_lo_u32_b32", "v_perm_b32", "v_xor3_b32 (v_xad_u32)", "v_fma_f32 + s_add_u32 pairs (VALU counted)",
    "v_add_f32", "v_sub_f32", "v_max_f32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_lshlrev_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_sub_u32", "v_min_u32", "v_max_i32", "v_mul_u32_u24", "v_ldexp_f32", "v_pk_add_f32", "v_fmac_f32", "v_cvt_pkrtz_f16_f32", "v_add3_u32", "v_and_or_b32", "v_or3_b32", "v_bfe_u32", "v_mad_u32_u24", "v_min3_f32", "v_max3_f32", "v_alignbit_b32", "v_lshl_or_b32", "v_bfi_b32", "v_add_lshl_u32", "v_mad_i32_i24", "v_mov_b32", "v_cvt_f32_u32", "v_cvt_u32_f32", "v_cvt_f32_i32", "v_cvt_i32_f32", "v_rndne_f32", "v_floor_f32", "v_fract_f32", "v_trunc_f32", "v_frexp_mant_f32", "v_not_b32", "v_bfrev_b32", "v_ffbh_u32", "v_rcp_f64", "v_sqrt_f64", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_cmp_lt_f32 vcc (alone)", "v_cndmask_b32 (alone, vcc)", "v_cmp_lt_f32 s[20:21] (VOP3)", "s_add_u32 (SALU only; counted as instructions)", "v_add_co_u32 + v_addc_co_u32 (2 instr)", "v_readfirstlane_b32", "v_writelane_b32"};
static int op_instr_per_step(int op) { return op == CMP_CNDMASK || op == X_ADDC_PAIR ? 2 : 1; }

// eight instructions in ONE inline-assembly block (between separate blocks the compiler puts an s_nop): on the eight accumulators
// (CHAIN = 8) or all on accumulator 0 (CHAIN = 1).  c, d: loop-invariant operands; a2 / c2 / d2: 64-bit register pairs.
#define REP8(F) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define DEP8(F) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0)
#define ASM32(F) do { if (CHAIN == 1) asm volatile(DEP8(F) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "v"(d) : "vcc", "s20", "scc"); \
                      else asm volatile(REP8(F) : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(c), "v"(d) : "vcc", "s20", "scc"); } while (0)
#define ASM64(F) do { if (CHAIN == 1) asm volatile(DEP8(F) : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "+v"(a2[7]) : "v"(c2), "v"(d2) : "vcc"); \
                      else asm volatile(REP8(F) : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "+v"(a2[7]) : "v"(c2), "v"(d2) : "vcc"); } while (0)
#define I_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
#define I_MUL(n) "v_mul_f32 %" #n ", %" #n ", %8\n\t"
#define I_ADDU(n) "v_add_u32 %" #n ", %" #n ", %8\n\t"
#define I_MIN(n) "v_min_f32 %" #n ", %" #n ", %8\n\t"
#define I_CMPSEL(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n\tv_cndmask_b32 %" #n ", %" #n ", %9, vcc\n\t"
#define I_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n\t"
#define I_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n\t"
#define I_FMA64(n) "v_fma_f64 %" #n ", %" #n ", %8, %9\n\t"
#define I_ADD64(n) "v_add_f64 %" #n ", %" #n ", %8\n\t"
#define I_MUL64(n) "v_mul_f64 %" #n ", %" #n ", %8\n\t"
#define I_RCP(n) "v_rcp_f32 %" #n ", %" #n "\n\t"
#define I_SQRT(n) "v_sqrt_f32 %" #n ", %" #n "\n\t"
// (a DPP read of a VGPR written by the previous VALU instruction needs two wait states; the assembler does not look into inline
// assembly, so the dependent chain carries its own s_nop -- the independent chains are eight instructions apart)
#define I_DPPMOV(n) "v_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPMOV_D(n) "s_nop 1\n\tv_mov_b32_dpp %" #n ", %" #n " row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPADD(n) "v_add_f32_dpp %" #n ", %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_DPPADD_D(n) "s_nop 1\n\tv_add_f32_dpp %" #n ", %" #n ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n\t"
#define I_READLANE(n) "v_readlane_b32 s20, %" #n ", 3\n\t"
#define I_MAD64(n) "v_mad_u64_u32 %" #n ", vcc, %8, %9, %" #n "\n\t"
#define I_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n\t"
#define I_MULHI(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n\t"
#define I_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n\t"
#define I_MED3(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n\t"
#define I_CVT64(n) "v_cvt_f64_f32 %" #n ", %8\n\t"
#define I_MBCNT(n) "v_mbcnt_lo_u32_b32 %" #n ", %8, %" #n "\n\t"
#define I_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n\t"
#define I_XAD(n) "v_xad_u32 %" #n ", %" #n ", %8, %9\n\t"
#define I_MIXS(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n\ts_add_u32 s20, s20, 1\n\t"
#define I_X_V_ADD_F32(n) "v_add_f32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_SUB_F32(n) "v_sub_f32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_MAX_F32(n) "v_max_f32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_AND_B32(n) "v_and_b32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_OR_B32(n) "v_or_b32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_XOR_B32(n) "v_xor_b32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_LSHLREV_B32(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n\t"
#define I_X_V_LSHRREV_B32(n) "v_lshrrev_b32 %" #n ", 1, %" #n "\n\t"
#define I_X_V_ASHRREV_I32(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n\t"
#define I_X_V_SUB_U32(n) "v_sub_u32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_MIN_U32(n) "v_min_u32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_MAX_I32(n) "v_max_i32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_MUL_U32_U24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n\t"
#define I_X_V_LDEXP_F32(n) "v_ldexp_f32 %" #n ", %" #n ", 1\n\t"
#define I_X_V_PK_ADD_F32(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_FMAC_F32(n) "v_fmac_f32 %" #n ", %8, %9\n\t"
#define I_X_V_CVT_PKRTZ_F16_F32(n) "v_cvt_pkrtz_f16_f32 %" #n ", %" #n ", %8\n\t"
#define I_X_V_ADD3_U32(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_AND_OR_B32(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_OR3_B32(n) "v_or3_b32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_BFE_U32(n) "v_bfe_u32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_MAD_U32_U24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_MIN3_F32(n) "v_min3_f32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_MAX3_F32(n) "v_max3_f32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_ALIGNBIT_B32(n) "v_alignbit_b32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_LSHL_OR_B32(n) "v_lshl_or_b32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_BFI_B32(n) "v_bfi_b32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_ADD_LSHL_U32(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_MAD_I32_I24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n\t"
#define I_X_V_MOV_B32(n) "v_mov_b32 %" #n ", %" #n "\n\t"
#define I_X_V_CVT_F32_U32(n) "v_cvt_f32_u32 %" #n ", %" #n "\n\t"
#define I_X_V_CVT_U32_F32(n) "v_cvt_u32_f32 %" #n ", %" #n "\n\t"
#define I_X_V_CVT_F32_I32(n) "v_cvt_f32_i32 %" #n ", %" #n "\n\t"
#define I_X_V_CVT_I32_F32(n) "v_cvt_i32_f32 %" #n ", %" #n "\n\t"
#define I_X_V_RNDNE_F32(n) "v_rndne_f32 %" #n ", %" #n "\n\t"
#define I_X_V_FLOOR_F32(n) "v_floor_f32 %" #n ", %" #n "\n\t"
#define I_X_V_FRACT_F32(n) "v_fract_f32 %" #n ", %" #n "\n\t"
#define I_X_V_TRUNC_F32(n) "v_trunc_f32 %" #n ", %" #n "\n\t"
#define I_X_V_FREXP_MANT_F32(n) "v_frexp_mant_f32 %" #n ", %" #n "\n\t"
#define I_X_V_NOT_B32(n) "v_not_b32 %" #n ", %" #n "\n\t"
#define I_X_V_BFREV_B32(n) "v_bfrev_b32 %" #n ", %" #n "\n\t"
#define I_X_V_FFBH_U32(n) "v_ffbh_u32 %" #n ", %" #n "\n\t"
#define I_X_V_RCP_F64(n) "v_rcp_f64 %" #n ", %" #n "\n\t"
#define I_X_V_SQRT_F64(n) "v_sqrt_f64 %" #n ", %" #n "\n\t"
#define I_X_V_RSQ_F32(n) "v_rsq_f32 %" #n ", %" #n "\n\t"
#define I_X_V_EXP_F32(n) "v_exp_f32 %" #n ", %" #n "\n\t"
#define I_X_V_LOG_F32(n) "v_log_f32 %" #n ", %" #n "\n\t"
#define I_X_CMP_ONLY(n) "v_cmp_lt_f32 vcc, %" #n ", %8\n\t"
#define I_X_CNDMASK_ONLY(n) "v_cndmask_b32 %" #n ", %" #n ", %9, vcc\n\t"
#define I_X_CMP_SGPR(n) "v_cmp_lt_f32 s[20:21], %" #n ", %8\n\t"
#define I_X_SALU_ONLY(n) "s_add_u32 s20, s20, 1\n\t"
#define I_X_ADDC_PAIR(n) "v_add_co_u32 %" #n ", vcc, %" #n ", %8\n\tv_addc_co_u32 %" #n ", vcc, %" #n ", %9, vcc\n\t"
#define I_X_READFIRSTLANE(n) "v_readfirstlane_b32 s20, %" #n "\n\t"
#define I_X_WRITELANE(n) "v_writelane_b32 %" #n ", s20, 5\n\t"
template <int OP, int CHAIN>
__device__ __forceinline__ void step8(float (&a)[8], double (&a2)[8], float c, float d, double c2, double d2) {
    if (OP == FMA_F32) ASM32(I_FMA);
    if (OP == MUL_F32) ASM32(I_MUL);
    if (OP == ADD_U32) ASM32(I_ADDU);
    if (OP == MIN_F32) ASM32(I_MIN);
    if (OP == CMP_CNDMASK) ASM32(I_CMPSEL);
    if (OP == PK_FMA_F32) ASM64(I_PKFMA);
    if (OP == PK_MUL_F32) ASM64(I_PKMUL);
    if (OP == FMA_F64) ASM64(I_FMA64);
    if (OP == ADD_F64) ASM64(I_ADD64);
    if (OP == MUL_F64) ASM64(I_MUL64);
    if (OP == RCP_F32) ASM32(I_RCP);
    if (OP == SQRT_F32) ASM32(I_SQRT);
    if (OP == DPP_MOV && CHAIN == 1) ASM32(I_DPPMOV_D);
    if (OP == DPP_ADD && CHAIN == 1) ASM32(I_DPPADD_D);
    if (OP == DPP_MOV && CHAIN != 1) ASM32(I_DPPMOV);
    if (OP == DPP_ADD && CHAIN != 1) ASM32(I_DPPADD);
    if (OP == READLANE) ASM32(I_READLANE);
    if (OP == MAD_U64_U32) { const float c = __int_as_float(12345), d = __int_as_float(6789);
        if (CHAIN == 1) asm volatile(DEP8(I_MAD64) : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "+v"(a2[7]) : "v"(c), "v"(d) : "vcc");
        else asm volatile(REP8(I_MAD64) : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "+v"(a2[7]) : "v"(c), "v"(d) : "vcc"); }
    if (OP == MUL_LO_U32) ASM32(I_MULLO);
    if (OP == MUL_HI_U32) ASM32(I_MULHI);
    if (OP == LSHL_ADD) ASM32(I_LSHLADD);
    if (OP == MED3_F32) ASM32(I_MED3);
    if (OP == CVT_F64_F32) { if (CHAIN == 1) asm volatile(DEP8(I_CVT64) : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "+v"(a2[7]) : "v"(c));
        else asm volatile(REP8(I_CVT64) : "+v"(a2[0]), "+v"(a2[1]), "+v"(a2[2]), "+v"(a2[3]), "+v"(a2[4]), "+v"(a2[5]), "+v"(a2[6]), "+v"(a2[7]) : "v"(c)); }
    if (OP == MBCNT) ASM32(I_MBCNT);
    if (OP == PERM_B32) ASM32(I_PERM);
    if (OP == XOR3) ASM32(I_XAD);
    if (OP == MIX_SALU) ASM32(I_MIXS);
    if (OP == X_V_ADD_F32) ASM32(I_X_V_ADD_F32);
    if (OP == X_V_SUB_F32) ASM32(I_X_V_SUB_F32);
    if (OP == X_V_MAX_F32) ASM32(I_X_V_MAX_F32);
    if (OP == X_V_AND_B32) ASM32(I_X_V_AND_B32);
    if (OP == X_V_OR_B32) ASM32(I_X_V_OR_B32);
    if (OP == X_V_XOR_B32) ASM32(I_X_V_XOR_B32);
    if (OP == X_V_LSHLREV_B32) ASM32(I_X_V_LSHLREV_B32);
    if (OP == X_V_LSHRREV_B32) ASM32(I_X_V_LSHRREV_B32);
    if (OP == X_V_ASHRREV_I32) ASM32(I_X_V_ASHRREV_I32);
    if (OP == X_V_SUB_U32) ASM32(I_X_V_SUB_U32);
    if (OP == X_V_MIN_U32) ASM32(I_X_V_MIN_U32);
    if (OP == X_V_MAX_I32) ASM32(I_X_V_MAX_I32);
    if (OP == X_V_MUL_U32_U24) ASM32(I_X_V_MUL_U32_U24);
    if (OP == X_V_LDEXP_F32) ASM32(I_X_V_LDEXP_F32);
    if (OP == X_V_PK_ADD_F32) ASM64(I_X_V_PK_ADD_F32);
    if (OP == X_V_FMAC_F32) ASM32(I_X_V_FMAC_F32);
    if (OP == X_V_CVT_PKRTZ_F16_F32) ASM32(I_X_V_CVT_PKRTZ_F16_F32);
    if (OP == X_V_ADD3_U32) ASM32(I_X_V_ADD3_U32);
    if (OP == X_V_AND_OR_B32) ASM32(I_X_V_AND_OR_B32);
    if (OP == X_V_OR3_B32) ASM32(I_X_V_OR3_B32);
    if (OP == X_V_BFE_U32) ASM32(I_X_V_BFE_U32);
    if (OP == X_V_MAD_U32_U24) ASM32(I_X_V_MAD_U32_U24);
    if (OP == X_V_MIN3_F32) ASM32(I_X_V_MIN3_F32);
    if (OP == X_V_MAX3_F32) ASM32(I_X_V_MAX3_F32);
    if (OP == X_V_ALIGNBIT_B32) ASM32(I_X_V_ALIGNBIT_B32);
    if (OP == X_V_LSHL_OR_B32) ASM32(I_X_V_LSHL_OR_B32);
    if (OP == X_V_BFI_B32) ASM32(I_X_V_BFI_B32);
    if (OP == X_V_ADD_LSHL_U32) ASM32(I_X_V_ADD_LSHL_U32);
    if (OP == X_V_MAD_I32_I24) ASM32(I_X_V_MAD_I32_I24);
    if (OP == X_V_MOV_B32) ASM32(I_X_V_MOV_B32);
    if (OP == X_V_CVT_F32_U32) ASM32(I_X_V_CVT_F32_U32);
    if (OP == X_V_CVT_U32_F32) ASM32(I_X_V_CVT_U32_F32);
    if (OP == X_V_CVT_F32_I32) ASM32(I_X_V_CVT_F32_I32);
    if (OP == X_V_CVT_I32_F32) ASM32(I_X_V_CVT_I32_F32);
    if (OP == X_V_RNDNE_F32) ASM32(I_X_V_RNDNE_F32);
    if (OP == X_V_FLOOR_F32) ASM32(I_X_V_FLOOR_F32);
    if (OP == X_V_FRACT_F32) ASM32(I_X_V_FRACT_F32);
    if (OP == X_V_TRUNC_F32) ASM32(I_X_V_TRUNC_F32);
    if (OP == X_V_FREXP_MANT_F32) ASM32(I_X_V_FREXP_MANT_F32);
    if (OP == X_V_NOT_B32) ASM32(I_X_V_NOT_B32);
    if (OP == X_V_BFREV_B32) ASM32(I_X_V_BFREV_B32);
    if (OP == X_V_FFBH_U32) ASM32(I_X_V_FFBH_U32);
    if (OP == X_V_RCP_F64) ASM64(I_X_V_RCP_F64);
    if (OP == X_V_SQRT_F64) ASM64(I_X_V_SQRT_F64);
    if (OP == X_V_RSQ_F32) ASM32(I_X_V_RSQ_F32);
    if (OP == X_V_EXP_F32) ASM32(I_X_V_EXP_F32);
    if (OP == X_V_LOG_F32) ASM32(I_X_V_LOG_F32);
    if (OP == X_CMP_ONLY) ASM32(I_X_CMP_ONLY);
    if (OP == X_CNDMASK_ONLY) ASM32(I_X_CNDMASK_ONLY);
    if (OP == X_CMP_SGPR) ASM32(I_X_CMP_SGPR);
    if (OP == X_SALU_ONLY) ASM32(I_X_SALU_ONLY);
    if (OP == X_ADDC_PAIR) ASM32(I_X_ADDC_PAIR);
    if (OP == X_READFIRSTLANE) ASM32(I_X_READFIRSTLANE);
    if (OP == X_WRITELANE) ASM32(I_X_WRITELANE);
}

// CHAIN = 8: eight independent accumulators; CHAIN = 1: every instruction depends on the one before
template <int OP, int CHAIN>
__global__ __launch_bounds__(1024) void peak_kernel(float *out, long long *clk, int iters, float c, float d) {
    float a[8];
    double a2[8];
    const double c2 = __hiloint2double(__float_as_int(c), __float_as_int(c)), d2 = __hiloint2double(__float_as_int(d), __float_as_int(d));
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = 1.0f + 0.001f * (float)(threadIdx.x + i); a2[i] = __hiloint2double(__float_as_int(a[i]), __float_as_int(a[i])); }
    const long long t0 = (long long)__builtin_readcyclecounter();   // s_memtime
    const long long w0 = (long long)wall_clock64();                 // 100 MHz
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++) step8<OP, CHAIN>(a, a2, c, d, c2, d2);
    }
    const long long t1 = (long long)__builtin_readcyclecounter();
    const long long w1 = (long long)wall_clock64();
    float r = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; i++) r += a[i] + (float)a2[i];
    if (r == 123.456f) out[0] = r;   // keeps the accumulators alive
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = w1 - w0; }
}

template <int OP, int CHAIN>
static double run(int waves_per_simd, int iters, float *d_out, long long *d_clk, double *ghz) {
    // W wavefronts per SIMD: W <= 4 -> one workgroup of 256 W threads per CU; W = 8 -> two workgroups of 1024
    const int threads = waves_per_simd <= 4 ? 256 * waves_per_simd : 1024, grid = waves_per_simd <= 4 ? 256 : 256 * (waves_per_simd / 4);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((peak_kernel<OP, CHAIN>), dim3(grid), dim3(threads), 0, 0, d_out, d_clk, iters / 8, 1.0000001f, 1e-9f);   // warm-up
    CHECK(hipDeviceSynchronize());
    double best = 0.0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((peak_kernel<OP, CHAIN>), dim3(grid), dim3(threads), 0, 0, d_out, d_clk, iters, 1.0000001f, 1e-9f);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        long long clk[2];
        CHECK(hipMemcpy(clk, d_clk, sizeof(clk), hipMemcpyDeviceToHost));
        const double waves = (double)grid * threads / 64.0, insts = waves * (double)iters * 64.0 * op_instr_per_step(OP);
        const double rate = insts / (ms * 1e-3);
        if (rate > best) { best = rate; *ghz = (double)clk[0] / ((double)clk[1] * 10.0); }   // ticks per 10 ns
    }
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
    return best;
}

template <int OP>
static void bench_op(float *d_out, long long *d_clk, int iters) {
    printf("| %2d | %-44s |", OP, op_name[OP]);
    const int ws[5] = {1, 2, 4, 8, 0};
    double ghz = 0.0, last = 0.0, last_ghz = 0.0;
    for (int i = 0; ws[i]; i++) {
        last = run<OP, 8>(ws[i], iters, d_out, d_clk, &ghz);
        last_ghz = ghz;
        printf(" %7.1f |", last / 1e9);
    }
    // cycles one SIMD spends per instruction at 8 wavefronts per SIMD and the measured clock
    printf(" %5.2f | %5.3f |", 1024.0 * last_ghz * 1e9 / last, last_ghz);
    const double dep = run<OP, 1>(1, iters, d_out, d_clk, &ghz);
    printf(" %6.2f |\n", 1024.0 * ghz * 1e9 / dep);   // one wavefront per SIMD, dependent chain: cycles per instruction = latency
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, %d CUs, clockRate %d kHz\n\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    float *d_out; long long *d_clk;
    CHECK(hipMalloc(&d_out, 64)); CHECK(hipMalloc(&d_clk, 64));
    printf("G wave-instructions/s on the whole chip (256 CUs x 4 SIMDs) with W wavefronts per SIMD, eight independent chains per wavefront;\n"
           "cyc/instr = SIMD cycles per wave64 instruction at W = 8 and the measured clock; dep = cycles per instruction of ONE dependent chain, W = 1\n\n");
    printf("| %2s | %-44s | %7s | %7s | %7s | %7s | %5s | %5s | %6s |\n", "op", "instruction", "W=1", "W=2", "W=4", "W=8", "cyc", "GHz", "dep");
    printf("|---|---|---|---|---|---|---|---|---|\n");
    bench_op<FMA_F32>(d_out, d_clk, iters);
    bench_op<MUL_F32>(d_out, d_clk, iters);
    bench_op<ADD_U32>(d_out, d_clk, iters);
    bench_op<MIN_F32>(d_out, d_clk, iters);
    bench_op<MED3_F32>(d_out, d_clk, iters);
    bench_op<LSHL_ADD>(d_out, d_clk, iters);
    bench_op<XOR3>(d_out, d_clk, iters);
    bench_op<PERM_B32>(d_out, d_clk, iters);
    bench_op<MBCNT>(d_out, d_clk, iters);
    bench_op<CMP_CNDMASK>(d_out, d_clk, iters);
    bench_op<PK_FMA_F32>(d_out, d_clk, iters);
    bench_op<PK_MUL_F32>(d_out, d_clk, iters);
    bench_op<FMA_F64>(d_out, d_clk, iters);
    bench_op<ADD_F64>(d_out, d_clk, iters);
    bench_op<MUL_F64>(d_out, d_clk, iters);
    bench_op<CVT_F64_F32>(d_out, d_clk, iters);
    bench_op<RCP_F32>(d_out, d_clk, iters);
    bench_op<SQRT_F32>(d_out, d_clk, iters);
    bench_op<MUL_LO_U32>(d_out, d_clk, iters);
    bench_op<MUL_HI_U32>(d_out, d_clk, iters);
    bench_op<MAD_U64_U32>(d_out, d_clk, iters);
    bench_op<DPP_MOV>(d_out, d_clk, iters);
    bench_op<DPP_ADD>(d_out, d_clk, iters);
    bench_op<READLANE>(d_out, d_clk, iters);
    bench_op<MIX_SALU>(d_out, d_clk, iters);
    bench_op<X_V_ADD_F32>(d_out, d_clk, iters);
    bench_op<X_V_SUB_F32>(d_out, d_clk, iters);
    bench_op<X_V_MAX_F32>(d_out, d_clk, iters);
    bench_op<X_V_AND_B32>(d_out, d_clk, iters);
    bench_op<X_V_OR_B32>(d_out, d_clk, iters);
    bench_op<X_V_XOR_B32>(d_out, d_clk, iters);
    bench_op<X_V_LSHLREV_B32>(d_out, d_clk, iters);
    bench_op<X_V_LSHRREV_B32>(d_out, d_clk, iters);
    bench_op<X_V_ASHRREV_I32>(d_out, d_clk, iters);
    bench_op<X_V_SUB_U32>(d_out, d_clk, iters);
    bench_op<X_V_MIN_U32>(d_out, d_clk, iters);
    bench_op<X_V_MAX_I32>(d_out, d_clk, iters);
    bench_op<X_V_MUL_U32_U24>(d_out, d_clk, iters);
    bench_op<X_V_LDEXP_F32>(d_out, d_clk, iters);
    bench_op<X_V_PK_ADD_F32>(d_out, d_clk, iters);
    bench_op<X_V_FMAC_F32>(d_out, d_clk, iters);
    bench_op<X_V_CVT_PKRTZ_F16_F32>(d_out, d_clk, iters);
    bench_op<X_V_ADD3_U32>(d_out, d_clk, iters);
    bench_op<X_V_AND_OR_B32>(d_out, d_clk, iters);
    bench_op<X_V_OR3_B32>(d_out, d_clk, iters);
    bench_op<X_V_BFE_U32>(d_out, d_clk, iters);
    bench_op<X_V_MAD_U32_U24>(d_out, d_clk, iters);
    bench_op<X_V_MIN3_F32>(d_out, d_clk, iters);
    bench_op<X_V_MAX3_F32>(d_out, d_clk, iters);
    bench_op<X_V_ALIGNBIT_B32>(d_out, d_clk, iters);
    bench_op<X_V_LSHL_OR_B32>(d_out, d_clk, iters);
    bench_op<X_V_BFI_B32>(d_out, d_clk, iters);
    bench_op<X_V_ADD_LSHL_U32>(d_out, d_clk, iters);
    bench_op<X_V_MAD_I32_I24>(d_out, d_clk, iters);
    bench_op<X_V_MOV_B32>(d_out, d_clk, iters);
    bench_op<X_V_CVT_F32_U32>(d_out, d_clk, iters);
    bench_op<X_V_CVT_U32_F32>(d_out, d_clk, iters);
    bench_op<X_V_CVT_F32_I32>(d_out, d_clk, iters);
    bench_op<X_V_CVT_I32_F32>(d_out, d_clk, iters);
    bench_op<X_V_RNDNE_F32>(d_out, d_clk, iters);
    bench_op<X_V_FLOOR_F32>(d_out, d_clk, iters);
    bench_op<X_V_FRACT_F32>(d_out, d_clk, iters);
    bench_op<X_V_TRUNC_F32>(d_out, d_clk, iters);
    bench_op<X_V_FREXP_MANT_F32>(d_out, d_clk, iters);
    bench_op<X_V_NOT_B32>(d_out, d_clk, iters);
    bench_op<X_V_BFREV_B32>(d_out, d_clk, iters);
    bench_op<X_V_FFBH_U32>(d_out, d_clk, iters);
    bench_op<X_V_RCP_F64>(d_out, d_clk, iters);
    bench_op<X_V_SQRT_F64>(d_out, d_clk, iters);
    bench_op<X_V_RSQ_F32>(d_out, d_clk, iters);
    bench_op<X_V_EXP_F32>(d_out, d_clk, iters);
    bench_op<X_V_LOG_F32>(d_out, d_clk, iters);
    bench_op<X_CMP_ONLY>(d_out, d_clk, iters);
    bench_op<X_CNDMASK_ONLY>(d_out, d_clk, iters);
    bench_op<X_CMP_SGPR>(d_out, d_clk, iters);
    bench_op<X_SALU_ONLY>(d_out, d_clk, iters);
    bench_op<X_ADDC_PAIR>(d_out, d_clk, iters);
    bench_op<X_READFIRSTLANE>(d_out, d_clk, iters);
    bench_op<X_WRITELANE>(d_out, d_clk, iters);
    return 0;
}
