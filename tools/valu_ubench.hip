// Cycles per wave64 instruction on one gfx950 SIMD, per instruction class — the calibration behind bench.py's
// issue-cycle roofline. Every kernel runs ITERS x 256 copies of one instruction over 8 independent
// register chains (throughput) or one chain (dependent latency) on W waves per SIMD of every CU, timed with
// s_memtime around the loop; cycles per instruction = cycles of the slowest wave / (W x instructions of one wave).
// One block per CU (the block asks for more than half of the LDS), 256 x W threads = W waves on each SIMD.
//
//   hipcc --offload-arch=gfx950 -O2 -o facebook360_dep_amd/bin/valu_ubench tools/valu_ubench.hip
//   valu_ubench            -> one JSON object per line
//   rocprofv3 --pmc ... -- valu_ubench --classes   (one dispatch per class at W = 2: which counter sees which class)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <class T>
__device__ __forceinline__ T seed_of(int i);
template <>
__device__ __forceinline__ float seed_of<float>(int i) {
  return 1.0f + 1e-3f * (float)(threadIdx.x + i);
}
template <>
__device__ __forceinline__ double seed_of<double>(int i) {
  return 1.0 + 1e-3 * (double)(threadIdx.x + i);
}
template <>
__device__ __forceinline__ unsigned seed_of<unsigned>(int i) {
  return 0x01010101u * (threadIdx.x + i + 1);
}
template <>
__device__ __forceinline__ v2f seed_of<v2f>(int i) {
  return (v2f){1.0f + 1e-3f * (float)(threadIdx.x + i), 1.0f + 2e-3f * (float)(threadIdx.x + i)};
}
template <>
__device__ __forceinline__ unsigned long long seed_of<unsigned long long>(int i) {
  return 0x0101010101010101ull * (threadIdx.x + i + 1);
}
__device__ __forceinline__ unsigned fold(float v) { return __float_as_uint(v); }
__device__ __forceinline__ unsigned fold(double v) { return (unsigned)__double_as_longlong(v); }
__device__ __forceinline__ unsigned fold(unsigned v) { return v; }
__device__ __forceinline__ unsigned fold(v2f v) { return __float_as_uint(v.x) ^ __float_as_uint(v.y); }
__device__ __forceinline__ unsigned fold(unsigned long long v) { return (unsigned)v ^ (unsigned)(v >> 32); }

#define X8(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define X64_(M) X8(M) X8(M) X8(M) X8(M) X8(M) X8(M) X8(M) X8(M)
#define X64(M) X64_(M) X64_(M) X64_(M) X64_(M)
#define D8(M) M(0) M(0) M(0) M(0) M(0) M(0) M(0) M(0)
#define D64_(M) D8(M) D8(M) D8(M) D8(M) D8(M) D8(M) D8(M) D8(M)
#define D64(M) D64_(M) D64_(M) D64_(M) D64_(M)

// NAME: kernel; T: register type of the chains; OP(i): asm text of one instruction on chain i (%8, %9 = two loop
// invariant operands of type T, %10 = a per-lane LDS byte address); REP: X64 (throughput) or D64 (latency)
#define KERNEL(NAME, T, OP, REP, TAIL)                                                                          \
  __global__ void __launch_bounds__(1024) NAME(unsigned long long* out, int iters) {                             \
    extern __shared__ unsigned long long lds[];                                                                  \
    T r0 = seed_of<T>(0), r1 = seed_of<T>(1), r2 = seed_of<T>(2), r3 = seed_of<T>(3), r4 = seed_of<T>(4),        \
      r5 = seed_of<T>(5), r6 = seed_of<T>(6), r7 = seed_of<T>(7);                                                \
    const T a = seed_of<T>(64), b = seed_of<T>(65);                                                              \
    const unsigned addr = threadIdx.x * 8u;                                                                      \
    lds[threadIdx.x] = threadIdx.x * 8u;                                                                         \
    __syncthreads();                                                                                             \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                  \
    for (int i = 0; i < iters; ++i) {                                                                            \
      asm volatile(REP(OP) TAIL                                                                                  \
                   : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7)              \
                   : "v"(a), "v"(b), "v"(addr)                                                                   \
                   : "vcc", "s40", "s41", "v100", "v101", "memory");                                                               \
    }                                                                                                            \
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                  \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                  \
    const unsigned sink = fold(r0) ^ fold(r1) ^ fold(r2) ^ fold(r3) ^ fold(r4) ^ fold(r5) ^ fold(r6) ^ fold(r7); \
    const int wave = (int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6);                                        \
    if ((threadIdx.x & 63) == 0) {                                                                               \
      out[2 * wave] = t1 - t0;                                                                                   \
      out[2 * wave + 1] = sink;                                                                                  \
    }                                                                                                            \
  }

// ---- fp32
#define O_ADD_F32(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define O_MUL_F32(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define O_FMA_F32(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define O_MAX_F32(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define O_TRUNC_F32(i) "v_trunc_f32 %" #i ", %" #i "\n"
#define O_RNDNE_F32(i) "v_rndne_f32 %" #i ", %" #i "\n"
#define O_CVT_F32_U32(i) "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define O_CVT_F32_U32_SDWA(i) "v_cvt_f32_u32_sdwa %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define O_CVT_I32_F32(i) "v_cvt_i32_f32 %" #i ", %" #i "\n"
#define O_RCP_F32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define O_SQRT_F32(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define O_EXP_F32(i) "v_exp_f32 %" #i ", %" #i "\n"
#define O_DIV_SCALE_F32(i) "v_div_scale_f32 %" #i ", vcc, %" #i ", %8, %9\n"
#define O_DIV_FMAS_F32(i) "v_div_fmas_f32 %" #i ", %" #i ", %8, %9\n"
#define O_DIV_FIXUP_F32(i) "v_div_fixup_f32 %" #i ", %" #i ", %8, %9\n"
#define O_CMP_LT_F32(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
// ---- packed fp32 (register pairs)
#define O_PK_ADD_F32(i) "v_pk_add_f32 %" #i ", %" #i ", %8\n"
#define O_PK_MUL_F32(i) "v_pk_mul_f32 %" #i ", %" #i ", %8\n"
#define O_PK_FMA_F32(i) "v_pk_fma_f32 %" #i ", %" #i ", %8, %9\n"
// ---- fp64
#define O_ADD_F64(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define O_MUL_F64(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define O_FMA_F64(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define O_MAX_F64(i) "v_max_f64 %" #i ", %" #i ", %8\n"
#define O_RCP_F64(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define O_RSQ_F64(i) "v_rsq_f64 %" #i ", %" #i "\n"
#define O_SQRT_F64(i) "v_sqrt_f64 %" #i ", %" #i "\n"
#define O_DIV_SCALE_F64(i) "v_div_scale_f64 %" #i ", vcc, %" #i ", %8, %9\n"
#define O_DIV_FMAS_F64(i) "v_div_fmas_f64 %" #i ", %" #i ", %8, %9\n"
#define O_DIV_FIXUP_F64(i) "v_div_fixup_f64 %" #i ", %" #i ", %8, %9\n"
#define O_CMP_LT_F64(i) "v_cmp_lt_f64 vcc, %" #i ", %8\n"
#define O_CMP_LT_F64_S(i) "v_cmp_lt_f64 s[40:41], %" #i ", %8\n"
#define O_LDEXP_F64(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define O_FLOOR_F64(i) "v_floor_f64 %" #i ", %" #i "\n"
// ---- integer / moves / selects (32-bit chains)
#define O_MOV_B32(i) "v_mov_b32 %" #i ", %8\n"
#define O_AND_B32(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define O_LSHRREV_B32(i) "v_lshrrev_b32 %" #i ", 1, %" #i "\n"
#define O_ADD_U32(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define O_LSHL_ADD_U32(i) "v_lshl_add_u32 %" #i ", %" #i ", 1, %8\n"
#define O_BFE_U32(i) "v_bfe_u32 %" #i ", %" #i ", 3, 16\n"
#define O_PERM_B32(i) "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define O_MUL_LO_U32(i) "v_mul_lo_u32 %" #i ", %" #i ", %8\n"
#define O_MUL_HI_U32(i) "v_mul_hi_u32 %" #i ", %" #i ", %8\n"
#define O_CNDMASK_B32(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define O_ADD_CO_U32(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8\n"
#define O_READFIRSTLANE(i) "v_readfirstlane_b32 s40, %" #i "\n"

// ---- selects, exec-mask moves, fused-multiply VOP2 forms, cross-lane
#define O_CNDMASK_E64(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[40:41]\n"
#define O_CMP_CNDMASK(i) "v_cmp_lt_f32 vcc, %" #i ", %8\nv_cndmask_b32 %" #i ", %" #i ", %9, vcc\n"
#define O_CMP_CNDMASK_E64(i) "v_cmp_lt_f32_e64 s[40:41], %" #i ", %8\nv_cndmask_b32_e64 %" #i ", %" #i ", %9, s[40:41]\n"
#define O_SAVEEXEC_MOV(i) "v_cmp_lt_f32 vcc, %" #i ", %8\ns_and_saveexec_b64 s[40:41], vcc\nv_mov_b32 %" #i ", %9\ns_or_b64 exec, exec, s[40:41]\n"
#define O_BFI_B32(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define O_FMAC_F32(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define O_FMAC_F64(i) "v_fmac_f64 %" #i ", %8, %9\n"
#define O_SUB_F32(i) "v_sub_f32 %" #i ", %" #i ", %8\n"
#define O_MIN_F32(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define O_OR_B32(i) "v_or_b32 %" #i ", %" #i ", %8\n"
#define O_XOR_B32(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define O_LSHLREV_B32(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define O_ADD3_U32(i) "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define O_AND_OR_B32(i) "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define O_MAX_I32(i) "v_max_i32 %" #i ", %" #i ", %8\n"
#define O_ADDC_CO_U32(i) "v_addc_co_u32 %" #i ", vcc, %" #i ", %8, vcc\n"
#define O_MOV_DPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define O_ADD_F32_DPP(i) "v_add_f32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define O_PK_MOV_B32(i) "v_pk_mov_b32 %" #i ", %" #i ", %8 op_sel:[1,0]\n"
#define O_MOV_B64(i) "v_mov_b64 %" #i ", %8\n"
#define O_LSHL_ADD_U64(i) "v_lshl_add_u64 %" #i ", %" #i ", 3, %8\n"
#define O_MUL_F64_S(i) "v_mul_f64 %" #i ", %" #i ", s[40:41]\n"
#define O_DS_READ_B128(i) "ds_read_b64 %" #i ", %10\n"
#define O_DS_BPERMUTE(i) "ds_bpermute_b32 %" #i ", %10, %" #i "\n"
// mixed streams: the shape of the SSD block (per 8: 3 pk_mul, 2 pk_add / add, 2 cvt, 1 trunc)
#define O_MIX_FMA64_ADD32(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\nv_mov_b32 v100, v101\n"
#define O_MIX_FMA64_SMOV(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\ns_mov_b32 s40, 5\n"
#define O_MIX_FMA64_2SMOV(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\ns_mov_b32 s40, 5\ns_mov_b32 s41, 6\n"
#define O_MIX_TRUNC_ADD32(i) "v_trunc_f32 %" #i ", %" #i "\nv_add_f32 %" #i ", %" #i ", %8\n"
#define O_MIX_ADD32_MUL32(i) "v_mul_f32 %" #i ", %" #i ", %8\nv_add_f32 %" #i ", %" #i ", %9\n"
// ---- 64-bit integer chains
#define O_LSHLREV_B64(i) "v_lshlrev_b64 %" #i ", 1, %" #i "\n"
#define O_MAD_U64_U32(i) "v_mad_u64_u32 %" #i ", vcc, s40, 5, %" #i "\n"
// ---- LDS (8-byte slots, lane-linear: conflict free). Latency: the loaded value is the next address.
#define O_DS_READ_B64(i) "ds_read_b64 %" #i ", %10\n"
#define O_DS_WRITE_B64(i) "ds_write_b64 %10, %" #i "\n"
#define O_DS_READ_B32_DEP(i) "ds_read_b32 %" #i ", %" #i "\ns_waitcnt lgkmcnt(0)\n"
#define WAIT_LGKM "s_waitcnt lgkmcnt(0)\n"

KERNEL(k_add_f32, float, O_ADD_F32, X64, "")
KERNEL(k_mul_f32, float, O_MUL_F32, X64, "")
KERNEL(k_fma_f32, float, O_FMA_F32, X64, "")
KERNEL(k_max_f32, float, O_MAX_F32, X64, "")
KERNEL(k_trunc_f32, float, O_TRUNC_F32, X64, "")
KERNEL(k_rndne_f32, float, O_RNDNE_F32, X64, "")
KERNEL(k_cvt_f32_u32, float, O_CVT_F32_U32, X64, "")
KERNEL(k_cvt_f32_u32_sdwa, float, O_CVT_F32_U32_SDWA, X64, "")
KERNEL(k_cvt_i32_f32, float, O_CVT_I32_F32, X64, "")
KERNEL(k_rcp_f32, float, O_RCP_F32, X64, "")
KERNEL(k_sqrt_f32, float, O_SQRT_F32, X64, "")
KERNEL(k_exp_f32, float, O_EXP_F32, X64, "")
KERNEL(k_div_scale_f32, float, O_DIV_SCALE_F32, X64, "")
KERNEL(k_div_fmas_f32, float, O_DIV_FMAS_F32, X64, "")
KERNEL(k_div_fixup_f32, float, O_DIV_FIXUP_F32, X64, "")
KERNEL(k_cmp_lt_f32, float, O_CMP_LT_F32, X64, "")
KERNEL(k_pk_add_f32, v2f, O_PK_ADD_F32, X64, "")
KERNEL(k_pk_mul_f32, v2f, O_PK_MUL_F32, X64, "")
KERNEL(k_pk_fma_f32, v2f, O_PK_FMA_F32, X64, "")
KERNEL(k_add_f64, double, O_ADD_F64, X64, "")
KERNEL(k_mul_f64, double, O_MUL_F64, X64, "")
KERNEL(k_fma_f64, double, O_FMA_F64, X64, "")
KERNEL(k_max_f64, double, O_MAX_F64, X64, "")
KERNEL(k_rcp_f64, double, O_RCP_F64, X64, "")
KERNEL(k_rsq_f64, double, O_RSQ_F64, X64, "")
KERNEL(k_sqrt_f64, double, O_SQRT_F64, X64, "")
KERNEL(k_div_scale_f64, double, O_DIV_SCALE_F64, X64, "")
KERNEL(k_div_fmas_f64, double, O_DIV_FMAS_F64, X64, "")
KERNEL(k_div_fixup_f64, double, O_DIV_FIXUP_F64, X64, "")
KERNEL(k_cmp_lt_f64, double, O_CMP_LT_F64, X64, "")
KERNEL(k_cmp_lt_f64_sgpr, double, O_CMP_LT_F64_S, X64, "")
KERNEL(k_ldexp_f64, double, O_LDEXP_F64, X64, "")
KERNEL(k_floor_f64, double, O_FLOOR_F64, X64, "")
KERNEL(k_mov_b32, unsigned, O_MOV_B32, X64, "")
KERNEL(k_and_b32, unsigned, O_AND_B32, X64, "")
KERNEL(k_lshrrev_b32, unsigned, O_LSHRREV_B32, X64, "")
KERNEL(k_add_u32, unsigned, O_ADD_U32, X64, "")
KERNEL(k_lshl_add_u32, unsigned, O_LSHL_ADD_U32, X64, "")
KERNEL(k_bfe_u32, unsigned, O_BFE_U32, X64, "")
KERNEL(k_perm_b32, unsigned, O_PERM_B32, X64, "")
KERNEL(k_mul_lo_u32, unsigned, O_MUL_LO_U32, X64, "")
KERNEL(k_mul_hi_u32, unsigned, O_MUL_HI_U32, X64, "")
KERNEL(k_cndmask_b32, unsigned, O_CNDMASK_B32, X64, "")
KERNEL(k_add_co_u32, unsigned, O_ADD_CO_U32, X64, "")
KERNEL(k_readfirstlane, unsigned, O_READFIRSTLANE, X64, "")
KERNEL(k_lshlrev_b64, unsigned long long, O_LSHLREV_B64, X64, "")
KERNEL(k_mad_u64_u32, unsigned long long, O_MAD_U64_U32, X64, "")
KERNEL(k_ds_read_b64, unsigned long long, O_DS_READ_B64, X64, WAIT_LGKM)
KERNEL(k_ds_write_b64, unsigned long long, O_DS_WRITE_B64, X64, WAIT_LGKM)

KERNEL(k_cndmask_e64, unsigned, O_CNDMASK_E64, X64, "")
KERNEL(k_cmp_cndmask, float, O_CMP_CNDMASK, X64, "")
KERNEL(k_cmp_cndmask_e64, float, O_CMP_CNDMASK_E64, X64, "")
KERNEL(k_saveexec_mov, float, O_SAVEEXEC_MOV, X64, "")
KERNEL(k_bfi_b32, unsigned, O_BFI_B32, X64, "")
KERNEL(k_fmac_f32, float, O_FMAC_F32, X64, "")
KERNEL(k_fmac_f64, double, O_FMAC_F64, X64, "")
KERNEL(k_sub_f32, float, O_SUB_F32, X64, "")
KERNEL(k_min_f32, float, O_MIN_F32, X64, "")
KERNEL(k_or_b32, unsigned, O_OR_B32, X64, "")
KERNEL(k_xor_b32, unsigned, O_XOR_B32, X64, "")
KERNEL(k_lshlrev_b32, unsigned, O_LSHLREV_B32, X64, "")
KERNEL(k_add3_u32, unsigned, O_ADD3_U32, X64, "")
KERNEL(k_and_or_b32, unsigned, O_AND_OR_B32, X64, "")
KERNEL(k_max_i32, unsigned, O_MAX_I32, X64, "")
KERNEL(k_addc_co_u32, unsigned, O_ADDC_CO_U32, X64, "")
KERNEL(k_mov_dpp, unsigned, O_MOV_DPP, X64, "")
KERNEL(k_add_f32_dpp, float, O_ADD_F32_DPP, X64, "")
KERNEL(k_pk_mov_b32, v2f, O_PK_MOV_B32, X64, "")
KERNEL(k_mov_b64, double, O_MOV_B64, X64, "")
KERNEL(k_lshl_add_u64, unsigned long long, O_LSHL_ADD_U64, X64, "")
KERNEL(k_mul_f64_sgpr, double, O_MUL_F64_S, X64, "")
KERNEL(k_ds_bpermute, unsigned, O_DS_BPERMUTE, X64, WAIT_LGKM)
KERNEL(k_mix_fma64_add32, double, O_MIX_FMA64_ADD32, X64, "")
KERNEL(k_mix_fma64_smov, double, O_MIX_FMA64_SMOV, X64, "")
KERNEL(k_mix_fma64_2smov, double, O_MIX_FMA64_2SMOV, X64, "")
KERNEL(k_mix_trunc_add32, float, O_MIX_TRUNC_ADD32, X64, "")
KERNEL(k_mix_add32_mul32, float, O_MIX_ADD32_MUL32, X64, "")
// dependent chains (latency)
KERNEL(k_dep_fma_f32, float, O_FMA_F32, D64, "")
KERNEL(k_dep_mul_f32, float, O_MUL_F32, D64, "")
KERNEL(k_dep_pk_mul_f32, v2f, O_PK_MUL_F32, D64, "")
KERNEL(k_dep_fma_f64, double, O_FMA_F64, D64, "")
KERNEL(k_dep_add_f64, double, O_ADD_F64, D64, "")
KERNEL(k_dep_rcp_f64, double, O_RCP_F64, D64, "")
KERNEL(k_dep_ds_read_b32, unsigned, O_DS_READ_B32_DEP, D64, "")

// pointer chases through memory: scalar cache (s_load_dword), vector L1 / L2 (global_load_dword); the buffer
// holds its own byte offsets (next = (i + stride) mod n)
__global__ void __launch_bounds__(64) k_chase_scalar(const unsigned* buf, unsigned long long* out, int steps) {
  unsigned off = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < steps; ++i) {
    asm volatile("s_load_dword %0, %1, %0\ns_waitcnt lgkmcnt(0)" : "+s"(off) : "s"(buf) : "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = off;
  }
}
__global__ void __launch_bounds__(64) k_chase_vector(const unsigned* buf, unsigned long long* out, int steps) {
  unsigned off = 0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < steps; ++i) {
    asm volatile("global_load_dword %0, %0, %1\ns_waitcnt vmcnt(0)" : "+v"(off) : "s"(buf) : "memory");
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = off;
  }
}

struct Entry {
  const char* name;
  const char* cls;   // the class bench.py's roofline charges this instruction to
  void (*fn)(unsigned long long*, int);
  bool latency;
};
#define E(k, c) {#k, c, k, false}
#define L(k, c) {#k, c, k, true}
static const Entry kEntries[] = {
    E(k_add_f32, "f32"), E(k_mul_f32, "f32"), E(k_fma_f32, "f32"), E(k_max_f32, "f32"), E(k_trunc_f32, "f32"),
    E(k_rndne_f32, "f32"), E(k_cvt_f32_u32, "cvt"), E(k_cvt_f32_u32_sdwa, "cvt"), E(k_cvt_i32_f32, "cvt"),
    E(k_rcp_f32, "trans_f32"), E(k_sqrt_f32, "trans_f32"), E(k_exp_f32, "trans_f32"), E(k_div_scale_f32, "f32"),
    E(k_div_fmas_f32, "f32"), E(k_div_fixup_f32, "f32"), E(k_cmp_lt_f32, "f32"), E(k_pk_add_f32, "pk_f32"),
    E(k_pk_mul_f32, "pk_f32"), E(k_pk_fma_f32, "pk_f32"), E(k_add_f64, "f64"), E(k_mul_f64, "f64"),
    E(k_fma_f64, "f64"), E(k_max_f64, "f64"), E(k_rcp_f64, "trans_f64"), E(k_rsq_f64, "trans_f64"),
    E(k_sqrt_f64, "trans_f64"), E(k_div_scale_f64, "f64"), E(k_div_fmas_f64, "f64"), E(k_div_fixup_f64, "f64"),
    E(k_cmp_lt_f64, "f64"), E(k_cmp_lt_f64_sgpr, "f64"), E(k_ldexp_f64, "f64"), E(k_floor_f64, "f64"),
    E(k_mov_b32, "int32"), E(k_and_b32, "int32"),
    E(k_lshrrev_b32, "int32"), E(k_add_u32, "int32"), E(k_lshl_add_u32, "int32"), E(k_bfe_u32, "int32"),
    E(k_perm_b32, "int32"), E(k_mul_lo_u32, "int32"), E(k_mul_hi_u32, "int32"), E(k_cndmask_b32, "int32"),
    E(k_add_co_u32, "int32"), E(k_readfirstlane, "int32"), E(k_lshlrev_b64, "int64"), E(k_mad_u64_u32, "int64"),
    E(k_ds_read_b64, "lds"), E(k_ds_write_b64, "lds"),
    E(k_cndmask_e64, "int32"), E(k_cmp_cndmask, "pair"), E(k_cmp_cndmask_e64, "pair"), E(k_saveexec_mov, "pair"),
    E(k_bfi_b32, "int32"), E(k_fmac_f32, "f32"), E(k_fmac_f64, "f64"), E(k_sub_f32, "f32"), E(k_min_f32, "f32"),
    E(k_or_b32, "int32"), E(k_xor_b32, "int32"), E(k_lshlrev_b32, "int32"), E(k_add3_u32, "int32"),
    E(k_and_or_b32, "int32"), E(k_max_i32, "int32"), E(k_addc_co_u32, "int32"), E(k_mov_dpp, "int32"),
    E(k_add_f32_dpp, "f32"), E(k_pk_mov_b32, "int32"), E(k_mov_b64, "int32"), E(k_lshl_add_u64, "int64"),
    E(k_mul_f64_sgpr, "f64"), E(k_ds_bpermute, "lds"), E(k_mix_fma64_add32, "pair"), E(k_mix_fma64_smov, "pair"),
    E(k_mix_fma64_2smov, "pair"), E(k_mix_trunc_add32, "pair"), E(k_mix_add32_mul32, "pair"),
    L(k_dep_fma_f32, "f32"), L(k_dep_mul_f32, "f32"), L(k_dep_pk_mul_f32, "pk_f32"), L(k_dep_fma_f64, "f64"),
    L(k_dep_add_f64, "f64"), L(k_dep_rcp_f64, "trans_f64"), L(k_dep_ds_read_b32, "lds"),
};

int main(int argc, char** argv) {
  bool classes = false;
  const char* only = nullptr;
  int iters = 128;
  for (int i = 1; i < argc; ++i) {
    if (!strcmp(argv[i], "--classes")) {
      classes = true;
    } else if (!strncmp(argv[i], "--only=", 7)) {
      only = argv[i] + 7;
    } else if (!strncmp(argv[i], "--iters=", 8)) {
      iters = atoi(argv[i] + 8);
    }
  }
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  unsigned long long* out;
  CK(hipMalloc(&out, sizeof(unsigned long long) * 2 * cus * 16));
  std::vector<unsigned long long> h(2 * cus * 16);
  const size_t lds = 96 * 1024;  // more than half of the CU's 160 KB: one block per CU
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"iters\": %d, \"insts_per_iter\": 256}\n", prop.name, cus,
         prop.clockRate, iters);
  for (const Entry& e : kEntries) {
    if (only && !strstr(e.name, only)) {
      continue;
    }
    CK(hipFuncSetAttribute((const void*)e.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int wlist[] = {1, 2, 3, 4};
    for (int w : wlist) {
      if (classes && w != 2) {
        continue;
      }
      if (e.latency && w != 1) {
        continue;
      }
      const int threads = 256 * w;
      hipLaunchKernelGGL(e.fn, dim3(cus), dim3(threads), lds, 0, out, 8);  // warm-up (instruction cache)
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(e.fn, dim3(cus), dim3(threads), lds, 0, out, iters);
      CK(hipEventRecord(e1));
      CK(hipDeviceSynchronize());
      float ms = 0;
      CK(hipEventElapsedTime(&ms, e0, e1));
      const int waves = cus * threads / 64;
      CK(hipMemcpy(h.data(), out, sizeof(unsigned long long) * 2 * waves, hipMemcpyDeviceToHost));
      unsigned long long mx = 0, sum = 0;
      for (int i = 0; i < waves; ++i) {
        mx = h[2 * i] > mx ? h[2 * i] : mx;
        sum += h[2 * i];
      }
      const double n = (double)iters * 256;
      // s_memtime ticks per instruction issued on the SIMD (w waves share it)
      printf("{\"op\": \"%s\", \"class\": \"%s\", \"mode\": \"%s\", \"waves_per_simd\": %d, \"ticks_per_inst_max\": %.3f, "
             "\"ticks_per_inst_mean\": %.3f, \"kernel_ms\": %.4f, \"ns_per_inst\": %.4f}\n",
             e.name + 2, e.cls, e.latency ? "latency" : "throughput", w, (double)mx / (n * w),
             (double)sum / waves / (n * w), ms, (double)ms * 1e6 / (n * w));
    }
  }
  if (!classes && !only) {
    // s_memtime tick length: a long fma_f32 kernel against the event clock
    // (reported so that ticks can be converted into shader cycles if they are not the same thing)
    for (int kb : {4, 256, 8192, 262144}) {  // footprints: scalar cache / L1, L2, beyond L2 (MALL / HBM)
      const size_t n = (size_t)kb * 1024 / 4;
      std::vector<unsigned> idx(n);
      const size_t stride = 64;  // one 256-byte step: a new cache line every time
      for (size_t i = 0; i < n; ++i) {
        idx[i] = (unsigned)(((i + stride) % n) * 4);
      }
      unsigned* buf;
      CK(hipMalloc(&buf, n * 4));
      CK(hipMemcpy(buf, idx.data(), n * 4, hipMemcpyHostToDevice));
      const int steps = 20000;
      for (int vec = 0; vec < 2; ++vec) {
        for (int rep = 0; rep < 2; ++rep) {  // second pass: warm
          if (vec) {
            hipLaunchKernelGGL(k_chase_vector, dim3(1), dim3(64), 0, 0, buf, out, steps);
          } else {
            hipLaunchKernelGGL(k_chase_scalar, dim3(1), dim3(64), 0, 0, buf, out, steps);
          }
          CK(hipDeviceSynchronize());
        }
        CK(hipMemcpy(h.data(), out, 16, hipMemcpyDeviceToHost));
        printf("{\"op\": \"chase_%s\", \"footprint_kb\": %d, \"mode\": \"latency\", \"ticks_per_load\": %.1f}\n",
               vec ? "global_load_dword" : "s_load_dword", kb, (double)h[0] / steps);
      }
      CK(hipFree(buf));
    }
  }
  return 0;
}
