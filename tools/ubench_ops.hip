// Per-op VALU/LDS issue cost on gfx950 (development aid; not part of the library).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP8(x) x x x x x x x x

// OPS(a,b,c,d): 8 instructions over 4 rotating registers
#define KERNEL(NAME, TYPE, CONSTR, BODY)                                                           \
    __global__ __launch_bounds__(512) void NAME(float* out, int iters, float seed) {               \
        TYPE r0 = (TYPE)(seed + threadIdx.x), r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3;               \
        __shared__ float lds[8192];                                                                \
        for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;                           \
        __syncthreads();                                                                           \
        uint32_t la = (threadIdx.x & 63) * 16u;                                                    \
        for (int it = 0; it < iters; ++it) {                                                       \
            REP8(asm volatile(BODY : "+" CONSTR(r0), "+" CONSTR(r1), "+" CONSTR(r2), "+" CONSTR(r3) : "v"(la) : "vcc", "scc", "s20", "s21", "s22", "s23", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118");)     \
        }                                                                                          \
        float r = (float)r0 + (float)r1 + (float)r2 + (float)r3;                                   \
        if (r == 123.456f) out[0] = r;                                                             \
    }

#define OP4(x) x(0,1,2,3) x(1,2,3,0) x(2,3,0,1) x(3,0,1,2) x(0,1,2,3) x(1,2,3,0) x(2,3,0,1) x(3,0,1,2)

KERNEL(k_add_f32, float, "v", "v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0\n v_add_f32 %0, %0, %1\n v_add_f32 %1, %1, %2\n v_add_f32 %2, %2, %3\n v_add_f32 %3, %3, %0\n")
KERNEL(k_mul_f32, float, "v", "v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0\n v_mul_f32 %0, %0, %1\n v_mul_f32 %1, %1, %2\n v_mul_f32 %2, %2, %3\n v_mul_f32 %3, %3, %0\n")
KERNEL(k_fmac_f32, float, "v", "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1\n")
KERNEL(k_fma_sgpr, float, "v", "v_fma_f32 %0, s20, %1, %0\n v_fma_f32 %1, s21, %2, %1\n v_fma_f32 %2, s22, %3, %2\n v_fma_f32 %3, s23, %0, %3\n v_fma_f32 %0, s20, %1, %0\n v_fma_f32 %1, s21, %2, %1\n v_fma_f32 %2, s22, %3, %2\n v_fma_f32 %3, s23, %0, %3\n")
KERNEL(k_min_f32, float, "v", "v_min_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_max_f32 %3, %3, %0\n v_min_f32 %0, %0, %1\n v_max_f32 %1, %1, %2\n v_min_f32 %2, %2, %3\n v_max_f32 %3, %3, %0\n")
KERNEL(k_min3_f32, float, "v", "v_min3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_min3_f32 %3, %3, %0, %1\n v_min3_f32 %0, %0, %1, %2\n v_med3_f32 %1, %1, %2, %3\n v_max3_f32 %2, %2, %3, %0\n v_min3_f32 %3, %3, %0, %1\n")
KERNEL(k_cmp_f32, float, "v", "v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_lt_f32 s[20:21], %2, %3\n v_cmp_lt_f32 s[22:23], %3, %0\n v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_lt_f32 s[22:23], %1, %2\n v_cmp_lt_f32 s[20:21], %2, %3\n v_cmp_lt_f32 s[22:23], %3, %0\n")
KERNEL(k_cmp_vcc, float, "v", "v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 vcc, %1, %2\n v_cmp_lt_f32 vcc, %2, %3\n v_cmp_lt_f32 vcc, %3, %0\n")
KERNEL(k_cndmask, float, "v", "v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
KERNEL(k_and_b32, uint32_t, "v", "v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0\n v_and_b32 %0, %0, %1\n v_and_b32 %1, %1, %2\n v_and_b32 %2, %2, %3\n v_and_b32 %3, %3, %0\n")
KERNEL(k_bfe_u32, uint32_t, "v", "v_bfe_u32 %0, %1, 8, 8\n v_bfe_u32 %1, %2, 8, 8\n v_bfe_u32 %2, %3, 8, 8\n v_bfe_u32 %3, %0, 8, 8\n v_bfe_u32 %0, %1, 8, 8\n v_bfe_u32 %1, %2, 8, 8\n v_bfe_u32 %2, %3, 8, 8\n v_bfe_u32 %3, %0, 8, 8\n")
KERNEL(k_add_u32, uint32_t, "v", "v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n v_add_u32 %0, %0, %1\n v_add_u32 %1, %1, %2\n v_add_u32 %2, %2, %3\n v_add_u32 %3, %3, %0\n")
KERNEL(k_lshlrev, uint32_t, "v", "v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %1, 3, %2\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %3, 3, %0\n v_lshlrev_b32 %0, 3, %1\n v_lshlrev_b32 %1, 3, %2\n v_lshlrev_b32 %2, 3, %3\n v_lshlrev_b32 %3, 3, %0\n")
KERNEL(k_mov_b32, uint32_t, "v", "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %0\n")
KERNEL(k_cvt_ubyte, float, "v", "v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte1 %1, %2\n v_cvt_f32_ubyte2 %2, %3\n v_cvt_f32_ubyte3 %3, %0\n v_cvt_f32_ubyte0 %0, %1\n v_cvt_f32_ubyte1 %1, %2\n v_cvt_f32_ubyte2 %2, %3\n v_cvt_f32_ubyte3 %3, %0\n")
KERNEL(k_cvt_u32_f32, float, "v", "v_cvt_u32_f32 %0, %1\n v_cvt_f32_u32 %1, %2\n v_cvt_u32_f32 %2, %3\n v_cvt_f32_u32 %3, %0\n v_cvt_u32_f32 %0, %1\n v_cvt_f32_u32 %1, %2\n v_cvt_u32_f32 %2, %3\n v_cvt_f32_u32 %3, %0\n")
KERNEL(k_floor_f32, float, "v", "v_floor_f32 %0, %1\n v_floor_f32 %1, %2\n v_floor_f32 %2, %3\n v_floor_f32 %3, %0\n v_floor_f32 %0, %1\n v_floor_f32 %1, %2\n v_floor_f32 %2, %3\n v_floor_f32 %3, %0\n")
KERNEL(k_add_f64, double, "v", "v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0\n v_add_f64 %0, %0, %1\n v_add_f64 %1, %1, %2\n v_add_f64 %2, %2, %3\n v_add_f64 %3, %3, %0\n")
KERNEL(k_sdwa_shl, uint32_t, "v", "v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %1, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_lshlrev_b32_sdwa %2, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_lshlrev_b32_sdwa %3, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n"
       "v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1\n v_lshlrev_b32_sdwa %1, %2, %3 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n v_lshlrev_b32_sdwa %2, %3, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_3\n v_lshlrev_b32_sdwa %3, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n")
KERNEL(k_dpp_add, uint32_t, "v", "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %0, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n"
       "v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf\n v_add_u32_dpp %3, %0, %3 row_shr:8 row_mask:0xf bank_mask:0xf\n")
// mixes: 4 fp32 fma + 4 perm (do they overlap across waves?)
KERNEL(k_mix_fma_perm, uint32_t, "v", "v_fma_f32 %0, %0, %0, %0\n v_perm_b32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %2, %2\n v_perm_b32 %3, %3, %0, %1\n v_fma_f32 %0, %0, %0, %0\n v_perm_b32 %1, %1, %2, %3\n v_fma_f32 %2, %2, %2, %2\n v_perm_b32 %3, %3, %0, %1\n")
KERNEL(k_mix_fma_f64, double, "v", "v_fma_f32 v100, v100, v100, v100\n v_fma_f64 %1, %1, %2, %1\n v_fma_f32 v101, v101, v101, v101\n v_fma_f64 %3, %3, %1, %3\n v_fma_f32 v102, v102, v102, v102\n v_fma_f64 %0, %0, %2, %0\n v_fma_f32 v103, v103, v103, v103\n v_fma_f64 %2, %2, %1, %2\n")
// LDS widths (conflict-free: lane*16 B)
KERNEL(k_ds_b96, float, "v", "ds_read_b96 v[100:102], %4\n ds_read_b96 v[104:106], %4 offset:1024\n ds_read_b96 v[100:102], %4 offset:2048\n ds_read_b96 v[104:106], %4 offset:3072\n ds_read_b96 v[100:102], %4 offset:4096\n ds_read_b96 v[104:106], %4 offset:5120\n ds_read_b96 v[100:102], %4 offset:6144\n ds_read_b96 v[104:106], %4 offset:7168\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_ds_b128, float, "v", "ds_read_b128 v[100:103], %4\n ds_read_b128 v[104:107], %4 offset:1024\n ds_read_b128 v[100:103], %4 offset:2048\n ds_read_b128 v[104:107], %4 offset:3072\n ds_read_b128 v[100:103], %4 offset:4096\n ds_read_b128 v[104:107], %4 offset:5120\n ds_read_b128 v[100:103], %4 offset:6144\n ds_read_b128 v[104:107], %4 offset:7168\n s_waitcnt lgkmcnt(0)\n")
KERNEL(k_ds_b64, float, "v", "ds_read_b64 v[100:101], %4\n ds_read_b64 v[104:105], %4 offset:1024\n ds_read_b64 v[100:101], %4 offset:2048\n ds_read_b64 v[104:105], %4 offset:3072\n ds_read_b64 v[100:101], %4 offset:4096\n ds_read_b64 v[104:105], %4 offset:5120\n ds_read_b64 v[100:101], %4 offset:6144\n ds_read_b64 v[104:105], %4 offset:7168\n s_waitcnt lgkmcnt(0)\n")
// 6 VALU + 2 LDS: does LDS issue hide behind VALU?
KERNEL(k_mix_valu_lds, uint32_t, "v", "ds_read_b64 v[100:101], %4\n v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n ds_read_b64 v[104:105], %4 offset:1024\n v_perm_b32 %3, %3, %0, %1\n v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n s_waitcnt lgkmcnt(0)\n")

KERNEL(k_fmac_sgpr, float, "v", "v_fmac_f32 %0, s20, %1\n v_fmac_f32 %1, s20, %2\n v_fmac_f32 %2, s20, %3\n v_fmac_f32 %3, s20, %0\n v_fmac_f32 %0, s20, %1\n v_fmac_f32 %1, s20, %2\n v_fmac_f32 %2, s20, %3\n v_fmac_f32 %3, s20, %0\n ")
KERNEL(k_mul_sgpr, float, "v", "v_mul_f32 %0, s20, %1\n v_mul_f32 %1, s20, %2\n v_mul_f32 %2, s20, %3\n v_mul_f32 %3, s20, %0\n v_mul_f32 %0, s20, %1\n v_mul_f32 %1, s20, %2\n v_mul_f32 %2, s20, %3\n v_mul_f32 %3, s20, %0\n ")
KERNEL(k_fma_lit, float, "v", "v_fma_f32 %0, %1, 2.0, %0\n v_fma_f32 %1, %2, 2.0, %1\n v_fma_f32 %2, %3, 2.0, %2\n v_fma_f32 %3, %0, 2.0, %3\n v_fma_f32 %0, %1, 2.0, %0\n v_fma_f32 %1, %2, 2.0, %1\n v_fma_f32 %2, %3, 2.0, %2\n v_fma_f32 %3, %0, 2.0, %3\n ")
KERNEL(k_fma_neg, float, "v", "v_fma_f32 %0, -%1, %2, %0\n v_fma_f32 %1, -%2, %3, %1\n v_fma_f32 %2, -%3, %0, %2\n v_fma_f32 %3, -%0, %1, %3\n v_fma_f32 %0, -%1, %2, %0\n v_fma_f32 %1, -%2, %3, %1\n v_fma_f32 %2, -%3, %0, %2\n v_fma_f32 %3, -%0, %1, %3\n ")
KERNEL(k_add_abs, float, "v", "v_add_f32_e64 %0, %1, |%2|\n v_add_f32_e64 %1, %2, |%3|\n v_add_f32_e64 %2, %3, |%0|\n v_add_f32_e64 %3, %0, |%1|\n v_add_f32_e64 %0, %1, |%2|\n v_add_f32_e64 %1, %2, |%3|\n v_add_f32_e64 %2, %3, |%0|\n v_add_f32_e64 %3, %0, |%1|\n ")
KERNEL(k_mul_u24, uint32_t, "v", "v_mul_u32_u24 %0, %1, %2\n v_mul_u32_u24 %1, %2, %3\n v_mul_u32_u24 %2, %3, %0\n v_mul_u32_u24 %3, %0, %1\n v_mul_u32_u24 %0, %1, %2\n v_mul_u32_u24 %1, %2, %3\n v_mul_u32_u24 %2, %3, %0\n v_mul_u32_u24 %3, %0, %1\n ")
KERNEL(k_or_b32, uint32_t, "v", "v_or_b32 %0, %1, %2\n v_or_b32 %1, %2, %3\n v_or_b32 %2, %3, %0\n v_or_b32 %3, %0, %1\n v_or_b32 %0, %1, %2\n v_or_b32 %1, %2, %3\n v_or_b32 %2, %3, %0\n v_or_b32 %3, %0, %1\n ")
KERNEL(k_xor_b32, uint32_t, "v", "v_xor_b32 %0, %1, %2\n v_xor_b32 %1, %2, %3\n v_xor_b32 %2, %3, %0\n v_xor_b32 %3, %0, %1\n v_xor_b32 %0, %1, %2\n v_xor_b32 %1, %2, %3\n v_xor_b32 %2, %3, %0\n v_xor_b32 %3, %0, %1\n ")
KERNEL(k_sub_u32, uint32_t, "v", "v_sub_u32 %0, %1, %2\n v_sub_u32 %1, %2, %3\n v_sub_u32 %2, %3, %0\n v_sub_u32 %3, %0, %1\n v_sub_u32 %0, %1, %2\n v_sub_u32 %1, %2, %3\n v_sub_u32 %2, %3, %0\n v_sub_u32 %3, %0, %1\n ")
KERNEL(k_lshl_add, uint32_t, "v", "v_lshl_add_u32 %0, %1, 3, %2\n v_lshl_add_u32 %1, %2, 3, %3\n v_lshl_add_u32 %2, %3, 3, %0\n v_lshl_add_u32 %3, %0, 3, %1\n v_lshl_add_u32 %0, %1, 3, %2\n v_lshl_add_u32 %1, %2, 3, %3\n v_lshl_add_u32 %2, %3, 3, %0\n v_lshl_add_u32 %3, %0, 3, %1\n ")
KERNEL(k_and_or, uint32_t, "v", "v_and_or_b32 %0, %1, %2, %0\n v_and_or_b32 %1, %2, %3, %1\n v_and_or_b32 %2, %3, %0, %2\n v_and_or_b32 %3, %0, %1, %3\n v_and_or_b32 %0, %1, %2, %0\n v_and_or_b32 %1, %2, %3, %1\n v_and_or_b32 %2, %3, %0, %2\n v_and_or_b32 %3, %0, %1, %3\n ")
KERNEL(k_add3, uint32_t, "v", "v_add3_u32 %0, %1, %2, %0\n v_add3_u32 %1, %2, %3, %1\n v_add3_u32 %2, %3, %0, %2\n v_add3_u32 %3, %0, %1, %3\n v_add3_u32 %0, %1, %2, %0\n v_add3_u32 %1, %2, %3, %1\n v_add3_u32 %2, %3, %0, %2\n v_add3_u32 %3, %0, %1, %3\n ")
KERNEL(k_cnd_indep, float, "v", "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %3, %3, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %1, %1, %3, vcc\n v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %3, %3, %1, vcc\n v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %1, %1, %3, vcc\n ")
KERNEL(k_cnd_sgpr, float, "v", "v_cndmask_b32_e64 %0, %1, %2, s[20:21]\n v_cndmask_b32_e64 %1, %2, %3, s[20:21]\n v_cndmask_b32_e64 %2, %3, %0, s[20:21]\n v_cndmask_b32_e64 %3, %0, %1, s[20:21]\n v_cndmask_b32_e64 %0, %1, %2, s[20:21]\n v_cndmask_b32_e64 %1, %2, %3, s[20:21]\n v_cndmask_b32_e64 %2, %3, %0, s[20:21]\n v_cndmask_b32_e64 %3, %0, %1, s[20:21]\n ")
KERNEL(k_max_u32, uint32_t, "v", "v_max_u32 %0, %1, %2\n v_max_u32 %1, %2, %3\n v_max_u32 %2, %3, %0\n v_max_u32 %3, %0, %1\n v_max_u32 %0, %1, %2\n v_max_u32 %1, %2, %3\n v_max_u32 %2, %3, %0\n v_max_u32 %3, %0, %1\n ")
KERNEL(k_ldexp, float, "v", "v_ldexp_f32 %0, %1, 3\n v_ldexp_f32 %1, %2, 3\n v_ldexp_f32 %2, %3, 3\n v_ldexp_f32 %3, %0, 3\n v_ldexp_f32 %0, %1, 3\n v_ldexp_f32 %1, %2, 3\n v_ldexp_f32 %2, %3, 3\n v_ldexp_f32 %3, %0, 3\n ")
KERNEL(k_mix_fma_cmp, float, "v", "v_fmac_f32 %0, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_fmac_f32 %2, %3, %0\n v_cmp_lt_f32 vcc, %3, %0\n v_fmac_f32 %0, %1, %2\n v_cmp_lt_f32 vcc, %1, %2\n v_fmac_f32 %2, %3, %0\n v_cmp_lt_f32 vcc, %3, %0\n ")
KERNEL(k_mix_fma_bfe, uint32_t, "v", "v_fmac_f32 %0, %1, %2\n v_bfe_u32 %2, %3, 8, 8\n v_fmac_f32 %2, %3, %0\n v_bfe_u32 %0, %1, 8, 8\n v_fmac_f32 %0, %1, %2\n v_bfe_u32 %2, %3, 8, 8\n v_fmac_f32 %2, %3, %0\n v_bfe_u32 %0, %1, 8, 8\n ")
KERNEL(k_mix_3fma_perm, uint32_t, "v", "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_perm_b32 %3, %0, %1, %3\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_perm_b32 %3, %0, %1, %3\n ")
KERNEL(k_mix_fma_exp, float, "v", "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_exp_f32 %3, %0\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_exp_f32 %3, %0\n ")
KERNEL(k_exp, float, "v", "v_exp_f32 %0, %1\n v_exp_f32 %1, %2\n v_exp_f32 %2, %3\n v_exp_f32 %3, %0\n v_exp_f32 %0, %1\n v_exp_f32 %1, %2\n v_exp_f32 %2, %3\n v_exp_f32 %3, %0\n ")
KERNEL(k_perm_ssel, uint32_t, "v", "v_perm_b32 %0, %1, %2, s20\n v_perm_b32 %1, %2, %3, s20\n v_perm_b32 %2, %3, %0, s20\n v_perm_b32 %3, %0, %1, s20\n v_perm_b32 %0, %1, %2, s20\n v_perm_b32 %1, %2, %3, s20\n v_perm_b32 %2, %3, %0, s20\n v_perm_b32 %3, %0, %1, s20\n ")
KERNEL(k_fmac_lit, float, "v", "v_fmac_f32 %0, 0x4459c000, %1\n v_fmac_f32 %1, 0x4459c000, %2\n v_fmac_f32 %2, 0x4459c000, %3\n v_fmac_f32 %3, 0x4459c000, %0\n v_fmac_f32 %0, 0x4459c000, %1\n v_fmac_f32 %1, 0x4459c000, %2\n v_fmac_f32 %2, 0x4459c000, %3\n v_fmac_f32 %3, 0x4459c000, %0\n ")
KERNEL(k_mul_lit, float, "v", "v_mul_f32 %0, 0x4459c000, %1\n v_mul_f32 %1, 0x4459c000, %2\n v_mul_f32 %2, 0x4459c000, %3\n v_mul_f32 %3, 0x4459c000, %0\n v_mul_f32 %0, 0x4459c000, %1\n v_mul_f32 %1, 0x4459c000, %2\n v_mul_f32 %2, 0x4459c000, %3\n v_mul_f32 %3, 0x4459c000, %0\n ")
KERNEL(k_and_lit, uint32_t, "v", "v_and_b32 %0, 0xffffff, %1\n v_and_b32 %1, 0xffffff, %2\n v_and_b32 %2, 0xffffff, %3\n v_and_b32 %3, 0xffffff, %0\n v_and_b32 %0, 0xffffff, %1\n v_and_b32 %1, 0xffffff, %2\n v_and_b32 %2, 0xffffff, %3\n v_and_b32 %3, 0xffffff, %0\n ")
KERNEL(k_cmp_sgpr, float, "v", "v_cmp_lt_f32 vcc, s20, %0\n v_cmp_lt_f32 vcc, s20, %1\n v_cmp_lt_f32 vcc, s20, %2\n v_cmp_lt_f32 vcc, s20, %3\n v_cmp_lt_f32 vcc, s20, %0\n v_cmp_lt_f32 vcc, s20, %1\n v_cmp_lt_f32 vcc, s20, %2\n v_cmp_lt_f32 vcc, s20, %3\n ")
KERNEL(k_cmp_zero, float, "v", "v_cmp_lt_f32 vcc, 0, %0\n v_cmp_lt_f32 vcc, 0, %1\n v_cmp_lt_f32 vcc, 0, %2\n v_cmp_lt_f32 vcc, 0, %3\n v_cmp_lt_f32 vcc, 0, %0\n v_cmp_lt_f32 vcc, 0, %1\n v_cmp_lt_f32 vcc, 0, %2\n v_cmp_lt_f32 vcc, 0, %3\n ")
KERNEL(k_alignbit, uint32_t, "v", "v_alignbit_b32 %0, %1, %2, 24\n v_alignbit_b32 %1, %2, %3, 24\n v_alignbit_b32 %2, %3, %0, 24\n v_alignbit_b32 %3, %0, %1, 24\n v_alignbit_b32 %0, %1, %2, 24\n v_alignbit_b32 %1, %2, %3, 24\n v_alignbit_b32 %2, %3, %0, 24\n v_alignbit_b32 %3, %0, %1, 24\n ")
KERNEL(k_mbcnt, uint32_t, "v", "v_mbcnt_lo_u32_b32 %0, s20, %1\n v_mbcnt_hi_u32_b32 %1, s21, %2\n v_mbcnt_lo_u32_b32 %2, s20, %3\n v_mbcnt_hi_u32_b32 %3, s21, %0\n v_mbcnt_lo_u32_b32 %0, s20, %1\n v_mbcnt_hi_u32_b32 %1, s21, %2\n v_mbcnt_lo_u32_b32 %2, s20, %3\n v_mbcnt_hi_u32_b32 %3, s21, %0\n ")
KERNEL(k_addc, uint32_t, "v", "v_addc_co_u32 %0, vcc, 0, %1, s[20:21]\n v_addc_co_u32 %1, vcc, 0, %2, s[20:21]\n v_addc_co_u32 %2, vcc, 0, %3, s[20:21]\n v_addc_co_u32 %3, vcc, 0, %0, s[20:21]\n v_addc_co_u32 %0, vcc, 0, %1, s[20:21]\n v_addc_co_u32 %1, vcc, 0, %2, s[20:21]\n v_addc_co_u32 %2, vcc, 0, %3, s[20:21]\n v_addc_co_u32 %3, vcc, 0, %0, s[20:21]\n ")
KERNEL(k_readlane, uint32_t, "v", "v_readlane_b32 s22, %0, 5\n v_readlane_b32 s22, %1, 5\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s22, %3, 5\n v_readlane_b32 s22, %0, 5\n v_readlane_b32 s22, %1, 5\n v_readlane_b32 s22, %2, 5\n v_readlane_b32 s22, %3, 5\n ")
KERNEL(k_mix_f64_perm, double, "v", "v_fma_f64 %0, %0, %1, %0\n v_perm_b32 v100, v101, v102, v103\n v_fma_f64 %2, %2, %3, %2\n v_perm_b32 v100, v101, v102, v103\n v_fma_f64 %0, %0, %1, %0\n v_perm_b32 v100, v101, v102, v103\n v_fma_f64 %2, %2, %3, %2\n v_perm_b32 v100, v101, v102, v103\n ")
KERNEL(k_mix_exp_perm, float, "v", "v_exp_f32 %0, %1\n v_perm_b32 v100, v101, v102, v103\n v_exp_f32 %2, %3\n v_perm_b32 v100, v101, v102, v103\n v_exp_f32 %0, %1\n v_perm_b32 v100, v101, v102, v103\n v_exp_f32 %2, %3\n v_perm_b32 v100, v101, v102, v103\n ")
KERNEL(k_mix_exp_2fma, float, "v", "v_exp_f32 %0, %1\n v_fmac_f32 v100, v101, v102\n v_fmac_f32 v104, v101, v102\n v_perm_b32 v105, v101, v102, v103\n v_exp_f32 %0, %1\n v_fmac_f32 v100, v101, v102\n v_fmac_f32 v104, v101, v102\n v_perm_b32 v105, v101, v102, v103\n ")
KERNEL(k_mix_f64_fma, double, "v", "v_fma_f64 %0, %0, %1, %0\n v_fmac_f32 v100, v101, v102\n v_perm_b32 v105, v101, v102, v103\n v_fmac_f32 v104, v101, v102\n v_fma_f64 %0, %0, %1, %0\n v_fmac_f32 v100, v101, v102\n v_perm_b32 v105, v101, v102, v103\n v_fmac_f32 v104, v101, v102\n ")
KERNEL(k_mfma4, float, "v", "v_mfma_f32_4x4x1_16b_f32 v[100:103], %0, %1, v[100:103]\n v_mfma_f32_4x4x1_16b_f32 v[104:107], %1, %2, v[104:107]\n v_mfma_f32_4x4x1_16b_f32 v[108:111], %2, %3, v[108:111]\n v_mfma_f32_4x4x1_16b_f32 v[112:115], %3, %0, v[112:115]\n v_mfma_f32_4x4x1_16b_f32 v[100:103], %0, %1, v[100:103]\n v_mfma_f32_4x4x1_16b_f32 v[104:107], %1, %2, v[104:107]\n v_mfma_f32_4x4x1_16b_f32 v[108:111], %2, %3, v[108:111]\n v_mfma_f32_4x4x1_16b_f32 v[112:115], %3, %0, v[112:115]\n")
KERNEL(k_mfma4_perm, uint32_t, "v", "v_mfma_f32_4x4x1_16b_f32 v[100:103], %0, %1, v[100:103]\n v_perm_b32 v116, %1, %2, %3\n v_perm_b32 v117, %1, %2, %3\n v_perm_b32 v118, %1, %2, %3\n v_mfma_f32_4x4x1_16b_f32 v[104:107], %1, %2, v[104:107]\n v_perm_b32 v116, %1, %2, %3\n v_perm_b32 v117, %1, %2, %3\n v_perm_b32 v118, %1, %2, %3\n")
KERNEL(k_mfma4_fma, float, "v", "v_mfma_f32_4x4x1_16b_f32 v[100:103], %0, %1, v[100:103]\n v_fmac_f32 v116, %1, %2\n v_fmac_f32 v117, %1, %2\n v_fmac_f32 v118, %1, %2\n v_mfma_f32_4x4x1_16b_f32 v[104:107], %1, %2, v[104:107]\n v_fmac_f32 v116, %1, %2\n v_fmac_f32 v117, %1, %2\n v_fmac_f32 v118, %1, %2\n")
// ---- round 4: the matrix pipe once more, in the shape the north star has in mind for the pixel x 3 products (bf16 16x16x16, 4 passes
// of 4 cycles): alone, and interleaved 1:3 with VALU work from the SAME wave (does the MFMA issue hide behind the VALU instructions
// of its own wave?).  With 2..8 waves per SIMD the other waves' VALU work can overlap too -- that is the w/SIMD sweep.
KERNEL(k_mfma16bf, float, "v", "v_mfma_f32_16x16x16_bf16 v[100:103], v[116:117], v[104:105], v[100:103]\n v_mfma_f32_16x16x16_bf16 v[106:109], v[116:117], v[104:105], v[106:109]\n v_mfma_f32_16x16x16_bf16 v[110:113], v[116:117], v[104:105], v[110:113]\n v_mfma_f32_16x16x16_bf16 v[100:103], v[116:117], v[104:105], v[100:103]\n v_mfma_f32_16x16x16_bf16 v[106:109], v[116:117], v[104:105], v[106:109]\n v_mfma_f32_16x16x16_bf16 v[110:113], v[116:117], v[104:105], v[110:113]\n v_mfma_f32_16x16x16_bf16 v[100:103], v[116:117], v[104:105], v[100:103]\n v_mfma_f32_16x16x16_bf16 v[106:109], v[116:117], v[104:105], v[106:109]\n")
KERNEL(k_mfma16bf_3fma, float, "v", "v_mfma_f32_16x16x16_bf16 v[100:103], v[116:117], v[104:105], v[100:103]\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_mfma_f32_16x16x16_bf16 v[106:109], v[116:117], v[104:105], v[106:109]\n v_fmac_f32 %3, %0, %1\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n")
KERNEL(k_mfma16bf_3perm, uint32_t, "v", "v_mfma_f32_16x16x16_bf16 v[100:103], v[116:117], v[104:105], v[100:103]\n v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %1, %2, %3, %0\n v_perm_b32 %2, %3, %0, %1\n v_mfma_f32_16x16x16_bf16 v[106:109], v[116:117], v[104:105], v[106:109]\n v_perm_b32 %3, %0, %1, %2\n v_perm_b32 %0, %1, %2, %3\n v_perm_b32 %1, %2, %3, %0\n")
KERNEL(k_only_6fma, float, "v", "v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n")
KERNEL(k_mfma16bf_7fma, float, "v", "v_mfma_f32_16x16x16_bf16 v[100:103], v[116:117], v[104:105], v[100:103]\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n")
typedef void (*kern_t)(float*, int, float);
static void run(const char* name, kern_t kf, int instr_per_rep, float* d_out) {
    const int iters = 2048;
    printf("%-22s", name);
    for (int b : {1, 2, 4}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int grid = 256 * b;
        hipLaunchKernelGGL(kf, dim3(grid), dim3(512), 0, 0, d_out, 8, 1.0f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kf, dim3(grid), dim3(512), 0, 0, d_out, iters, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double winst = (double)instr_per_rep * 8 * iters * (2.0 * b);
        printf("  w/SIMD %d: %6.3f G/s", 2 * b, winst / (ms * 1e-3) * 1e-9);
    }
    printf("\n");
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* d_out; hipMalloc(&d_out, 1024);
#define RUN(k, n) run(#k, k, n, d_out)
    RUN(k_mfma16bf,8);RUN(k_mfma16bf_3fma,8);RUN(k_mfma16bf_3perm,8);RUN(k_only_6fma,6);RUN(k_mfma16bf_7fma,8);
    RUN(k_mfma4,8);RUN(k_mfma4_perm,8);RUN(k_mfma4_fma,8);
    RUN(k_perm_ssel,8);RUN(k_fmac_lit,8);RUN(k_mul_lit,8);RUN(k_and_lit,8);RUN(k_cmp_sgpr,8);RUN(k_cmp_zero,8);RUN(k_alignbit,8);RUN(k_mbcnt,8);RUN(k_addc,8);RUN(k_readlane,8);RUN(k_mix_f64_perm,8);RUN(k_mix_exp_perm,8);RUN(k_mix_exp_2fma,8);RUN(k_mix_f64_fma,8);
    RUN(k_fmac_sgpr,8);RUN(k_mul_sgpr,8);RUN(k_fma_lit,8);RUN(k_fma_neg,8);RUN(k_add_abs,8);RUN(k_mul_u24,8);RUN(k_or_b32,8);RUN(k_xor_b32,8);RUN(k_sub_u32,8);RUN(k_lshl_add,8);RUN(k_and_or,8);RUN(k_add3,8);RUN(k_cnd_indep,8);RUN(k_cnd_sgpr,8);RUN(k_max_u32,8);RUN(k_ldexp,8);RUN(k_mix_fma_cmp,8);RUN(k_mix_fma_bfe,8);RUN(k_mix_3fma_perm,8);RUN(k_mix_fma_exp,8);RUN(k_exp,8);
    RUN(k_add_f32, 8); RUN(k_mul_f32, 8); RUN(k_fmac_f32, 8); RUN(k_fma_sgpr, 8); RUN(k_min_f32, 8); RUN(k_min3_f32, 8);
    RUN(k_cmp_f32, 8); RUN(k_cmp_vcc, 8); RUN(k_cndmask, 8); RUN(k_and_b32, 8); RUN(k_bfe_u32, 8); RUN(k_add_u32, 8);
    RUN(k_lshlrev, 8); RUN(k_mov_b32, 8); RUN(k_cvt_ubyte, 8); RUN(k_cvt_u32_f32, 8); RUN(k_floor_f32, 8); RUN(k_add_f64, 8);
    RUN(k_sdwa_shl, 8); RUN(k_dpp_add, 8); RUN(k_mix_fma_perm, 8); RUN(k_mix_fma_f64, 8);
    RUN(k_ds_b64, 8); RUN(k_ds_b96, 8); RUN(k_ds_b128, 8); RUN(k_mix_valu_lds, 8);
    return 0;
}
