// Micro-benchmarks of per-SIMD issue rates on gfx950 (development aid; not part of the library).
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_issue tools/ubench_issue.hip && build/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

constexpr int kIters = 256;

template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    uint32_t u0 = threadIdx.x, u1 = u0 * 3, u2 = u0 * 5, u3 = u0 * 7;
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i;
    __syncthreads();
    uint32_t la = (threadIdx.x * 4) & 16383u;
    uint32_t s0 = 1, s1 = 2;
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {  // 64 independent-ish fma (8 chains)
            REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                              "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 1) {  // 64 fma interleaved with 64 SALU
            REP8(asm volatile("v_fma_f32 %0, %0, %0, %0\n s_add_u32 %8, %8, %9\n v_fma_f32 %1, %1, %1, %1\n s_and_b32 %9, %9, %8\n v_fma_f32 %2, %2, %2, %2\n s_add_u32 %8, %8, %9\n v_fma_f32 %3, %3, %3, %3\n s_and_b32 %9, %9, %8\n"
                              "v_fma_f32 %4, %4, %4, %4\n s_add_u32 %8, %8, %9\n v_fma_f32 %5, %5, %5, %5\n s_and_b32 %9, %9, %8\n v_fma_f32 %6, %6, %6, %6\n s_add_u32 %8, %8, %9\n v_fma_f32 %7, %7, %7, %7\n s_and_b32 %9, %9, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), "+s"(s1) : : "scc");)
        } else if (KIND == 2) {  // 64 v_exp_f32
            REP8(asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) {  // 64 v_fma_f64 (4 chains)
            REP8(asm volatile("v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                              "v_fma_f64 %0, %0, %0, %0\n v_fma_f64 %1, %1, %1, %1\n v_fma_f64 %2, %2, %2, %2\n v_fma_f64 %3, %3, %3, %3\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (KIND == 4) {  // 64 v_mad_u32_u24
            REP8(asm volatile("v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1\n"
                              "v_mad_u32_u24 %0, %0, %1, %2\n v_mad_u32_u24 %1, %1, %2, %3\n v_mad_u32_u24 %2, %2, %3, %0\n v_mad_u32_u24 %3, %3, %0, %1\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
        } else if (KIND == 5) {  // 64 v_perm_b32
            REP8(asm volatile("v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1\n"
                              "v_perm_b32 %0, %0, %1, %2\n v_perm_b32 %1, %1, %2, %3\n v_perm_b32 %2, %2, %3, %0\n v_perm_b32 %3, %3, %0, %1\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
        } else if (KIND == 6) {  // 64 v_cmp + v_cndmask pairs (32 each)
            REP8(asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_f32 vcc, %1, %2\n v_cndmask_b32 %3, %3, %0, vcc\n"
                              "v_cmp_lt_f32 vcc, %2, %3\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_lt_f32 vcc, %3, %0\n v_cndmask_b32 %1, %1, %2, vcc\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : : "vcc");)
        } else if (KIND == 7) {  // 56 fma + 8 ds_read_b32 (random-ish addresses), results consumed
            REP8(asm volatile("ds_read_b32 %8, %9\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                              "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n s_waitcnt lgkmcnt(0)\n v_add_f32 %7, %7, %8\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u1) : "v"(la));)
        } else if (KIND == 8) {  // 64 v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                              "v_pk_fma_f32 %0, %0, %0, %0\n v_pk_fma_f32 %1, %1, %1, %1\n v_pk_fma_f32 %2, %2, %2, %2\n v_pk_fma_f32 %3, %3, %3, %3\n"
                              : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));)
        } else if (KIND == 9) {  // 64 v_lshl_or_b32
            REP8(asm volatile("v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 3, %2\n v_lshl_or_b32 %2, %2, 3, %3\n v_lshl_or_b32 %3, %3, 3, %0\n"
                              "v_lshl_or_b32 %0, %0, 3, %1\n v_lshl_or_b32 %1, %1, 3, %2\n v_lshl_or_b32 %2, %2, 3, %3\n v_lshl_or_b32 %3, %3, 3, %0\n"
                              : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
        } else if (KIND == 10) {  // 64 v_cvt_pk_u8_f32
            REP8(asm volatile("v_cvt_pk_u8_f32 %4, %0, 0, %4\n v_cvt_pk_u8_f32 %5, %1, 1, %5\n v_cvt_pk_u8_f32 %6, %2, 2, %6\n v_cvt_pk_u8_f32 %7, %3, 3, %7\n"
                              "v_cvt_pk_u8_f32 %4, %0, 0, %4\n v_cvt_pk_u8_f32 %5, %1, 1, %5\n v_cvt_pk_u8_f32 %6, %2, 2, %6\n v_cvt_pk_u8_f32 %7, %3, 3, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3));)
        } else if (KIND == 11) {  // 64 ds_read_b64 only (LDS issue rate), conflict-free addresses
            REP8(asm volatile("ds_read_b64 %0, %2\n ds_read_b64 %1, %2 offset:512\n ds_read_b64 %0, %2 offset:1024\n ds_read_b64 %1, %2 offset:1536\n"
                              "ds_read_b64 %0, %2 offset:2048\n ds_read_b64 %1, %2 offset:2560\n ds_read_b64 %0, %2 offset:3072\n ds_read_b64 %1, %2 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                              : "+v"(d0), "+v"(d1) : "v"((threadIdx.x & 63) * 8u));)
        } else if (KIND == 12) {  // 64 ds_read_b32 only
            REP8(asm volatile("ds_read_b32 %0, %2\n ds_read_b32 %1, %2 offset:512\n ds_read_b32 %0, %2 offset:1024\n ds_read_b32 %1, %2 offset:1536\n"
                              "ds_read_b32 %0, %2 offset:2048\n ds_read_b32 %1, %2 offset:2560\n ds_read_b32 %0, %2 offset:3072\n ds_read_b32 %1, %2 offset:3584\n s_waitcnt lgkmcnt(0)\n"
                              : "+v"(a0), "+v"(a1) : "v"((threadIdx.x & 63) * 4u));)
        }
    }
    float r = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(u0 ^ u1 ^ u2 ^ u3) + (float)(s0 + s1);
    if (r == 123.456f) out[0] = r;
}

// rounding behaviour of v_cvt_pk_u8_f32 under MODE.fp_round = RTZ
__global__ void k_round(const float* in, uint32_t* out, int n) {
    int i = threadIdx.x;
    if (i >= n) return;
    float x = in[i];
    uint32_t a = 0, b = 0;
    asm volatile("v_cvt_pk_u8_f32 %0, %1, 0, %0" : "+v"(a) : "v"(x));
    asm volatile("s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 3\n v_cvt_pk_u8_f32 %0, %1, 0, %0\n s_setreg_imm32_b32 hwreg(HW_REG_MODE, 0, 2), 0" : "+v"(b) : "v"(x));
    out[2 * i] = a; out[2 * i + 1] = b;
}

template <int KIND>
static void run(const char* name, int instr_per_iter, float* d_out) {
    const int blocks_per_cu[] = {1, 2, 4};
    for (int b : blocks_per_cu) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        const int grid = 256 * b;
        k<KIND><<<grid, 512>>>(d_out, 8, 1.0f);
        hipEventRecord(e0);
        k<KIND><<<grid, 512>>>(d_out, kIters * 16, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        // wave-instructions per SIMD: each SIMD hosts 2*b waves (8 waves per block over 4 SIMDs)
        const double winst = (double)instr_per_iter * kIters * 16 * (2.0 * b);
        printf("%-28s waves/SIMD %d  %8.3f ms  %7.3f G wave-instr/s per SIMD\n", name, 2 * b, ms, winst / (ms * 1e-3) * 1e-9);
    }
}

int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* d_out; hipMalloc(&d_out, 1024);
    run<0>("v_fma_f32", 64, d_out);
    run<1>("v_fma_f32 + SALU 1:1 (VALU)", 64, d_out);
    run<2>("v_exp_f32", 64, d_out);
    run<3>("v_fma_f64", 64, d_out);
    run<4>("v_mad_u32_u24", 64, d_out);
    run<5>("v_perm_b32", 64, d_out);
    run<6>("v_cmp+v_cndmask", 64, d_out);
    run<7>("7 fma + ds_read + add", 72, d_out);
    run<8>("v_pk_fma_f32", 64, d_out);
    run<9>("v_lshl_or_b32", 64, d_out);
    run<10>("v_cvt_pk_u8_f32", 64, d_out);
    run<11>("ds_read_b64", 64, d_out);
    run<12>("ds_read_b32", 64, d_out);
    // rounding test
    std::vector<float> h = {0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 2.9999f, 3.0f, 254.5f, 254.99f, 255.0f, 255.7f, 256.0f, 300.0f, -0.5f};
    float* d_in; uint32_t* d_o;
    hipMalloc(&d_in, h.size() * 4); hipMalloc(&d_o, h.size() * 8);
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    k_round<<<1, 64>>>(d_in, d_o, (int)h.size());
    std::vector<uint32_t> o(h.size() * 2);
    hipMemcpy(o.data(), d_o, o.size() * 4, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); ++i) printf("cvt_pk_u8(%g): default %u  rtz %u\n", h[i], o[2 * i], o[2 * i + 1]);
    return 0;
}
