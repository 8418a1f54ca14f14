// Development micro-benchmark: variants of the apply sweep, timed with hipEvents.
// Not product code.  Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I stainlib_amd/csrc tools/kbench_apply.hip -o gpurun_out/kbench_apply
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "apply_kernels.hpp"
using namespace sl;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int VEC>
__global__ __launch_bounds__(256) void k_copy(const uint32_t* __restrict__ a, uint32_t* __restrict__ b, size_t nvec) {
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        if (VEC == 3) { reinterpret_cast<Chunk*>(b)[i] = reinterpret_cast<const Chunk*>(a)[i]; }
        else { reinterpret_cast<uint4*>(b)[i] = reinterpret_cast<const uint4*>(a)[i]; }
    }
}

// generalised apply kernel with knobs
template <int THREADS, int REPL, int U, bool NT, int ODMODE, int WORK>
__global__ __launch_bounds__(THREADS) void k_apply_x(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out,
                                                     int P, int parts, const double* __restrict__ M_src,
                                                     const double* __restrict__ maxC_src,
                                                     const double* __restrict__ M_tgt,
                                                     const double* __restrict__ maxC_tgt, double lam) {
    __shared__ float s_od[256 * REPL];
    if (ODMODE == 0) fill_od_lut<REPL>(s_od);
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const int tid = threadIdx.x;
    const uint32_t cp = tid & (REPL - 1);
    LassoK L;
    lasso_consts(M_src + 6 * (size_t)tile, lam, L);
    uni(L);
    ReconK R;
    for (int i = 0; i < 2; ++i) {
        const double ratio = maxC_tgt[i] / maxC_src[2 * (size_t)tile + i];
        for (int c = 0; c < 3; ++c) R.q[i][c] = uni((float)(-1.4426950408889634 * ratio * M_tgt[3 * i + c]));
    }
    __syncthreads();
    const size_t nbytes = (size_t)P * 3;
    const Chunk* src = reinterpret_cast<const Chunk*>(rgb + (size_t)tile * nbytes);
    Chunk* dst = reinterpret_cast<Chunk*>(out + (size_t)tile * nbytes);
    const int nch = P >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span;
    const int c1 = min(nch, c0 + span);
    for (int c = c0 + tid; c < c1; c += THREADS * U) {
        Chunk in[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + u * THREADS;
            if (cc < c1) {
                if (NT) { in[u].w0 = __builtin_nontemporal_load(&src[cc].w0); in[u].w1 = __builtin_nontemporal_load(&src[cc].w1); in[u].w2 = __builtin_nontemporal_load(&src[cc].w2); }
                else in[u] = src[cc];
            } else in[u] = Chunk{0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + u * THREADS;
            uint32_t ob[12];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                float x, y, z;
                const uint32_t r = chunk_byte(in[u], 3 * px), g = chunk_byte(in[u], 3 * px + 1), b = chunk_byte(in[u], 3 * px + 2);
                if (ODMODE == 0) { x = lut<REPL>(s_od, r, cp); y = lut<REPL>(s_od, g, cp); z = lut<REPL>(s_od, b, cp); }
                else {  // log-based OD (no table): -ln(max(v,1)/255)
                    const float k = 0.69314718056f;
                    x = fmaxf((7.99435343686f - __builtin_amdgcn_logf(fmaxf((float)r, 1.f))) * k, 1e-6f);
                    y = fmaxf((7.99435343686f - __builtin_amdgcn_logf(fmaxf((float)g, 1.f))) * k, 1e-6f);
                    z = fmaxf((7.99435343686f - __builtin_amdgcn_logf(fmaxf((float)b, 1.f))) * k, 1e-6f);
                }
                float a1, a2, v[3];
                if (WORK >= 1) lasso2(L, x, y, z, a1, a2); else { a1 = x + y; a2 = z; }
                if (WORK >= 2) recon_px<false>(R, a1, a2, v); else { v[0] = a1 * 100.f; v[1] = a2 * 100.f; v[2] = (a1 + a2) * 50.f; }
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) ob[3 * px + ch] = trunc_u8(v[ch]);
            }
            Chunk o;
            o.w0 = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
            o.w1 = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
            o.w2 = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
            if (cc < c1) {
                if (NT) { __builtin_nontemporal_store(o.w0, &dst[cc].w0); __builtin_nontemporal_store(o.w1, &dst[cc].w1); __builtin_nontemporal_store(o.w2, &dst[cc].w2); }
                else dst[cc] = o;
            }
        }
    }
}

struct Ctx { uint8_t *in, *out; double *M, *mc, *Mt, *mct; int n, P; size_t bytes; };

template <class F> float timeit(F&& f, int reps = 20) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 3; ++i) f();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; ++i) f();
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

template <int THREADS, int REPL, int U, bool NT, int ODMODE, int WORK>
void run(const Ctx& c, int parts, const char* name) {
    float ms = timeit([&] { hipLaunchKernelGGL((k_apply_x<THREADS, REPL, U, NT, ODMODE, WORK>), dim3(c.n * parts), dim3(THREADS), 0, 0,
                                               c.in, c.out, c.P, parts, c.M, c.mc, c.Mt, c.mct, 0.01); });
    printf("%-44s parts=%3d  %.3f ms  %7.1f GB/s  frac=%.3f\n", name, parts, ms, 2.0 * c.bytes / ms / 1e6, 2.0 * c.bytes / ms / 1e6 / 8000);
}


// software-pipelined loads: next trip's chunks are requested before this trip's arithmetic
template <int THREADS, int REPL, int U>
__global__ __launch_bounds__(THREADS) void k_apply_pipe(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out,
                                                        int P, int parts, const double* __restrict__ M_src,
                                                        const double* __restrict__ maxC_src,
                                                        const double* __restrict__ M_tgt,
                                                        const double* __restrict__ maxC_tgt, double lam) {
    __shared__ float s_od[256 * REPL];
    fill_od_lut<REPL>(s_od);
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const int tid = threadIdx.x;
    const uint32_t cp = tid & (REPL - 1);
    LassoK L;
    lasso_consts(M_src + 6 * (size_t)tile, lam, L);
    uni(L);
    ReconK R;
    for (int i = 0; i < 2; ++i) {
        const double ratio = maxC_tgt[i] / maxC_src[2 * (size_t)tile + i];
        for (int c = 0; c < 3; ++c) R.q[i][c] = uni((float)(-1.4426950408889634 * ratio * M_tgt[3 * i + c]));
    }
    __syncthreads();
    const size_t nbytes = (size_t)P * 3;
    const Chunk* src = reinterpret_cast<const Chunk*>(rgb + (size_t)tile * nbytes);
    Chunk* dst = reinterpret_cast<Chunk*>(out + (size_t)tile * nbytes);
    const int nch = P >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span;
    const int c1 = min(nch, c0 + span);
    Chunk cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int cc = c0 + tid + u * THREADS; cur[u] = cc < c1 ? src[cc] : Chunk{0, 0, 0}; }
    for (int c = c0 + tid; c < c1; c += THREADS * U) {
        Chunk nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int cc = c + THREADS * U + u * THREADS; nxt[u] = cc < c1 ? src[cc] : Chunk{0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + u * THREADS;
            uint32_t ob[12];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const uint32_t r = chunk_byte(cur[u], 3 * px), g = chunk_byte(cur[u], 3 * px + 1), b = chunk_byte(cur[u], 3 * px + 2);
                const float x = lut<REPL>(s_od, r, cp), y = lut<REPL>(s_od, g, cp), z = lut<REPL>(s_od, b, cp);
                float a1, a2, v[3];
                lasso2(L, x, y, z, a1, a2);
                recon_px<false>(R, a1, a2, v);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) ob[3 * px + ch] = trunc_u8(v[ch]);
            }
            Chunk o;
            o.w0 = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
            o.w1 = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
            o.w2 = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
            if (cc < c1) dst[cc] = o;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
}

template <int THREADS, int REPL, int U>
void run_pipe(const Ctx& c, int parts, const char* name) {
    float ms = timeit([&] { hipLaunchKernelGGL((k_apply_pipe<THREADS, REPL, U>), dim3(c.n * parts), dim3(THREADS), 0, 0,
                                               c.in, c.out, c.P, parts, c.M, c.mc, c.Mt, c.mct, 0.01); });
    printf("%-44s parts=%3d  %.3f ms  %7.1f GB/s  frac=%.3f\n", name, parts, ms, 2.0 * c.bytes / ms / 1e6, 2.0 * c.bytes / ms / 1e6 / 8000);
}

// ---- packed-math variant: two pixels per v_pk_* instruction
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float a) { return v2f{a, a}; }
__device__ __forceinline__ void lasso2_x2(const LassoK& k, v2f x, v2f y, v2f z, v2f& c1, v2f& c2) {
    const v2f b1 = fma2(splat(k.m[0][2]), z, fma2(splat(k.m[0][1]), y, fma2(splat(k.m[0][0]), x, splat(-k.lam))));
    const v2f b2 = fma2(splat(k.m[1][2]), z, fma2(splat(k.m[1][1]), y, fma2(splat(k.m[1][0]), x, splat(-k.lam))));
    const v2f a1 = fma2(splat(k.i12), b2, splat(k.i11) * b1);
    const v2f a2 = fma2(splat(k.i12), b1, splat(k.i22) * b2);
    const v2f s1 = b1 * splat(k.r1);
    v2f s2 = b2 * splat(k.r2);
    s2 = __builtin_elementwise_max(s2, splat(0.0f));
    const v2f t = fma2(splat(-k.g12), s1, b2);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const bool both = (a1[i] >= 0.0f) & (a2[i] >= 0.0f);
        const bool only1 = (b1[i] > 0.0f) & (t[i] <= 0.0f);
        c1[i] = both ? a1[i] : (only1 ? s1[i] : 0.0f);
        c2[i] = both ? a2[i] : (only1 ? 0.0f : s2[i]);
    }
}
template <int THREADS, int U>
__global__ __launch_bounds__(THREADS) void k_apply_pk(const uint8_t* __restrict__ rgb, uint8_t* __restrict__ out,
                                                      int P, int parts, const double* __restrict__ M_src,
                                                      const double* __restrict__ maxC_src,
                                                      const double* __restrict__ M_tgt,
                                                      const double* __restrict__ maxC_tgt, double lam) {
    __shared__ float s_od[256];
    fill_od_lut<1>(s_od);
    const int tile = blockIdx.x / parts, part = blockIdx.x % parts;
    const int tid = threadIdx.x;
    LassoK L;
    lasso_consts(M_src + 6 * (size_t)tile, lam, L);
    uni(L);
    ReconK R;
    for (int i = 0; i < 2; ++i) {
        const double ratio = maxC_tgt[i] / maxC_src[2 * (size_t)tile + i];
        for (int c = 0; c < 3; ++c) R.q[i][c] = uni((float)(-1.4426950408889634 * ratio * M_tgt[3 * i + c]));
    }
    __syncthreads();
    const size_t nbytes = (size_t)P * 3;
    const Chunk* src = reinterpret_cast<const Chunk*>(rgb + (size_t)tile * nbytes);
    Chunk* dst = reinterpret_cast<Chunk*>(out + (size_t)tile * nbytes);
    const int nch = P >> 2;
    const int span = (nch + parts - 1) / parts;
    const int c0 = part * span;
    const int c1 = min(nch, c0 + span);
    Chunk cur[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const int cc = c0 + tid + u * THREADS; cur[u] = cc < c1 ? src[cc] : Chunk{0, 0, 0}; }
    for (int c = c0 + tid; c < c1; c += THREADS * U) {
        Chunk nxt[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const int cc = c + THREADS * U + u * THREADS; nxt[u] = cc < c1 ? src[cc] : Chunk{0, 0, 0}; }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int cc = c + u * THREADS;
            uint32_t ob[12];
#pragma unroll
            for (int pp = 0; pp < 2; ++pp) {      // pixel pair (2pp, 2pp+1)
                v2f x, y, z;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int px = 2 * pp + i;
                    x[i] = s_od[chunk_byte(cur[u], 3 * px)]; y[i] = s_od[chunk_byte(cur[u], 3 * px + 1)]; z[i] = s_od[chunk_byte(cur[u], 3 * px + 2)];
                }
                v2f a1, a2;
                lasso2_x2(L, x, y, z, a1, a2);
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) {
                    const v2f e = fma2(a1, splat(R.q[0][ch]), a2 * splat(R.q[1][ch]));
                    v2f t;
                    t[0] = __builtin_amdgcn_exp2f(e[0]); t[1] = __builtin_amdgcn_exp2f(e[1]);
                    t = t * splat(255.0f);
                    ob[3 * (2 * pp) + ch] = trunc_u8(t[0]);
                    ob[3 * (2 * pp + 1) + ch] = trunc_u8(t[1]);
                }
            }
            Chunk o;
            o.w0 = ob[0] | (ob[1] << 8) | (ob[2] << 16) | (ob[3] << 24);
            o.w1 = ob[4] | (ob[5] << 8) | (ob[6] << 16) | (ob[7] << 24);
            o.w2 = ob[8] | (ob[9] << 8) | (ob[10] << 16) | (ob[11] << 24);
            if (cc < c1) dst[cc] = o;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) cur[u] = nxt[u];
    }
}
template <int THREADS, int U>
void run_pk(const Ctx& c, int parts, const char* name) {
    float ms = timeit([&] { hipLaunchKernelGGL((k_apply_pk<THREADS, U>), dim3(c.n * parts), dim3(THREADS), 0, 0,
                                               c.in, c.out, c.P, parts, c.M, c.mc, c.Mt, c.mct, 0.01); });
    printf("%-44s parts=%3d  %.3f ms  %7.1f GB/s  frac=%.3f\n", name, parts, ms, 2.0 * c.bytes / ms / 1e6, 2.0 * c.bytes / ms / 1e6 / 8000);
}
int main(int argc, char** argv) {
    Ctx c; c.n = argc > 1 ? atoi(argv[1]) : 256; c.P = 1024 * 1024; c.bytes = (size_t)c.n * c.P * 3;
    CK(hipMalloc(&c.in, c.bytes)); CK(hipMalloc(&c.out, c.bytes));
    std::vector<uint8_t> h(c.bytes);
    uint32_t s = 12345; for (size_t i = 0; i < c.bytes; ++i) { s = s * 1664525u + 1013904223u; h[i] = 40 + ((s >> 24) % 200); }
    CK(hipMemcpy(c.in, h.data(), c.bytes, hipMemcpyHostToDevice));
    std::vector<double> M(6 * c.n), mc(2 * c.n);
    double m0[6] = {0.626, 0.727, 0.283, 0.106, 0.987, 0.122};
    for (int t = 0; t < c.n; ++t) { for (int i = 0; i < 6; ++i) M[6 * t + i] = m0[i]; mc[2 * t] = 1.9; mc[2 * t + 1] = 1.5; }
    double Mt[6] = {0.55, 0.75, 0.35, 0.10, 0.95, 0.20}, mct[2] = {2.0, 1.4};
    CK(hipMalloc(&c.M, M.size() * 8)); CK(hipMalloc(&c.mc, mc.size() * 8)); CK(hipMalloc(&c.Mt, 48)); CK(hipMalloc(&c.mct, 16));
    CK(hipMemcpy(c.M, M.data(), M.size() * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(c.mc, mc.data(), mc.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemcpy(c.Mt, Mt, 48, hipMemcpyHostToDevice)); CK(hipMemcpy(c.mct, mct, 16, hipMemcpyHostToDevice));

    for (int g : {2048, 8192}) {
        float ms = timeit([&] { hipLaunchKernelGGL(k_copy<3>, dim3(g), dim3(256), 0, 0, (const uint32_t*)c.in, (uint32_t*)c.out, c.bytes / 12); });
        printf("copy dwordx3 grid=%d: %.3f ms %.1f GB/s\n", g, ms, 2.0 * c.bytes / ms / 1e6);
        ms = timeit([&] { hipLaunchKernelGGL(k_copy<4>, dim3(g), dim3(256), 0, 0, (const uint32_t*)c.in, (uint32_t*)c.out, c.bytes / 16); });
        printf("copy dwordx4 grid=%d: %.3f ms %.1f GB/s\n", g, ms, 2.0 * c.bytes / ms / 1e6);
    }
    run<256, 1, 4, false, 0, 2>(c, 32, "T256 R1 U4 (product)");
    run_pk<256, 2>(c, 32, "PK   T256 R1 U2");
    run_pk<256, 1>(c, 32, "PK   T256 R1 U1");
    run_pk<512, 2>(c, 16, "PK   T512 R1 U2");
    run_pipe<256, 1, 4>(c, 32, "PIPE T256 R1 U4");
    run_pipe<256, 1, 2>(c, 32, "PIPE T256 R1 U2");
    run_pipe<256, 1, 1>(c, 32, "PIPE T256 R1 U1");
    run_pipe<512, 1, 2>(c, 16, "PIPE T512 R1 U2");
    run_pipe<256, 8, 2>(c, 32, "PIPE T256 R8 U2");
    run_pipe<256, 32, 2>(c, 8, "PIPE T256 R32 U2 parts8");
    run<256, 32, 4, false, 0, 2>(c, 32, "T256 R32 U4");
    run<256, 32, 4, false, 0, 2>(c, 16, "T256 R32 U4");
    run<256, 32, 4, false, 0, 2>(c, 8, "T256 R32 U4");
    run<512, 32, 4, false, 0, 2>(c, 16, "T512 R32 U4");
    run<1024, 32, 4, false, 0, 2>(c, 8, "T1024 R32 U4");
    run<1024, 32, 2, false, 0, 2>(c, 8, "T1024 R32 U2");
    run<1024, 32, 1, false, 0, 2>(c, 8, "T1024 R32 U1");
    run<256, 1, 4, false, 0, 2>(c, 32, "T256 R1 U4");
    run<256, 8, 4, false, 0, 2>(c, 32, "T256 R8 U4");
    run<256, 16, 4, false, 0, 2>(c, 32, "T256 R16 U4");
    run<512, 16, 4, false, 0, 2>(c, 16, "T512 R16 U4");
    run<256, 32, 4, true, 0, 2>(c, 32, "T256 R32 U4 NT");
    run<1024, 32, 4, true, 0, 2>(c, 8, "T1024 R32 U4 NT");
    run<256, 1, 4, false, 1, 2>(c, 32, "T256 logOD U4");
    run<1024, 1, 4, false, 1, 2>(c, 8, "T1024 logOD U4");
    run<256, 32, 4, false, 0, 1>(c, 32, "T256 R32 U4 no-recon");
    run<256, 32, 4, false, 0, 0>(c, 32, "T256 R32 U4 lut+pack only");
    run<1024, 32, 4, false, 0, 0>(c, 8, "T1024 R32 U4 lut+pack only");
    run<256, 1, 4, false, 1, 0>(c, 32, "T256 logOD pack only");
    return 0;
}
