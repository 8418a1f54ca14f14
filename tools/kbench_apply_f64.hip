// What would the OD + reconstruction pass (k_apply, normalizer.py:45-50) cost with binary64 per-pixel arithmetic, as the reference
// computes it?  The same memory pattern (512 persistent workgroups, dwordx3 chunks, next trip in flight, one read and one write per
// pixel), the same arithmetic -- optical density from a table, the two-atom non-negative lasso in closed form (g12 >= 0), the rescale
// by maxC_tgt / maxC_src, 255 * exp(-C . M_tgt), truncating cast -- once in binary32 (what the product does: v_exp_f32, FMAs at full
// rate) and once in binary64 (v_fma_f64 at a quarter of that rate on both pipes, exp() as a library polynomial).  Development aid:
// the answer to "dtype f32 because it passes the tolerance" as a stated trade (bench.py's `arithmetic` key quotes the two times).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
struct Chunk { uint32_t w0, w1, w2; };

template <typename T> struct Consts { T wa1[3], ka1, wa2[3], ka2, r1, r2, q[2][3]; };

template <typename T> __device__ __forceinline__ T my_exp(T x);
template <> __device__ __forceinline__ float my_exp<float>(float x) { return __expf(x); }
template <> __device__ __forceinline__ double my_exp<double>(double x) { return exp(x); }

template <typename T>
__global__ __launch_bounds__(512, 2) void k_apply_t(const uint8_t* src, uint8_t* dst, int nch, int n_tiles, Consts<T> K) {
    __shared__ T s_od[256];
    for (int i = threadIdx.x; i < 256; i += 512) s_od[i] = (T)fmax(-log(fmax((double)i, 1.0) / 255.0), 1e-6);
    __syncthreads();
    const int t = threadIdx.x;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const Chunk* s = reinterpret_cast<const Chunk*>(src + (size_t)tile * nch * 12);
        Chunk* d = reinterpret_cast<Chunk*>(dst + (size_t)tile * nch * 12);
        Chunk cur = s[min(t, nch - 1)], nx = s[min(t + 512, nch - 1)];
        for (int c = t; c < nch; c += 512) {
            const Chunk ch = cur;
            cur = nx;
            nx = s[min(c + 1024, nch - 1)];
            uint32_t b[12];
#pragma unroll
            for (int i = 0; i < 12; ++i) b[i] = ((i < 4 ? ch.w0 : (i < 8 ? ch.w1 : ch.w2)) >> (8 * (i & 3))) & 255u;
            uint32_t o[12];
#pragma unroll
            for (int px = 0; px < 4; ++px) {
                const T x = s_od[b[3 * px]], y = s_od[b[3 * px + 1]], z = s_od[b[3 * px + 2]];
                const T a1 = K.wa1[2] * z + (K.wa1[1] * y + (K.wa1[0] * x + K.ka1));
                const T a2 = K.wa2[2] * z + (K.wa2[1] * y + (K.wa2[0] * x + K.ka2));
                const T c1 = fmax(a1 + K.r1 * fmin(a2, (T)0), (T)0), c2 = fmax(a2 + K.r2 * fmin(a1, (T)0), (T)0);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const T v = (T)255 * my_exp<T>(c1 * K.q[0][k] + c2 * K.q[1][k]);
                    o[3 * px + k] = (uint32_t)v & 255u;
                }
            }
            Chunk r;
            r.w0 = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
            r.w1 = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
            r.w2 = o[8] | (o[9] << 8) | (o[10] << 16) | (o[11] << 24);
            d[c] = r;
        }
    }
}

template <typename T> Consts<T> make_consts() {
    const double M[6] = {0.65, 0.70, 0.29, 0.07, 0.99, 0.11}, Mt[6] = {0.55, 0.75, 0.35, 0.10, 0.95, 0.20};
    const double g11 = M[0] * M[0] + M[1] * M[1] + M[2] * M[2], g22 = M[3] * M[3] + M[4] * M[4] + M[5] * M[5];
    const double g12 = M[0] * M[3] + M[1] * M[4] + M[2] * M[5], det = g11 * g22 - g12 * g12, lam = 0.01;
    Consts<T> K;
    for (int c = 0; c < 3; ++c) {
        K.wa1[c] = (T)((g22 * M[c] - g12 * M[3 + c]) / det);
        K.wa2[c] = (T)((g11 * M[3 + c] - g12 * M[c]) / det);
        K.q[0][c] = (T)(-1.1 * Mt[c]);
        K.q[1][c] = (T)(-0.9 * Mt[3 + c]);
    }
    K.ka1 = (T)(-lam * (g22 - g12) / det);
    K.ka2 = (T)(-lam * (g11 - g12) / det);
    K.r1 = (T)(g12 / g11);
    K.r2 = (T)(g12 / g22);
    return K;
}

int main() {
    const int n = 512, nch = 1024 * 1024 / 4;
    const size_t bytes = (size_t)n * nch * 12;
    uint8_t *src, *dst;
    hipMalloc(&src, bytes);
    hipMalloc(&dst, bytes);
    {   // bytes that make plausible pixels (a repeating ramp with noise-like variation): the table lookups must not all hit one entry
        uint8_t* h = (uint8_t*)malloc(bytes);
        uint32_t x = 12345u;
        for (size_t i = 0; i < bytes; ++i) { x = x * 1664525u + 1013904223u; h[i] = (uint8_t)(60 + ((x >> 24) % 180)); }
        hipMemcpy(src, h, bytes, hipMemcpyHostToDevice);
        free(h);
    }
    auto run = [&](const char* name, auto kern, auto K) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        for (int i = 0; i < 3; ++i) kern<<<512, 512>>>(src, dst, nch, n, K);
        hipEventRecord(e0);
        for (int i = 0; i < 5; ++i) kern<<<512, 512>>>(src, dst, nch, n, K);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        printf("%-44s %.3f ms per 512 tiles of 1024^2  = %.2f TB/s at 6 B/px  (%.1f k tiles/s)\n", name, ms, 2.0 * bytes / ms * 1e-9, n / ms);
        return ms;
    };
    const float f = run("apply arithmetic in binary32", k_apply_t<float>, make_consts<float>());
    const float d = run("apply arithmetic in binary64", k_apply_t<double>, make_consts<double>());
    printf("binary64 / binary32 = %.2f x\n", d / f);
    return 0;
}
