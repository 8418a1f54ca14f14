// LDS gather rates for the row-table layout (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int WIDTH, int OFF, int COPIES, int ESIZE, bool RANDOM>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
    extern __shared__ uint32_t lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    __syncthreads();
    const uint32_t c = (threadIdx.x & (COPIES - 1)) * ESIZE + OFF;
    uint32_t x = threadIdx.x * 2654435761u;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            uint32_t row = RANDOM ? ((x >> (3 * u + 4)) & 255u) : (uint32_t)((it + u) & 255);
            const uint32_t a = row * 256u + c;
            if (WIDTH == 4) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(0)"); acc += v; }
            if (WIDTH == 8) { double v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(0)"); acc += (float)__double_as_longlong(v); }
            if (WIDTH == 16) { float4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); asm volatile("s_waitcnt lgkmcnt(0)"); acc += v.x + v.w; }
        }
        x = x * 1664525u + 1013904223u;
    }
    if (acc == 123.456f) out[0] = acc;
}
template <int WIDTH, int OFF, int COPIES, int ESIZE, bool RANDOM>
static void run(const char* name, float* d) {
    printf("%-44s", name);
    for (int b : {1, 2}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipFuncSetAttribute((const void*)k<WIDTH, OFF, COPIES, ESIZE, RANDOM>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
        k<WIDTH, OFF, COPIES, ESIZE, RANDOM><<<256 * b, 512, 65536>>>(d, 4);
        hipEventRecord(e0);
        k<WIDTH, OFF, COPIES, ESIZE, RANDOM><<<256 * b, 512, 65536>>>(d, 2048);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("  WG/CU %d: %6.3f G reads/s per CU", b, 8.0 * 2048 * 8 * b / (ms * 1e-3) * 1e-9);
    }
    printf("\n");
}
int main() {
    setvbuf(stdout, nullptr, _IONBF, 0);
    float* d; hipMalloc(&d, 1024);
    run<16, 0, 16, 16, false>("b128 16 copies x16B uniform rows", d);
    run<16, 0, 16, 16, true>("b128 16 copies x16B random rows", d);
    run<8, 8, 16, 16, true>("b64 @+8 16 copies x16B random rows", d);
    run<4, 12, 16, 16, true>("b32 @+12 16 copies x16B random rows", d);
    run<8, 0, 32, 8, true>("b64 32 copies x8B random rows", d);
    run<8, 0, 16, 8, true>("b64 16 copies x8B random rows", d);
    run<4, 0, 32, 4, true>("b32 32 copies x4B random rows", d);
    run<4, 0, 64, 4, true>("b32 64 copies x4B random rows", d);
    run<4, 0, 16, 4, true>("b32 16 copies x4B random rows", d);
    run<4, 0, 1, 4, true>("b32 1 copy random rows", d);
    return 0;
}
