// Achievable MFMA rate on the box: waves per CU x independent accumulators, no memory traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang diagnostic ignored "-Wunused-value"
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, unsigned seed) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    f16x8 a, b; bf16x8 ab, bb;
    unsigned h = (blockIdx.x * 977u + threadIdx.x) * 2654435761u + seed;
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u; const float u = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
        h = h * 1664525u + 1013904223u; const float v = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
        a[i] = (_Float16)(seed ? u : 1.f); b[i] = (_Float16)(seed ? v : 1.f); ab[i] = (__bf16)(seed ? u : 1.f); bb[i] = (__bf16)(seed ? v : 1.f);
    }
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
        } else if (KIND == 1) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
        } else {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f * threadIdx.x, 2.0f, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f * threadIdx.x, 2.0f, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f * threadIdx.x, 2.0f, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(1.0f * threadIdx.x, 2.0f, c3, 0, 0, 0);
        }
    }
    c0 += c1 + c2 + c3;
    float s = 0; for (int i = 0; i < 16; ++i) s += c0[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND> void run(const char* name, double flop_per_mfma, int threads, unsigned seed = 12345u, int iters = 20000) {
    float* out; hipMalloc(&out, 4096 * 512 * 4);
    const int blocks = 256 * 4 * (512 / threads);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, iters, seed);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double fl = (double)blocks * (threads / 64) * iters * 4 * flop_per_mfma;
        printf("%s seed %u iters %d threads/block %d: %.3f ms  %.1f TFLOP/s\n", name, seed, iters, threads, ms, fl / ms / 1e9);
    }
    hipFree(out);
}
int main() {
    run<0>("f16 32x32x16", 32.0 * 32 * 16 * 2, 512, 0u);
    run<0>("f16 32x32x16", 32.0 * 32 * 16 * 2, 512);
    run<0>("f16 32x32x16", 32.0 * 32 * 16 * 2, 512, 12345u, 200000);
    run<1>("bf16 32x32x16", 32.0 * 32 * 16 * 2, 512);
    run<2>("f32 32x32x2", 32.0 * 32 * 2 * 2, 512);
    return 0;
}
