// Diagnostic translation unit, NOT part of libglfusion_hip.so: built into lib/libglfusion_diag.so (include/glfusion_diag.h).
// bench.py loads it for `roofline.power_limited_peak_measured`; nothing of the product path links or loads it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../../include/glfusion_diag.h"

// measurement aid: the MFMA rate the box sustains on random fp16 operands (no memory traffic)
typedef _Float16 probe_f16x8 __attribute__((ext_vector_type(8)));
typedef float probe_f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(512) void probe_mfma_f16_kernel(float* __restrict__ out, int iters, unsigned seed) {
    probe_f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    probe_f16x8 a, b;
    unsigned h = (blockIdx.x * 977u + threadIdx.x) * 2654435761u + seed;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        h = h * 1664525u + 1013904223u; const float u = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
        h = h * 1664525u + 1013904223u; const float v = ((h >> 8) & 0xffff) / 65536.f - 0.5f;
        a[i] = (_Float16)(seed ? u : 1.f); b[i] = (_Float16)(seed ? v : 1.f);
    }
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
    }
    c0 += c1 + c2 + c3;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) sum += c0[i];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = sum;
}
extern "C" int glf_probe_mfma_f16(float* out, int blocks, int iters, uint32_t seed, void* s) {
    if (!out) return -5;                                                              // GLF_ERR_NULL
    if (!(blocks > 0 && blocks <= 65536 && iters > 0 && iters <= (1 << 22))) return -1;  // GLF_ERR_BAD_SHAPE
    hipLaunchKernelGGL(probe_mfma_f16_kernel, dim3(blocks), dim3(512), 0, reinterpret_cast<hipStream_t>(s), out, iters, (unsigned)seed);
    return hipGetLastError() == hipSuccess ? 0 : -4;                                  // GLF_ERR_LAUNCH
}

