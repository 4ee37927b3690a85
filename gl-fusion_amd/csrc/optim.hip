// Fused multi-tensor Adam: torch.optim.Adam as the reference constructs it (main.py:162-165: lr,
// weight_decay as L2-in-gradient, default betas / eps, no amsgrad) over every parameter in ONE launch.
// HBM-bound: 16 B read + 12 B written per element.
#include "glf_common.h"
#include <cmath>

namespace {

// One row per chunk of one parameter: pointers and element count.
struct AdamRow { long long p, g, m, v, n; };

// Same operation order as torch.optim._functional.adam (torch 1.8.1); every operation individually rounded
// (__f*_rn keep hipcc from contracting them into fmas) so the update equals the ATen CPU kernels' bit for bit up to
// the rounding of sqrt / division.
struct AdamK { float b1, b2, omb1, omb2, eps, wd, step_size, sqrt_bc2; };

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, const AdamK k) {
    const float b1 = k.b1, b2 = k.b2, eps = k.eps, wd = k.wd, step_size = k.step_size, sqrt_bc2 = k.sqrt_bc2;
    if (wd != 0.f) g = __fadd_rn(g, __fmul_rn(wd, p));
    m = __fadd_rn(__fmul_rn(m, b1), __fmul_rn(k.omb1, g));                       // mul_(beta1).add_(grad, alpha = 1 - beta1)
    v = __fadd_rn(__fmul_rn(v, b2), __fmul_rn(__fmul_rn(k.omb2, g), g));         // mul_(beta2).addcmul_(grad, grad, value = 1 - beta2)
    const float denom = __fadd_rn(__fdiv_rn(__fsqrt_rn(v), sqrt_bc2), eps);
    p = __fadd_rn(p, __fmul_rn(-step_size, __fdiv_rn(m, denom)));
}

__global__ __launch_bounds__(256) void adam_kernel(const AdamRow* __restrict__ table, int n_rows, const AdamK k) {
    for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
        const AdamRow r = table[row];
        float* __restrict__ p = reinterpret_cast<float*>(r.p);
        const float* __restrict__ g = reinterpret_cast<const float*>(r.g);
        float* __restrict__ m = reinterpret_cast<float*>(r.m);
        float* __restrict__ v = reinterpret_cast<float*>(r.v);
        const int n = (int)r.n;
        const bool vec = (((r.p | r.g | r.m | r.v) & 15) == 0);
        const int n4 = vec ? n >> 2 : 0;
        for (int i = threadIdx.x; i < n4; i += blockDim.x) {
            float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
            const float4 gg = reinterpret_cast<const float4*>(g)[i];
            adam1(pp.x, gg.x, mm.x, vv.x, k);
            adam1(pp.y, gg.y, mm.y, vv.y, k);
            adam1(pp.z, gg.z, mm.z, vv.z, k);
            adam1(pp.w, gg.w, mm.w, vv.w, k);
            reinterpret_cast<float4*>(p)[i] = pp; reinterpret_cast<float4*>(m)[i] = mm; reinterpret_cast<float4*>(v)[i] = vv;
        }
        for (int i = 4 * n4 + threadIdx.x; i < n; i += blockDim.x) {
            float pp = p[i], mm = m[i], vv = v[i];
            adam1(pp, g[i], mm, vv, k);
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
    }
}

}  // namespace

extern "C" int glf_adam_step(const int64_t* table, int n_rows, double lr, double beta1, double beta2, double eps,
                             double weight_decay, int64_t step, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(table != nullptr, GLF_ERR_NULL, "adam_step: null table");
    GLF_REQUIRE(n_rows > 0, GLF_ERR_BAD_SHAPE, "adam_step: n_rows must be > 0");
    GLF_REQUIRE(step >= 1, GLF_ERR_BAD_SHAPE, "adam_step: step counts from 1");
    // scalars exactly as torch derives them: in double on the host, rounded to float once
    AdamK k;
    k.b1 = (float)beta1; k.b2 = (float)beta2; k.omb1 = (float)(1.0 - beta1); k.omb2 = (float)(1.0 - beta2);
    k.eps = (float)eps; k.wd = (float)weight_decay;
    k.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
    k.sqrt_bc2 = (float)sqrt(1.0 - pow(beta2, (double)step));
    GLF_REQUIRE((reinterpret_cast<uintptr_t>(table) & 7u) == 0, GLF_ERR_BAD_SHAPE, "adam_step: table must be 8-byte aligned");
    const int blocks = n_rows < 8192 ? n_rows : 8192;
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, glf::S(s), reinterpret_cast<const AdamRow*>(table), n_rows, k);
    return glf::check_launch("adam_step");
}
