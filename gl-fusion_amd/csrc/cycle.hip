// Temporal cycle-consistency loss: Trainer.seg_cycle / Trainer.dense_seg_cycle of the reference
// (main.py:650-798) on the pooled fusion features feat [T][F] (T = 40 frames, F = 2048 at the shipped settings),
// forward AND gradient in one single-workgroup launch.  The arithmetic is tiny (a few hundred F-long dot products);
// what the fusion saves is the ~60 small ATen kernels and index tensors the reference builds per call.
//
// Frames [0, R) are queries, [R, T) keys.  For a start frame s: D[k][j] = |key_k - query_{s+j}|^2, the chunk
// matches tot[b] = sum_j D[(b+j) % Kn][j] are soft-maxed (beta), W[j] = sum_b beta_b key_{(b+off+j) % Kn} is the
// soft nearest key chunk, QD[m][j] = |query_{off+m} - W[j]|^2 and the logits are z_i = -(temp / (F c)) sum_j
// QD[(i+j) % Rc][j]; loss = mean_i BCE-with-logits(z_i, [i == s]).  Small reductions run in double.
#include "glf_common.h"

namespace {

constexpr int MAXR = 128, MAXC = 8;       // limits of target_region / key frames and of chunk_size

__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void seg_cycle_kernel(const float* __restrict__ feat, int T, int F, int R, int off, int c,
                                                        float temperature, int start0, int n_starts, int stride, float weight,
                                                        int soft, float* __restrict__ loss_out, float* __restrict__ dfeat) {
    extern __shared__ float smem_c[];
    float* W = smem_c;                    // [c][F] weighted key chunk
    float* dW = smem_c + (size_t)c * F;   // [c][F] its gradient
    __shared__ double D[MAXR * MAXC], QD[MAXR * MAXC], gq[MAXR * MAXC], dD[MAXR * MAXC];
    __shared__ double beta[MAXR], dbeta[MAXR], dz[MAXR];
    __shared__ double loss_acc;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Kn = T - R, P = R - (c + off) + 1, NB = Kn - (c + off) + 1, Rc = R - off;
    const float* __restrict__ Qc = feat + (size_t)off * F;
    const float* __restrict__ K = feat + (size_t)R * F;
    const double sc = -(double)temperature / (double)F / (double)c;
    if (dfeat)
        for (long long i = tid; i < (long long)T * F; i += 256) dfeat[i] = 0.f;
    if (tid == 0) loss_acc = 0.0;
    __syncthreads();
    for (int si = 0; si < n_starts; ++si) {
        const int s = start0 + si * stride;
        const float* __restrict__ q = feat + (size_t)s * F;
        // key / query-chunk distances (main.py:665-667)
        for (int pair = wave; pair < Kn * c; pair += 4) {
            const int k = pair / c, j = pair - k * c;
            double a = 0.0;
            for (int f = lane; f < F; f += 64) { const float d = K[(size_t)k * F + f] - q[(size_t)j * F + f]; a += (double)d * d; }
            a = wave_sum_d(a);
            if (lane == 0) D[pair] = a;
        }
        __syncthreads();
        if (tid == 0) {                                                   // chunk matches and their softmax (main.py:670-679)
            double mx = -1e300;
            for (int b = 0; b < NB; ++b) {
                double t = 0.0;
                for (int j = 0; j < c; ++j) t += D[((b + j) % Kn) * c + j];
                beta[b] = sc * t;
                mx = fmax(mx, beta[b]);
            }
            double sum = 0.0;
            for (int b = 0; b < NB; ++b) { beta[b] = exp(beta[b] - mx); sum += beta[b]; }
            for (int b = 0; b < NB; ++b) beta[b] /= sum;
        }
        __syncthreads();
        for (int f = tid; f < F; f += 256)                                // soft nearest key chunk (main.py:684-691)
            for (int j = 0; j < c; ++j) {
                double w = 0.0;
                for (int b = 0; b < NB; ++b) w += beta[b] * (double)K[(size_t)((b + off + j) % Kn) * F + f];
                W[(size_t)j * F + f] = (float)w;
            }
        __syncthreads();
        for (int pair = wave; pair < Rc * c; pair += 4) {                 // back to the query frames (main.py:695-697)
            const int m = pair / c, j = pair - m * c;
            double a = 0.0;
            for (int f = lane; f < F; f += 64) { const float d = Qc[(size_t)m * F + f] - W[(size_t)j * F + f]; a += (double)d * d; }
            a = wave_sum_d(a);
            if (lane == 0) QD[pair] = a;
        }
        __syncthreads();
        if (tid == 0) {                                                   // logits, BCE, d loss / d logits (main.py:699-716)
            for (int i = 0; i < Rc * c; ++i) gq[i] = 0.0;
            double l = 0.0;
            for (int i = 0; i < P; ++i) {
                double t = 0.0;
                for (int j = 0; j < c; ++j) t += QD[((i + j) % Rc) * c + j];
                const double z = sc * t;
                const double y = soft ? (i == s ? 0.8 : 0.2 / (double)(P - 1)) : (i == s ? 1.0 : 0.0);
                l += fmax(z, 0.0) - z * y + log1p(exp(-fabs(z)));
                const double sig = z >= 0.0 ? 1.0 / (1.0 + exp(-z)) : exp(z) / (1.0 + exp(z));
                dz[i] = (sig - y) / (double)P * (double)weight;
                for (int j = 0; j < c; ++j) gq[((i + j) % Rc) * c + j] += sc * dz[i];
            }
            loss_acc += (double)weight * l / (double)P;
        }
        __syncthreads();
        if (!dfeat) continue;
        for (int f = tid; f < F; f += 256)                                // d QD -> d query frames, d W
            for (int j = 0; j < c; ++j) {
                const float wjf = W[(size_t)j * F + f];
                float acc = 0.f;
                for (int m = 0; m < Rc; ++m) {
                    const float g = (float)(2.0 * gq[m * c + j]) * (Qc[(size_t)m * F + f] - wjf);
                    dfeat[(size_t)(off + m) * F + f] += g;
                    acc -= g;
                }
                dW[(size_t)j * F + f] = acc;
            }
        __syncthreads();
        for (int b = wave; b < NB; b += 4) {                              // d beta
            double a = 0.0;
            for (int j = 0; j < c; ++j) {
                const float* __restrict__ kr = K + (size_t)((b + off + j) % Kn) * F;
                for (int f = lane; f < F; f += 64) a += (double)dW[(size_t)j * F + f] * (double)kr[f];
            }
            a = wave_sum_d(a);
            if (lane == 0) dbeta[b] = a;
        }
        __syncthreads();
        if (tid == 0) {                                                   // softmax backward -> d D
            double dot = 0.0;
            for (int b = 0; b < NB; ++b) dot += beta[b] * dbeta[b];
            for (int i = 0; i < Kn * c; ++i) dD[i] = 0.0;
            for (int b = 0; b < NB; ++b) {
                const double dtot = sc * beta[b] * (dbeta[b] - dot);
                for (int j = 0; j < c; ++j) dD[((b + j) % Kn) * c + j] += dtot;
            }
        }
        __syncthreads();
        for (int f = tid; f < F; f += 256)                                // d D, d W -> d key frames, d query chunk
            for (int j = 0; j < c; ++j) {
                const float qjf = q[(size_t)j * F + f];
                const float dwjf = dW[(size_t)j * F + f];
                float dq = 0.f;
                for (int k = 0; k < Kn; ++k) {
                    const float g = (float)(2.0 * dD[k * c + j]) * (K[(size_t)k * F + f] - qjf);
                    dfeat[(size_t)(R + k) * F + f] += g;
                    dq -= g;
                }
                dfeat[(size_t)(s + j) * F + f] += dq;
                for (int b = 0; b < NB; ++b) dfeat[(size_t)(R + (b + off + j) % Kn) * F + f] += (float)beta[b] * dwjf;
            }
        __syncthreads();
    }
    if (tid == 0) *loss_out = (float)loss_acc;
}

// y = x * scale * (*scale_dev)   (the autograd upstream gradient of a scalar loss lives on the device)
__global__ __launch_bounds__(256) void scale_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float scale,
                                                    const float* __restrict__ scale_dev) {
    const float k = scale * (scale_dev ? *scale_dev : 1.f);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = x[i] * k;
}

// out = a * x + b * y
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out,
                                                    float a, float b, long long n4, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        const float4 xv = reinterpret_cast<const float4*>(x)[i], yv = reinterpret_cast<const float4*>(y)[i];
        reinterpret_cast<float4*>(out)[i] = make_float4(a * xv.x + b * yv.x, a * xv.y + b * yv.y, a * xv.z + b * yv.z, a * xv.w + b * yv.w);
    }
    for (long long i = 4 * n4 + blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = a * x[i] + b * y[i];
}

}  // namespace

extern "C" int glf_axpby(const float* x, const float* y, float* out, float a, float b, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y && out, GLF_ERR_NULL, "axpby: null argument");
    GLF_REQUIRE(numel > 0, GLF_ERR_BAD_SHAPE, "axpby: numel must be > 0");
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(out)) & 15u) == 0;
    const long long n4 = vec ? numel / 4 : 0;
    long long blocks = (numel / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)blocks), dim3(256), 0, glf::S(s), x, y, out, a, b, n4, (long long)numel);
    return glf::check_launch("axpby");
}

extern "C" int glf_seg_cycle(const float* feat, int T, int F, int target_region, int cyc_off, int chunk_size, float temperature,
                             int start0, int n_starts, int stride, float weight, int soft_label, float* loss_out, float* dfeat,
                             glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(feat && loss_out, GLF_ERR_NULL, "seg_cycle: null argument");
    const int R = target_region, c = chunk_size, off = cyc_off;
    GLF_REQUIRE(T > 0 && F > 0 && R > 0 && c > 0 && off >= 0, GLF_ERR_BAD_SHAPE, "seg_cycle: bad sizes");
    const int Kn = T - R, P = R - (c + off) + 1, NB = Kn - (c + off) + 1;
    GLF_REQUIRE(Kn > 0 && P > 0 && NB > 0, GLF_ERR_BAD_SHAPE,
                "seg_cycle: needs T > target_region and both regions longer than chunk_size + cyc_off (T=%d, target_region=%d)", T, R);
    GLF_REQUIRE(R <= MAXR && Kn <= MAXR && c <= MAXC, GLF_ERR_UNSUPPORTED, "seg_cycle: target_region / key frames <= %d, chunk_size <= %d", MAXR, MAXC);
    GLF_REQUIRE(n_starts >= 1 && stride >= 1 && start0 >= 0 && start0 + (n_starts - 1) * stride < P, GLF_ERR_BAD_SHAPE,
                "seg_cycle: start frames must lie in [0, %d)", P);
    GLF_REQUIRE(!soft_label || P > 1, GLF_ERR_BAD_SHAPE, "seg_cycle: soft labels need more than one candidate");
    const size_t smem = (size_t)2 * c * F * sizeof(float);
    GLF_REQUIRE(smem <= 96 * 1024, GLF_ERR_UNSUPPORTED, "seg_cycle: chunk_size * F too large for the workgroup's LDS (%zu bytes)", smem);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(seg_cycle_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "hipFuncSetAttribute(seg_cycle_kernel): %s", hipGetErrorString(e));
        attr_set = true;
    }
    hipLaunchKernelGGL(seg_cycle_kernel, dim3(1), dim3(256), smem, glf::S(s), feat, T, F, R, off, c, temperature, start0, n_starts,
                       stride, weight, soft_label, loss_out, dfeat);
    return glf::check_launch("seg_cycle");
}

extern "C" int glf_scale(const float* x, float* y, int64_t numel, float scale, const float* scale_dev, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y, GLF_ERR_NULL, "scale: null argument");
    GLF_REQUIRE(numel > 0, GLF_ERR_BAD_SHAPE, "scale: numel must be > 0");
    long long blocks = (numel + 1023) / 1024;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, glf::S(s), x, y, (long long)numel, scale, scale_dev);
    return glf::check_launch("scale");
}

extern "C" int glf_zero(void* p, int64_t bytes, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(p != nullptr && bytes >= 0, GLF_ERR_NULL, "zero: null pointer / negative size");
    if (bytes == 0) return GLF_OK;
    hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, glf::S(s));
    if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "hipMemsetAsync: %s", hipGetErrorString(e));
    return GLF_OK;
}
