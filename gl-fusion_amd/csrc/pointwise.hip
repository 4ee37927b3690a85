// HBM-bound pieces of the GL-Fusion path (gfx950): stem conv (Cin = 1), pooling, local gate,
// view stacking, bilinear up-sampling, loss and metric reductions, weight re-layout.
// All channels-last; 16-byte accesses wherever the channel count allows; 64-wide wavefront
// reductions via __shfl_xor.
#include "glf_common.h"
#include "split_f16.h"

namespace {

inline int stream_grid(long long total, int block) {
    long long g = (total + block - 1) / block;
    const long long cap = (long long)glf::num_cus() * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}
inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---------------------------------------------------------------------------------------
// weight re-layout
// ---------------------------------------------------------------------------------------
__global__ void oihw_to_tap_kernel(const float* __restrict__ w, float* __restrict__ out, long long co_ci, int taps) {
    // out[t][i] = w[i][t], i = co*cin + ci
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < co_ci * taps; o += (long long)gridDim.x * blockDim.x) {
        const long long t = o / co_ci, i = o - t * co_ci;
        out[o] = w[i * taps + t];
    }
}
__global__ void tap_to_oihw_kernel(const float* __restrict__ w, float* __restrict__ out, long long co_ci, int taps) {
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < co_ci * taps; o += (long long)gridDim.x * blockDim.x) {
        const long long i = o / taps, t = o - i * taps;
        out[o] = w[t * co_ci + i];
    }
}

// out[t][ci][co] = w[co][ci][t]  (the dgrad operand of the split-bf16 path: k = co contiguous)
__global__ void oihw_to_tap_t_kernel(const float* __restrict__ w, float* __restrict__ out, int cout, int cin, int taps) {
    const long long total = (long long)cout * cin * taps;
    for (long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x; o < total; o += (long long)gridDim.x * blockDim.x) {
        const int co = (int)(o % cout); long long r = o / cout;
        const int ci = (int)(r % cin); const int t = (int)(r / cin);
        out[o] = w[((long long)co * cin + ci) * taps + t];
    }
}
// batched 2-D transpose through a padded 32x32 LDS tile: dst[b][c][r] = src[b][r][c]
__global__ __launch_bounds__(256) void transpose2d_kernel(const float* __restrict__ src, float* __restrict__ dst, int rows, int cols) {
    __shared__ float tile[32][33];
    const long long boff = (long long)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[boff + (long long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[boff + (long long)c * rows + r] = tile[tx][i];
    }
}

// the same with strides: src[b][r][c] at src + b * bs_src + r * ld_src + c; dst[b][c][r] at dst + b * bs_dst + c * ld_dst + r for
// r < rows_pad, ZERO for rows <= r < rows_pad (a reduction dimension padded to the contraction kernels' K granule)
__global__ __launch_bounds__(256) void transpose2d_strided_kernel(const float* __restrict__ src, long long ld_src, long long bs_src,
                                                                  float* __restrict__ dst, long long ld_dst, long long bs_dst, int rows, int cols,
                                                                  int rows_pad) {
    __shared__ float tile[32][33];
    const float* sb = src + (long long)blockIdx.z * bs_src;
    float* db = dst + (long long)blockIdx.z * bs_dst;
    const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? sb[(long long)r * ld_src + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows_pad) db[(long long)c * ld_dst + r] = tile[tx][i];
    }
}

// ---------------------------------------------------------------------------------------
// stem 7x7, Cin = 1.  One workgroup = 16x16 output pixels; the 22x22 input patch and the
// 49 x Cout weights sit in LDS; each thread owns one pixel x 16 output channels per pass.
// ---------------------------------------------------------------------------------------
constexpr int ST = 16, SP = ST + 6;
// 16-bit storage (glf_s16_stem7x7_*): the conv output / its gradient are bf16; arithmetic stays fp32
__device__ __forceinline__ unsigned stem_pack2(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
template <int COUT, bool OUT16>
__global__ __launch_bounds__(256) void stem_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                       void* __restrict__ yv, int h, int wd, int ho, int wo, int pad) {
    __shared__ float patch[SP * SP];
    __shared__ __attribute__((aligned(16))) float wt[49 * COUT];       // [tap][co]
    __shared__ __attribute__((aligned(16))) float bs[COUT];
    const int tid = threadIdx.x;
    const int n = blockIdx.z, oy0 = blockIdx.y * ST, ox0 = blockIdx.x * ST;
    for (int i = tid; i < 49 * COUT; i += 256) { const int co = i / 49, t = i - co * 49; wt[t * COUT + co] = w[i]; }
    for (int i = tid; i < COUT; i += 256) bs[i] = bias ? bias[i] : 0.f;
    for (int i = tid; i < SP * SP; i += 256) {
        const int py = i / SP, px = i - py * SP;
        const int iy = oy0 - pad + py, ix = ox0 - pad + px;
        patch[i] = (iy >= 0 && iy < h && ix >= 0 && ix < wd) ? x[((long long)n * h + iy) * wd + ix] : 0.f;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    const int oy = oy0 + ty, ox = ox0 + tx;
    float xin[49];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) xin[ky * 7 + kx] = patch[(ty + ky) * SP + tx + kx];
    if (oy >= ho || ox >= wo) return;
    float* dst = static_cast<float*>(yv) + (((long long)n * ho + oy) * wo + ox) * COUT;
    unsigned short* dst16 = static_cast<unsigned short*>(yv) + (((long long)n * ho + oy) * wo + ox) * COUT;
#pragma unroll 1
    for (int c0 = 0; c0 < COUT; c0 += 16) {
        float4 a0 = *reinterpret_cast<const float4*>(bs + c0), a1 = *reinterpret_cast<const float4*>(bs + c0 + 4);
        float4 a2 = *reinterpret_cast<const float4*>(bs + c0 + 8), a3 = *reinterpret_cast<const float4*>(bs + c0 + 12);
#pragma unroll
        for (int t = 0; t < 49; ++t) {
            const float v = xin[t];
            const float4 w0 = *reinterpret_cast<const float4*>(wt + t * COUT + c0);
            const float4 w1 = *reinterpret_cast<const float4*>(wt + t * COUT + c0 + 4);
            const float4 w2 = *reinterpret_cast<const float4*>(wt + t * COUT + c0 + 8);
            const float4 w3 = *reinterpret_cast<const float4*>(wt + t * COUT + c0 + 12);
            a0.x = fmaf(v, w0.x, a0.x); a0.y = fmaf(v, w0.y, a0.y); a0.z = fmaf(v, w0.z, a0.z); a0.w = fmaf(v, w0.w, a0.w);
            a1.x = fmaf(v, w1.x, a1.x); a1.y = fmaf(v, w1.y, a1.y); a1.z = fmaf(v, w1.z, a1.z); a1.w = fmaf(v, w1.w, a1.w);
            a2.x = fmaf(v, w2.x, a2.x); a2.y = fmaf(v, w2.y, a2.y); a2.z = fmaf(v, w2.z, a2.z); a2.w = fmaf(v, w2.w, a2.w);
            a3.x = fmaf(v, w3.x, a3.x); a3.y = fmaf(v, w3.y, a3.y); a3.z = fmaf(v, w3.z, a3.z); a3.w = fmaf(v, w3.w, a3.w);
        }
        if (OUT16) {
            *reinterpret_cast<uint4*>(dst16 + c0) = make_uint4(stem_pack2(a0.x, a0.y), stem_pack2(a0.z, a0.w), stem_pack2(a1.x, a1.y), stem_pack2(a1.z, a1.w));
            *reinterpret_cast<uint4*>(dst16 + c0 + 8) = make_uint4(stem_pack2(a2.x, a2.y), stem_pack2(a2.z, a2.w), stem_pack2(a3.x, a3.y), stem_pack2(a3.z, a3.w));
        } else {
            *reinterpret_cast<float4*>(dst + c0) = a0; *reinterpret_cast<float4*>(dst + c0 + 4) = a1;
            *reinterpret_cast<float4*>(dst + c0 + 8) = a2; *reinterpret_cast<float4*>(dst + c0 + 12) = a3;
        }
    }
}

// stem wgrad: dW[co][tap] = sum_pixels dy[p][co] * x[p + tap]; db[co] = sum dy.
// One workgroup = one 16x16 output tile of one image; thread = (co = tid & 63, tap group = tid >> 6).
// partial[block][50][64] (49 taps + bias row), folded by stem_wgrad_finalize.
constexpr int STEM_CO = 64;
template <bool DY16>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ x, const void* __restrict__ dyv, float* __restrict__ partial,
                                                         int h, int wd, int ho, int wo, int pad) {
    const float* __restrict__ dy = static_cast<const float*>(dyv);
    const unsigned short* __restrict__ dy16 = static_cast<const unsigned short*>(dyv);
    __shared__ float patch[SP * SP];
    const int tid = threadIdx.x;
    const int n = blockIdx.z, oy0 = blockIdx.y * ST, ox0 = blockIdx.x * ST;
    for (int i = tid; i < SP * SP; i += 256) {
        const int py = i / SP, px = i - py * SP;
        const int iy = oy0 - pad + py, ix = ox0 - pad + px;
        patch[i] = (iy >= 0 && iy < h && ix >= 0 && ix < wd) ? x[((long long)n * h + iy) * wd + ix] : 0.f;
    }
    __syncthreads();
    const int co = tid & 63, tg = tid >> 6;          // taps tg, tg+4, ... (13 for tg 0, 12 otherwise)
    float acc[13];
#pragma unroll
    for (int i = 0; i < 13; ++i) acc[i] = 0.f;
    float accb = 0.f;
    const int ny = min(ST, ho - oy0), nx = min(ST, wo - ox0);
    for (int py = 0; py < ny; ++py)
        for (int px = 0; px < nx; ++px) {
            const long long go = (((long long)n * ho + oy0 + py) * wo + ox0 + px) * STEM_CO + co;
            const float g = DY16 ? __uint_as_float((unsigned)dy16[go] << 16) : dy[go];
            accb += g;
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                const int t = tg + 4 * i;
                if (t < 49) { const int ky = t / 7, kx = t - ky * 7; acc[i] = fmaf(g, patch[(py + ky) * SP + px + kx], acc[i]); }
            }
        }
    const long long blk = ((long long)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    float* dst = partial + blk * 50 * STEM_CO;
#pragma unroll
    for (int i = 0; i < 13; ++i) { const int t = tg + 4 * i; if (t < 49) dst[t * STEM_CO + co] = acc[i]; }
    if (tg == 0) dst[49 * STEM_CO + co] = accb;
}
// one workgroup per tap row t (64 channels x 16 block-lanes, fixed summation order => bitwise reproducible)
__global__ __launch_bounds__(1024) void stem_wgrad_finalize(const float* __restrict__ partial, long long nblk, float* __restrict__ dw, float* __restrict__ db) {
    __shared__ double sh[16][STEM_CO];
    const int t = blockIdx.x, co = threadIdx.x & 63, ln = threadIdx.x >> 6;
    double s = 0;
    for (long long b = ln; b < nblk; b += 16) s += partial[(b * 50 + t) * STEM_CO + co];
    sh[ln][co] = s;
    __syncthreads();
    if (ln == 0) {
        for (int i = 1; i < 16; ++i) s += sh[i][co];
        if (t < 49) dw[co * 49 + t] = (float)s; else if (db) db[co] = (float)s;
    }
}

// ---------------------------------------------------------------------------------------
// maxpool 3x3 stride 2 pad 1
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, uint8_t* __restrict__ idx,
                                                          int n, int h, int w, int c4, int ho, int wo) {
    const long long total = (long long)n * ho * wo * c4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4); long long p = i / c4;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho); const int nn = (int)(p / ho);
        float4 best = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
        int b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        bool first = true;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
                if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (((long long)nn * h + iy) * w + ix) * (c4 * 4) + cc * 4);
                const int t = ky * 3 + kx;
                // ATen: take the first in-range element, then strictly-greater (or NaN) replaces
                if (first || v.x > best.x || v.x != v.x) { best.x = v.x; b0 = t; }
                if (first || v.y > best.y || v.y != v.y) { best.y = v.y; b1 = t; }
                if (first || v.z > best.z || v.z != v.z) { best.z = v.z; b2 = t; }
                if (first || v.w > best.w || v.w != v.w) { best.w = v.w; b3 = t; }
                first = false;
            }
        *reinterpret_cast<float4*>(y + i * 4) = best;
        *reinterpret_cast<uchar4*>(idx + i * 4) = make_uchar4((unsigned char)b0, (unsigned char)b1, (unsigned char)b2, (unsigned char)b3);
    }
}
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ idx, float* __restrict__ dx,
                                                          int n, int h, int w, int c4, int ho, int wo) {
    const long long total = (long long)n * h * w * c4;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4); long long p = i / c4;
        const int ix = (int)(p % w); p /= w;
        const int iy = (int)(p % h); const int nn = (int)(p / h);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        // windows (oy,ox) that contain (iy,ix): iy = 2*oy - 1 + ky
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ny = iy + 1 - ky;
            if (ny < 0 || (ny & 1)) continue;
            const int oy = ny >> 1;
            if (oy >= ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int nx = ix + 1 - kx;
                if (nx < 0 || (nx & 1)) continue;
                const int ox = nx >> 1;
                if (ox >= wo) continue;
                const long long o = ((((long long)nn * ho + oy) * wo + ox) * c4 + cc) * 4;
                const uchar4 t = *reinterpret_cast<const uchar4*>(idx + o);
                const float4 d = *reinterpret_cast<const float4*>(dy + o);
                const int me = ky * 3 + kx;
                if (t.x == me) g.x += d.x;
                if (t.y == me) g.y += d.y;
                if (t.z == me) g.z += d.z;
                if (t.w == me) g.w += d.w;
            }
        }
        *reinterpret_cast<float4*>(dx + i * 4) = g;
    }
}

// ---------------------------------------------------------------------------------------
// per-frame row sums / broadcasts (ASPP pooling branch)
// ---------------------------------------------------------------------------------------
// y[n][c] = scale * sum_p x[n][p][c]   (row stride ld)
__global__ __launch_bounds__(256) void sum_rows_kernel(const float* __restrict__ x, int ld, float* __restrict__ y, float scale, int p, int c) {
    __shared__ float sh[256];
    const int n = blockIdx.y;
    const int cpb = 64;                                   // channels per block
    const int c0 = blockIdx.x * cpb + (threadIdx.x & 63);
    const int pl = threadIdx.x >> 6;                      // 4 row lanes
    float s = 0.f;
    if (c0 < c)
        for (int r = pl; r < p; r += 4) s += x[((long long)n * p + r) * ld + c0];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (pl == 0 && c0 < c) y[(long long)n * c + c0] = scale * (sh[threadIdx.x] + sh[threadIdx.x + 64] + sh[threadIdx.x + 128] + sh[threadIdx.x + 192]);
}
// 16-byte form: 16 channel quads x 16 row lanes per workgroup, four rows' loads in flight per thread (the scalar form above walks
// 196 dependent 4-byte loads per thread: 2.4 TB/s on the ASPP pooled branch's 411 MB maps)
__global__ __launch_bounds__(256) void sum_rows4_kernel(const float* __restrict__ x, int ld, float* __restrict__ y, float scale, int p, int c4) {
    // sums in double: the ASPP pooled branch feeds these N per-frame averages to a BatchNorm over N samples, which amplifies their
    // rounding by |mean| / spread -- the kernel is memory-bound, the wider adds are free
    __shared__ double sh[4][256];
    const int n = blockIdx.y;
    const int cq = blockIdx.x * 16 + (threadIdx.x & 15);
    const int pl = threadIdx.x >> 4;                      // 16 row lanes
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    if (cq < c4) {
        const float* base = x + (long long)n * p * ld + 4 * cq;
        int r = pl;
        for (; r + 48 < p; r += 64) {
            const float4 a = *reinterpret_cast<const float4*>(base + (long long)r * ld);
            const float4 b = *reinterpret_cast<const float4*>(base + (long long)(r + 16) * ld);
            const float4 d = *reinterpret_cast<const float4*>(base + (long long)(r + 32) * ld);
            const float4 e = *reinterpret_cast<const float4*>(base + (long long)(r + 48) * ld);
            s0 += ((double)a.x + (double)b.x) + ((double)d.x + (double)e.x);
            s1 += ((double)a.y + (double)b.y) + ((double)d.y + (double)e.y);
            s2 += ((double)a.z + (double)b.z) + ((double)d.z + (double)e.z);
            s3 += ((double)a.w + (double)b.w) + ((double)d.w + (double)e.w);
        }
        for (; r < p; r += 16) {
            const float4 a = *reinterpret_cast<const float4*>(base + (long long)r * ld);
            s0 += a.x; s1 += a.y; s2 += a.z; s3 += a.w;
        }
    }
    sh[0][threadIdx.x] = s0; sh[1][threadIdx.x] = s1; sh[2][threadIdx.x] = s2; sh[3][threadIdx.x] = s3;
    __syncthreads();
    if (pl == 0 && cq < c4) {
#pragma unroll
        for (int q = 1; q < 16; ++q) { s0 += sh[0][q * 16 + threadIdx.x]; s1 += sh[1][q * 16 + threadIdx.x]; s2 += sh[2][q * 16 + threadIdx.x]; s3 += sh[3][q * 16 + threadIdx.x]; }
        const double sc = (double)scale;
        *reinterpret_cast<float4*>(y + (long long)n * 4 * c4 + 4 * cq) = make_float4((float)(sc * s0), (float)(sc * s1), (float)(sc * s2), (float)(sc * s3));
    }
}
__global__ __launch_bounds__(256) void bcast_rows4_kernel(const float* __restrict__ x, float* __restrict__ y, int ld, float scale, int p, int c4, long long total4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c4); const long long row = i / c4;
        const long long n = row / p;
        const float4 v = *reinterpret_cast<const float4*>(x + (n * c4 + cc) * 4);
        *reinterpret_cast<float4*>(y + row * ld + 4 * cc) = make_float4(scale * v.x, scale * v.y, scale * v.z, scale * v.w);
    }
}
__global__ __launch_bounds__(256) void bcast_rows_kernel(const float* __restrict__ x, float* __restrict__ y, int ld, float scale, int p, int c, long long total) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c); const long long row = i / c;
        const long long n = row / p;
        y[row * ld + cc] = scale * x[n * c + cc];
    }
}

// ---------------------------------------------------------------------------------------
// dropout (counter-based; same mask recomputed in backward from the seed)
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 40);                           // 24 bits
}
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float p, float scale,
                                                      unsigned long long seed, const unsigned long long* __restrict__ step) {
    // `step`: a device counter the caller advances once per training step.  A launch recorded in a hipGraph replays with the
    // SAME `seed` argument every time; the counter is what gives every replay its own mask (forward and backward of one step
    // read the same value).
    if (step) seed += *step * 0xD1B54A32D192ED03ull;
    const unsigned thr = (unsigned)(p * 16777216.0f);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        y[i] = (mix64(seed * 0x100000001B3ull + (unsigned long long)i) >= thr) ? x[i] * scale : 0.f;
}

__global__ __launch_bounds__(256) void relu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) y[i] = fmaxf(x[i], 0.f);
}
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y, float* __restrict__ dx, long long n) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) dx[i] = y[i] > 0.f ? dy[i] : 0.f;
}

// ---------------------------------------------------------------------------------------
// local gate
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gate_fwd_kernel(const float* __restrict__ cls, int ncls, const float* __restrict__ ctr,
                                                       const float* __restrict__ f, float* __restrict__ y, float* __restrict__ a_out,
                                                       int* __restrict__ amax, float weight, int rows, int c) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // max over classes of sigmoid(cls): scan order, strictly greater replaces (first maximum wins)
    float best = sigmoidf_(cls[(long long)row * ncls]);
    int bi = 0;
    for (int k = 1; k < ncls; ++k) { const float s = sigmoidf_(cls[(long long)row * ncls + k]); if (s > best) { best = s; bi = k; } }
    const float cc = sigmoidf_(ctr[row]);
    const float a = sigmoidf_(weight * best * cc);
    if (lane == 0) { a_out[row] = a; amax[row] = bi; }
    const float4* src = reinterpret_cast<const float4*>(f + (long long)row * c);
    float4* dst = reinterpret_cast<float4*>(y + (long long)row * c);
    for (int i = lane; i < (c >> 2); i += 64) { float4 v = src[i]; v.x *= a; v.y *= a; v.z *= a; v.w *= a; dst[i] = v; }
}
__global__ __launch_bounds__(256) void gate_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ f, const float* __restrict__ cls, int ncls,
                                                       const float* __restrict__ ctr, const float* __restrict__ a_in, const int* __restrict__ amax,
                                                       float weight, float* __restrict__ df, float* __restrict__ dcls, float* __restrict__ dctr,
                                                       int rows, int c) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float a = a_in[row];
    const float4* g4 = reinterpret_cast<const float4*>(dy + (long long)row * c);
    const float4* f4 = reinterpret_cast<const float4*>(f + (long long)row * c);
    float4* d4 = reinterpret_cast<float4*>(df + (long long)row * c);
    float s = 0.f;
    for (int i = lane; i < (c >> 2); i += 64) {
        const float4 g = g4[i], v = f4[i];
        s += (g.x * v.x + g.y * v.y) + (g.z * v.z + g.w * v.w);
        d4[i] = make_float4(g.x * a, g.y * a, g.z * a, g.w * a);
    }
    const float da = wave_sum(s);
    if (lane == 0) {
        const int bi = amax[row];
        const float m = sigmoidf_(cls[(long long)row * ncls + bi]);
        const float cc = sigmoidf_(ctr[row]);
        const float dt = da * a * (1.f - a) * weight;            // d/d(m*cc)
        for (int k = 0; k < ncls; ++k) dcls[(long long)row * ncls + k] = (k == bi) ? dt * cc * m * (1.f - m) : 0.f;
        dctr[row] = dt * m * cc * (1.f - cc);
    }
}

// ---------------------------------------------------------------------------------------
// frame-strided copies / adds
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_frames_kernel(const float4* __restrict__ src, long long sfs, float4* __restrict__ dst, long long dfs,
                                                          long long inner4, long long total4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / inner4, r = i - n * inner4;
        dst[n * dfs + r] = src[n * sfs + r];
    }
}
// the same copy that also writes the packed pre-split fp16 image of what it copies (scaled by *amax, an upper bound of the
// sources' maxima known BEFORE the copy): the stacked fusion-block input is read by contractions only through that image
__global__ __launch_bounds__(256) void copy_frames_split_kernel(const float4* __restrict__ src, long long sfs, float4* __restrict__ dst,
                                                                float4* __restrict__ dst_pk, long long dfs, long long inner4, long long total4,
                                                                const float* __restrict__ amax) {
    float sc, inv;
    pow2_scale(amax, sc, inv);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / inner4, r = i - n * inner4;
        const float4 v = src[n * sfs + r];
        dst[n * dfs + r] = v;
        const SplitH sp = split4h(v, sc);
        const float2 h = __builtin_bit_cast(float2, sp.h), l = __builtin_bit_cast(float2, sp.l);
        dst_pk[n * dfs + r] = make_float4(h.x, h.y, l.x, l.y);
    }
}
__global__ __launch_bounds__(256) void add_frames_kernel(const float4* __restrict__ a, long long afs, const float4* __restrict__ b, long long bfs,
                                                         float4* __restrict__ dst, long long dfs, long long inner4, long long total4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / inner4, r = i - n * inner4;
        const float4 u = a[n * afs + r], v = b[n * bfs + r];
        dst[n * dfs + r] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
    }
}

// n-ary sum (gradient fan-in of a tensor consumed by several branches): one pass, k reads + 1 write
struct AddNPtrs { const float4* p[8]; };
__global__ __launch_bounds__(256) void add_n_kernel(AddNPtrs in, int k, float4* __restrict__ out, long long n4) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
        float4 a = in.p[0][i];
        for (int j = 1; j < k; ++j) { const float4 b = in.p[j][i]; a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }
        out[i] = a;
    }
}

// ---------------------------------------------------------------------------------------
// row softmax (embedded mode), one workgroup per row
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float block_reduce(float v, bool is_max, float* sh) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const float t = __shfl_xor(v, o, 64); v = is_max ? fmaxf(v, t) : v + t; }
    const int wv = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[wv] = v;
    __syncthreads();
    float r = sh[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, sh[i]) : r + sh[i];
    return r;
}
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* __restrict__ x, int cols, int ld) {
    __shared__ float sh[4];
    float* row = x + (long long)blockIdx.x * ld;
    float m = -INFINITY;
    for (int i = threadIdx.x; i < cols; i += 256) m = fmaxf(m, row[i]);
    m = block_reduce(m, true, sh);
    float s = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) { const float e = expf(row[i] - m); row[i] = e; s += e; }
    s = block_reduce(s, false, sh);
    const float inv = 1.0f / s;
    for (int i = threadIdx.x; i < cols; i += 256) row[i] *= inv;
    for (int i = cols + threadIdx.x; i < ld; i += 256) row[i] = 0.f;          // padding columns of a row stride > cols read as zero probabilities
}
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float* __restrict__ p, float* __restrict__ dp, int cols, int ld) {
    __shared__ float sh[4];
    const float* pr = p + (long long)blockIdx.x * ld;
    float* dr = dp + (long long)blockIdx.x * ld;
    float s = 0.f;
    for (int i = threadIdx.x; i < cols; i += 256) s += pr[i] * dr[i];
    s = block_reduce(s, false, sh);
    for (int i = threadIdx.x; i < cols; i += 256) dr[i] = pr[i] * (dr[i] - s);
    for (int i = cols + threadIdx.x; i < ld; i += 256) dr[i] = 0.f;
}

// ---------------------------------------------------------------------------------------
// bilinear (align_corners = False), channels-last in -> NCHW out, and the adjoint
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void src_index(int o, float scale, int in_size, int& i0, int& i1, float& l1) {
    float s = ((float)o + 0.5f) * scale - 0.5f;
    if (s < 0.f) s = 0.f;
    i0 = (int)s;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
    l1 = s - (float)i0;
}
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int n, int h, int w, int c,
                                                           int ho, int wo, float sy, float sx) {
    const long long total = (long long)n * c * ho * wo;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int ox = (int)(i % wo); long long t = i / wo;
        const int oy = (int)(t % ho); t /= ho;
        const int cc = (int)(t % c); const int nn = (int)(t / c);
        int y0, y1, x0, x1; float ly, lx;
        src_index(oy, sy, h, y0, y1, ly);
        src_index(ox, sx, w, x0, x1, lx);
        const float* b = x + (long long)nn * h * w * c + cc;
        const float v00 = b[((long long)y0 * w + x0) * c], v01 = b[((long long)y0 * w + x1) * c];
        const float v10 = b[((long long)y1 * w + x0) * c], v11 = b[((long long)y1 * w + x1) * c];
        y[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
    }
}
// gather form of the adjoint: input pixel (iy,ix) collects from every output whose y0/y1 (x0/x1) hits it
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int n, int h, int w, int c,
                                                           int ho, int wo, float sy, float sx, int ry, int rx) {
    const long long total = (long long)n * h * w * c;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c); long long t = i / c;
        const int ix = (int)(t % w); t /= w;
        const int iy = (int)(t % h); const int nn = (int)(t / h);
        const float* g = dy + ((long long)nn * c + cc) * ho * wo;
        // candidate outputs: o with src in (i-1, i+1)  =>  o in ((i-1+0.5)/scale - 0.5, (i+1+0.5)/scale - 0.5)
        const int oy_lo = max(0, (int)floorf(((float)iy - 0.5f) / sy - 0.5f) - 1), oy_hi = min(ho - 1, oy_lo + ry);
        const int ox_lo = max(0, (int)floorf(((float)ix - 0.5f) / sx - 0.5f) - 1), ox_hi = min(wo - 1, ox_lo + rx);
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            int y0, y1; float ly;
            src_index(oy, sy, h, y0, y1, ly);
            float wy = 0.f;
            if (y0 == iy) wy += 1.f - ly;
            if (y1 == iy) wy += ly;
            if (wy == 0.f) continue;
            float rowacc = 0.f;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                int x0, x1; float lx;
                src_index(ox, sx, w, x0, x1, lx);
                float wx = 0.f;
                if (x0 == ix) wx += 1.f - lx;
                if (x1 == ix) wx += lx;
                if (wx != 0.f) rowacc += wx * g[(long long)oy * wo + ox];
            }
            acc += wy * rowacc;
        }
        dx[i] = acc;
    }
}

// ---------------------------------------------------------------------------------------
// BCE-with-logits (sum) and overlap counts
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bce_kernel(const float* __restrict__ x, const float* __restrict__ t, double* __restrict__ loss,
                                                  float* __restrict__ dx, float gscale, const float* __restrict__ gsd, long long n) {
    __shared__ double sh[4];
    double acc = 0;
    if (gsd) gscale *= *gsd;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float xv = x[i], tv = t[i];
        // max(x,0) - x*t + log1p(exp(-|x|))  (the stable form ATen uses)
        acc += (double)(fmaxf(xv, 0.f) - xv * tv + log1pf(expf(-fabsf(xv))));
        if (dx) dx[i] = (sigmoidf_(xv) - tv) * gscale;
    }
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(loss, sh[0] + sh[1] + sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void overlap_kernel(const float* __restrict__ x, const float* __restrict__ t, unsigned long long* __restrict__ counts, long long n) {
    unsigned long long tp = 0, fp = 0, fn = 0, tn = 0;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const bool pr = sigmoidf_(x[i]) > 0.5f;        // main.py:250,385
        const bool gt = t[i] != 0.f;
        tp += pr && gt; fp += pr && !gt; fn += !pr && gt; tn += !pr && !gt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tp += __shfl_xor(tp, o, 64); fp += __shfl_xor(fp, o, 64); fn += __shfl_xor(fn, o, 64); tn += __shfl_xor(tn, o, 64);
    }
    if ((threadIdx.x & 63) == 0) { atomicAdd(counts + 0, tp); atomicAdd(counts + 1, fp); atomicAdd(counts + 2, fn); atomicAdd(counts + 3, tn); }
}

}  // namespace

// =========================================================================================
// C ABI
// =========================================================================================
extern "C" int glf_oihw_to_tap_major(const float* w, float* out, int cout, int cin, int taps, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(w && out, GLF_ERR_NULL, "oihw_to_tap_major: null argument");
    GLF_REQUIRE(cout > 0 && cin > 0 && taps > 0, GLF_ERR_BAD_SHAPE, "oihw_to_tap_major: bad shape");
    const long long cc = (long long)cout * cin;
    hipLaunchKernelGGL(oihw_to_tap_kernel, dim3(stream_grid(cc * taps, 256)), dim3(256), 0, glf::S(s), w, out, cc, taps);
    return glf::check_launch("oihw_to_tap_major");
}
extern "C" int glf_tap_major_to_oihw(const float* w, float* out, int cout, int cin, int taps, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(w && out, GLF_ERR_NULL, "tap_major_to_oihw: null argument");
    GLF_REQUIRE(cout > 0 && cin > 0 && taps > 0, GLF_ERR_BAD_SHAPE, "tap_major_to_oihw: bad shape");
    const long long cc = (long long)cout * cin;
    hipLaunchKernelGGL(tap_to_oihw_kernel, dim3(stream_grid(cc * taps, 256)), dim3(256), 0, glf::S(s), w, out, cc, taps);
    return glf::check_launch("tap_major_to_oihw");
}

extern "C" int glf_oihw_to_tap_major_t(const float* w, float* out, int cout, int cin, int taps, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(w && out, GLF_ERR_NULL, "oihw_to_tap_major_t: null argument");
    GLF_REQUIRE(cout > 0 && cin > 0 && taps > 0, GLF_ERR_BAD_SHAPE, "oihw_to_tap_major_t: bad shape");
    hipLaunchKernelGGL(oihw_to_tap_t_kernel, dim3(stream_grid((long long)cout * cin * taps, 256)), dim3(256), 0, glf::S(s), w, out, cout, cin, taps);
    return glf::check_launch("oihw_to_tap_major_t");
}
extern "C" int glf_transpose2d(const float* src, float* dst, int rows, int cols, int batch, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(src && dst, GLF_ERR_NULL, "transpose2d: null argument");
    GLF_REQUIRE(rows > 0 && cols > 0 && batch > 0 && batch <= 65535 && (rows + 31) / 32 <= 65535, GLF_ERR_BAD_SHAPE, "transpose2d: bad shape");
    hipLaunchKernelGGL(transpose2d_kernel, dim3((cols + 31) / 32, (rows + 31) / 32, batch), dim3(256), 0, glf::S(s), src, dst, rows, cols);
    return glf::check_launch("transpose2d");
}

extern "C" int glf_transpose2d_strided(const float* src, int64_t ld_src, int64_t batch_stride_src, float* dst, int64_t ld_dst,
                                       int64_t batch_stride_dst, int rows, int cols, int rows_pad, int batch, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(src && dst, GLF_ERR_NULL, "transpose2d_strided: null argument");
    GLF_REQUIRE(rows > 0 && cols > 0 && batch > 0 && batch <= 65535 && rows_pad >= rows && (rows_pad + 31) / 32 <= 65535 && ld_src >= cols &&
                ld_dst >= rows_pad, GLF_ERR_BAD_SHAPE, "transpose2d_strided: bad shape");
    hipLaunchKernelGGL(transpose2d_strided_kernel, dim3((cols + 31) / 32, (rows_pad + 31) / 32, batch), dim3(256), 0, glf::S(s), src, (long long)ld_src,
                       (long long)batch_stride_src, dst, (long long)ld_dst, (long long)batch_stride_dst, rows, cols, rows_pad);
    return glf::check_launch("transpose2d_strided");
}

extern "C" int glf_stem7x7_fwd(const float* x, const float* w, const float* bias, float* y,
                               int n, int h, int wdt, int cout, int pad, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && w && y, GLF_ERR_NULL, "stem7x7_fwd: null argument");
    GLF_REQUIRE(cout == STEM_CO, GLF_ERR_UNSUPPORTED, "stem7x7: Cout must be 64 (got %d)", cout);
    const int ho = h + 2 * pad - 6, wo = wdt + 2 * pad - 6;
    GLF_REQUIRE(n > 0 && n <= 65535 && ho > 0 && wo > 0 && pad >= 0 && pad <= 3, GLF_ERR_BAD_SHAPE, "stem7x7_fwd: bad shape");
    GLF_REQUIRE(al16(y), GLF_ERR_BAD_SHAPE, "stem7x7_fwd: y must be 16-byte aligned");
    dim3 grid((wo + ST - 1) / ST, (ho + ST - 1) / ST, n);
    hipLaunchKernelGGL((stem_fwd_kernel<STEM_CO, false>), grid, dim3(256), 0, glf::S(s), x, w, bias, y, h, wdt, ho, wo, pad);
    return glf::check_launch("stem7x7_fwd");
}
extern "C" int glf_s16_stem7x7_fwd(const float* x, const float* w, const float* bias, void* y,
                                   int n, int h, int wdt, int cout, int pad, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && w && y, GLF_ERR_NULL, "s16_stem7x7_fwd: null argument");
    GLF_REQUIRE(cout == STEM_CO, GLF_ERR_UNSUPPORTED, "stem7x7: Cout must be 64 (got %d)", cout);
    const int ho = h + 2 * pad - 6, wo = wdt + 2 * pad - 6;
    GLF_REQUIRE(n > 0 && n <= 65535 && ho > 0 && wo > 0 && pad >= 0 && pad <= 3, GLF_ERR_BAD_SHAPE, "s16_stem7x7_fwd: bad shape");
    GLF_REQUIRE(al16(y), GLF_ERR_BAD_SHAPE, "s16_stem7x7_fwd: y must be 16-byte aligned");
    dim3 grid((wo + ST - 1) / ST, (ho + ST - 1) / ST, n);
    hipLaunchKernelGGL((stem_fwd_kernel<STEM_CO, true>), grid, dim3(256), 0, glf::S(s), x, w, bias, y, h, wdt, ho, wo, pad);
    return glf::check_launch("s16_stem7x7_fwd");
}
// Inference-mode stem in ONE kernel: conv7x7 (Cin = 1) + bias -> BatchNorm (given mean / invstd: the running statistics) -> ReLU ->
// max-pool 3x3 stride 2 pad 1.  One workgroup = a 16 x 16 tile of conv outputs starting one row / column before an even
// coordinate = the 3x3 windows of 7 x 7 pooled outputs (15 of its 16 rows / columns are used).  The conv tile is evaluated like
// stem_fwd_kernel's (same patch, same tap order: bit-identical values), 16 channels at a time, normalised, parked in LDS and
// pooled from there: the [N][Ho][Wo][64] conv output and its normalised copy never reach memory (2 x 3 MB per 112 x 112 frame).
constexpr int SPOOL = 7;                   // pooled outputs per tile edge
__global__ __launch_bounds__(256) void stem_bn_relu_pool_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                                 const float* __restrict__ mean, const float* __restrict__ invstd,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 float* __restrict__ y, int h, int wd, int ho, int wo, int hp, int wp, int pad,
                                                                 float* __restrict__ amax_out) {
    __shared__ float patch[SP * SP];
    __shared__ __attribute__((aligned(16))) float wt[49 * STEM_CO];       // [tap][co]
    __shared__ __attribute__((aligned(16))) float coef[5 * STEM_CO];      // bias, mean, invstd, gamma, beta
    constexpr int TP = ST * ST + 1;                                       // pixel stride of a channel row (odd: no bank conflicts)
    __shared__ float tile[16 * TP];                                       // [16 channels][conv pixel], normalised + ReLU
    const int tid = threadIdx.x;
    const int n = blockIdx.z, py0 = blockIdx.y * SPOOL, px0 = blockIdx.x * SPOOL;
    const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;                       // first conv row / column of the tile (may be -1)
    for (int i = tid; i < 49 * STEM_CO; i += 256) { const int co = i / 49, t = i - co * 49; wt[t * STEM_CO + co] = w[i]; }
    for (int i = tid; i < STEM_CO; i += 256) {
        coef[i] = bias ? bias[i] : 0.f; coef[STEM_CO + i] = mean[i]; coef[2 * STEM_CO + i] = invstd[i];
        coef[3 * STEM_CO + i] = gamma[i]; coef[4 * STEM_CO + i] = beta[i];
    }
    for (int i = tid; i < SP * SP; i += 256) {
        const int qy = i / SP, qx = i - qy * SP;
        const int iy = oy0 - pad + qy, ix = ox0 - pad + qx;
        patch[i] = (iy >= 0 && iy < h && ix >= 0 && ix < wd) ? x[((long long)n * h + iy) * wd + ix] : 0.f;
    }
    __syncthreads();
    const int ty = tid >> 4, tx = tid & 15;
    float xin[49];
#pragma unroll
    for (int ky = 0; ky < 7; ++ky)
#pragma unroll
        for (int kx = 0; kx < 7; ++kx) xin[ky * 7 + kx] = patch[(ty + ky) * SP + tx + kx];
    // pooling role of this thread: pooled pixel (tid >> 2) of the 7 x 7 (threads 0 .. 195), channels 4 (tid & 3) .. of the group
    const int pp = tid >> 2, pc = (tid & 3) * 4;
    const int ppy = pp / SPOOL, ppx = pp - ppy * SPOOL;
    const bool pool_ok = pp < SPOOL * SPOOL && py0 + ppy < hp && px0 + ppx < wp;
    float am = 0.f;
#pragma unroll 1
    for (int c0 = 0; c0 < STEM_CO; c0 += 16) {
        float4 a0 = *reinterpret_cast<const float4*>(coef + c0), a1 = *reinterpret_cast<const float4*>(coef + c0 + 4);
        float4 a2 = *reinterpret_cast<const float4*>(coef + c0 + 8), a3 = *reinterpret_cast<const float4*>(coef + c0 + 12);
#pragma unroll
        for (int t = 0; t < 49; ++t) {
            const float v = xin[t];
            const float4 w0 = *reinterpret_cast<const float4*>(wt + t * STEM_CO + c0);
            const float4 w1 = *reinterpret_cast<const float4*>(wt + t * STEM_CO + c0 + 4);
            const float4 w2 = *reinterpret_cast<const float4*>(wt + t * STEM_CO + c0 + 8);
            const float4 w3 = *reinterpret_cast<const float4*>(wt + t * STEM_CO + c0 + 12);
            a0.x = fmaf(v, w0.x, a0.x); a0.y = fmaf(v, w0.y, a0.y); a0.z = fmaf(v, w0.z, a0.z); a0.w = fmaf(v, w0.w, a0.w);
            a1.x = fmaf(v, w1.x, a1.x); a1.y = fmaf(v, w1.y, a1.y); a1.z = fmaf(v, w1.z, a1.z); a1.w = fmaf(v, w1.w, a1.w);
            a2.x = fmaf(v, w2.x, a2.x); a2.y = fmaf(v, w2.y, a2.y); a2.z = fmaf(v, w2.z, a2.z); a2.w = fmaf(v, w2.w, a2.w);
            a3.x = fmaf(v, w3.x, a3.x); a3.y = fmaf(v, w3.y, a3.y); a3.z = fmaf(v, w3.z, a3.z); a3.w = fmaf(v, w3.w, a3.w);
        }
        // (x - mean) * invstd * gamma + beta, the expression of glf_bn_apply.  Tile positions outside the conv output hold a finite
        // value of no meaning: the pooling loop below skips them by coordinate (selecting here, per element, drove the register
        // allocator to 1.4 KB of scratch per lane)
        auto park = [&](const float4& a, int cb) {
            const float4 mu = *reinterpret_cast<const float4*>(coef + STEM_CO + c0 + cb), is = *reinterpret_cast<const float4*>(coef + 2 * STEM_CO + c0 + cb);
            const float4 ga = *reinterpret_cast<const float4*>(coef + 3 * STEM_CO + c0 + cb), be = *reinterpret_cast<const float4*>(coef + 4 * STEM_CO + c0 + cb);
            const float v0 = (a.x - mu.x) * is.x * ga.x + be.x, v1 = (a.y - mu.y) * is.y * ga.y + be.y;
            const float v2 = (a.z - mu.z) * is.z * ga.z + be.z, v3 = (a.w - mu.w) * is.w * ga.w + be.w;
            tile[(cb + 0) * TP + tid] = fmaxf(v0, 0.f); tile[(cb + 1) * TP + tid] = fmaxf(v1, 0.f);
            tile[(cb + 2) * TP + tid] = fmaxf(v2, 0.f); tile[(cb + 3) * TP + tid] = fmaxf(v3, 0.f);
        };
        park(a0, 0); park(a1, 4); park(a2, 8); park(a3, 12);
        __syncthreads();
        if (pool_ok) {
            float m[4] = {-1.f, -1.f, -1.f, -1.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int pix = (2 * ppy + dy) * ST + 2 * ppx + dx;
                    const int cy = oy0 + 2 * ppy + dy, cx = ox0 + 2 * ppx + dx;
                    const bool in = cy >= 0 && cy < ho && cx >= 0 && cx < wo;       // the window hangs over the conv output at the borders
#pragma unroll
                    for (int k = 0; k < 4; ++k) m[k] = in ? fmaxf(m[k], tile[(pc + k) * TP + pix]) : m[k];
                }
            *reinterpret_cast<float4*>(y + (((long long)n * hp + py0 + ppy) * wp + px0 + ppx) * STEM_CO + c0 + pc) = make_float4(m[0], m[1], m[2], m[3]);
            am = fmaxf(fmaxf(am, fmaxf(m[0], m[1])), fmaxf(m[2], m[3]));
        }
        __syncthreads();
    }
    if (amax_out) block_amax(am, amax_out);
}

extern "C" int glf_stem7x7_bn_relu_pool(const float* x, const float* w, const float* bias, const float* mean, const float* invstd,
                                        const float* gamma, const float* beta, float* y, int n, int h, int wdt, int cout, int pad,
                                        float* amax_out, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && w && mean && invstd && gamma && beta && y, GLF_ERR_NULL, "stem7x7_bn_relu_pool: null argument");
    GLF_REQUIRE(cout == STEM_CO, GLF_ERR_UNSUPPORTED, "stem7x7: Cout must be 64 (got %d)", cout);
    const int ho = h + 2 * pad - 6, wo = wdt + 2 * pad - 6;
    GLF_REQUIRE(n > 0 && n <= 65535 && ho > 0 && wo > 0 && pad >= 0 && pad <= 3, GLF_ERR_BAD_SHAPE, "stem7x7_bn_relu_pool: bad shape");
    const int hp = (ho - 1) / 2 + 1, wp = (wo - 1) / 2 + 1;             // MaxPool2d(3, stride 2, padding 1)
    GLF_REQUIRE(al16(y), GLF_ERR_BAD_SHAPE, "stem7x7_bn_relu_pool: y must be 16-byte aligned");
    dim3 grid((wp + SPOOL - 1) / SPOOL, (hp + SPOOL - 1) / SPOOL, n);
    hipLaunchKernelGGL(stem_bn_relu_pool_kernel, grid, dim3(256), 0, glf::S(s), x, w, bias, mean, invstd, gamma, beta, y, h, wdt, ho, wo, hp, wp, pad, amax_out);
    return glf::check_launch("stem7x7_bn_relu_pool");
}
extern "C" size_t glf_stem7x7_wgrad_workspace(int n, int h, int wdt, int cout, int pad) {
    const int ho = h + 2 * pad - 6, wo = wdt + 2 * pad - 6;
    if (n <= 0 || ho <= 0 || wo <= 0) return 0;
    return (size_t)n * ((ho + ST - 1) / ST) * ((wo + ST - 1) / ST) * 50 * (size_t)cout;
}
extern "C" int glf_stem7x7_wgrad(const float* x, const float* dy, float* dw, float* db, float* partial,
                                 int n, int h, int wdt, int cout, int pad, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && dy && dw && partial, GLF_ERR_NULL, "stem7x7_wgrad: null argument");
    GLF_REQUIRE(cout == STEM_CO, GLF_ERR_UNSUPPORTED, "stem7x7: Cout must be 64 (got %d)", cout);
    const int ho = h + 2 * pad - 6, wo = wdt + 2 * pad - 6;
    GLF_REQUIRE(n > 0 && n <= 65535 && ho > 0 && wo > 0 && pad >= 0 && pad <= 3, GLF_ERR_BAD_SHAPE, "stem7x7_wgrad: bad shape");
    dim3 grid((wo + ST - 1) / ST, (ho + ST - 1) / ST, n);
    hipLaunchKernelGGL(stem_wgrad_kernel<false>, grid, dim3(256), 0, glf::S(s), x, dy, partial, h, wdt, ho, wo, pad);
    if (int rc = glf::check_launch("stem7x7_wgrad")) return rc;
    const long long nblk = (long long)grid.x * grid.y * grid.z;
    hipLaunchKernelGGL(stem_wgrad_finalize, dim3(50), dim3(1024), 0, glf::S(s), partial, nblk, dw, db);
    return glf::check_launch("stem7x7_wgrad_finalize");
}
extern "C" int glf_s16_stem7x7_wgrad(const float* x, const void* dy, float* dw, float* db, float* partial,
                                     int n, int h, int wdt, int cout, int pad, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && dy && dw && partial, GLF_ERR_NULL, "s16_stem7x7_wgrad: null argument");
    GLF_REQUIRE(cout == STEM_CO, GLF_ERR_UNSUPPORTED, "stem7x7: Cout must be 64 (got %d)", cout);
    const int ho = h + 2 * pad - 6, wo = wdt + 2 * pad - 6;
    GLF_REQUIRE(n > 0 && n <= 65535 && ho > 0 && wo > 0 && pad >= 0 && pad <= 3, GLF_ERR_BAD_SHAPE, "s16_stem7x7_wgrad: bad shape");
    dim3 grid((wo + ST - 1) / ST, (ho + ST - 1) / ST, n);
    hipLaunchKernelGGL(stem_wgrad_kernel<true>, grid, dim3(256), 0, glf::S(s), x, dy, partial, h, wdt, ho, wo, pad);
    if (int rc = glf::check_launch("s16_stem7x7_wgrad")) return rc;
    const long long nblk = (long long)grid.x * grid.y * grid.z;
    hipLaunchKernelGGL(stem_wgrad_finalize, dim3(50), dim3(1024), 0, glf::S(s), partial, nblk, dw, db);
    return glf::check_launch("s16_stem7x7_wgrad_finalize");
}

extern "C" int glf_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y && idx, GLF_ERR_NULL, "maxpool_fwd: null argument");
    GLF_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && (c % 4) == 0, GLF_ERR_BAD_SHAPE, "maxpool_fwd: bad shape (C %% 4 == 0 required)");
    GLF_REQUIRE(al16(x) && al16(y) && ((reinterpret_cast<uintptr_t>(idx) & 3u) == 0), GLF_ERR_BAD_SHAPE, "maxpool_fwd: alignment");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * ho * wo * (c / 4);
    hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, glf::S(s), x, y, idx, n, h, w, c / 4, ho, wo);
    return glf::check_launch("maxpool_fwd");
}
extern "C" int glf_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int w, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && idx && dx, GLF_ERR_NULL, "maxpool_bwd: null argument");
    GLF_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && (c % 4) == 0, GLF_ERR_BAD_SHAPE, "maxpool_bwd: bad shape (C %% 4 == 0 required)");
    GLF_REQUIRE(al16(dy) && al16(dx), GLF_ERR_BAD_SHAPE, "maxpool_bwd: alignment");
    const int ho = (h + 2 - 3) / 2 + 1, wo = (w + 2 - 3) / 2 + 1;
    const long long total = (long long)n * h * w * (c / 4);
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, glf::S(s), dy, idx, dx, n, h, w, c / 4, ho, wo);
    return glf::check_launch("maxpool_bwd");
}

extern "C" int glf_sum_rows_fwd(const float* dy, int lddy, float* dx, float scale, int n, int p, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && dx, GLF_ERR_NULL, "sum_rows: null argument");
    GLF_REQUIRE(n > 0 && n <= 65535 && p > 0 && c > 0 && lddy >= c, GLF_ERR_BAD_SHAPE, "sum_rows: bad shape");
    if (c % 4 == 0 && lddy % 4 == 0 && al16(dy) && al16(dx))
        hipLaunchKernelGGL(sum_rows4_kernel, dim3((c / 4 + 15) / 16, n), dim3(256), 0, glf::S(s), dy, lddy, dx, scale, p, c / 4);
    else
        hipLaunchKernelGGL(sum_rows_kernel, dim3((c + 63) / 64, n), dim3(256), 0, glf::S(s), dy, lddy, dx, scale, p, c);
    return glf::check_launch("sum_rows");
}
extern "C" int glf_avgpool_fwd(const float* x, float* y, int n, int p, int c, glf_stream_t s) {
    return glf_sum_rows_fwd(x, c, y, p > 0 ? 1.0f / (float)p : 0.f, n, p, c, s);
}
extern "C" int glf_bcast_rows_scaled(const float* x, float* y, int ldy, float scale, int n, int p, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y, GLF_ERR_NULL, "bcast_rows: null argument");
    GLF_REQUIRE(n > 0 && p > 0 && c > 0 && ldy >= c, GLF_ERR_BAD_SHAPE, "bcast_rows: bad shape");
    const long long total = (long long)n * p * c;
    if (c % 4 == 0 && ldy % 4 == 0 && al16(x) && al16(y))
        hipLaunchKernelGGL(bcast_rows4_kernel, dim3(stream_grid(total / 4, 256)), dim3(256), 0, glf::S(s), x, y, ldy, scale, p, c / 4, total / 4);
    else
        hipLaunchKernelGGL(bcast_rows_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, glf::S(s), x, y, ldy, scale, p, c, total);
    return glf::check_launch("bcast_rows");
}
extern "C" int glf_bcast_rows_fwd(const float* x, float* y, int ldy, int n, int p, int c, glf_stream_t s) {
    return glf_bcast_rows_scaled(x, y, ldy, 1.0f, n, p, c, s);
}

extern "C" int glf_dropout(const float* x, float* y, int64_t numel, float p, uint64_t seed, const uint64_t* step_counter, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y, GLF_ERR_NULL, "dropout: null argument");
    GLF_REQUIRE(numel > 0 && p >= 0.f && p < 1.f, GLF_ERR_BAD_SHAPE, "dropout: numel > 0 and 0 <= p < 1 required");
    hipLaunchKernelGGL(dropout_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, glf::S(s), x, y, (long long)numel, p, 1.0f / (1.0f - p),
                       (unsigned long long)seed, reinterpret_cast<const unsigned long long*>(step_counter));
    return glf::check_launch("dropout");
}
__global__ void amax_combine_kernel(const float* a, const float* b, float scale, int sum, float* out) {
    const float va = fabsf(*a), vb = b ? fabsf(*b) : 0.f;
    const float v = scale * (sum ? va + vb : fmaxf(va, vb));
    if (v > *out) *out = v;
}
extern "C" int glf_amax_combine(const float* a, const float* b, float scale, int sum, float* out, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(a && out, GLF_ERR_NULL, "amax_combine: null argument");
    GLF_REQUIRE(scale >= 0.f, GLF_ERR_BAD_SHAPE, "amax_combine: scale must be >= 0");
    hipLaunchKernelGGL(amax_combine_kernel, dim3(1), dim3(1), 0, glf::S(s), a, b, scale, sum, out);
    return glf::check_launch("amax_combine");
}
__global__ void counter_add_kernel(unsigned long long* c, unsigned long long inc) { *c += inc; }
extern "C" int glf_counter_add(uint64_t* counter, uint64_t inc, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(counter, GLF_ERR_NULL, "counter_add: null argument");
    hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, glf::S(s), reinterpret_cast<unsigned long long*>(counter), (unsigned long long)inc);
    return glf::check_launch("counter_add");
}

extern "C" int glf_relu_fwd(const float* x, float* y, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y, GLF_ERR_NULL, "relu_fwd: null argument");
    GLF_REQUIRE(numel > 0, GLF_ERR_BAD_SHAPE, "relu_fwd: numel must be > 0");
    hipLaunchKernelGGL(relu_fwd_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, glf::S(s), x, y, (long long)numel);
    return glf::check_launch("relu_fwd");
}
extern "C" int glf_relu_bwd(const float* dy, const float* y, float* dx, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && y && dx, GLF_ERR_NULL, "relu_bwd: null argument");
    GLF_REQUIRE(numel > 0, GLF_ERR_BAD_SHAPE, "relu_bwd: numel must be > 0");
    hipLaunchKernelGGL(relu_bwd_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, glf::S(s), dy, y, dx, (long long)numel);
    return glf::check_launch("relu_bwd");
}

extern "C" int glf_gate_fwd(const float* cls, int ncls, const float* ctr, const float* f, float* y, float* a,
                            int32_t* argmax, float weight, int rows, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(cls && ctr && f && y && a && argmax, GLF_ERR_NULL, "gate_fwd: null argument");
    GLF_REQUIRE(rows > 0 && ncls > 0 && c > 0 && (c % 4) == 0, GLF_ERR_BAD_SHAPE, "gate_fwd: bad shape (C %% 4 == 0 required)");
    GLF_REQUIRE(al16(f) && al16(y), GLF_ERR_BAD_SHAPE, "gate_fwd: alignment");
    hipLaunchKernelGGL(gate_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, glf::S(s), cls, ncls, ctr, f, y, a, argmax, weight, rows, c);
    return glf::check_launch("gate_fwd");
}
extern "C" int glf_gate_bwd(const float* dy, const float* f, const float* cls, int ncls, const float* ctr,
                            const float* a, const int32_t* argmax, float weight,
                            float* df, float* dcls, float* dctr, int rows, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && f && cls && ctr && a && argmax && df && dcls && dctr, GLF_ERR_NULL, "gate_bwd: null argument");
    GLF_REQUIRE(rows > 0 && ncls > 0 && c > 0 && (c % 4) == 0, GLF_ERR_BAD_SHAPE, "gate_bwd: bad shape (C %% 4 == 0 required)");
    GLF_REQUIRE(al16(dy) && al16(f) && al16(df), GLF_ERR_BAD_SHAPE, "gate_bwd: alignment");
    hipLaunchKernelGGL(gate_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, glf::S(s), dy, f, cls, ncls, ctr, a, argmax, weight, df, dcls, dctr, rows, c);
    return glf::check_launch("gate_bwd");
}

extern "C" int glf_copy_frames(const float* src, int64_t src_fs, float* dst, int64_t dst_fs, int n, int64_t inner, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(src && dst, GLF_ERR_NULL, "copy_frames: null argument");
    GLF_REQUIRE(n > 0 && inner > 0 && (inner % 4) == 0 && (src_fs % 4) == 0 && (dst_fs % 4) == 0, GLF_ERR_BAD_SHAPE,
                "copy_frames: sizes and strides must be positive multiples of 4");
    GLF_REQUIRE(al16(src) && al16(dst), GLF_ERR_BAD_SHAPE, "copy_frames: alignment");
    const long long total4 = (long long)n * (inner / 4);
    hipLaunchKernelGGL(copy_frames_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, glf::S(s), reinterpret_cast<const float4*>(src),
                       (long long)(src_fs / 4), reinterpret_cast<float4*>(dst), (long long)(dst_fs / 4), (long long)(inner / 4), total4);
    return glf::check_launch("copy_frames");
}
extern "C" int glf_copy_frames_split(const float* src, int64_t src_fs, float* dst, float* dst_packed, int64_t dst_fs, int n, int64_t inner,
                                     const float* amax, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(src && dst && dst_packed && amax, GLF_ERR_NULL, "copy_frames_split: null argument");
    GLF_REQUIRE(n > 0 && inner > 0 && (inner % 4) == 0 && (src_fs % 4) == 0 && (dst_fs % 4) == 0, GLF_ERR_BAD_SHAPE,
                "copy_frames_split: sizes and strides must be positive multiples of 4");
    GLF_REQUIRE(al16(src) && al16(dst) && al16(dst_packed), GLF_ERR_BAD_SHAPE, "copy_frames_split: alignment");
    const long long total4 = (long long)n * (inner / 4);
    hipLaunchKernelGGL(copy_frames_split_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, glf::S(s), reinterpret_cast<const float4*>(src),
                       (long long)(src_fs / 4), reinterpret_cast<float4*>(dst), reinterpret_cast<float4*>(dst_packed), (long long)(dst_fs / 4),
                       (long long)(inner / 4), total4, amax);
    return glf::check_launch("copy_frames_split");
}
extern "C" int glf_add_frames(const float* a, int64_t a_fs, const float* b, int64_t b_fs, float* dst, int64_t dst_fs,
                              int n, int64_t inner, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(a && b && dst, GLF_ERR_NULL, "add_frames: null argument");
    GLF_REQUIRE(n > 0 && inner > 0 && (inner % 4) == 0 && (a_fs % 4) == 0 && (b_fs % 4) == 0 && (dst_fs % 4) == 0, GLF_ERR_BAD_SHAPE,
                "add_frames: sizes and strides must be positive multiples of 4");
    GLF_REQUIRE(al16(a) && al16(b) && al16(dst), GLF_ERR_BAD_SHAPE, "add_frames: alignment");
    const long long total4 = (long long)n * (inner / 4);
    hipLaunchKernelGGL(add_frames_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, glf::S(s), reinterpret_cast<const float4*>(a),
                       (long long)(a_fs / 4), reinterpret_cast<const float4*>(b), (long long)(b_fs / 4), reinterpret_cast<float4*>(dst),
                       (long long)(dst_fs / 4), (long long)(inner / 4), total4);
    return glf::check_launch("add_frames");
}

extern "C" int glf_add_n(const float* const* inputs, int k, float* out, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(inputs && out, GLF_ERR_NULL, "add_n: null argument");
    GLF_REQUIRE(k >= 1 && k <= 8 && numel > 0 && (numel % 4) == 0, GLF_ERR_BAD_SHAPE, "add_n: 1 <= k <= 8 and numel %% 4 == 0 required");
    AddNPtrs ptrs;
    for (int j = 0; j < 8; ++j) {
        ptrs.p[j] = reinterpret_cast<const float4*>(inputs[j < k ? j : 0]);
        GLF_REQUIRE(ptrs.p[j] != nullptr && al16(ptrs.p[j]), GLF_ERR_BAD_SHAPE, "add_n: inputs must be non-null and 16-byte aligned");
    }
    GLF_REQUIRE(al16(out), GLF_ERR_BAD_SHAPE, "add_n: out must be 16-byte aligned");
    hipLaunchKernelGGL(add_n_kernel, dim3(stream_grid(numel / 4, 256)), dim3(256), 0, glf::S(s), ptrs, k, reinterpret_cast<float4*>(out), (long long)(numel / 4));
    return glf::check_launch("add_n");
}

extern "C" int glf_softmax_rows(float* x, int64_t rows, int cols, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x, GLF_ERR_NULL, "softmax_rows: null argument");
    GLF_REQUIRE(rows > 0 && rows < 2147483647LL && cols > 0, GLF_ERR_BAD_SHAPE, "softmax_rows: bad shape");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, glf::S(s), x, cols, cols);
    return glf::check_launch("softmax_rows");
}
extern "C" int glf_softmax_rows_ld(float* x, int64_t rows, int cols, int ld, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x, GLF_ERR_NULL, "softmax_rows_ld: null argument");
    GLF_REQUIRE(rows > 0 && rows < 2147483647LL && cols > 0 && ld >= cols, GLF_ERR_BAD_SHAPE, "softmax_rows_ld: bad shape");
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, glf::S(s), x, cols, ld);
    return glf::check_launch("softmax_rows_ld");
}
extern "C" int glf_softmax_rows_bwd_ld(const float* p, float* dp_inout, int64_t rows, int cols, int ld, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(p && dp_inout, GLF_ERR_NULL, "softmax_rows_bwd_ld: null argument");
    GLF_REQUIRE(rows > 0 && rows < 2147483647LL && cols > 0 && ld >= cols, GLF_ERR_BAD_SHAPE, "softmax_rows_bwd_ld: bad shape");
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, glf::S(s), p, dp_inout, cols, ld);
    return glf::check_launch("softmax_rows_bwd_ld");
}
extern "C" int glf_softmax_rows_bwd(const float* p, float* dp_inout, int64_t rows, int cols, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(p && dp_inout, GLF_ERR_NULL, "softmax_rows_bwd: null argument");
    GLF_REQUIRE(rows > 0 && rows < 2147483647LL && cols > 0, GLF_ERR_BAD_SHAPE, "softmax_rows_bwd: bad shape");
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)rows), dim3(256), 0, glf::S(s), p, dp_inout, cols, cols);
    return glf::check_launch("softmax_rows_bwd");
}

extern "C" int glf_bilinear_up_fwd(const float* x, float* y, int n, int h, int w, int c, int ho, int wo, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y, GLF_ERR_NULL, "bilinear_up_fwd: null argument");
    GLF_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && ho > 0 && wo > 0, GLF_ERR_BAD_SHAPE, "bilinear_up_fwd: bad shape");
    const long long total = (long long)n * c * ho * wo;
    hipLaunchKernelGGL(bilinear_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, glf::S(s), x, y, n, h, w, c, ho, wo,
                       (float)h / (float)ho, (float)w / (float)wo);
    return glf::check_launch("bilinear_up_fwd");
}
extern "C" int glf_bilinear_up_bwd(const float* dy, float* dx, int n, int h, int w, int c, int ho, int wo, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && dx, GLF_ERR_NULL, "bilinear_up_bwd: null argument");
    GLF_REQUIRE(n > 0 && h > 0 && w > 0 && c > 0 && ho >= h && wo >= w, GLF_ERR_BAD_SHAPE, "bilinear_up_bwd: up-sampling only (ho >= h, wo >= w)");
    const long long total = (long long)n * h * w * c;
    // an input pixel is touched by outputs within 2/scale (+ margin) of its centre
    const int ry = (int)(2.0f * (float)ho / (float)h) + 4, rx = (int)(2.0f * (float)wo / (float)w) + 4;
    hipLaunchKernelGGL(bilinear_bwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, glf::S(s), dy, dx, n, h, w, c, ho, wo,
                       (float)h / (float)ho, (float)w / (float)wo, ry, rx);
    return glf::check_launch("bilinear_up_bwd");
}

extern "C" int glf_bce_logits_sum(const float* x, const float* t, double* loss_out, float* dx, float grad_scale,
                                  const float* grad_scale_dev, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && t && loss_out, GLF_ERR_NULL, "bce_logits_sum: null argument");
    GLF_REQUIRE(numel > 0, GLF_ERR_BAD_SHAPE, "bce_logits_sum: numel must be > 0");
    hipError_t e = hipMemsetAsync(loss_out, 0, sizeof(double), glf::S(s));
    if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "bce_logits_sum: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(bce_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, glf::S(s), x, t, loss_out, dx, grad_scale, grad_scale_dev, (long long)numel);
    return glf::check_launch("bce_logits_sum");
}
extern "C" int glf_overlap_counts(const float* logits, const float* target, int64_t* counts, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(logits && target && counts, GLF_ERR_NULL, "overlap_counts: null argument");
    GLF_REQUIRE(numel > 0, GLF_ERR_BAD_SHAPE, "overlap_counts: numel must be > 0");
    hipError_t e = hipMemsetAsync(counts, 0, 4 * sizeof(int64_t), glf::S(s));
    if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "overlap_counts: memset: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(overlap_kernel, dim3(stream_grid(numel, 256)), dim3(256), 0, glf::S(s), logits, target,
                       reinterpret_cast<unsigned long long*>(counts), (long long)numel);
    return glf::check_launch("overlap_counts");
}
