// 16-bit-storage ("S16") streaming kernels of the GL-Fusion path: everything between the contractions when activations,
// saved tensors and activation gradients live in HBM as bf16 (BASELINE configs 3 / 5).  All HBM-bound: 16-byte accesses
// (eight channels per lane), fp32 arithmetic in registers, one rounding to bf16 at the store.  Channels-last [rows][C],
// C % 8 == 0.  Statistics (BatchNorm forward and backward sums) are accumulated in double and meet in ONE f64 atomic per
// column and workgroup in a zero-filled [2][C] buffer -- the same contract as glf_gemm_params.colstats, so a BatchNorm takes
// its sums from the producing contraction's epilogue or from s16_colstats_kernel alike, and every apply kernel finishes the
// statistics itself (no finalize launches).
#include "glf_common.h"

namespace {

typedef unsigned short u16;

struct F8 { float v[8]; };

__device__ __forceinline__ F8 ld8(const u16* p) {
    const uint4 q = *reinterpret_cast<const uint4*>(p);
    F8 o;
    o.v[0] = __uint_as_float(q.x << 16); o.v[1] = __uint_as_float(q.x & 0xffff0000u);
    o.v[2] = __uint_as_float(q.y << 16); o.v[3] = __uint_as_float(q.y & 0xffff0000u);
    o.v[4] = __uint_as_float(q.z << 16); o.v[5] = __uint_as_float(q.z & 0xffff0000u);
    o.v[6] = __uint_as_float(q.w << 16); o.v[7] = __uint_as_float(q.w & 0xffff0000u);
    return o;
}
__device__ __forceinline__ unsigned pack2(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
__device__ __forceinline__ void st8(u16* p, const F8& o) {
    *reinterpret_cast<uint4*>(p) = make_uint4(pack2(o.v[0], o.v[1]), pack2(o.v[2], o.v[3]), pack2(o.v[4], o.v[5]), pack2(o.v[6], o.v[7]));
}
__device__ __forceinline__ F8 ldf8(const float* p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    F8 o;
    o.v[0] = a.x; o.v[1] = a.y; o.v[2] = a.z; o.v[3] = a.w; o.v[4] = b.x; o.v[5] = b.y; o.v[6] = b.z; o.v[7] = b.w;
    return o;
}
__device__ __forceinline__ float bf2f(u16 h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ u16 f2bf(float f) { return (u16)(pack2(f, 0.f) & 0xffffu); }

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
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// the forward value of one BatchNorm element: ONE definition, so the ReLU mask recomputed in backward is the forward's
__device__ __forceinline__ float bn_val(float x, float mu, float is, float ga, float be) { return (x - mu) * is * ga + be; }

// ----------------------------------------------------------------------------------------------------------------------
// column statistics: sums[c] += sum_r x[r][c], sums[C + c] += sum_r x[r][c]^2   (doubles, zero-filled by the caller)
// ----------------------------------------------------------------------------------------------------------------------
constexpr int RT = 256;

// Grid shape of the column reductions (profiles/r04_bn_reduce_sweep.txt): a workgroup covers RED_TPR * 8 columns with 256 / RED_TPR row
// lanes that are folded in LDS, 16 rows per thread, at most 1 024 workgroups.  What it optimises is the number of f64 atomics that
// land on ONE address: same-address atomics are serialised by the L2 (~25-30 ns each; 1 024 workgroups adding to the same 512
// doubles took longer than streaming the tensor), so narrow column blocks -- many blocks, few row slices per block -- beat whole
// rows: 193 600 x 256: 37.8 -> 27.5 us, 50 176 x 1 024: 45.4 -> 20.3 us.  More, smaller workgroups only add atomics.
constexpr int g_red_rpt = 16, g_red_cap = 1024, g_red_tpr = 16;
inline void red_init() {}
inline int red_slices(int rows, int c) {
    // A workgroup streams `per` rows of all its channels; the grid must put enough loads in flight to reach the HBM rate (these
    // reductions are latency-bound at low occupancy: 512 workgroups of 12 dependent iterations each ran at 1 TB/s), so: ~64 rows
    // per row lane, at most 1 024 workgroups (one f64 atomic per column and slice).
    const int c8 = c > 8 ? c / 8 : 1;
    const int tpr = c8 < g_red_tpr ? c8 : g_red_tpr;
    const int rpp = 256 / tpr;                        // rows per pass of a workgroup
    int s = (rows + g_red_rpt * rpp - 1) / (g_red_rpt * rpp);
    const int cblocks = (c8 + tpr - 1) / tpr;
    int cap = g_red_cap / cblocks;
    if (cap < 1) cap = 1;
    return s < 1 ? 1 : (s > cap ? cap : s);
}

// generic two-value column reduction over rows: op(r, c8, a[8], b[8]); grid (slices, channel blocks).  Per thread: fp32 partial
// sums over its <= ~16-64 rows, four rows' loads in flight at a time; the row lanes of a workgroup are folded in double through LDS
// and ONE f64 atomic per column, statistic and workgroup reaches memory.
template <class Op>
__global__ __launch_bounds__(RT) void s16_colreduce_kernel(Op op, int rows, int c, int slices, double* __restrict__ sums, int tpr_max) {
    __shared__ float sh[RT * 16];
    const int tid = threadIdx.x;
    const int c8 = c >> 3;
    const int tpr = c8 < tpr_max ? c8 : tpr_max;  // threads per row
    const int rpp = RT / tpr;                     // rows per pass
    const int ct = tid % tpr, rl = tid / tpr;
    const int slice = blockIdx.x;
    const int per = (rows + slices - 1) / slices;
    const int r0 = slice * per, r1 = min(rows, r0 + per);
    for (int cb = blockIdx.y * tpr; cb < c8; cb += gridDim.y * tpr) {
        const int cc = cb + ct;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        if (cc < c8 && rl < rpp) {
            int r = r0 + rl;
            for (; r + 3 * rpp < r1; r += 4 * rpp) {
                float a0[8], b0[8], a1[8], b1[8], a2[8], b2[8], a3[8], b3[8];
                op(r, cc * 8, a0, b0); op(r + rpp, cc * 8, a1, b1); op(r + 2 * rpp, cc * 8, a2, b2); op(r + 3 * rpp, cc * 8, a3, b3);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[j] += (a0[j] + a1[j]) + (a2[j] + a3[j]); acc[8 + j] += (b0[j] + b1[j]) + (b2[j] + b3[j]); }
            }
            for (; r < r1; r += rpp) {
                float a[8], b[8];
                op(r, cc * 8, a, b);
#pragma unroll
                for (int j = 0; j < 8; ++j) { acc[j] += a[j]; acc[8 + j] += b[j]; }
            }
        }
        // fold the row lanes through LDS ([statistic][row lane][column]) and let consecutive threads own consecutive COLUMNS: a wave's
        // 64 atomics then cover 512 contiguous bytes (scattered f64 atomics -- one thread adding its 8 columns, lanes 64 bytes
        // apart -- ran at a tenth of that rate and were most of this kernel's time)
        const int ncol = tpr * 8;
        if (rl < rpp) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sh[(0 * rpp + rl) * ncol + ct * 8 + j] = acc[j]; sh[(1 * rpp + rl) * ncol + ct * 8 + j] = acc[8 + j]; }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * ncol; idx += RT) {
            const int st = idx / ncol, col = idx - st * ncol;
            const int gc = cb * 8 + col;
            if (gc < c) {
                double d = 0;
                for (int q = 0; q < rpp; ++q) d += sh[(st * rpp + q) * ncol + col];
                atomicAdd(sums + st * c + gc, d);
            }
        }
        __syncthreads();
    }
}

struct OpStats16 {          // (x, x^2)
    const u16* x; int ldx;
    __device__ void operator()(int r, int c, float* a, float* b) const {
        const F8 v = ld8(x + (long long)r * ldx + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = v.v[j]; b[j] = v.v[j] * v.v[j]; }
    }
};

struct OpBnBwd16 {          // (dy', dy' * xhat), dy' = (dy + dy2) * relu-mask
    const u16* dy; int lddy; const u16* dy2; int lddy2; const u16* x; int ldx;
    const float* mean; const float* invstd; const float* gamma; const float* beta; int relu;
    const unsigned char* mask; int c8;          // relu with a residual: the forward's sign bits, one byte per 8 channels
    __device__ void operator()(int r, int c, float* a, float* b) const {
        F8 g = ld8(dy + (long long)r * lddy + c);
        if (dy2) {
            const F8 g2 = ld8(dy2 + (long long)r * lddy2 + c);
#pragma unroll
            for (int j = 0; j < 8; ++j) g.v[j] += g2.v[j];
        }
        const F8 xx = ld8(x + (long long)r * ldx + c);
        const F8 mu = ldf8(mean + c), is = ldf8(invstd + c);
        unsigned m = 0xffu;
        if (relu && mask) {
            m = mask[(long long)r * c8 + (c >> 3)];
        } else if (relu) {
            const F8 ga = ldf8(gamma + c), be = ldf8(beta + c);
            m = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) m |= (bn_val(xx.v[j], mu.v[j], is.v[j], ga.v[j], be.v[j]) > 0.f ? 1u : 0u) << j;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float gj = ((m >> j) & 1u) ? g.v[j] : 0.f;
            a[j] = gj;
            b[j] = gj * (xx.v[j] - mu.v[j]) * is.v[j];
        }
    }
};

__device__ __forceinline__ F8 unpack8(const uint4 q) {
    F8 o;
    o.v[0] = __uint_as_float(q.x << 16); o.v[1] = __uint_as_float(q.x & 0xffff0000u);
    o.v[2] = __uint_as_float(q.y << 16); o.v[3] = __uint_as_float(q.y & 0xffff0000u);
    o.v[4] = __uint_as_float(q.z << 16); o.v[5] = __uint_as_float(q.z & 0xffff0000u);
    o.v[6] = __uint_as_float(q.w << 16); o.v[7] = __uint_as_float(q.w & 0xffff0000u);
    return o;
}

// BatchNorm backward column reduce, the form that runs: the generic kernel above with OpBnBwd16 compiled into a chain of dependent
// loads (dy -> wait -> dy2 -> wait -> x -> wait -> mask -> the four coefficient vectors again for every row: the ISA had five
// s_waitcnt vmcnt(0) per row and 2.3 TB/s).  Here the coefficient vectors are loaded once per column block, the loads of FOUR rows
// (8-12 x 16 bytes per lane) are issued back to back with nothing between them, and the arithmetic follows.
// HAS2: a second gradient addend; RM: 0 = no ReLU, 1 = sign bytes, 2 = sign recomputed from x and the BatchNorm coefficients.
template <bool HAS2, int RM>
__global__ __launch_bounds__(RT) void s16_bnbwd_reduce_kernel(OpBnBwd16 op, int rows, int c, int slices, double* __restrict__ sums, int tpr_max) {
    __shared__ float sh[RT * 16];
    const int tid = threadIdx.x;
    const int c8 = c >> 3;
    const int tpr = c8 < tpr_max ? c8 : tpr_max;
    const int rpp = RT / tpr;
    const int ct = tid % tpr, rl = tid / tpr;
    const int slice = blockIdx.x;
    const int per = (rows + slices - 1) / slices;
    const int r0 = slice * per, r1 = min(rows, r0 + per);
    for (int cb = blockIdx.y * tpr; cb < c8; cb += gridDim.y * tpr) {
        const int cc = cb + ct;
        float acc[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] = 0.f;
        if (cc < c8 && rl < rpp) {
            const int c0 = cc * 8;
            const F8 mu = ldf8(op.mean + c0), is = ldf8(op.invstd + c0);
            F8 ga = mu, be = mu;
            if (RM == 2) { ga = ldf8(op.gamma + c0); be = ldf8(op.beta + c0); }
            auto fold = [&](const uint4 qg, const uint4 qg2, const uint4 qx, unsigned m) __attribute__((always_inline)) {
                F8 g = unpack8(qg);
                if (HAS2) {
                    const F8 g2 = unpack8(qg2);
#pragma unroll
                    for (int j = 0; j < 8; ++j) g.v[j] += g2.v[j];
                }
                const F8 xx = unpack8(qx);
                if (RM == 2) {
                    m = 0;
#pragma unroll
                    for (int j = 0; j < 8; ++j) m |= (bn_val(xx.v[j], mu.v[j], is.v[j], ga.v[j], be.v[j]) > 0.f ? 1u : 0u) << j;
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float gj = (RM == 0 || ((m >> j) & 1u)) ? g.v[j] : 0.f;
                    acc[j] += gj;
                    acc[8 + j] += gj * (xx.v[j] - mu.v[j]) * is.v[j];
                }
            };
            int r = r0 + rl;
            for (; r + 3 * rpp < r1; r += 4 * rpp) {
                uint4 qg[4], qg2[4], qx[4];
                unsigned m[4] = {0xffu, 0xffu, 0xffu, 0xffu};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long row = r + u * rpp;
                    qg[u] = *reinterpret_cast<const uint4*>(op.dy + row * op.lddy + c0);
                    if (HAS2) qg2[u] = *reinterpret_cast<const uint4*>(op.dy2 + row * op.lddy2 + c0);
                    qx[u] = *reinterpret_cast<const uint4*>(op.x + row * op.ldx + c0);
                    if (RM == 1) m[u] = op.mask[row * op.c8 + cc];
                }
                __builtin_amdgcn_sched_barrier(0);       // all loads of the four rows are in flight before the first use (hipcc sinks them to their uses)
#pragma unroll
                for (int u = 0; u < 4; ++u) fold(qg[u], HAS2 ? qg2[u] : qg[u], qx[u], m[u]);
            }
            for (; r < r1; r += rpp) {
                const long long row = r;
                const uint4 qg = *reinterpret_cast<const uint4*>(op.dy + row * op.lddy + c0);
                const uint4 qg2 = HAS2 ? *reinterpret_cast<const uint4*>(op.dy2 + row * op.lddy2 + c0) : qg;
                const uint4 qx = *reinterpret_cast<const uint4*>(op.x + row * op.ldx + c0);
                const unsigned m = RM == 1 ? op.mask[row * op.c8 + cc] : 0xffu;
                fold(qg, qg2, qx, m);
            }
        }
        const int ncol = tpr * 8;
        if (rl < rpp) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { sh[(0 * rpp + rl) * ncol + ct * 8 + j] = acc[j]; sh[(1 * rpp + rl) * ncol + ct * 8 + j] = acc[8 + j]; }
        }
        __syncthreads();
        for (int idx = tid; idx < 2 * ncol; idx += RT) {
            const int st = idx / ncol, col = idx - st * ncol;
            const int gc = cb * 8 + col;
            if (gc < c) {
                double d = 0;
                for (int q = 0; q < rpp; ++q) d += sh[(st * rpp + q) * ncol + col];
                atomicAdd(sums + st * c + gc, d);
            }
        }
        __syncthreads();
    }
}

struct OpColsum16 {         // (dy, 0)
    const u16* dy; int ld;
    __device__ void operator()(int r, int c, float* a, float* b) const {
        const F8 v = ld8(dy + (long long)r * ld + c);
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] = v.v[j]; b[j] = 0.f; }
    }
};

template <class Op>
int launch_colreduce16(Op op, int rows, int c, double* sums, hipStream_t s) {
    red_init();
    const int slices = red_slices(rows, c);
    const int c8 = c / 8, tpr = c8 < g_red_tpr ? c8 : g_red_tpr;
    hipLaunchKernelGGL((s16_colreduce_kernel<Op>), dim3(slices, (c8 + tpr - 1) / tpr), dim3(RT), 0, s, op, rows, c, slices, sums, g_red_tpr);
    return glf::check_launch("s16_colreduce");
}

int launch_bnbwd_reduce16(const OpBnBwd16& op, int rows, int c, double* sums, hipStream_t s) {
    red_init();
    const int slices = red_slices(rows, c);
    const int c8 = c / 8, tpr = c8 < g_red_tpr ? c8 : g_red_tpr;
    const dim3 grid(slices, (c8 + tpr - 1) / tpr), block(RT);
    const int rm = !op.relu ? 0 : (op.mask ? 1 : 2);
#define GLF_BNR(H2, RM_) hipLaunchKernelGGL((s16_bnbwd_reduce_kernel<H2, RM_>), grid, block, 0, s, op, rows, c, slices, sums, g_red_tpr)
    if (op.dy2) { if (rm == 0) GLF_BNR(true, 0); else if (rm == 1) GLF_BNR(true, 1); else GLF_BNR(true, 2); }
    else { if (rm == 0) GLF_BNR(false, 0); else if (rm == 1) GLF_BNR(false, 1); else GLF_BNR(false, 2); }
#undef GLF_BNR
    return glf::check_launch("s16_bn_bwd_reduce");
}

__global__ void f64_to_f32_kernel(const double* __restrict__ src, float* __restrict__ dst, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (float)src[i];
}

// ----------------------------------------------------------------------------------------------------------------------
// BatchNorm apply (+ residual, + ReLU), statistics finished in the kernel
// ----------------------------------------------------------------------------------------------------------------------
constexpr int APPLY_MAX_C = 4096;
__global__ __launch_bounds__(256) void s16_bn_apply_kernel(const u16* __restrict__ x, int ldx, const u16* __restrict__ res, int ldr,
                                                           u16* __restrict__ y, int ldy, const double* __restrict__ sums, int rows, int c,
                                                           float eps, float momentum, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ mean_io, float* __restrict__ invstd_io, float* rmean, float* rvar,
                                                           long long* nbt, long long total8, int c8, int relu, unsigned char* __restrict__ mask) {
    // sums != null (train): every workgroup turns (sum x, sum x^2) into mean / invstd for all channels in LDS; workgroup 0 writes
    // them to mean_io / invstd_io for the backward pass and updates the running statistics.  sums == null (eval): mean_io /
    // invstd_io are inputs (glf_bn_eval_coeffs).
    extern __shared__ __attribute__((aligned(16))) float s_coef[];       // [4][c]: scale, shift (y = x * scale + shift)
    float* s_sc = s_coef;
    float* s_sh = s_coef + c;
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        float mf, isf;
        if (sums) {
            const double m = sums[ch] / rows;
            double var = sums[c + ch] / rows - m * m;
            if (var < 0) var = 0;
            mf = (float)m; isf = (float)(1.0 / sqrt(var + (double)eps));
            if (blockIdx.x == 0) {
                mean_io[ch] = mf; invstd_io[ch] = isf;
                if (rmean) {
                    const double unb = rows > 1 ? var * rows / (rows - 1) : var;
                    rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)m;
                    rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
                }
            }
        } else {
            mf = mean_io[ch]; isf = invstd_io[ch];
        }
        s_sc[ch] = mf; s_sh[ch] = isf;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt && sums) *nbt += 1;
    __syncthreads();
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c8;
        const int cc = (int)(i - r * c8) * 8;
        const F8 xx = ld8(x + r * ldx + cc);
        const F8 mu = ldf8(s_sc + cc), is = ldf8(s_sh + cc), ga = ldf8(gamma + cc), be = ldf8(beta + cc);
        F8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o.v[j] = bn_val(xx.v[j], mu.v[j], is.v[j], ga.v[j], be.v[j]);
        if (res) {
            const F8 rr = ld8(res + r * ldr + cc);
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] += rr.v[j];
        }
        if (mask) {
            unsigned m = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) m |= (o.v[j] > 0.f ? 1u : 0u) << j;
            mask[i] = (unsigned char)m;
        }
        if (relu) {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = fmaxf(o.v[j], 0.f);
        }
        st8(y + r * ldy + cc, o);
    }
}

// dx = gamma*invstd*(dy' - [sum_dy/n + xhat*sum_dyx/n]) ; dres = dy'.  The sums arrive unfinished (doubles [2][c] from the
// reduction kernel's atomics); workgroup 0 also writes dgamma = sum dy' xhat and dbeta = sum dy'.
__global__ __launch_bounds__(256) void s16_bnbwd_apply_kernel(const u16* __restrict__ dy, int lddy, const u16* __restrict__ dy2, int lddy2,
                                                              const u16* __restrict__ x, int ldx, const float* __restrict__ mean,
                                                              const float* __restrict__ invstd, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                              const double* __restrict__ sums, u16* __restrict__ dx, int lddx, u16* __restrict__ dres,
                                                              int lddres, float* __restrict__ dgamma, float* __restrict__ dbeta, long long total8, int c8, int c,
                                                              int relu, int training, float inv_n, const unsigned char* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) float s_k[];        // [2][c]: k1 = gamma invstd, (training) s1 / n, s2 / n packed below
    float* s_s1 = s_k;
    float* s_s2 = s_k + c;
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        const float a = (float)sums[ch], b = (float)sums[c + ch];
        s_s1[ch] = a * inv_n; s_s2[ch] = b * inv_n;
        if (blockIdx.x == 0) { if (dbeta) dbeta[ch] = a; if (dgamma) dgamma[ch] = b; }
    }
    __syncthreads();
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c8;
        const int cc = (int)(i - r * c8) * 8;
        F8 g = ld8(dy + r * lddy + cc);
        if (dy2) {
            const F8 g2 = ld8(dy2 + r * lddy2 + cc);
#pragma unroll
            for (int j = 0; j < 8; ++j) g.v[j] += g2.v[j];
        }
        const F8 is = ldf8(invstd + cc), ga = ldf8(gamma + cc);
        F8 xx, mu;
        const bool need_x = training || (relu && !mask);
        if (need_x) { xx = ld8(x + r * ldx + cc); mu = ldf8(mean + cc); }
        unsigned m = 0xffu;
        if (relu && mask) {
            m = mask[i];
        } else if (relu) {
            const F8 be = ldf8(beta + cc);
            m = 0;
#pragma unroll
            for (int j = 0; j < 8; ++j) m |= (bn_val(xx.v[j], mu.v[j], is.v[j], ga.v[j], be.v[j]) > 0.f ? 1u : 0u) << j;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) g.v[j] = ((m >> j) & 1u) ? g.v[j] : 0.f;
        if (dres) st8(dres + r * lddres + cc, g);
        F8 o;
        if (training) {
            const F8 s1 = ldf8(s_s1 + cc), s2 = ldf8(s_s2 + cc);
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = ga.v[j] * is.v[j] * (g.v[j] - (s1.v[j] + (xx.v[j] - mu.v[j]) * is.v[j] * s2.v[j]));
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = g.v[j] * ga.v[j] * is.v[j];
        }
        st8(dx + r * lddx + cc, o);
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// TPAVI tail z = LayerNorm_C(BN(w) + x): one wavefront per row, C <= 2048
// ----------------------------------------------------------------------------------------------------------------------
constexpr int LN_NV = 4;       // 8 channels x 64 lanes x 4 = 2048
constexpr int LN_ROWS = 64;    // backward: rows per workgroup (4 waves x 16 rows), one slab of column partials per workgroup

__global__ __launch_bounds__(256) void s16_bn_res_ln_fwd_kernel(const u16* __restrict__ w, const u16* __restrict__ x, const float* __restrict__ bn_mean,
                                                                const float* __restrict__ bn_invstd, const float* __restrict__ bn_g,
                                                                const float* __restrict__ bn_b, const float* __restrict__ ln_g,
                                                                const float* __restrict__ ln_b, float eps, u16* __restrict__ z,
                                                                float* __restrict__ row_mean, float* __restrict__ row_rstd, int rows, int c) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int c8 = c >> 3;
    const long long base = (long long)row * c;
    F8 u[LN_NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < LN_NV; ++v) {
        const int cc = lane + 64 * v;
        if (cc < c8) {
            const int ch = cc * 8;
            const F8 ww = ld8(w + base + ch), xx = ld8(x + base + ch);
            const F8 mu = ldf8(bn_mean + ch), is = ldf8(bn_invstd + ch), ga = ldf8(bn_g + ch), be = ldf8(bn_b + ch);
#pragma unroll
            for (int j = 0; j < 8; ++j) { u[v].v[j] = (ww.v[j] - mu.v[j]) * is.v[j] * ga.v[j] + be.v[j] + xx.v[j]; s += u[v].v[j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) u[v].v[j] = 0.f;
        }
    }
    const float mean = wave_sum(s) / c;
    float q = 0.f;
#pragma unroll
    for (int v = 0; v < LN_NV; ++v)
        if (lane + 64 * v < c8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = u[v].v[j] - mean; q += d * d; }
        }
    const float rstd = 1.0f / sqrtf(wave_sum(q) / c + eps);
    if (lane == 0) { row_mean[row] = mean; row_rstd[row] = rstd; }
#pragma unroll
    for (int v = 0; v < LN_NV; ++v) {
        const int cc = lane + 64 * v;
        if (cc < c8) {
            const int ch = cc * 8;
            const F8 g = ldf8(ln_g + ch), b = ldf8(ln_b + ch);
            F8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) o.v[j] = (u[v].v[j] - mean) * rstd * g.v[j] + b.v[j];
            st8(z + base + ch, o);
        }
    }
}

// backward: du (gradient of u = BN(w) + x) per row, and per-workgroup partial column sums (dz * uhat, dz) of its LN_ROWS rows
// into slab[blockIdx.x][2][c] (floats); s16_ln_param_finalize folds the slabs in order.
__global__ __launch_bounds__(256) void s16_bn_res_ln_bwd_kernel(const u16* __restrict__ dz, const u16* __restrict__ w, const u16* __restrict__ x,
                                                                const float* __restrict__ bn_mean, const float* __restrict__ bn_invstd,
                                                                const float* __restrict__ bn_g, const float* __restrict__ bn_b,
                                                                const float* __restrict__ ln_g, const float* __restrict__ row_mean,
                                                                const float* __restrict__ row_rstd, u16* __restrict__ du, float* __restrict__ slab,
                                                                int rows, int c) {
    __shared__ float sh[2 * 2048];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int c8 = c >> 3;
    F8 pg[LN_NV], pb[LN_NV];                     // partial sums of dz * uhat and dz over this wave's rows
#pragma unroll
    for (int v = 0; v < LN_NV; ++v)
#pragma unroll
        for (int j = 0; j < 8; ++j) { pg[v].v[j] = 0.f; pb[v].v[j] = 0.f; }
    const int row0 = blockIdx.x * LN_ROWS + wv * (LN_ROWS / 4);
    for (int rr = 0; rr < LN_ROWS / 4; ++rr) {
        const int row = row0 + rr;
        if (row >= rows) break;
        const long long base = (long long)row * c;
        const float mean = row_mean[row], rstd = row_rstd[row];
        F8 uh[LN_NV], gz[LN_NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < LN_NV; ++v) {
            const int cc = lane + 64 * v;
            if (cc < c8) {
                const int ch = cc * 8;
                const F8 ww = ld8(w + base + ch), xx = ld8(x + base + ch), d = ld8(dz + base + ch);
                const F8 mu = ldf8(bn_mean + ch), is = ldf8(bn_invstd + ch), ga = ldf8(bn_g + ch), be = ldf8(bn_b + ch), g = ldf8(ln_g + ch);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float uu = (ww.v[j] - mu.v[j]) * is.v[j] * ga.v[j] + be.v[j] + xx.v[j];
                    uh[v].v[j] = (uu - mean) * rstd;
                    gz[v].v[j] = d.v[j] * g.v[j];
                    s1 += gz[v].v[j];
                    s2 += gz[v].v[j] * uh[v].v[j];
                    pg[v].v[j] += d.v[j] * uh[v].v[j];
                    pb[v].v[j] += d.v[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) { uh[v].v[j] = 0.f; gz[v].v[j] = 0.f; }
            }
        }
        const float m1 = wave_sum(s1) / c, m2 = wave_sum(s2) / c;
#pragma unroll
        for (int v = 0; v < LN_NV; ++v) {
            const int cc = lane + 64 * v;
            if (cc < c8) {
                F8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o.v[j] = rstd * (gz[v].v[j] - m1 - uh[v].v[j] * m2);
                st8(du + base + cc * 8, o);
            }
        }
    }
    // fold the four waves' partials through LDS, wave after wave (same lane <-> channel map in every wave)
    for (int k = 0; k < 4; ++k) {
        if (wv == k) {
#pragma unroll
            for (int v = 0; v < LN_NV; ++v) {
                const int cc = lane + 64 * v;
                if (cc < c8) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int ch = cc * 8 + j;
                        if (k == 0) { sh[ch] = pg[v].v[j]; sh[2048 + ch] = pb[v].v[j]; }
                        else { sh[ch] += pg[v].v[j]; sh[2048 + ch] += pb[v].v[j]; }
                    }
                }
            }
        }
        __syncthreads();
    }
    float* dst = slab + (long long)blockIdx.x * 2 * c;
    for (int ch = threadIdx.x; ch < c; ch += 256) { dst[ch] = sh[ch]; dst[c + ch] = sh[2048 + ch]; }
}
__global__ __launch_bounds__(256) void s16_ln_param_finalize(const float* __restrict__ slab, int nslab, int c, float* __restrict__ dg, float* __restrict__ db) {
    // a workgroup = 16 (statistic, channel) columns x 16 slab lanes; every lane adds its slabs in order in double, lane 0 folds the
    // 16 lane sums in order: a fixed summation tree => bitwise reproducible
    __shared__ double sh[256];
    const int cx = threadIdx.x & 15, lane = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + cx;
    double s = 0;
    if (i < 2 * c)
        for (int k = lane; k < nslab; k += 16) s += slab[(long long)k * 2 * c + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (lane == 0 && i < 2 * c) {
        for (int l = 1; l < 16; ++l) s += sh[l * 16 + cx];
        if (i < c) dg[i] = (float)s; else db[i - c] = (float)s;
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// pooling / broadcast / dropout / gate / adds / casts
// ----------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void s16_maxpool_fwd_kernel(const u16* __restrict__ x, u16* __restrict__ y, uint8_t* __restrict__ idx,
                                                              int n, int h, int w, int c8, int ho, int wo) {
    const long long total = (long long)n * ho * wo * c8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8); long long p = i / c8;
        const int ox = (int)(p % wo); p /= wo;
        const int oy = (int)(p % ho); const int nn = (int)(p / ho);
        F8 best;
        unsigned char bi[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { best.v[j] = -INFINITY; bi[j] = 0; }
        bool first = true;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
                if (iy < 0 || iy >= h || ix < 0 || ix >= w) continue;
                const F8 v = ld8(x + (((long long)nn * h + iy) * w + ix) * (c8 * 8) + cc * 8);
                const int t = ky * 3 + kx;
                // ATen: the first in-range element, then strictly greater (or NaN) replaces
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (first || v.v[j] > best.v[j] || v.v[j] != v.v[j]) { best.v[j] = v.v[j]; bi[j] = (unsigned char)t; }
                first = false;
            }
        st8(y + i * 8, best);
        *reinterpret_cast<uint2*>(idx + i * 8) = make_uint2(bi[0] | (bi[1] << 8) | (bi[2] << 16) | ((unsigned)bi[3] << 24),
                                                            bi[4] | (bi[5] << 8) | (bi[6] << 16) | ((unsigned)bi[7] << 24));
    }
}
__global__ __launch_bounds__(256) void s16_maxpool_bwd_kernel(const u16* __restrict__ dy, const uint8_t* __restrict__ idx, u16* __restrict__ dx,
                                                              int n, int h, int w, int c8, int ho, int wo) {
    const long long total = (long long)n * h * w * c8;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8); long long p = i / c8;
        const int ix = (int)(p % w); p /= w;
        const int iy = (int)(p % h); const int nn = (int)(p / h);
        F8 g;
#pragma unroll
        for (int j = 0; j < 8; ++j) g.v[j] = 0.f;
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
                const long long o = ((((long long)nn * ho + oy) * wo + ox) * c8 + cc) * 8;
                const uint2 t = *reinterpret_cast<const uint2*>(idx + o);
                const F8 d = ld8(dy + o);
                const unsigned me = ky * 3 + kx;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const unsigned tj = ((j < 4 ? t.x : t.y) >> (8 * (j & 3))) & 0xffu;
                    if (tj == me) g.v[j] += d.v[j];
                }
            }
        }
        st8(dx + i * 8, g);
    }
}

// y[n][c] = scale * sum_p x[n][p][c]   (x row stride ld); y bf16 or fp32 (the ASPP pooled branch keeps its N per-frame vectors
// in fp32: its BatchNorm normalises over the N frames, whose averages differ by less than a few bf16 steps)
__global__ __launch_bounds__(256) void s16_sum_rows_kernel(const u16* __restrict__ x, int ld, void* __restrict__ yv, int y_f32, float scale, int p, int c) {
    // workgroup = 32 channel groups of 8 x 8 row lanes; grid (c / 256, n)
    __shared__ float sh[256 * 8];
    const int n = blockIdx.y;
    const int cg = blockIdx.x * 32 + (threadIdx.x & 31);       // group of 8 channels
    const int pl = threadIdx.x >> 5;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (cg * 8 < c) {
        const u16* base = x + (long long)n * p * ld + cg * 8;
        int r = pl;
        for (; r + 24 < p; r += 32) {                                   // four rows' loads in flight
            const uint4 q0 = *reinterpret_cast<const uint4*>(base + (long long)r * ld);
            const uint4 q1 = *reinterpret_cast<const uint4*>(base + (long long)(r + 8) * ld);
            const uint4 q2 = *reinterpret_cast<const uint4*>(base + (long long)(r + 16) * ld);
            const uint4 q3 = *reinterpret_cast<const uint4*>(base + (long long)(r + 24) * ld);
            const F8 v0 = unpack8(q0), v1 = unpack8(q1), v2 = unpack8(q2), v3 = unpack8(q3);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += (v0.v[j] + v1.v[j]) + (v2.v[j] + v3.v[j]);
        }
        for (; r < p; r += 8) {
            const F8 v = ld8(base + (long long)r * ld);
#pragma unroll
            for (int j = 0; j < 8; ++j) s[j] += v.v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) sh[j * 256 + threadIdx.x] = s[j];
    __syncthreads();
    if (pl == 0 && cg * 8 < c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = s[j];
            for (int q = 1; q < 8; ++q) v += sh[j * 256 + q * 32 + (threadIdx.x & 31)];
            v *= scale;
            if (y_f32) static_cast<float*>(yv)[(long long)n * c + cg * 8 + j] = v;
            else static_cast<u16*>(yv)[(long long)n * c + cg * 8 + j] = f2bf(v);
        }
    }
}
__global__ __launch_bounds__(256) void s16_bcast_rows_kernel(const void* __restrict__ xv, int x_f32, u16* __restrict__ y, int ld, float scale, int p, int c8, long long total8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % c8); const long long row = i / c8;
        const long long n = row / p;
        F8 v = x_f32 ? ldf8(static_cast<const float*>(xv) + (n * c8 + cc) * 8) : ld8(static_cast<const u16*>(xv) + (n * c8 + cc) * 8);
        if (scale != 1.f) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v.v[j] *= scale;
        }
        st8(y + row * ld + cc * 8, v);
    }
}

__device__ __forceinline__ unsigned mix64(unsigned long long z) {
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (unsigned)(z >> 40);                           // 24 bits
}
__global__ __launch_bounds__(256) void s16_dropout_kernel(const u16* __restrict__ x, u16* __restrict__ y, long long n8, float p, float scale,
                                                          unsigned long long seed, const unsigned long long* __restrict__ step) {
    if (step) seed += *step * 0xD1B54A32D192ED03ull;
    const unsigned thr = (unsigned)(p * 16777216.0f);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        F8 v = ld8(x + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v.v[j] = (mix64(seed * 0x100000001B3ull + (unsigned long long)(i * 8 + j)) >= thr) ? v.v[j] * scale : 0.f;
        st8(y + i * 8, v);
    }
}

__global__ __launch_bounds__(256) void s16_relu_fwd_kernel(const u16* __restrict__ x, u16* __restrict__ y, long long n8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        F8 v = ld8(x + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v.v[j] = fmaxf(v.v[j], 0.f);
        st8(y + i * 8, v);
    }
}
__global__ __launch_bounds__(256) void s16_relu_bwd_kernel(const u16* __restrict__ dy, const u16* __restrict__ y, u16* __restrict__ dx, long long n8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        F8 g = ld8(dy + i * 8);
        const F8 v = ld8(y + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) g.v[j] = v.v[j] > 0.f ? g.v[j] : 0.f;
        st8(dx + i * 8, g);
    }
}
// out = a * x + b * y
__global__ __launch_bounds__(256) void s16_axpby_kernel(const u16* __restrict__ x, const u16* __restrict__ y, u16* __restrict__ out, float a, float b, long long n8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        F8 u = ld8(x + i * 8);
        const F8 v = ld8(y + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) u.v[j] = a * u.v[j] + b * v.v[j];
        st8(out + i * 8, u);
    }
}

__global__ __launch_bounds__(256) void s16_gate_fwd_kernel(const float* __restrict__ cls, int ncls, const float* __restrict__ ctr,
                                                           const u16* __restrict__ f, u16* __restrict__ y, float* __restrict__ a_out,
                                                           int* __restrict__ amax, float weight, int rows, int c) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float best = sigmoidf_(cls[(long long)row * ncls]);
    int bi = 0;
    for (int k = 1; k < ncls; ++k) { const float s = sigmoidf_(cls[(long long)row * ncls + k]); if (s > best) { best = s; bi = k; } }
    const float cc = sigmoidf_(ctr[row]);
    const float a = sigmoidf_(weight * best * cc);
    if (lane == 0) { a_out[row] = a; amax[row] = bi; }
    for (int i = lane; i < (c >> 3); i += 64) {
        F8 v = ld8(f + (long long)row * c + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) v.v[j] *= a;
        st8(y + (long long)row * c + i * 8, v);
    }
}
__global__ __launch_bounds__(256) void s16_gate_bwd_kernel(const u16* __restrict__ dy, const u16* __restrict__ f, const float* __restrict__ cls, int ncls,
                                                           const float* __restrict__ ctr, const float* __restrict__ a_in, const int* __restrict__ amax,
                                                           float weight, u16* __restrict__ df, float* __restrict__ dcls, float* __restrict__ dctr,
                                                           int rows, int c) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float a = a_in[row];
    float s = 0.f;
    for (int i = lane; i < (c >> 3); i += 64) {
        F8 g = ld8(dy + (long long)row * c + i * 8);
        const F8 v = ld8(f + (long long)row * c + i * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s += g.v[j] * v.v[j]; g.v[j] *= a; }
        st8(df + (long long)row * c + i * 8, g);
    }
    const float da = wave_sum(s);
    if (lane == 0) {
        const int bi = amax[row];
        const float m = sigmoidf_(cls[(long long)row * ncls + bi]);
        const float cc = sigmoidf_(ctr[row]);
        const float dt = da * a * (1.f - a) * weight;
        for (int k = 0; k < ncls; ++k) dcls[(long long)row * ncls + k] = (k == bi) ? dt * cc * m * (1.f - m) : 0.f;
        dctr[row] = dt * m * cc * (1.f - cc);
    }
}

__global__ __launch_bounds__(256) void s16_add_frames_kernel(const u16* __restrict__ a, long long afs, const u16* __restrict__ b, long long bfs,
                                                             u16* __restrict__ dst, long long dfs, long long inner8, long long total8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total8; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / inner8, r = i - n * inner8;
        F8 u = ld8(a + n * afs + r * 8);
        const F8 v = ld8(b + n * bfs + r * 8);
#pragma unroll
        for (int j = 0; j < 8; ++j) u.v[j] += v.v[j];
        st8(dst + n * dfs + r * 8, u);
    }
}
struct AddN16 { const u16* p[8]; };
__global__ __launch_bounds__(256) void s16_add_n_kernel(AddN16 in, int k, u16* __restrict__ out, long long n8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        F8 a = ld8(in.p[0] + i * 8);
        for (int j = 1; j < k; ++j) {
            const F8 b = ld8(in.p[j] + i * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) a.v[e] += b.v[e];
        }
        st8(out + i * 8, a);
    }
}
__global__ __launch_bounds__(256) void s16_to_f32_kernel(const u16* __restrict__ x, float* __restrict__ y, long long n8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) {
        const F8 v = ld8(x + i * 8);
        *reinterpret_cast<float4*>(y + i * 8) = make_float4(v.v[0], v.v[1], v.v[2], v.v[3]);
        *reinterpret_cast<float4*>(y + i * 8 + 4) = make_float4(v.v[4], v.v[5], v.v[6], v.v[7]);
    }
}
__global__ __launch_bounds__(256) void s16_from_f32_kernel(const float* __restrict__ x, u16* __restrict__ y, long long n8) {
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n8; i += (long long)gridDim.x * blockDim.x) st8(y + i * 8, ldf8(x + i * 8));
}
// batched 2-D transpose of 16-bit elements through a padded LDS tile: dst[b][c][r] = src[b][r][c]
__global__ __launch_bounds__(256) void s16_transpose2d_kernel(const u16* __restrict__ src, u16* __restrict__ dst, int rows, int cols) {
    __shared__ u16 tile[64][66];
    const long long boff = (long long)blockIdx.z * rows * cols;
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? src[boff + (long long)r * cols + c] : (u16)0;
    }
    __syncthreads();
    for (int i = ty; i < 64; i += 4) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) dst[boff + (long long)c * rows + r] = tile[tx][i];
    }
}

}  // namespace

#define REQ_C8(c) GLF_REQUIRE((c) > 0 && ((c) % 8) == 0, GLF_ERR_BAD_SHAPE, "channel count must be a positive multiple of 8 (got %d)", (c))
#define REQ_AL(p, name) GLF_REQUIRE(al16(p), GLF_ERR_BAD_SHAPE, name " must be 16-byte aligned")
#define REQ_LD(ld, name) GLF_REQUIRE(((ld) % 8) == 0, GLF_ERR_BAD_SHAPE, name " must be a multiple of 8")

extern "C" int glf_s16_colstats(const void* x, int ldx, int rows, int c, double* sums, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && sums, GLF_ERR_NULL, "s16_colstats: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "s16_colstats: rows must be > 0");
    REQ_C8(c); REQ_AL(x, "x"); REQ_LD(ldx, "ldx");
    return launch_colreduce16(OpStats16{static_cast<const u16*>(x), ldx}, rows, c, sums, glf::S(s));
}

extern "C" int glf_s16_bn_apply(const void* x, int ldx, const void* residual, int ldr, void* y, int ldy, const double* sums,
                                int rows, int c, float eps, float momentum, const float* gamma, const float* beta,
                                float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                int relu, uint8_t* relu_mask, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y && mean && invstd && gamma && beta, GLF_ERR_NULL, "s16_bn_apply: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "s16_bn_apply: rows must be > 0");
    REQ_C8(c); REQ_AL(x, "x"); REQ_AL(y, "y"); REQ_LD(ldx, "ldx"); REQ_LD(ldy, "ldy");
    GLF_REQUIRE(c <= APPLY_MAX_C, GLF_ERR_UNSUPPORTED, "s16_bn_apply: C must be <= %d", APPLY_MAX_C);
    GLF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), GLF_ERR_NULL, "s16_bn_apply: running_mean/var must both be set or both NULL");
    if (residual) { REQ_AL(residual, "residual"); REQ_LD(ldr, "ldr"); }
    const long long total8 = (long long)rows * (c / 8);
    hipLaunchKernelGGL(s16_bn_apply_kernel, dim3(stream_grid(total8, 256)), dim3(256), (size_t)2 * c * sizeof(float), glf::S(s),
                       static_cast<const u16*>(x), ldx, static_cast<const u16*>(residual), ldr, static_cast<u16*>(y), ldy, sums, rows, c, eps, momentum,
                       gamma, beta, mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), total8, c / 8, relu,
                       relu_mask);
    return glf::check_launch("s16_bn_apply");
}

extern "C" int glf_s16_bn_bwd(const void* dy, int lddy, const void* dy2, int lddy2, const void* x, int ldx,
                              const float* mean, const float* invstd, const float* gamma, const float* beta,
                              void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta,
                              int rows, int c, int relu, int training, double* sums, const uint8_t* relu_mask, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && x && mean && invstd && gamma && dx && sums, GLF_ERR_NULL, "s16_bn_bwd: null argument");
    GLF_REQUIRE(!relu || beta || relu_mask, GLF_ERR_NULL, "s16_bn_bwd: relu != 0 needs relu_mask or beta (to recompute the sign from x)");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "s16_bn_bwd: rows must be > 0");
    REQ_C8(c); REQ_AL(dy, "dy"); REQ_AL(x, "x"); REQ_AL(dx, "dx"); REQ_LD(lddy, "lddy"); REQ_LD(ldx, "ldx"); REQ_LD(lddx, "lddx");
    GLF_REQUIRE(c <= APPLY_MAX_C, GLF_ERR_UNSUPPORTED, "s16_bn_bwd: C must be <= %d", APPLY_MAX_C);
    if (dres) { REQ_AL(dres, "dres"); REQ_LD(lddres, "lddres"); }
    if (dy2) { REQ_AL(dy2, "dy2"); REQ_LD(lddy2, "lddy2"); }
    const OpBnBwd16 op{static_cast<const u16*>(dy), lddy, static_cast<const u16*>(dy2), lddy2, static_cast<const u16*>(x), ldx, mean, invstd, gamma, beta,
                       relu, relu_mask, c / 8};
    if (int rc = launch_bnbwd_reduce16(op, rows, c, sums, glf::S(s))) return rc;
    const long long total8 = (long long)rows * (c / 8);
    hipLaunchKernelGGL(s16_bnbwd_apply_kernel, dim3(stream_grid(total8, 256)), dim3(256), (size_t)2 * c * sizeof(float), glf::S(s),
                       static_cast<const u16*>(dy), lddy, static_cast<const u16*>(dy2), lddy2, static_cast<const u16*>(x), ldx, mean, invstd, gamma, beta,
                       sums, static_cast<u16*>(dx), lddx, static_cast<u16*>(dres), lddres, dgamma, dbeta, total8, c / 8, c, relu, training,
                       1.0f / (float)rows, relu_mask);
    return glf::check_launch("s16_bn_bwd_apply");
}

extern "C" int glf_s16_colsum(const void* dy, int lddy, float* db, int rows, int c, double* workspace, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && db && workspace, GLF_ERR_NULL, "s16_colsum: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "s16_colsum: rows must be > 0");
    REQ_C8(c); REQ_AL(dy, "dy"); REQ_LD(lddy, "lddy");
    hipError_t e = hipMemsetAsync(workspace, 0, (size_t)2 * c * sizeof(double), glf::S(s));
    if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "s16_colsum: hipMemsetAsync: %s", hipGetErrorString(e));
    if (int rc = launch_colreduce16(OpColsum16{static_cast<const u16*>(dy), lddy}, rows, c, workspace, glf::S(s))) return rc;
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3((c + 255) / 256), dim3(256), 0, glf::S(s), workspace, db, c);
    return glf::check_launch("s16_colsum");
}

extern "C" int glf_s16_bn_res_ln_fwd(const void* w, const void* x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                                     const float* bn_beta, const float* ln_gamma, const float* ln_beta, float ln_eps, void* z,
                                     float* row_mean, float* row_rstd, int rows, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(w && x && bn_mean && bn_invstd && bn_gamma && bn_beta && ln_gamma && ln_beta && z && row_mean && row_rstd, GLF_ERR_NULL,
                "s16_bn_res_ln_fwd: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "s16_bn_res_ln_fwd: rows must be > 0");
    REQ_C8(c); GLF_REQUIRE(c <= 64 * 8 * LN_NV, GLF_ERR_UNSUPPORTED, "s16_bn_res_ln: C must be <= %d", 64 * 8 * LN_NV);
    REQ_AL(w, "w"); REQ_AL(x, "x"); REQ_AL(z, "z");
    hipLaunchKernelGGL(s16_bn_res_ln_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, glf::S(s), static_cast<const u16*>(w), static_cast<const u16*>(x),
                       bn_mean, bn_invstd, bn_gamma, bn_beta, ln_gamma, ln_beta, ln_eps, static_cast<u16*>(z), row_mean, row_rstd, rows, c);
    return glf::check_launch("s16_bn_res_ln_fwd");
}

extern "C" size_t glf_s16_bn_res_ln_workspace(int rows, int c) {
    return (size_t)((rows + LN_ROWS - 1) / LN_ROWS) * 2 * (size_t)(c > 0 ? c : 0) * sizeof(float);
}

extern "C" int glf_s16_bn_res_ln_bwd(const void* dz, const void* w, const void* x, const float* bn_mean, const float* bn_invstd,
                                     const float* bn_gamma, const float* bn_beta, const float* ln_gamma, const float* row_mean,
                                     const float* row_rstd, void* du, float* dln_gamma, float* dln_beta, int rows, int c,
                                     float* workspace, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dz && w && x && bn_mean && bn_invstd && bn_gamma && bn_beta && ln_gamma && row_mean && row_rstd && du && dln_gamma && dln_beta &&
                workspace, GLF_ERR_NULL, "s16_bn_res_ln_bwd: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "s16_bn_res_ln_bwd: rows must be > 0");
    REQ_C8(c); GLF_REQUIRE(c <= 64 * 8 * LN_NV, GLF_ERR_UNSUPPORTED, "s16_bn_res_ln: C must be <= %d", 64 * 8 * LN_NV);
    REQ_AL(dz, "dz"); REQ_AL(w, "w"); REQ_AL(x, "x"); REQ_AL(du, "du");
    const int nslab = (rows + LN_ROWS - 1) / LN_ROWS;
    hipLaunchKernelGGL(s16_bn_res_ln_bwd_kernel, dim3(nslab), dim3(256), 0, glf::S(s), static_cast<const u16*>(dz), static_cast<const u16*>(w),
                       static_cast<const u16*>(x), bn_mean, bn_invstd, bn_gamma, bn_beta, ln_gamma, row_mean, row_rstd, static_cast<u16*>(du), workspace,
                       rows, c);
    if (int rc = glf::check_launch("s16_bn_res_ln_bwd")) return rc;
    hipLaunchKernelGGL(s16_ln_param_finalize, dim3((2 * c + 15) / 16), dim3(256), 0, glf::S(s), workspace, nslab, c, dln_gamma, dln_beta);
    return glf::check_launch("s16_ln_param_finalize");
}

extern "C" int glf_s16_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int n, int h, int w, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y && idx, GLF_ERR_NULL, "s16_maxpool_fwd: null argument");
    GLF_REQUIRE(n > 0 && h > 0 && w > 0, GLF_ERR_BAD_SHAPE, "s16_maxpool_fwd: bad extents");
    REQ_C8(c); REQ_AL(x, "x"); REQ_AL(y, "y");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    const long long total = (long long)n * ho * wo * (c / 8);
    hipLaunchKernelGGL(s16_maxpool_fwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, glf::S(s), static_cast<const u16*>(x), static_cast<u16*>(y), idx,
                       n, h, w, c / 8, ho, wo);
    return glf::check_launch("s16_maxpool_fwd");
}
extern "C" int glf_s16_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int n, int h, int w, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && dx && idx, GLF_ERR_NULL, "s16_maxpool_bwd: null argument");
    GLF_REQUIRE(n > 0 && h > 0 && w > 0, GLF_ERR_BAD_SHAPE, "s16_maxpool_bwd: bad extents");
    REQ_C8(c); REQ_AL(dy, "dy"); REQ_AL(dx, "dx");
    const int ho = (h - 1) / 2 + 1, wo = (w - 1) / 2 + 1;
    const long long total = (long long)n * h * w * (c / 8);
    hipLaunchKernelGGL(s16_maxpool_bwd_kernel, dim3(stream_grid(total, 256)), dim3(256), 0, glf::S(s), static_cast<const u16*>(dy), idx, static_cast<u16*>(dx),
                       n, h, w, c / 8, ho, wo);
    return glf::check_launch("s16_maxpool_bwd");
}

extern "C" int glf_s16_sum_rows(const void* x, int ldx, void* y, int y_dtype, float scale, int n, int p, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y, GLF_ERR_NULL, "s16_sum_rows: null argument");
    GLF_REQUIRE(n > 0 && p > 0, GLF_ERR_BAD_SHAPE, "s16_sum_rows: bad extents");
    REQ_C8(c); REQ_AL(x, "x"); REQ_LD(ldx, "ldx");
    GLF_REQUIRE(y_dtype == GLF_DT_F32 || y_dtype == GLF_DT_BF16, GLF_ERR_UNSUPPORTED, "s16_sum_rows: y_dtype must be GLF_DT_F32 or GLF_DT_BF16");
    hipLaunchKernelGGL(s16_sum_rows_kernel, dim3((c + 255) / 256, n), dim3(256), 0, glf::S(s), static_cast<const u16*>(x), ldx, y, y_dtype == GLF_DT_F32, scale, p, c);
    return glf::check_launch("s16_sum_rows");
}
extern "C" int glf_s16_bcast_rows(const void* x, int x_dtype, void* y, int ldy, float scale, int n, int p, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y, GLF_ERR_NULL, "s16_bcast_rows: null argument");
    GLF_REQUIRE(n > 0 && p > 0, GLF_ERR_BAD_SHAPE, "s16_bcast_rows: bad extents");
    REQ_C8(c); REQ_AL(x, "x"); REQ_AL(y, "y"); REQ_LD(ldy, "ldy");
    const long long total8 = (long long)n * p * (c / 8);
    GLF_REQUIRE(x_dtype == GLF_DT_F32 || x_dtype == GLF_DT_BF16, GLF_ERR_UNSUPPORTED, "s16_bcast_rows: x_dtype must be GLF_DT_F32 or GLF_DT_BF16");
    hipLaunchKernelGGL(s16_bcast_rows_kernel, dim3(stream_grid(total8, 256)), dim3(256), 0, glf::S(s), x, x_dtype == GLF_DT_F32, static_cast<u16*>(y), ldy,
                       scale, p, c / 8, total8);
    return glf::check_launch("s16_bcast_rows");
}

extern "C" int glf_s16_dropout(const void* x, void* y, int64_t numel, float p, uint64_t seed, const uint64_t* step_counter, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y, GLF_ERR_NULL, "s16_dropout: null argument");
    GLF_REQUIRE(numel > 0 && numel % 8 == 0 && p >= 0.f && p < 1.f, GLF_ERR_BAD_SHAPE, "s16_dropout: numel must be a positive multiple of 8, 0 <= p < 1");
    REQ_AL(x, "x"); REQ_AL(y, "y");
    hipLaunchKernelGGL(s16_dropout_kernel, dim3(stream_grid(numel / 8, 256)), dim3(256), 0, glf::S(s), static_cast<const u16*>(x), static_cast<u16*>(y),
                       (long long)(numel / 8), p, 1.0f / (1.0f - p), (unsigned long long)seed, reinterpret_cast<const unsigned long long*>(step_counter));
    return glf::check_launch("s16_dropout");
}

extern "C" int glf_s16_relu_fwd(const void* x, void* y, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y, GLF_ERR_NULL, "s16_relu_fwd: null argument");
    GLF_REQUIRE(numel > 0 && numel % 8 == 0, GLF_ERR_BAD_SHAPE, "s16_relu_fwd: numel must be a positive multiple of 8");
    REQ_AL(x, "x"); REQ_AL(y, "y");
    hipLaunchKernelGGL(s16_relu_fwd_kernel, dim3(stream_grid(numel / 8, 256)), dim3(256), 0, glf::S(s), static_cast<const u16*>(x), static_cast<u16*>(y), (long long)(numel / 8));
    return glf::check_launch("s16_relu_fwd");
}
extern "C" int glf_s16_relu_bwd(const void* dy, const void* y, void* dx, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && y && dx, GLF_ERR_NULL, "s16_relu_bwd: null argument");
    GLF_REQUIRE(numel > 0 && numel % 8 == 0, GLF_ERR_BAD_SHAPE, "s16_relu_bwd: numel must be a positive multiple of 8");
    REQ_AL(dy, "dy"); REQ_AL(y, "y"); REQ_AL(dx, "dx");
    hipLaunchKernelGGL(s16_relu_bwd_kernel, dim3(stream_grid(numel / 8, 256)), dim3(256), 0, glf::S(s), static_cast<const u16*>(dy), static_cast<const u16*>(y),
                       static_cast<u16*>(dx), (long long)(numel / 8));
    return glf::check_launch("s16_relu_bwd");
}
extern "C" int glf_s16_axpby(const void* x, const void* y, void* out, float a, float b, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y && out, GLF_ERR_NULL, "s16_axpby: null argument");
    GLF_REQUIRE(numel > 0 && numel % 8 == 0, GLF_ERR_BAD_SHAPE, "s16_axpby: numel must be a positive multiple of 8");
    REQ_AL(x, "x"); REQ_AL(y, "y"); REQ_AL(out, "out");
    hipLaunchKernelGGL(s16_axpby_kernel, dim3(stream_grid(numel / 8, 256)), dim3(256), 0, glf::S(s), static_cast<const u16*>(x), static_cast<const u16*>(y),
                       static_cast<u16*>(out), a, b, (long long)(numel / 8));
    return glf::check_launch("s16_axpby");
}

extern "C" int glf_s16_gate_fwd(const float* cls, int ncls, const float* ctr, const void* f, void* y, float* a, int32_t* argmax, float weight,
                                int rows, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(cls && ctr && f && y && a && argmax, GLF_ERR_NULL, "s16_gate_fwd: null argument");
    GLF_REQUIRE(rows > 0 && ncls > 0, GLF_ERR_BAD_SHAPE, "s16_gate_fwd: bad extents");
    REQ_C8(c); REQ_AL(f, "f"); REQ_AL(y, "y");
    hipLaunchKernelGGL(s16_gate_fwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, glf::S(s), cls, ncls, ctr, static_cast<const u16*>(f), static_cast<u16*>(y), a,
                       argmax, weight, rows, c);
    return glf::check_launch("s16_gate_fwd");
}
extern "C" int glf_s16_gate_bwd(const void* dy, const void* f, const float* cls, int ncls, const float* ctr, const float* a, const int32_t* argmax,
                                float weight, void* df, float* dcls, float* dctr, int rows, int c, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && f && cls && ctr && a && argmax && df && dcls && dctr, GLF_ERR_NULL, "s16_gate_bwd: null argument");
    GLF_REQUIRE(rows > 0 && ncls > 0, GLF_ERR_BAD_SHAPE, "s16_gate_bwd: bad extents");
    REQ_C8(c); REQ_AL(dy, "dy"); REQ_AL(f, "f"); REQ_AL(df, "df");
    hipLaunchKernelGGL(s16_gate_bwd_kernel, dim3((rows + 3) / 4), dim3(256), 0, glf::S(s), static_cast<const u16*>(dy), static_cast<const u16*>(f), cls, ncls,
                       ctr, a, argmax, weight, static_cast<u16*>(df), dcls, dctr, rows, c);
    return glf::check_launch("s16_gate_bwd");
}

extern "C" int glf_s16_add_frames(const void* a, int64_t a_fs, const void* b, int64_t b_fs, void* dst, int64_t dst_fs, int n, int64_t inner,
                                  glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(a && b && dst, GLF_ERR_NULL, "s16_add_frames: null argument");
    GLF_REQUIRE(n > 0 && inner > 0 && inner % 8 == 0 && a_fs % 8 == 0 && b_fs % 8 == 0 && dst_fs % 8 == 0, GLF_ERR_BAD_SHAPE,
                "s16_add_frames: inner and the frame strides must be multiples of 8");
    REQ_AL(a, "a"); REQ_AL(b, "b"); REQ_AL(dst, "dst");
    const long long total8 = (long long)n * (inner / 8);
    hipLaunchKernelGGL(s16_add_frames_kernel, dim3(stream_grid(total8, 256)), dim3(256), 0, glf::S(s), static_cast<const u16*>(a), (long long)a_fs,
                       static_cast<const u16*>(b), (long long)b_fs, static_cast<u16*>(dst), (long long)dst_fs, (long long)(inner / 8), total8);
    return glf::check_launch("s16_add_frames");
}
extern "C" int glf_s16_add_n(const void* const* inputs, int k, void* out, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(inputs && out, GLF_ERR_NULL, "s16_add_n: null argument");
    GLF_REQUIRE(k >= 1 && k <= 8 && numel > 0 && numel % 8 == 0, GLF_ERR_BAD_SHAPE, "s16_add_n: 1 <= k <= 8, numel a positive multiple of 8");
    AddN16 in;
    for (int i = 0; i < 8; ++i) in.p[i] = static_cast<const u16*>(inputs[i < k ? i : 0]);
    for (int i = 0; i < k; ++i) { GLF_REQUIRE(inputs[i], GLF_ERR_NULL, "s16_add_n: null input"); REQ_AL(inputs[i], "input"); }
    REQ_AL(out, "out");
    hipLaunchKernelGGL(s16_add_n_kernel, dim3(stream_grid(numel / 8, 256)), dim3(256), 0, glf::S(s), in, k, static_cast<u16*>(out), (long long)(numel / 8));
    return glf::check_launch("s16_add_n");
}

extern "C" int glf_s16_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t numel, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(src && dst, GLF_ERR_NULL, "s16_cast: null argument");
    GLF_REQUIRE(numel > 0 && numel % 8 == 0, GLF_ERR_BAD_SHAPE, "s16_cast: numel must be a positive multiple of 8");
    REQ_AL(src, "src"); REQ_AL(dst, "dst");
    if (src_dtype == GLF_DT_BF16 && dst_dtype == GLF_DT_F32)
        hipLaunchKernelGGL(s16_to_f32_kernel, dim3(stream_grid(numel / 8, 256)), dim3(256), 0, glf::S(s), static_cast<const u16*>(src), static_cast<float*>(dst),
                           (long long)(numel / 8));
    else if (src_dtype == GLF_DT_F32 && dst_dtype == GLF_DT_BF16)
        hipLaunchKernelGGL(s16_from_f32_kernel, dim3(stream_grid(numel / 8, 256)), dim3(256), 0, glf::S(s), static_cast<const float*>(src), static_cast<u16*>(dst),
                           (long long)(numel / 8));
    else
        return glf::fail(GLF_ERR_UNSUPPORTED, "s16_cast: built for bf16 -> f32 and f32 -> bf16");
    return glf::check_launch("s16_cast");
}

extern "C" int glf_s16_transpose2d(const void* src, void* dst, int rows, int cols, int batch, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(src && dst, GLF_ERR_NULL, "s16_transpose2d: null argument");
    GLF_REQUIRE(rows > 0 && cols > 0 && batch > 0 && batch <= 65535, GLF_ERR_BAD_SHAPE, "s16_transpose2d: bad extents");
    hipLaunchKernelGGL(s16_transpose2d_kernel, dim3((cols + 63) / 64, (rows + 63) / 64, batch), dim3(256), 0, glf::S(s), static_cast<const u16*>(src),
                       static_cast<u16*>(dst), rows, cols);
    return glf::check_launch("s16_transpose2d");
}
