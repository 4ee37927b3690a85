// Normalisation kernels of the GL-Fusion path (HBM-bound; channels-last [rows][C], C % 4 == 0).
//   * column reductions (BatchNorm statistics, BatchNorm / LayerNorm / bias gradients) share one
//     two-stage scheme: stage 1 streams a slice of rows per workgroup with 16-byte loads and
//     accumulates in f64 (sum / sum-of-squares cancellation-free), stage 2 folds the per-slice
//     partials per channel.  No atomics => bitwise reproducible.
//   * BatchNorm apply (+residual, +ReLU) and the backward apply are single streaming passes.
//   * the TPAVI tail z = LayerNorm_C(BN(W_z y) + x) runs one wavefront per row with shuffle
//     reductions only (no LDS, no barrier).
#include "glf_common.h"
#include "split_f16.h"

namespace {

constexpr int RT = 256;               // threads of a reduction workgroup
constexpr int MAX_SLICES = 1024;

// row slices of the stage-1 grid: enough workgroups to stream at the HBM rate (grid = slices x channel blocks of
// up to 1024 channels) while the f64 partials stay small (2 x slices x C x 8 B; 33 MB per BatchNorm at C = 2048 with
// 1024 slices made the stage-2 kernels 2 % of the step)
__host__ __device__ inline int n_slices_c(int rows, int c) {
    int s = (rows + 31) / 32;
    int cap = 524288 / (c > 0 ? c : 1);
    if (cap < 128) cap = 128;
    if (cap > MAX_SLICES) cap = MAX_SLICES;
    return s < 1 ? 1 : (s > cap ? cap : s);
}

struct Coef { const float* mean; const float* invstd; const float* gamma; const float* beta; };

// ---- functors: two values per element --------------------------------------------------
struct OpStats {          // (x, x^2)
    const float* x; int ldx;
    __device__ void operator()(int r, int c, float4& a, float4& b) const {
        a = *reinterpret_cast<const float4*>(x + (long long)r * ldx + c);
        b = make_float4(a.x * a.x, a.y * a.y, a.z * a.z, a.w * a.w);
    }
};
struct OpColsum {         // (dy, 0)
    const float* dy; int ld;
    __device__ void operator()(int r, int c, float4& a, float4& b) const {
        a = *reinterpret_cast<const float4*>(dy + (long long)r * ld + c);
        b = make_float4(0.f, 0.f, 0.f, 0.f);
    }
};
// the forward value of one element; ONE definition so that the ReLU mask recomputed in backward (y == NULL) is the
// mask the forward applied
__device__ __forceinline__ float bn_val(float x, float mu, float is, float ga, float be) { return (x - mu) * is * ga + be; }

struct OpBnBwd {          // (dy', dy' * xhat), dy' = dy * (y > 0) when relu; y == NULL: the mask is recomputed from x
    const float* dy; int lddy; const float* x; int ldx; const float* y; int ldy;
    const float* mean; const float* invstd; const float* gamma; const float* beta; int relu;
    const unsigned char* mask; int c4;          // relu with a residual: the forward's sign bits, one byte per float4 (instead of y)
    const float* dy2; int lddy2;                // second addend of the incoming gradient (a fan-in not yet summed), or null
    __device__ void operator()(int r, int c, float4& a, float4& b) const {
        a = *reinterpret_cast<const float4*>(dy + (long long)r * lddy + c);
        if (dy2) {
            const float4 a2 = *reinterpret_cast<const float4*>(dy2 + (long long)r * lddy2 + c);
            a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w;
        }
        const float4 xx = *reinterpret_cast<const float4*>(x + (long long)r * ldx + c);
        const float4 mu = *reinterpret_cast<const float4*>(mean + c);
        const float4 is = *reinterpret_cast<const float4*>(invstd + c);
        if (relu && mask) {
            const unsigned m = mask[(long long)r * c4 + (c >> 2)];
            a.x = (m & 1u) ? a.x : 0.f; a.y = (m & 2u) ? a.y : 0.f; a.z = (m & 4u) ? a.z : 0.f; a.w = (m & 8u) ? a.w : 0.f;
        } else if (relu) {
            float4 yy;
            if (y) {
                yy = *reinterpret_cast<const float4*>(y + (long long)r * ldy + c);
            } else {
                const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
                const float4 be = *reinterpret_cast<const float4*>(beta + c);
                yy = make_float4(bn_val(xx.x, mu.x, is.x, ga.x, be.x), bn_val(xx.y, mu.y, is.y, ga.y, be.y),
                                 bn_val(xx.z, mu.z, is.z, ga.z, be.z), bn_val(xx.w, mu.w, is.w, ga.w, be.w));
            }
            a.x = yy.x > 0.f ? a.x : 0.f; a.y = yy.y > 0.f ? a.y : 0.f;
            a.z = yy.z > 0.f ? a.z : 0.f; a.w = yy.w > 0.f ? a.w : 0.f;
        }
        b = make_float4(a.x * (xx.x - mu.x) * is.x, a.y * (xx.y - mu.y) * is.y,
                        a.z * (xx.z - mu.z) * is.z, a.w * (xx.w - mu.w) * is.w);
    }
};
struct OpLnParam {        // (dz * uhat, dz) with u = BN(w) + x and per-row LayerNorm statistics
    const float* dz; const float* w; const float* x; Coef bn; const float* row_mean; const float* row_rstd; int c_total;
    __device__ void operator()(int r, int c, float4& a, float4& b) const {
        const long long o = (long long)r * c_total + c;
        b = *reinterpret_cast<const float4*>(dz + o);
        const float4 ww = *reinterpret_cast<const float4*>(w + o);
        const float4 xx = *reinterpret_cast<const float4*>(x + o);
        const float4 mu = *reinterpret_cast<const float4*>(bn.mean + c);
        const float4 is = *reinterpret_cast<const float4*>(bn.invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(bn.gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(bn.beta + c);
        const float m = row_mean[r], rs = row_rstd[r];
        const float u0 = (ww.x - mu.x) * is.x * ga.x + be.x + xx.x;
        const float u1 = (ww.y - mu.y) * is.y * ga.y + be.y + xx.y;
        const float u2 = (ww.z - mu.z) * is.z * ga.z + be.z + xx.z;
        const float u3 = (ww.w - mu.w) * is.w * ga.w + be.w + xx.w;
        a = make_float4(b.x * (u0 - m) * rs, b.y * (u1 - m) * rs, b.z * (u2 - m) * rs, b.w * (u3 - m) * rs);
    }
};

// stage 1: partial[0][slice][c], partial[1][slice][c]
template <class Op>
__global__ __launch_bounds__(RT) void colreduce_kernel(Op op, int rows, int c, int slices, double* __restrict__ partial) {
    __shared__ double sh_a[RT * 4];
    __shared__ double sh_b[RT * 4];
    const int tid = threadIdx.x;
    const int c4 = c >> 2;
    const int tpr = c4 < RT ? c4 : RT;            // threads per row
    const int rpp = RT / tpr;                     // rows per pass
    const int ct = tid % tpr, rl = tid / tpr;
    const int slice = blockIdx.x;
    const int per = (rows + slices - 1) / slices;
    const int r0 = slice * per, r1 = min(rows, r0 + per);
    for (int cb = blockIdx.y * tpr; cb < c4; cb += gridDim.y * tpr) {
        const int cc = cb + ct;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        if (cc < c4 && rl < rpp) {
            for (int r = r0 + rl; r < r1; r += rpp) {
                float4 a, b;
                op(r, cc * 4, a, b);
                a0 += a.x; a1 += a.y; a2 += a.z; a3 += a.w;
                b0 += b.x; b1 += b.y; b2 += b.z; b3 += b.w;
            }
        }
        sh_a[tid * 4 + 0] = a0; sh_a[tid * 4 + 1] = a1; sh_a[tid * 4 + 2] = a2; sh_a[tid * 4 + 3] = a3;
        sh_b[tid * 4 + 0] = b0; sh_b[tid * 4 + 1] = b1; sh_b[tid * 4 + 2] = b2; sh_b[tid * 4 + 3] = b3;
        __syncthreads();
        if (rl == 0 && cc < c4) {
            for (int q = 1; q < rpp; ++q) {
                const int o = (q * tpr + ct) * 4;
                a0 += sh_a[o]; a1 += sh_a[o + 1]; a2 += sh_a[o + 2]; a3 += sh_a[o + 3];
                b0 += sh_b[o]; b1 += sh_b[o + 1]; b2 += sh_b[o + 2]; b3 += sh_b[o + 3];
            }
            double* pa = partial + (long long)slice * c + cc * 4;
            double* pb = partial + (long long)(slices + slice) * c + cc * 4;
            pa[0] = a0; pa[1] = a1; pa[2] = a2; pa[3] = a3;
            pb[0] = b0; pb[1] = b1; pb[2] = b2; pb[3] = b3;
        }
        __syncthreads();
    }
}

// BatchNorm-backward stage 1 with two more per-channel reductions over the same elements: max |dy'| and max |xhat|.  They cost
// no memory traffic (the values are in registers for the sums) and let the finalize kernel bound max |dx| BEFORE dx is
// written -- which is what allows bn_bwd_apply to emit dx directly as the packed pre-split fp16 image the consuming
// contractions read (the image's power-of-two scale must be known when the first element is written).
// pmax[0][slice][c] = max |dy'|, pmax[1][slice][c] = max |xhat| (floats, behind the f64 partials).
__global__ __launch_bounds__(RT) void bnbwd_reduce_kernel(OpBnBwd op, int rows, int c, int slices, double* __restrict__ partial,
                                                          float* __restrict__ pmax) {
    __shared__ double sh_a[RT * 4];
    __shared__ double sh_b[RT * 4];
    __shared__ float sh_m[RT * 8];
    const int tid = threadIdx.x;
    const int c4 = c >> 2;
    const int tpr = c4 < RT ? c4 : RT;            // threads per row
    const int rpp = RT / tpr;                     // rows per pass
    const int ct = tid % tpr, rl = tid / tpr;
    const int slice = blockIdx.x;
    const int per = (rows + slices - 1) / slices;
    const int r0 = slice * per, r1 = min(rows, r0 + per);
    for (int cb = blockIdx.y * tpr; cb < c4; cb += gridDim.y * tpr) {
        const int cc = cb + ct;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0, b0 = 0, b1 = 0, b2 = 0, b3 = 0;
        float mg[4] = {0.f, 0.f, 0.f, 0.f}, mx[4] = {0.f, 0.f, 0.f, 0.f};
        if (cc < c4 && rl < rpp) {
            const float4 mu = *reinterpret_cast<const float4*>(op.mean + cc * 4);
            const float4 is = *reinterpret_cast<const float4*>(op.invstd + cc * 4);
            for (int r = r0 + rl; r < r1; r += rpp) {
                float4 a, b;
                op(r, cc * 4, a, b);
                a0 += a.x; a1 += a.y; a2 += a.z; a3 += a.w;
                b0 += b.x; b1 += b.y; b2 += b.z; b3 += b.w;
                const float4 xx = *reinterpret_cast<const float4*>(op.x + (long long)r * op.ldx + cc * 4);   // (the load op() made: CSE'd)
                mg[0] = fmaxf(mg[0], fabsf(a.x)); mg[1] = fmaxf(mg[1], fabsf(a.y)); mg[2] = fmaxf(mg[2], fabsf(a.z)); mg[3] = fmaxf(mg[3], fabsf(a.w));
                mx[0] = fmaxf(mx[0], fabsf((xx.x - mu.x) * is.x)); mx[1] = fmaxf(mx[1], fabsf((xx.y - mu.y) * is.y));
                mx[2] = fmaxf(mx[2], fabsf((xx.z - mu.z) * is.z)); mx[3] = fmaxf(mx[3], fabsf((xx.w - mu.w) * is.w));
            }
        }
        sh_a[tid * 4 + 0] = a0; sh_a[tid * 4 + 1] = a1; sh_a[tid * 4 + 2] = a2; sh_a[tid * 4 + 3] = a3;
        sh_b[tid * 4 + 0] = b0; sh_b[tid * 4 + 1] = b1; sh_b[tid * 4 + 2] = b2; sh_b[tid * 4 + 3] = b3;
#pragma unroll
        for (int j = 0; j < 4; ++j) { sh_m[tid * 8 + j] = mg[j]; sh_m[tid * 8 + 4 + j] = mx[j]; }
        __syncthreads();
        if (rl == 0 && cc < c4) {
            for (int q = 1; q < rpp; ++q) {
                const int o = (q * tpr + ct) * 4;
                a0 += sh_a[o]; a1 += sh_a[o + 1]; a2 += sh_a[o + 2]; a3 += sh_a[o + 3];
                b0 += sh_b[o]; b1 += sh_b[o + 1]; b2 += sh_b[o + 2]; b3 += sh_b[o + 3];
#pragma unroll
                for (int j = 0; j < 4; ++j) { mg[j] = fmaxf(mg[j], sh_m[2 * o + j]); mx[j] = fmaxf(mx[j], sh_m[2 * o + 4 + j]); }
            }
            double* pa = partial + (long long)slice * c + cc * 4;
            double* pb = partial + (long long)(slices + slice) * c + cc * 4;
            pa[0] = a0; pa[1] = a1; pa[2] = a2; pa[3] = a3;
            pb[0] = b0; pb[1] = b1; pb[2] = b2; pb[3] = b3;
            *reinterpret_cast<float4*>(pmax + (long long)slice * c + cc * 4) = make_float4(mg[0], mg[1], mg[2], mg[3]);
            *reinterpret_cast<float4*>(pmax + (long long)(slices + slice) * c + cc * 4) = make_float4(mx[0], mx[1], mx[2], mx[3]);
        }
        __syncthreads();
    }
}

// ---- the same reduction WITHOUT a second stage (round 4): per-thread fp32 partial sums over its <= ~16-64 rows (four rows' loads
// in flight at a time), the workgroup's row lanes folded in double through LDS, then ONE f64 atomic per column, statistic and
// workgroup into a zero-filled [2][c] buffer (consecutive threads own consecutive columns: a wave's atomics cover 512 contiguous
// bytes), and one atomicMax per column for the two maxima the packed-gradient bound needs.  bn_bwd_apply_kernel finishes the sums
// in its prologue: two launches per BatchNorm backward instead of three (the finalize kernels were 221 launches of ~14 us per step).
// Column blocks are FUSED_TPR float4 wide (64 columns): what limits these reductions is the number of f64 atomics that land on one
// address -- same-address atomics are serialised by the L2, ~25-30 ns each -- so many narrow column blocks with few row slices each
// beat whole rows (profiles/r04_bn_reduce_sweep.txt, measured on the 16-bit twin of this kernel).
constexpr int FUSED_TPR = 16;
__host__ __device__ inline int fused_slices(int rows, int c) {
    const int c4 = c > 4 ? c / 4 : 1;
    const int tpr = c4 < FUSED_TPR ? c4 : FUSED_TPR;
    const int rpp = RT / tpr;
    int s = (rows + 16 * rpp - 1) / (16 * rpp);
    const int cblocks = (c4 + tpr - 1) / tpr;
    int cap = 1024 / cblocks;
    if (cap < 1) cap = 1;
    return s < 1 ? 1 : (s > cap ? cap : s);
}
// HAS2: second gradient addend; RM: 0 = no ReLU, 1 = sign bytes, 2 = sign recomputed from x, 3 = sign of the saved output y.
// The loads of FOUR rows are issued back to back before any arithmetic (through the generic op() the compiler emitted one
// dependent load -> wait chain per row, the coefficient vectors re-loaded for every row).
template <bool HAS2, int RM>
__global__ __launch_bounds__(RT) void bnbwd_reduce_atomic_kernel(OpBnBwd op, int rows, int c, int slices, double* __restrict__ sums,
                                                                 float* __restrict__ cmax) {
    __shared__ float sh[RT * 8];                  // [statistic][row lane][column] for the sums, reused for the maxima
    const int tid = threadIdx.x;
    const int c4 = c >> 2;
    const int tpr = c4 < FUSED_TPR ? c4 : FUSED_TPR;
    const int rpp = RT / tpr;
    const int ct = tid % tpr, rl = tid / tpr;
    const int slice = blockIdx.x;
    const int per = (rows + slices - 1) / slices;
    const int r0 = slice * per, r1 = min(rows, r0 + per);
    for (int cb = blockIdx.y * tpr; cb < c4; cb += gridDim.y * tpr) {
        const int cc = cb + ct;
        float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa, mg = sa, mx = sa;
        if (cc < c4 && rl < rpp) {
            const int c0 = cc * 4;
            const float4 mu = *reinterpret_cast<const float4*>(op.mean + c0);
            const float4 is = *reinterpret_cast<const float4*>(op.invstd + c0);
            float4 ga = mu, be = mu;
            if (RM == 2) { ga = *reinterpret_cast<const float4*>(op.gamma + c0); be = *reinterpret_cast<const float4*>(op.beta + c0); }
            auto fold = [&](float4 a, const float4 a2, const float4 xx, const float4 yy, unsigned m) __attribute__((always_inline)) {
                if (HAS2) { a.x += a2.x; a.y += a2.y; a.z += a2.z; a.w += a2.w; }
                if (RM == 2) {
                    m = (bn_val(xx.x, mu.x, is.x, ga.x, be.x) > 0.f ? 1u : 0u) | (bn_val(xx.y, mu.y, is.y, ga.y, be.y) > 0.f ? 2u : 0u) |
                        (bn_val(xx.z, mu.z, is.z, ga.z, be.z) > 0.f ? 4u : 0u) | (bn_val(xx.w, mu.w, is.w, ga.w, be.w) > 0.f ? 8u : 0u);
                } else if (RM == 3) {
                    m = (yy.x > 0.f ? 1u : 0u) | (yy.y > 0.f ? 2u : 0u) | (yy.z > 0.f ? 4u : 0u) | (yy.w > 0.f ? 8u : 0u);
                }
                if (RM != 0) { a.x = (m & 1u) ? a.x : 0.f; a.y = (m & 2u) ? a.y : 0.f; a.z = (m & 4u) ? a.z : 0.f; a.w = (m & 8u) ? a.w : 0.f; }
                const float4 xh = make_float4((xx.x - mu.x) * is.x, (xx.y - mu.y) * is.y, (xx.z - mu.z) * is.z, (xx.w - mu.w) * is.w);
                sa.x += a.x; sa.y += a.y; sa.z += a.z; sa.w += a.w;
                sb.x += a.x * (xx.x - mu.x) * is.x; sb.y += a.y * (xx.y - mu.y) * is.y; sb.z += a.z * (xx.z - mu.z) * is.z; sb.w += a.w * (xx.w - mu.w) * is.w;
                if (cmax) {
                    mg.x = fmaxf(mg.x, fabsf(a.x)); mg.y = fmaxf(mg.y, fabsf(a.y)); mg.z = fmaxf(mg.z, fabsf(a.z)); mg.w = fmaxf(mg.w, fabsf(a.w));
                    mx.x = fmaxf(mx.x, fabsf(xh.x)); mx.y = fmaxf(mx.y, fabsf(xh.y)); mx.z = fmaxf(mx.z, fabsf(xh.z)); mx.w = fmaxf(mx.w, fabsf(xh.w));
                }
            };
            int r = r0 + rl;
            for (; r + 3 * rpp < r1; r += 4 * rpp) {
                float4 qa[4], qa2[4], qx[4], qy[4];
                unsigned m[4] = {15u, 15u, 15u, 15u};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const long long row = r + u * rpp;
                    qa[u] = *reinterpret_cast<const float4*>(op.dy + row * op.lddy + c0);
                    if (HAS2) qa2[u] = *reinterpret_cast<const float4*>(op.dy2 + row * op.lddy2 + c0);
                    qx[u] = *reinterpret_cast<const float4*>(op.x + row * op.ldx + c0);
                    if (RM == 3) qy[u] = *reinterpret_cast<const float4*>(op.y + row * op.ldy + c0);
                    if (RM == 1) m[u] = op.mask[row * op.c4 + cc];
                }
                __builtin_amdgcn_sched_barrier(0);       // every load of the four rows is in flight before the first use
#pragma unroll
                for (int u = 0; u < 4; ++u) fold(qa[u], HAS2 ? qa2[u] : qa[u], qx[u], RM == 3 ? qy[u] : qx[u], m[u]);
            }
            for (; r < r1; r += rpp) {
                const long long row = r;
                const float4 qa = *reinterpret_cast<const float4*>(op.dy + row * op.lddy + c0);
                const float4 qa2 = HAS2 ? *reinterpret_cast<const float4*>(op.dy2 + row * op.lddy2 + c0) : qa;
                const float4 qx = *reinterpret_cast<const float4*>(op.x + row * op.ldx + c0);
                const float4 qy = RM == 3 ? *reinterpret_cast<const float4*>(op.y + row * op.ldy + c0) : qx;
                const unsigned m = RM == 1 ? op.mask[row * op.c4 + cc] : 15u;
                fold(qa, qa2, qx, qy, m);
            }
        }
        const int ncol = tpr * 4;
        for (int pass = 0; pass < (cmax ? 2 : 1); ++pass) {
            if (rl < rpp) {
                const float4 u = pass == 0 ? sa : mg, v = pass == 0 ? sb : mx;
                float* d0 = sh + (0 * rpp + rl) * ncol + ct * 4;
                float* d1 = sh + (1 * rpp + rl) * ncol + ct * 4;
                d0[0] = u.x; d0[1] = u.y; d0[2] = u.z; d0[3] = u.w;
                d1[0] = v.x; d1[1] = v.y; d1[2] = v.z; d1[3] = v.w;
            }
            __syncthreads();
            for (int idx = tid; idx < 2 * ncol; idx += RT) {
                const int st = idx / ncol, col = idx - st * ncol;
                const int gc = cb * 4 + col;
                if (gc < c) {
                    if (pass == 0) {
                        double d = 0;
                        for (int q = 0; q < rpp; ++q) d += sh[(st * rpp + q) * ncol + col];
                        atomicAdd(sums + st * c + gc, d);
                    } else {
                        float m = 0.f;
                        for (int q = 0; q < rpp; ++q) m = fmaxf(m, sh[(st * rpp + q) * ncol + col]);
                        if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(cmax + st * c + gc), __float_as_uint(m));
                    }
                }
            }
            __syncthreads();
        }
    }
}

// stage 2: fold the per-slice partials.  One workgroup = 16 channels x 16 slice lanes (a serial loop over
// up to 1024 slices per channel on a single thread was 13% of the whole step in the first profile).
constexpr int FIN_CH = 16, FIN_LANES = 16;

__device__ __forceinline__ void fold_partials(const double* __restrict__ partial, int slices, int c, int ch, int lane,
                                              double& s, double& q, double* sh) {
    s = 0; q = 0;
    if (ch < c) {
        double s1 = 0, q1 = 0, s2 = 0, q2 = 0, s3 = 0, q3 = 0;      // four independent chains: the loop is latency-bound
        int i = lane;
        for (; i + 3 * FIN_LANES < slices; i += 4 * FIN_LANES) {
            s += partial[(long long)i * c + ch];                     q += partial[(long long)(slices + i) * c + ch];
            s1 += partial[(long long)(i + FIN_LANES) * c + ch];      q1 += partial[(long long)(slices + i + FIN_LANES) * c + ch];
            s2 += partial[(long long)(i + 2 * FIN_LANES) * c + ch];  q2 += partial[(long long)(slices + i + 2 * FIN_LANES) * c + ch];
            s3 += partial[(long long)(i + 3 * FIN_LANES) * c + ch];  q3 += partial[(long long)(slices + i + 3 * FIN_LANES) * c + ch];
        }
        for (; i < slices; i += FIN_LANES) {
            s += partial[(long long)i * c + ch];
            q += partial[(long long)(slices + i) * c + ch];
        }
        s += (s1 + s2) + s3; q += (q1 + q2) + q3;
    }
    const int t = threadIdx.x;
    sh[t] = s; sh[256 + t] = q;
    __syncthreads();
    if (lane == 0) {
        for (int l = 1; l < FIN_LANES; ++l) { s += sh[t + l * FIN_CH]; q += sh[256 + t + l * FIN_CH]; }
    }
}

__global__ __launch_bounds__(256) void bn_stats_finalize(const double* __restrict__ partial, int slices, int c, int rows, float eps,
                                                         float momentum, float* mean, float* invstd, float* rmean, float* rvar, long long* nbt) {
    __shared__ double sh[512];
    const int cx = threadIdx.x % FIN_CH, lane = threadIdx.x / FIN_CH;
    const int ch = blockIdx.x * FIN_CH + cx;
    double s, q;
    fold_partials(partial, slices, c, ch, lane, s, q, sh);
    if (lane != 0 || ch >= c) return;
    if (ch == 0 && nbt) *nbt += 1;
    const double m = s / rows;
    double var = q / rows - m * m;
    if (var < 0) var = 0;
    mean[ch] = (float)m;
    invstd[ch] = (float)(1.0 / sqrt(var + (double)eps));
    if (rmean) {
        const double unb = rows > 1 ? var * rows / (rows - 1) : var;
        rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)m;
        rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
    }
}

// out_a[ch] = sum of first partial, out_b[ch] = sum of second (either may be NULL)
__global__ __launch_bounds__(256) void sum_finalize(const double* __restrict__ partial, int slices, int c, float* out_a, float* out_b) {
    __shared__ double sh[512];
    const int cx = threadIdx.x % FIN_CH, lane = threadIdx.x / FIN_CH;
    const int ch = blockIdx.x * FIN_CH + cx;
    double s, q;
    fold_partials(partial, slices, c, ch, lane, s, q, sh);
    if (lane != 0 || ch >= c) return;
    if (out_a) out_a[ch] = (float)s;
    if (out_b) out_b[ch] = (float)q;
}

// BatchNorm-backward stage 2 for the packed-dx path: the two sums, and an upper bound of max |dx| over the whole tensor into
// *bound (a zeroed device float; non-negative floats order like their bit patterns):
//   training: |dx| = |gamma invstd| |g - (s1 + xhat s2) / n| <= |gamma invstd| (max|g| + (|s1| + max|xhat| |s2|) / n)   per channel
//   eval    : |dx| = |gamma invstd| |g|
// (x 1.0001: the apply kernel rounds its fp32 expression differently; the image's power-of-two scale leaves two bits of
// fp16 headroom above the bound anyway)
__global__ __launch_bounds__(256) void bnbwd_finalize(const double* __restrict__ partial, const float* __restrict__ pmax, int slices, int c,
                                                      const float* __restrict__ gamma, const float* __restrict__ invstd, float inv_n,
                                                      int training, float* out_a, float* out_b, float* __restrict__ bound) {
    __shared__ double sh[512];
    __shared__ float shm[512];
    const int cx = threadIdx.x % FIN_CH, lane = threadIdx.x / FIN_CH;
    const int ch = blockIdx.x * FIN_CH + cx;
    double s, q;
    fold_partials(partial, slices, c, ch, lane, s, q, sh);
    float mg = 0.f, mx = 0.f;
    if (ch < c) {
        float g1 = 0.f, x1 = 0.f, g2 = 0.f, x2 = 0.f, g3 = 0.f, x3 = 0.f;       // independent chains: the loop is latency-bound
        int i = lane;
        for (; i + 3 * FIN_LANES < slices; i += 4 * FIN_LANES) {
            mg = fmaxf(mg, pmax[(long long)i * c + ch]);                      mx = fmaxf(mx, pmax[(long long)(slices + i) * c + ch]);
            g1 = fmaxf(g1, pmax[(long long)(i + FIN_LANES) * c + ch]);        x1 = fmaxf(x1, pmax[(long long)(slices + i + FIN_LANES) * c + ch]);
            g2 = fmaxf(g2, pmax[(long long)(i + 2 * FIN_LANES) * c + ch]);    x2 = fmaxf(x2, pmax[(long long)(slices + i + 2 * FIN_LANES) * c + ch]);
            g3 = fmaxf(g3, pmax[(long long)(i + 3 * FIN_LANES) * c + ch]);    x3 = fmaxf(x3, pmax[(long long)(slices + i + 3 * FIN_LANES) * c + ch]);
        }
        for (; i < slices; i += FIN_LANES) {
            mg = fmaxf(mg, pmax[(long long)i * c + ch]);
            mx = fmaxf(mx, pmax[(long long)(slices + i) * c + ch]);
        }
        mg = fmaxf(fmaxf(mg, g1), fmaxf(g2, g3)); mx = fmaxf(fmaxf(mx, x1), fmaxf(x2, x3));
    }
    shm[threadIdx.x] = mg; shm[256 + threadIdx.x] = mx;
    __syncthreads();
    if (lane != 0 || ch >= c) return;
    for (int l = 1; l < FIN_LANES; ++l) { mg = fmaxf(mg, shm[threadIdx.x + l * FIN_CH]); mx = fmaxf(mx, shm[256 + threadIdx.x + l * FIN_CH]); }
    if (out_a) out_a[ch] = (float)s;
    if (out_b) out_b[ch] = (float)q;
    const float k = fabsf(gamma[ch] * invstd[ch]);
    const float b = 1.0001f * k * (training ? mg + inv_n * (fabsf((float)s) + mx * fabsf((float)q)) : mg);
    if (b > 0.f && b < 3.0e38f) atomicMax(reinterpret_cast<unsigned*>(bound), __float_as_uint(b));
}

// running-statistics update of a train-mode BatchNorm replayed from its saved batch statistics (mean, invstd): what a second
// forward over the same input would have done to running_mean / running_var / num_batches_tracked
__global__ void bn_replay_kernel(const float* __restrict__ mean, const float* __restrict__ invstd, int rows, float eps, float momentum,
                                 float* rmean, float* rvar, long long* nbt, int c) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch == 0 && nbt) *nbt += 1;
    if (ch >= c) return;
    const double is = (double)invstd[ch];
    double var = 1.0 / (is * is) - (double)eps;
    if (var < 0) var = 0;
    const double unb = rows > 1 ? var * rows / (rows - 1) : var;
    rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * mean[ch];
    rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
}

__global__ void bn_eval_coeffs_kernel(const float* rm, const float* rv, float eps, float* mean, float* invstd, int c) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= c) return;
    mean[ch] = rm[ch];
    invstd[ch] = 1.0f / sqrtf(rv[ch] + eps);
}

// ---- streaming applies ------------------------------------------------------------------

__global__ __launch_bounds__(256) void bn_apply_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ res, int ldr,
                                                       float* __restrict__ y, int ldy, Coef k, long long total4, int c4, int relu,
                                                       float* __restrict__ amax_out, unsigned char* __restrict__ mask) {
    float am = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        const float4 xx = *reinterpret_cast<const float4*>(x + r * ldx + c);
        const float4 mu = *reinterpret_cast<const float4*>(k.mean + c);
        const float4 is = *reinterpret_cast<const float4*>(k.invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(k.gamma + c);
        const float4 be = *reinterpret_cast<const float4*>(k.beta + c);
        float4 o = make_float4(bn_val(xx.x, mu.x, is.x, ga.x, be.x), bn_val(xx.y, mu.y, is.y, ga.y, be.y),
                               bn_val(xx.z, mu.z, is.z, ga.z, be.z), bn_val(xx.w, mu.w, is.w, ga.w, be.w));
        if (res) {
            const float4 rr = *reinterpret_cast<const float4*>(res + r * ldr + c);
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        if (mask) mask[i] = (unsigned char)((o.x > 0.f) | ((o.y > 0.f) << 1) | ((o.z > 0.f) << 2) | ((o.w > 0.f) << 3));
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        *reinterpret_cast<float4*>(y + r * ldy + c) = o;
        am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
    }
    if (amax_out) block_amax(am, amax_out);
}

// BatchNorm apply with the statistics FINISHED IN THE KERNEL: the producing contraction's epilogue left (sum x, sum x^2) per
// channel in `sums` (glf_gemm_params.colstats); every workgroup turns them into mean / invstd for all channels in LDS (the
// same double-precision expressions as bn_stats_finalize: results are bit-identical to glf_bn_stats_from_sums + glf_bn_apply),
// workgroup 0 also writes them out for the backward pass and updates the running statistics.  One launch instead of two per
// BatchNorm, and no tiny kernel on the dependent chain conv -> statistics -> apply -> next conv.
constexpr int APPLY_MAX_C = 4096;
__global__ __launch_bounds__(256) void bn_apply_sums_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ res, int ldr,
                                                            float* __restrict__ y, int ldy, const double* __restrict__ sums, int rows, int c,
                                                            float eps, float momentum, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            float* __restrict__ mean_out, float* __restrict__ invstd_out, float* rmean, float* rvar,
                                                            long long* nbt, long long total4, int c4, int relu, float* __restrict__ amax_out,
                                                            unsigned char* __restrict__ mask, const float* __restrict__ colmax) {
    // colmax != null: y is written as the packed pre-split fp16 image (glf_split_f16_packed's format), scaled with an upper
    // bound of max |y| that every workgroup derives from the per-channel maxima of |x| BEFORE writing anything:
    //   |y_c| <= |gamma_c| invstd_c (max|x_c| + |mean_c|) + |beta_c|        (workgroup 0 stores the bound to *amax_out)
    extern __shared__ __attribute__((aligned(16))) float s_coef[];       // [2][c]: mean, invstd
    __shared__ float s_bound[4];
    float* s_mean = s_coef;
    float* s_is = s_coef + c;
    float bound = 0.f;
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        const double m = sums[ch] / rows;
        double var = sums[c + ch] / rows - m * m;
        if (var < 0) var = 0;
        const float mf = (float)m, isf = (float)(1.0 / sqrt(var + (double)eps));
        s_mean[ch] = mf; s_is[ch] = isf;
        if (colmax) bound = fmaxf(bound, fabsf(gamma[ch]) * isf * (colmax[ch] + fabsf(mf)) + fabsf(beta[ch]));
        if (blockIdx.x == 0) {
            mean_out[ch] = mf; invstd_out[ch] = isf;
            if (rmean) {
                const double unb = rows > 1 ? var * rows / (rows - 1) : var;
                rmean[ch] = (1.f - momentum) * rmean[ch] + momentum * (float)m;
                rvar[ch] = (1.f - momentum) * rvar[ch] + momentum * (float)unb;
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0 && nbt) *nbt += 1;
    float sc = 1.f, sc_inv = 1.f;
    if (colmax) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bound = fmaxf(bound, __shfl_xor(bound, o, 64));
        if ((threadIdx.x & 63) == 0) s_bound[threadIdx.x >> 6] = bound;
    }
    __syncthreads();
    if (colmax) {
        bound = 1.0001f * fmaxf(fmaxf(s_bound[0], s_bound[1]), fmaxf(s_bound[2], s_bound[3]));
        if (blockIdx.x == 0 && threadIdx.x == 0) *amax_out = bound;
        pow2_scale(&bound, sc, sc_inv);
    }
    float am = 0.f;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c4;
        const int cc = (int)(i - r * c4) * 4;
        const float4 xx = *reinterpret_cast<const float4*>(x + r * ldx + cc);
        const float4 mu = *reinterpret_cast<const float4*>(s_mean + cc);
        const float4 is = *reinterpret_cast<const float4*>(s_is + cc);
        const float4 ga = *reinterpret_cast<const float4*>(gamma + cc);
        const float4 be = *reinterpret_cast<const float4*>(beta + cc);
        float4 o = make_float4(bn_val(xx.x, mu.x, is.x, ga.x, be.x), bn_val(xx.y, mu.y, is.y, ga.y, be.y),
                               bn_val(xx.z, mu.z, is.z, ga.z, be.z), bn_val(xx.w, mu.w, is.w, ga.w, be.w));
        if (res) {
            const float4 rr = *reinterpret_cast<const float4*>(res + r * ldr + cc);
            o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
        }
        if (mask) mask[i] = (unsigned char)((o.x > 0.f) | ((o.y > 0.f) << 1) | ((o.z > 0.f) << 2) | ((o.w > 0.f) << 3));
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        if (colmax) {
            const SplitH sp = split4h(o, sc);
            const float2 h = __builtin_bit_cast(float2, sp.h), l = __builtin_bit_cast(float2, sp.l);
            o = make_float4(h.x, h.y, l.x, l.y);
        } else {
            am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        *reinterpret_cast<float4*>(y + r * ldy + cc) = o;
    }
    if (amax_out && !colmax) block_amax(am, amax_out);
}

// dx = gamma*invstd*(dy' - [sum_dy/n + xhat*sum_dyx/n]) ; dres = dy'
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const float* __restrict__ dy, int lddy, const float* __restrict__ x, int ldx,
                                                           const float* __restrict__ y, int ldy, Coef k,
                                                           const float* __restrict__ sum_dy, const float* __restrict__ sum_dyx,
                                                           float* __restrict__ dx, int lddx, float* __restrict__ dres, int lddres,
                                                           long long total4, int c4, int relu, int training, float inv_n,
                                                           float* __restrict__ amax_out, int packed, const unsigned char* __restrict__ mask,
                                                           const float* __restrict__ dy2, int lddy2, const double* __restrict__ fsums,
                                                           const float* __restrict__ fmax, float* __restrict__ out_dbeta, float* __restrict__ out_dgamma,
                                                           int c) {
    // packed != 0: dx is written as the packed pre-split fp16 image (glf_split_f16_packed's format) scaled by *amax_out, which
    // then holds an upper bound of max |dx| computed by bnbwd_finalize (not a by-product of this kernel)
    // fsums != null (fused form): the reduction left UNFINISHED sums (doubles [2][c]) and, for the packed form, per-channel maxima
    // (fmax [2][c]); every workgroup finishes them in LDS -- sum_dy / sum_dyx then point into LDS -- workgroup 0 writes dbeta /
    // dgamma, and the packed form's bound (bnbwd_finalize's expression) is derived by every workgroup before it writes anything
    extern __shared__ __attribute__((aligned(16))) float s_fin[];        // [2][c]
    __shared__ float s_bound[4];
    float bound_local = 0.f;
    if (fsums) {
        for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
            const float a = (float)fsums[ch], b = (float)fsums[c + ch];
            s_fin[ch] = a; s_fin[c + ch] = b;
            if (blockIdx.x == 0) { if (out_dbeta) out_dbeta[ch] = a; if (out_dgamma) out_dgamma[ch] = b; }
            if (packed) {
                const float kk = fabsf(k.gamma[ch] * k.invstd[ch]);
                const float bb = 1.0001f * kk * (training ? fmax[ch] + inv_n * (fabsf(a) + fmax[c + ch] * fabsf(b)) : fmax[ch]);
                if (bb < 3.0e38f) bound_local = fmaxf(bound_local, bb);
            }
        }
        if (packed) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) bound_local = fmaxf(bound_local, __shfl_xor(bound_local, o, 64));
            if ((threadIdx.x & 63) == 0) s_bound[threadIdx.x >> 6] = bound_local;
        }
        __syncthreads();
        sum_dy = s_fin; sum_dyx = s_fin + c;
    }
    float am = 0.f, sc = 1.f, sc_inv = 1.f;
    if (packed && fsums) {
        float bnd = fmaxf(fmaxf(s_bound[0], s_bound[1]), fmaxf(s_bound[2], s_bound[3]));
        if (blockIdx.x == 0 && threadIdx.x == 0) *amax_out = bnd;
        pow2_scale(&bnd, sc, sc_inv);
    } else if (packed) pow2_scale(amax_out, sc, sc_inv);
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c4;
        const int c = (int)(i - r * c4) * 4;
        float4 g = *reinterpret_cast<const float4*>(dy + r * lddy + c);
        if (dy2) {
            const float4 g2 = *reinterpret_cast<const float4*>(dy2 + r * lddy2 + c);
            g.x += g2.x; g.y += g2.y; g.z += g2.z; g.w += g2.w;
        }
        const float4 is = *reinterpret_cast<const float4*>(k.invstd + c);
        const float4 ga = *reinterpret_cast<const float4*>(k.gamma + c);
        float4 xx = make_float4(0.f, 0.f, 0.f, 0.f), mu = xx;
        if (training || (relu && !y && !mask)) {
            xx = *reinterpret_cast<const float4*>(x + r * ldx + c);
            mu = *reinterpret_cast<const float4*>(k.mean + c);
        }
        if (relu && mask) {
            const unsigned m = mask[i];
            g.x = (m & 1u) ? g.x : 0.f; g.y = (m & 2u) ? g.y : 0.f; g.z = (m & 4u) ? g.z : 0.f; g.w = (m & 8u) ? g.w : 0.f;
        } else if (relu) {
            float4 yy;
            if (y) {
                yy = *reinterpret_cast<const float4*>(y + r * ldy + c);
            } else {
                const float4 be = *reinterpret_cast<const float4*>(k.beta + c);
                yy = make_float4(bn_val(xx.x, mu.x, is.x, ga.x, be.x), bn_val(xx.y, mu.y, is.y, ga.y, be.y),
                                 bn_val(xx.z, mu.z, is.z, ga.z, be.z), bn_val(xx.w, mu.w, is.w, ga.w, be.w));
            }
            g.x = yy.x > 0.f ? g.x : 0.f; g.y = yy.y > 0.f ? g.y : 0.f;
            g.z = yy.z > 0.f ? g.z : 0.f; g.w = yy.w > 0.f ? g.w : 0.f;
        }
        if (dres) *reinterpret_cast<float4*>(dres + r * lddres + c) = g;
        float4 o;
        if (training) {
            const float4 s1 = *reinterpret_cast<const float4*>(sum_dy + c);
            const float4 s2 = *reinterpret_cast<const float4*>(sum_dyx + c);
            o.x = ga.x * is.x * (g.x - inv_n * (s1.x + (xx.x - mu.x) * is.x * s2.x));
            o.y = ga.y * is.y * (g.y - inv_n * (s1.y + (xx.y - mu.y) * is.y * s2.y));
            o.z = ga.z * is.z * (g.z - inv_n * (s1.z + (xx.z - mu.z) * is.z * s2.z));
            o.w = ga.w * is.w * (g.w - inv_n * (s1.w + (xx.w - mu.w) * is.w * s2.w));
        } else {
            o = make_float4(g.x * ga.x * is.x, g.y * ga.y * is.y, g.z * ga.z * is.z, g.w * ga.w * is.w);
        }
        if (packed) {
            const SplitH sp = split4h(o, sc);
            const float2 h = __builtin_bit_cast(float2, sp.h), l = __builtin_bit_cast(float2, sp.l);
            o = make_float4(h.x, h.y, l.x, l.y);
        } else {
            am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
        }
        *reinterpret_cast<float4*>(dx + r * lddx + c) = o;
    }
    if (amax_out && !packed) block_amax(am, amax_out);
}

// ---- TPAVI tail: one wavefront per row ---------------------------------------------------
constexpr int LN_NV = 8;      // float4 per lane => C <= 64*4*8 = 2048

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <bool BWD>
__global__ __launch_bounds__(256) void bn_res_ln_kernel(const float* __restrict__ w, const float* __restrict__ x, Coef bn,
                                                        const float* __restrict__ ln_g, const float* __restrict__ ln_b, float eps,
                                                        float* __restrict__ z, float* __restrict__ row_mean, float* __restrict__ row_rstd,
                                                        const float* __restrict__ dz, float* __restrict__ du, int rows, int c,
                                                        float* __restrict__ amax_out) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int c4 = c >> 2;
    const long long base = (long long)row * c;
    float4 u[LN_NV];
    float s = 0.f;
#pragma unroll
    for (int v = 0; v < LN_NV; ++v) {
        const int cc = lane + 64 * v;
        if (cc < c4) {
            const int ch = cc * 4;
            const float4 ww = *reinterpret_cast<const float4*>(w + base + ch);
            const float4 xx = *reinterpret_cast<const float4*>(x + base + ch);
            const float4 mu = *reinterpret_cast<const float4*>(bn.mean + ch);
            const float4 is = *reinterpret_cast<const float4*>(bn.invstd + ch);
            const float4 ga = *reinterpret_cast<const float4*>(bn.gamma + ch);
            const float4 be = *reinterpret_cast<const float4*>(bn.beta + ch);
            u[v] = make_float4((ww.x - mu.x) * is.x * ga.x + be.x + xx.x, (ww.y - mu.y) * is.y * ga.y + be.y + xx.y,
                               (ww.z - mu.z) * is.z * ga.z + be.z + xx.z, (ww.w - mu.w) * is.w * ga.w + be.w + xx.w);
            s += (u[v].x + u[v].y) + (u[v].z + u[v].w);
        } else {
            u[v] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float mean, rstd;
    if (!BWD) {
        mean = wave_sum(s) / c;
        float q = 0.f;
#pragma unroll
        for (int v = 0; v < LN_NV; ++v)
            if (lane + 64 * v < c4) {
                const float a = u[v].x - mean, b = u[v].y - mean, cc2 = u[v].z - mean, d = u[v].w - mean;
                q += (a * a + b * b) + (cc2 * cc2 + d * d);
            }
        rstd = 1.0f / sqrtf(wave_sum(q) / c + eps);
        if (lane == 0) { row_mean[row] = mean; row_rstd[row] = rstd; }
        float am = 0.f;
#pragma unroll
        for (int v = 0; v < LN_NV; ++v) {
            const int cc = lane + 64 * v;
            if (cc < c4) {
                const int ch = cc * 4;
                const float4 g = *reinterpret_cast<const float4*>(ln_g + ch);
                const float4 b = *reinterpret_cast<const float4*>(ln_b + ch);
                const float4 o = make_float4((u[v].x - mean) * rstd * g.x + b.x, (u[v].y - mean) * rstd * g.y + b.y,
                                             (u[v].z - mean) * rstd * g.z + b.z, (u[v].w - mean) * rstd * g.w + b.w);
                *reinterpret_cast<float4*>(z + base + ch) = o;
                am = fmaxf(fmaxf(am, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
            }
        }
        if (amax_out) {          // max |z| as a by-product (one wave per row: only a row that raises the maximum issues the atomic)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) am = fmaxf(am, __shfl_xor(am, o, 64));
            if (lane == 0 && am > *reinterpret_cast<volatile float*>(amax_out)) atomicMax(reinterpret_cast<unsigned*>(amax_out), __float_as_uint(am));
        }
    } else {
        mean = row_mean[row]; rstd = row_rstd[row];
        float4 gz[LN_NV];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int v = 0; v < LN_NV; ++v) {
            const int cc = lane + 64 * v;
            if (cc < c4) {
                const int ch = cc * 4;
                const float4 d = *reinterpret_cast<const float4*>(dz + base + ch);
                const float4 g = *reinterpret_cast<const float4*>(ln_g + ch);
                gz[v] = make_float4(d.x * g.x, d.y * g.y, d.z * g.z, d.w * g.w);
                u[v] = make_float4((u[v].x - mean) * rstd, (u[v].y - mean) * rstd, (u[v].z - mean) * rstd, (u[v].w - mean) * rstd);
                s1 += (gz[v].x + gz[v].y) + (gz[v].z + gz[v].w);
                s2 += (gz[v].x * u[v].x + gz[v].y * u[v].y) + (gz[v].z * u[v].z + gz[v].w * u[v].w);
            } else {
                gz[v] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
        const float m1 = wave_sum(s1) / c, m2 = wave_sum(s2) / c;
#pragma unroll
        for (int v = 0; v < LN_NV; ++v) {
            const int cc = lane + 64 * v;
            if (cc < c4)
                *reinterpret_cast<float4*>(du + base + cc * 4) =
                    make_float4(rstd * (gz[v].x - m1 - u[v].x * m2), rstd * (gz[v].y - m1 - u[v].y * m2),
                                rstd * (gz[v].z - m1 - u[v].z * m2), rstd * (gz[v].w - m1 - u[v].w * m2));
        }
    }
}

inline int stream_grid(long long total, int block) {
    long long g = (total + block - 1) / block;
    const long long cap = (long long)glf::num_cus() * 8;
    return (int)(g < 1 ? 1 : (g > cap ? cap : g));
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

template <class Op>
int launch_colreduce(Op op, int rows, int c, double* ws, hipStream_t s) {
    const int slices = n_slices_c(rows, c);
    const int c4 = c / 4, tpr = c4 < RT ? c4 : RT;
    hipLaunchKernelGGL((colreduce_kernel<Op>), dim3(slices, (c4 + tpr - 1) / tpr), dim3(RT), 0, s, op, rows, c, slices, ws);
    return glf::check_launch("colreduce");
}

}  // namespace

extern "C" size_t glf_bn_workspace(int rows, int c) {
    // doubles: 2 x slices x c partial sums, 2 x c finished sums, and (as floats) 2 x slices x c partial maxima of glf_bn_bwd's packed path
    return (size_t)3 * MAX_SLICES * (size_t)(c > 0 ? c : 0) + (size_t)2 * (c > 0 ? c : 0);
}

#define REQ_C4(c) GLF_REQUIRE((c) > 0 && ((c) % 4) == 0, GLF_ERR_BAD_SHAPE, "channel count must be a positive multiple of 4 (got %d)", (c))
#define REQ_AL(p, name) GLF_REQUIRE(al16(p), GLF_ERR_BAD_SHAPE, name " must be 16-byte aligned")
#define REQ_LD(ld, name) GLF_REQUIRE(((ld) % 4) == 0, GLF_ERR_BAD_SHAPE, name " must be a multiple of 4")

extern "C" int glf_bn_stats(const float* x, int ldx, int rows, int c, float eps, float momentum,
                            float* mean, float* invstd, float* running_mean, float* running_var,
                            int64_t* num_batches_tracked, double* workspace, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && mean && invstd && workspace, GLF_ERR_NULL, "bn_stats: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "bn_stats: rows must be > 0");
    REQ_C4(c); REQ_AL(x, "x"); REQ_LD(ldx, "ldx");
    GLF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), GLF_ERR_NULL, "bn_stats: running_mean/var must both be set or both NULL");
    if (int rc = launch_colreduce(OpStats{x, ldx}, rows, c, workspace, glf::S(s))) return rc;
    hipLaunchKernelGGL(bn_stats_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(256), 0, glf::S(s), workspace, n_slices_c(rows, c), c, rows,
                       eps, momentum, mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked));
    return glf::check_launch("bn_stats_finalize");
}

extern "C" int glf_bn_stats_from_sums(const double* sums, int rows, int c, float eps, float momentum, float* mean, float* invstd,
                                      float* running_mean, float* running_var, int64_t* num_batches_tracked, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(sums && mean && invstd, GLF_ERR_NULL, "bn_stats_from_sums: null argument");
    GLF_REQUIRE(rows > 0 && c > 0, GLF_ERR_BAD_SHAPE, "bn_stats_from_sums: rows and c must be > 0");
    GLF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), GLF_ERR_NULL, "bn_stats_from_sums: running_mean/var must both be set or both NULL");
    hipLaunchKernelGGL(bn_stats_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(256), 0, glf::S(s), sums, 1, c, rows,
                       eps, momentum, mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked));
    return glf::check_launch("bn_stats_from_sums");
}

extern "C" int glf_bn_replay_running(const float* mean, const float* invstd, int rows, int c, float eps, float momentum,
                                     float* running_mean, float* running_var, int64_t* num_batches_tracked, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(mean && invstd && running_mean && running_var, GLF_ERR_NULL, "bn_replay_running: null argument");
    GLF_REQUIRE(rows > 0 && c > 0, GLF_ERR_BAD_SHAPE, "bn_replay_running: rows and c must be > 0");
    hipLaunchKernelGGL(bn_replay_kernel, dim3((c + 255) / 256), dim3(256), 0, glf::S(s), mean, invstd, rows, eps, momentum,
                       running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked), c);
    return glf::check_launch("bn_replay_running");
}

extern "C" int glf_bn_eval_coeffs(const float* rm, const float* rv, float eps, float* mean, float* invstd, int c, glf_stream_t s) {
    GLF_REQUIRE(rm && rv && mean && invstd, GLF_ERR_NULL, "bn_eval_coeffs: null argument");
    GLF_REQUIRE(c > 0, GLF_ERR_BAD_SHAPE, "bn_eval_coeffs: c must be > 0");
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((c + 255) / 256), dim3(256), 0, glf::S(s), rm, rv, eps, mean, invstd, c);
    return glf::check_launch("bn_eval_coeffs");
}

extern "C" int glf_bn_apply(const float* x, int ldx, const float* residual, int ldr, float* y, int ldy,
                            const float* mean, const float* invstd, const float* gamma, const float* beta,
                            int rows, int c, int relu, float* amax_out, uint8_t* relu_mask, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y && mean && invstd && gamma && beta, GLF_ERR_NULL, "bn_apply: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "bn_apply: rows must be > 0");
    REQ_C4(c); REQ_AL(x, "x"); REQ_AL(y, "y"); REQ_LD(ldx, "ldx"); REQ_LD(ldy, "ldy");
    if (residual) { REQ_AL(residual, "residual"); REQ_LD(ldr, "ldr"); }
    const long long total4 = (long long)rows * (c / 4);
    hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, glf::S(s), x, ldx, residual, ldr, y, ldy,
                       Coef{mean, invstd, gamma, beta}, total4, c / 4, relu, amax_out, relu_mask);
    return glf::check_launch("bn_apply");
}

extern "C" int glf_bn_apply_from_sums(const float* x, int ldx, const float* residual, int ldr, float* y, int ldy, const double* sums,
                                      int rows, int c, float eps, float momentum, const float* gamma, const float* beta,
                                      float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                                      int relu, float* amax_out, uint8_t* relu_mask, const float* colmax, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && y && sums && mean && invstd && gamma && beta, GLF_ERR_NULL, "bn_apply_from_sums: null argument");
    GLF_REQUIRE(!colmax || (amax_out && !residual && y != x), GLF_ERR_BAD_SHAPE, "bn_apply_from_sums: the packed output (colmax) needs amax_out, no residual and y != x");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "bn_apply_from_sums: rows must be > 0");
    REQ_C4(c); REQ_AL(x, "x"); REQ_AL(y, "y"); REQ_LD(ldx, "ldx"); REQ_LD(ldy, "ldy");
    GLF_REQUIRE(c <= APPLY_MAX_C, GLF_ERR_UNSUPPORTED, "bn_apply_from_sums: C must be <= %d (use glf_bn_stats_from_sums + glf_bn_apply)", APPLY_MAX_C);
    GLF_REQUIRE((running_mean == nullptr) == (running_var == nullptr), GLF_ERR_NULL, "bn_apply_from_sums: running_mean/var must both be set or both NULL");
    if (residual) { REQ_AL(residual, "residual"); REQ_LD(ldr, "ldr"); }
    const long long total4 = (long long)rows * (c / 4);
    hipLaunchKernelGGL(bn_apply_sums_kernel, dim3(stream_grid(total4, 256)), dim3(256), (size_t)2 * c * sizeof(float), glf::S(s), x, ldx, residual, ldr, y, ldy, sums, rows, c,
                       eps, momentum, gamma, beta, mean, invstd, running_mean, running_var, reinterpret_cast<long long*>(num_batches_tracked),
                       total4, c / 4, relu, amax_out, relu_mask, colmax);
    return glf::check_launch("bn_apply_from_sums");
}

extern "C" int glf_bn_bwd(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy,
                          const float* mean, const float* invstd, const float* gamma, const float* beta,
                          float* dx, int lddx, float* dres, int lddres, float* dgamma, float* dbeta,
                          int rows, int c, int relu, int training, double* workspace, float* amax_out, int packed_dx,
                          const uint8_t* relu_mask, const float* dy2, int lddy2, double* fused_sums, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && x && mean && invstd && gamma && dx && (workspace || fused_sums), GLF_ERR_NULL, "bn_bwd: null argument");
    if (relu_mask) y = nullptr;
    GLF_REQUIRE(!packed_dx || amax_out, GLF_ERR_NULL, "bn_bwd: packed_dx needs amax_out (a zeroed device float that receives the bound the image is scaled with)");
    GLF_REQUIRE(!relu || y || beta || relu_mask, GLF_ERR_NULL, "bn_bwd: relu != 0 needs relu_mask, y (the forward output) or beta (to recompute its sign from x)");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "bn_bwd: rows must be > 0");
    REQ_C4(c); REQ_AL(dy, "dy"); REQ_AL(x, "x"); REQ_AL(dx, "dx"); REQ_LD(lddy, "lddy"); REQ_LD(ldx, "ldx"); REQ_LD(lddx, "lddx");
    if (relu && y) { REQ_AL(y, "y"); REQ_LD(ldy, "ldy"); }
    if (dres) { REQ_AL(dres, "dres"); REQ_LD(lddres, "lddres"); }
    if (dy2) { REQ_AL(dy2, "dy2"); REQ_LD(lddy2, "lddy2"); }
    const OpBnBwd op{dy, lddy, x, ldx, y, ldy, mean, invstd, gamma, beta, relu, relu_mask, c / 4, dy2, lddy2};
    const long long total4f = (long long)rows * (c / 4);
    if (fused_sums && c <= APPLY_MAX_C) {
        // two launches: reduction with atomics into the caller's ZERO-FILLED buffer (2 c doubles, then 2 c floats of maxima), apply
        float* fmax = reinterpret_cast<float*>(fused_sums + (size_t)2 * c);
        const int fs = fused_slices(rows, c);
        const int c4 = c / 4, tpr = c4 < FUSED_TPR ? c4 : FUSED_TPR;
        const dim3 rgrid(fs, (c4 + tpr - 1) / tpr);
        float* rmax = packed_dx ? fmax : (float*)nullptr;
        const int rm = !op.relu ? 0 : (op.mask ? 1 : (op.y ? 3 : 2));
#define GLF_BNR(H2, RM_) hipLaunchKernelGGL((bnbwd_reduce_atomic_kernel<H2, RM_>), rgrid, dim3(RT), 0, glf::S(s), op, rows, c, fs, fused_sums, rmax)
        if (op.dy2) { if (rm == 0) GLF_BNR(true, 0); else if (rm == 1) GLF_BNR(true, 1); else if (rm == 2) GLF_BNR(true, 2); else GLF_BNR(true, 3); }
        else { if (rm == 0) GLF_BNR(false, 0); else if (rm == 1) GLF_BNR(false, 1); else if (rm == 2) GLF_BNR(false, 2); else GLF_BNR(false, 3); }
#undef GLF_BNR
        if (int rc = glf::check_launch("bn_bwd_reduce(fused)")) return rc;
        hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(total4f, 256)), dim3(256), (size_t)2 * c * sizeof(float), glf::S(s), dy, lddy, x, ldx, y, ldy,
                           Coef{mean, invstd, gamma, beta}, (const float*)nullptr, (const float*)nullptr, dx, lddx, dres, lddres, total4f, c / 4, relu, training,
                           1.0f / (float)rows, amax_out, packed_dx, relu_mask, dy2, lddy2, fused_sums, fmax, dbeta, dgamma, c);
        return glf::check_launch("bn_bwd_apply(fused)");
    }
    GLF_REQUIRE(workspace, GLF_ERR_NULL, "bn_bwd: the three-launch form needs the workspace");
    const int slices = n_slices_c(rows, c);
    // per-channel sums live behind the partials in the workspace (as floats) when the caller does not want them
    float* sums = reinterpret_cast<float*>(workspace + (size_t)2 * slices * c);
    float* s_dy = dbeta ? dbeta : sums;
    float* s_dyx = dgamma ? dgamma : sums + c;
    if (packed_dx) {
        float* pmax = reinterpret_cast<float*>(workspace + (size_t)2 * slices * c + (size_t)2 * c);
        const int c4 = c / 4, tpr = c4 < RT ? c4 : RT;
        hipLaunchKernelGGL(bnbwd_reduce_kernel, dim3(slices, (c4 + tpr - 1) / tpr), dim3(RT), 0, glf::S(s), op, rows, c, slices, workspace, pmax);
        if (int rc = glf::check_launch("bn_bwd_reduce")) return rc;
        hipLaunchKernelGGL(bnbwd_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(256), 0, glf::S(s), workspace, pmax, slices, c, gamma, invstd,
                           1.0f / (float)rows, training, s_dy, s_dyx, amax_out);
    } else {
        if (int rc = launch_colreduce(op, rows, c, workspace, glf::S(s))) return rc;
        hipLaunchKernelGGL(sum_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(256), 0, glf::S(s), workspace, slices, c, s_dy, s_dyx);
    }
    if (int rc = glf::check_launch("bn_bwd_finalize")) return rc;
    const long long total4 = (long long)rows * (c / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(stream_grid(total4, 256)), dim3(256), 0, glf::S(s), dy, lddy, x, ldx, y, ldy,
                       Coef{mean, invstd, gamma, beta}, s_dy, s_dyx, dx, lddx, dres, lddres, total4, c / 4, relu, training,
                       1.0f / (float)rows, amax_out, packed_dx, relu_mask, dy2, lddy2, (const double*)nullptr, (const float*)nullptr, (float*)nullptr,
                       (float*)nullptr, c);
    return glf::check_launch("bn_bwd_apply");
}

extern "C" int glf_colsum(const float* dy, int lddy, float* db, int rows, int c, double* workspace, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dy && db && workspace, GLF_ERR_NULL, "colsum: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "colsum: rows must be > 0");
    REQ_C4(c); REQ_AL(dy, "dy"); REQ_LD(lddy, "lddy");
    if (int rc = launch_colreduce(OpColsum{dy, lddy}, rows, c, workspace, glf::S(s))) return rc;
    hipLaunchKernelGGL(sum_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(256), 0, glf::S(s), workspace, n_slices_c(rows, c), c, db, (float*)nullptr);
    return glf::check_launch("colsum_finalize");
}

extern "C" int glf_bn_res_ln_fwd(const float* w, const float* x, const float* bn_mean, const float* bn_invstd,
                                 const float* bn_gamma, const float* bn_beta, const float* ln_gamma,
                                 const float* ln_beta, float ln_eps, float* z, float* row_mean, float* row_rstd,
                                 int rows, int c, float* amax_out, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(w && x && bn_mean && bn_invstd && bn_gamma && bn_beta && ln_gamma && ln_beta && z && row_mean && row_rstd,
                GLF_ERR_NULL, "bn_res_ln_fwd: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "bn_res_ln_fwd: rows must be > 0");
    REQ_C4(c); GLF_REQUIRE(c <= 64 * 4 * LN_NV, GLF_ERR_UNSUPPORTED, "bn_res_ln: C must be <= %d", 64 * 4 * LN_NV);
    REQ_AL(w, "w"); REQ_AL(x, "x"); REQ_AL(z, "z");
    hipLaunchKernelGGL((bn_res_ln_kernel<false>), dim3((rows + 3) / 4), dim3(256), 0, glf::S(s), w, x,
                       Coef{bn_mean, bn_invstd, bn_gamma, bn_beta}, ln_gamma, ln_beta, ln_eps, z, row_mean, row_rstd,
                       (const float*)nullptr, (float*)nullptr, rows, c, amax_out);
    return glf::check_launch("bn_res_ln_fwd");
}

extern "C" int glf_bn_res_ln_bwd(const float* dz, const float* w, const float* x, const float* bn_mean,
                                 const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                                 const float* ln_gamma, const float* row_mean, const float* row_rstd,
                                 float* du, float* dln_gamma, float* dln_beta, int rows, int c,
                                 double* workspace, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(dz && w && x && bn_mean && bn_invstd && bn_gamma && bn_beta && ln_gamma && row_mean && row_rstd && du &&
                dln_gamma && dln_beta && workspace, GLF_ERR_NULL, "bn_res_ln_bwd: null argument");
    GLF_REQUIRE(rows > 0, GLF_ERR_BAD_SHAPE, "bn_res_ln_bwd: rows must be > 0");
    REQ_C4(c); GLF_REQUIRE(c <= 64 * 4 * LN_NV, GLF_ERR_UNSUPPORTED, "bn_res_ln: C must be <= %d", 64 * 4 * LN_NV);
    REQ_AL(dz, "dz"); REQ_AL(w, "w"); REQ_AL(x, "x"); REQ_AL(du, "du");
    const Coef bn{bn_mean, bn_invstd, bn_gamma, bn_beta};
    hipLaunchKernelGGL((bn_res_ln_kernel<true>), dim3((rows + 3) / 4), dim3(256), 0, glf::S(s), w, x, bn, ln_gamma,
                       (const float*)nullptr, 0.f, (float*)nullptr, const_cast<float*>(row_mean), const_cast<float*>(row_rstd),
                       dz, du, rows, c, (float*)nullptr);
    if (int rc = glf::check_launch("bn_res_ln_bwd")) return rc;
    if (int rc = launch_colreduce(OpLnParam{dz, w, x, bn, row_mean, row_rstd, c}, rows, c, workspace, glf::S(s))) return rc;
    hipLaunchKernelGGL(sum_finalize, dim3((c + FIN_CH - 1) / FIN_CH), dim3(256), 0, glf::S(s), workspace, n_slices_c(rows, c), c, dln_gamma, dln_beta);
    return glf::check_launch("ln_param_finalize");
}
