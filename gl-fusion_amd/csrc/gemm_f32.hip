// Implicit-GEMM contraction engine for the GL-Fusion hot path on MI355X (gfx950 / CDNA4).
//
//   * fp32 in / fp32 accumulate on the matrix cores: v_mfma_f32_32x32x2_f32 (exact f32,
//     64 cycles/SIMD, 157 TF chip peak).  One wave owns a 64x64 output tile = 2x2 MFMA tiles
//     (64 accumulator VGPRs); a 256-thread workgroup owns 128x128; K is walked 32 deep.
//   * NHWC, im2col-free: the A rows of a K-tile are gathered straight from the activation
//     tensor with the (tap, stride, pad, dilation) mapping; taps that are out of range for a
//     whole 128-row tile are skipped (ASPP rate 12/24/36 on a 28x28 map is mostly padding).
//   * Operands are staged global -> registers -> LDS (issue-early / write-late, one barrier per
//     K-tile, two LDS buffers).  LDS tiles are k-major ([k][m]) so that every MFMA operand read
//     is a conflict-free ds_read_b32 of 32 consecutive floats; k-contiguous sources are
//     transposed on the way in (row stride 129: conflict-free ds_write_b32), row-contiguous
//     sources go in with ds_write_b128 (row stride 132).
//   * blockIdx -> tile mapping is XCD-aware (8 XCDs, private L2s): each XCD walks a contiguous
//     range of tiles that share A rows.
//
// Three kernels share the MFMA core:
//   gemm_rows_kernel<0>  A[m][k] (gathered rows) x B[n][k]   -> conv forward, 1x1 / linear
//   gemm_rows_kernel<1>  A[m][k] (gathered rows) x B[k][n]   -> conv dgrad, y = theta @ M
//   gemm_tn_kernel       sum_r A[r][m] * B[src(r)][n]        -> conv wgrad, M = phi^T g
#include "glf_common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LD_T = 129;   // LDS row stride of a tile filled by the transposing scatter
constexpr int LD_V = 132;   // LDS row stride of a tile filled with ds_write_b128
constexpr int NTHREADS = 256;

struct Geo { int n_img, hs, ws, hd, wd, kh, kw, stride, pad, dil; };

struct GemmArgs {
    const float* A; const float* B; const float* bias; float* C;
    int M, N, K, lda, ldb, ldc, taps;
    unsigned tap_mask;
    long long tap_stride_b;
    int gather;
    Geo g;
    long long bsa, bsb, bsc;
    float alpha;
    int accumulate, split;
    int tiles_m, tiles_n;
    int vec_a, vec_b;
};

// source row of GEMM row (n,y,x) for `tap`, or -1 when the tap falls into the padding
__device__ __forceinline__ int map_src(const Geo& g, int gather, int n, int y, int x, int tap) {
    const int ky = tap / g.kw, kx = tap - ky * g.kw;
    int sy, sx;
    if (gather == 1) {
        sy = y * g.stride - g.pad + ky * g.dil;
        sx = x * g.stride - g.pad + kx * g.dil;
        if ((unsigned)sy >= (unsigned)g.hs || (unsigned)sx >= (unsigned)g.ws) return -1;
    } else {
        sy = y + g.pad - ky * g.dil;
        sx = x + g.pad - kx * g.dil;
        if (sy < 0 || sx < 0) return -1;
        if (g.stride > 1) {
            if ((sy % g.stride) != 0 || (sx % g.stride) != 0) return -1;
            sy /= g.stride; sx /= g.stride;
        }
        if (sy >= g.hs || sx >= g.ws) return -1;
    }
    return (n * g.hs + sy) * g.ws + sx;
}

__device__ __forceinline__ float4 ld4(const float* p, int nvalid, bool vec) {
    if (nvalid >= 4 && vec) return *reinterpret_cast<const float4*>(p);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (nvalid > 0) v.x = p[0];
    if (nvalid > 1) v.y = p[1];
    if (nvalid > 2) v.z = p[2];
    if (nvalid > 3) v.w = p[3];
    return v;
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// 64x64 per wave, K = 32 from LDS tiles laid out [k][m] / [k][n].
// Two-level accumulation: the 32-deep K-tile is summed by the MFMA chain into fresh accumulators
// (C = 0 inline constant), which are then added to the running sums on the VALU (hidden under the
// next MFMAs).  A single f32 fma chain over K = 9*2048 products has a ~sqrt(K) rounding walk;
// chains of 32 + K/32 keep the contraction at the accuracy of a blocked CPU GEMM.
template <int LDA, int LDB>
__device__ __forceinline__ void mma_ktile(const float* __restrict__ a_s, const float* __restrict__ b_s,
                                          f32x16& c00, f32x16& c01, f32x16& c10, f32x16& c11) {
    f32x16 t00 = {0}, t01 = {0}, t10 = {0}, t11 = {0};
#pragma unroll
    for (int k = 0; k < BK; k += 2) {
        const float a0 = a_s[k * LDA], a1 = a_s[k * LDA + 32];
        const float b0 = b_s[k * LDB], b1 = b_s[k * LDB + 32];
        t00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, t00, 0, 0, 0);
        t01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, t01, 0, 0, 0);
        t10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, t10, 0, 0, 0);
        t11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, t11, 0, 0, 0);
    }
    c00 += t00; c01 += t01; c10 += t10; c11 += t11;
}

__device__ __forceinline__ void scatter4(float* dst, int ld, const float4& v) {
    dst[0] = v.x; dst[ld] = v.y; dst[2 * ld] = v.z; dst[3 * ld] = v.w;
}

// ----------------------------------------------------------------------------------------
// rows kernel: C[m][n] = alpha * sum_tap sum_k A[src(m,tap)][k] * B_tap(k,n) + bias[n]
// BMODE 0: B_tap[n][k]; BMODE 1: B_tap[k][n]
// ----------------------------------------------------------------------------------------
template <int BMODE, bool GATHER>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_rows_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LDA = LD_T;
    constexpr int LDB = (BMODE == 0) ? LD_T : LD_V;
    constexpr int A_SZ = BK * LDA, B_SZ = BK * LDB;
    float* As = smem;
    float* Bs = smem + 2 * A_SZ;
    unsigned* s_mask = reinterpret_cast<unsigned*>(smem + 2 * A_SZ + 2 * B_SZ);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
    const int bz = blockIdx.z;
    const float* __restrict__ A = p.A + (long long)bz * p.bsa;
    const float* __restrict__ B = p.B + (long long)bz * p.bsb;
    float* __restrict__ C = p.C + (long long)bz * p.bsc;

    const int ac = tid & 7, ar = tid >> 3;          // k-contiguous staging: 8 float4 per 32-deep row
    const int bc = tid & 31, br = tid >> 5;         // row-contiguous staging (BMODE 1)

    int a_n[4], a_y[4], a_x[4];
    long long a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = tm * BM + ar + 32 * j;
        if (GATHER) {
            if (m < p.M) {
                const int hw = p.g.hd * p.g.wd;
                const int n = m / hw, rem = m - n * hw;
                a_n[j] = n; a_y[j] = rem / p.g.wd; a_x[j] = rem - a_y[j] * p.g.wd;
            } else { a_n[j] = -1; a_y[j] = 0; a_x[j] = 0; }
            a_off[j] = -1;
        } else {
            a_n[j] = 0; a_y[j] = 0; a_x[j] = 0;
            a_off[j] = (m < p.M) ? (long long)m * p.lda : -1;
        }
    }

    unsigned mask = p.tap_mask;
    if (GATHER && p.taps > 1) {
        if (tid == 0) *s_mask = 0u;
        __syncthreads();
        if (ac == 0) {
            unsigned local = 0;
            for (unsigned mm = mask; mm; mm &= mm - 1) {
                const int t = __ffs(mm) - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (a_n[j] >= 0 && map_src(p.g, p.gather, a_n[j], a_y[j], a_x[j], t) >= 0) local |= 1u << t;
            }
            if (local) atomicOr(s_mask, local);
        }
        __syncthreads();
        mask &= *s_mask;
    }

    const int nkc = (p.K + BK - 1) / BK;
    const int ntiles = __popc(mask) * nkc;

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};

    float4 ra[4], rb[4];
    unsigned rem_mask = mask;
    int tap = -1, kc = nkc;          // "before the first tile"
    const bool vec_a = p.vec_a, vec_b = p.vec_b;

    auto advance = [&]() {
        if (++kc >= nkc) {
            kc = 0;
            tap = __ffs(rem_mask) - 1;
            rem_mask &= rem_mask - 1;
            if (GATHER) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int s = (a_n[j] >= 0) ? map_src(p.g, p.gather, a_n[j], a_y[j], a_x[j], tap) : -1;
                    a_off[j] = (s >= 0) ? (long long)s * p.lda : -1;
                }
            }
        }
    };
    auto load_tile = [&]() {
        const int kbase = kc * BK + 4 * ac;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ra[j] = (a_off[j] >= 0) ? ld4(A + a_off[j] + kbase, p.K - kbase, vec_a) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float* Bt = B + (long long)tap * p.tap_stride_b;
        if (BMODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = tn * BN + ar + 32 * j;
                rb[j] = (n < p.N) ? ld4(Bt + (long long)n * p.ldb + kbase, p.K - kbase, vec_b) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int n0 = tn * BN + 4 * bc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kc * BK + br + 8 * j;
                rb[j] = (k < p.K) ? ld4(Bt + (long long)k * p.ldb + n0, p.N - n0, vec_b) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tile = [&](int buf) {
        float* a_d = As + buf * A_SZ + (4 * ac) * LDA + ar;
#pragma unroll
        for (int j = 0; j < 4; ++j) scatter4(a_d + 32 * j, LDA, ra[j]);
        if (BMODE == 0) {
            float* b_d = Bs + buf * B_SZ + (4 * ac) * LDB + ar;
#pragma unroll
            for (int j = 0; j < 4; ++j) scatter4(b_d + 32 * j, LDB, rb[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *reinterpret_cast<float4*>(Bs + buf * B_SZ + (br + 8 * j) * LDB + 4 * bc) = rb[j];
        }
    };

    if (ntiles > 0) {
        advance();
        load_tile();
        store_tile(0);
        __syncthreads();
        const int a_lane = (lane >> 5) * LDA + wm + (lane & 31);
        const int b_lane = (lane >> 5) * LDB + wn + (lane & 31);
        for (int it = 0; it < ntiles; ++it) {
            const int buf = it & 1;
            const bool has_next = (it + 1) < ntiles;
            if (has_next) { advance(); load_tile(); }
            mma_ktile<LDA, LDB>(As + buf * A_SZ + a_lane, Bs + buf * B_SZ + b_lane, c00, c01, c10, c11);
            if (has_next) store_tile(buf ^ 1);
            __syncthreads();
        }
    }

    // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    auto emit = [&](const f32x16& acc, int ti, int tj) {
        const int col = tn * BN + wn + 32 * tj + col_l;
        if (col >= p.N) return;
        const float bv = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * BM + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + row_l;
            if (row < p.M) {
                float* dst = C + (long long)row * p.ldc + col;
                float v = p.alpha * acc[r] + bv;
                if (p.accumulate) v += *dst;
                *dst = v;
            }
        }
    };
    emit(c00, 0, 0); emit(c01, 0, 1); emit(c10, 1, 0); emit(c11, 1, 1);
}

// ----------------------------------------------------------------------------------------
// TN kernel: C_tap[m][n] (+)= alpha * sum_{r in slice} A[r][m] * B[src(r,tap)][n]
// grid: x = tiles_m*tiles_n, y = active taps, z = batch*split
// ----------------------------------------------------------------------------------------
template <bool GATHER>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(const GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = LD_V;
    constexpr int T_SZ = BK * LD;
    float* As = smem;
    float* Bs = smem + 2 * T_SZ;
    int* vflag = reinterpret_cast<int*>(smem + 4 * T_SZ);      // [2][32] row-valid flags

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p.tiles_n, tm = bid / p.tiles_n;
    // nth set bit of the tap mask
    int tap;
    {
        unsigned mm = p.tap_mask;
        for (int i = 0; i < (int)blockIdx.y; ++i) mm &= mm - 1;
        tap = __ffs(mm) - 1;
    }
    const int bz = blockIdx.z / p.split, sl = blockIdx.z - bz * p.split;
    const float* __restrict__ A = p.A + (long long)bz * p.bsa;
    const float* __restrict__ B = p.B + (long long)bz * p.bsb;
    float* __restrict__ C = p.C + (long long)bz * p.bsc + (long long)tap * p.tap_stride_b;

    int chunk = (p.K + p.split - 1) / p.split;
    chunk = ((chunk + BK - 1) / BK) * BK;
    const int r0 = sl * chunk;
    const int r1 = min(p.K, r0 + chunk);
    if (r0 >= r1) return;                                      // block-uniform

    const int c4 = tid & 31, rr = tid >> 5;
    const int m0 = tm * BM + 4 * c4, n0 = tn * BN + 4 * c4;
    const bool vec_a = p.vec_a, vec_b = p.vec_b;
    const int hw = GATHER ? p.g.hd * p.g.wd : 1;

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    float4 ra[4], rb[4];
    int rvalid[4];

    auto load_tile = [&](int rbase) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = rbase + rr + 8 * j;
            long long src = -1;
            if (r < r1) {
                if (GATHER) {
                    const int n = r / hw, rem = r - n * hw;
                    const int y = rem / p.g.wd, x = rem - y * p.g.wd;
                    src = map_src(p.g, 1, n, y, x, tap);
                } else {
                    src = r;
                }
            }
            rvalid[j] = src >= 0;
            if (src >= 0) {
                ra[j] = ld4(A + (long long)r * p.lda + m0, p.M - m0, vec_a);
                rb[j] = ld4(B + src * p.ldb + n0, p.N - n0, vec_b);
            } else {
                ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *reinterpret_cast<float4*>(As + buf * T_SZ + (rr + 8 * j) * LD + 4 * c4) = ra[j];
            *reinterpret_cast<float4*>(Bs + buf * T_SZ + (rr + 8 * j) * LD + 4 * c4) = rb[j];
            if (c4 == 0) vflag[buf * 32 + rr + 8 * j] = rvalid[j];
        }
    };

    load_tile(r0);
    store_tile(0);
    __syncthreads();
    const int a_lane = (lane >> 5) * LD + wm + (lane & 31);
    const int b_lane = (lane >> 5) * LD + wn + (lane & 31);
    int it = 0;
    for (int rbase = r0; rbase < r1; rbase += BK, ++it) {
        const int buf = it & 1;
        const bool has_next = (rbase + BK) < r1;
        if (has_next) load_tile(rbase + BK);
        // all 32 rows of this K-tile in the padding (dilated taps): nothing to add
        const bool any = __ballot(vflag[buf * 32 + (lane & 31)] != 0) != 0ull;
        if (any)
            mma_ktile<LD, LD>(As + buf * T_SZ + a_lane, Bs + buf * T_SZ + b_lane, c00, c01, c10, c11);
        if (has_next) store_tile(buf ^ 1);
        __syncthreads();
    }

    const bool atomic = (p.split > 1) || p.accumulate;
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    auto emit = [&](const f32x16& acc, int ti, int tj) {
        const int col = tn * BN + wn + 32 * tj + col_l;
        if (col >= p.N) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * BM + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + row_l;
            if (row < p.M) {
                float* dst = C + (long long)row * p.ldc + col;
                const float v = p.alpha * acc[r];
                if (atomic) atomicAdd(dst, v); else *dst = v;
            }
        }
    };
    emit(c00, 0, 0); emit(c01, 0, 1); emit(c10, 1, 0); emit(c11, 1, 1);
}

constexpr size_t SMEM_ROWS_NT = (2 * BK * LD_T + 2 * BK * LD_T) * sizeof(float) + 16;
constexpr size_t SMEM_ROWS_NN = (2 * BK * LD_T + 2 * BK * LD_V) * sizeof(float) + 16;
constexpr size_t SMEM_TN = (4 * BK * LD_V) * sizeof(float) + 2 * 32 * sizeof(int);

int validate(const glf_gemm_params* p, const void* A, const void* B, const void* C) {
    GLF_REQUIRE(p && A && B && C, GLF_ERR_NULL, "gemm: null argument");
    GLF_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, GLF_ERR_BAD_SHAPE, "gemm: M,N,K must be > 0 (got %d,%d,%d)", p->M, p->N, p->K);
    GLF_REQUIRE(p->taps >= 1 && p->taps <= 32, GLF_ERR_BAD_SHAPE, "gemm: taps must be in [1,32] (got %d)", p->taps);
    GLF_REQUIRE(p->batch >= 1 && p->batch <= 65535, GLF_ERR_BAD_SHAPE, "gemm: batch out of range (%d)", p->batch);
    GLF_REQUIRE(p->gather >= 0 && p->gather <= 2, GLF_ERR_BAD_SHAPE, "gemm: gather must be 0,1,2");
    const unsigned full = p->taps == 32 ? 0xffffffffu : ((1u << p->taps) - 1u);
    GLF_REQUIRE((p->tap_mask & ~full) == 0, GLF_ERR_BAD_SHAPE, "gemm: tap_mask has bits beyond taps");
    if (p->gather) {
        GLF_REQUIRE(p->kh * p->kw == p->taps, GLF_ERR_BAD_SHAPE, "gemm: kh*kw != taps");
        GLF_REQUIRE(p->stride >= 1 && p->dil >= 1 && p->hs > 0 && p->ws > 0 && p->hd > 0 && p->wd > 0 && p->n_img > 0,
                    GLF_ERR_BAD_SHAPE, "gemm: bad conv geometry");
    } else {
        GLF_REQUIRE(p->taps == 1, GLF_ERR_BAD_SHAPE, "gemm: taps > 1 needs a gather mapping");
    }
    return GLF_OK;
}

GemmArgs make_args(const float* A, const float* B, const float* bias, float* C, const glf_gemm_params* p) {
    GemmArgs a;
    a.A = A; a.B = B; a.bias = bias; a.C = C;
    a.M = p->M; a.N = p->N; a.K = p->K; a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc; a.taps = p->taps;
    a.tap_mask = p->tap_mask; a.tap_stride_b = p->tap_stride_b; a.gather = p->gather;
    a.g = Geo{p->n_img, p->hs, p->ws, p->hd, p->wd, p->kh, p->kw, p->stride, p->pad, p->dil};
    a.bsa = p->batch_stride_a; a.bsb = p->batch_stride_b; a.bsc = p->batch_stride_c;
    a.alpha = p->alpha; a.accumulate = p->accumulate; a.split = p->split < 1 ? 1 : p->split;
    a.tiles_m = (p->M + BM - 1) / BM; a.tiles_n = (p->N + BN - 1) / BN;
    a.vec_a = 0; a.vec_b = 0;
    return a;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

namespace glf {
int init_gemm_attrs() {
    hipError_t e;
#define SET_ATTR(fn, bytes)                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
    if (e != hipSuccess) return fail(GLF_ERR_LAUNCH, "hipFuncSetAttribute(" #fn "): %s", hipGetErrorString(e));
    SET_ATTR((gemm_rows_kernel<0, false>), SMEM_ROWS_NT)
    SET_ATTR((gemm_rows_kernel<0, true>), SMEM_ROWS_NT)
    SET_ATTR((gemm_rows_kernel<1, false>), SMEM_ROWS_NN)
    SET_ATTR((gemm_rows_kernel<1, true>), SMEM_ROWS_NN)
    SET_ATTR((gemm_tn_kernel<false>), SMEM_TN)
    SET_ATTR((gemm_tn_kernel<true>), SMEM_TN)
#undef SET_ATTR
    return GLF_OK;
}
}  // namespace glf

extern "C" int glf_gemm_nt(const float* A, const float* B, const float* bias, float* C,
                           const glf_gemm_params* p, glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    if (int rc = validate(p, A, B, C)) return rc;
    if (p->gather) GLF_REQUIRE((long long)p->n_img * p->hd * p->wd == p->M, GLF_ERR_BAD_SHAPE,
                               "gemm_nt: M (%d) != n_img*hd*wd", p->M);
    if (p->tap_mask == 0) return glf::fail(GLF_ERR_BAD_SHAPE, "gemm_nt: empty tap_mask");
    GemmArgs a = make_args(A, B, bias, C, p);
    a.vec_a = aligned16(A) && (p->lda % 4 == 0) && (p->batch_stride_a % 4 == 0);
    a.vec_b = aligned16(B) && (p->ldb % 4 == 0) && (p->batch_stride_b % 4 == 0) && (p->tap_stride_b % 4 == 0);
    dim3 grid(a.tiles_m * a.tiles_n, 1, p->batch);
    if (p->gather)
        hipLaunchKernelGGL((gemm_rows_kernel<0, true>), grid, dim3(NTHREADS), SMEM_ROWS_NT, glf::S(stream), a);
    else
        hipLaunchKernelGGL((gemm_rows_kernel<0, false>), grid, dim3(NTHREADS), SMEM_ROWS_NT, glf::S(stream), a);
    return glf::check_launch("gemm_nt");
}

extern "C" int glf_gemm_nn(const float* A, const float* B, const float* bias, float* C,
                           const glf_gemm_params* p, glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    if (int rc = validate(p, A, B, C)) return rc;
    if (p->gather) GLF_REQUIRE((long long)p->n_img * p->hd * p->wd == p->M, GLF_ERR_BAD_SHAPE,
                               "gemm_nn: M (%d) != n_img*hd*wd", p->M);
    if (p->tap_mask == 0) return glf::fail(GLF_ERR_BAD_SHAPE, "gemm_nn: empty tap_mask");
    GemmArgs a = make_args(A, B, bias, C, p);
    a.vec_a = aligned16(A) && (p->lda % 4 == 0) && (p->batch_stride_a % 4 == 0);
    a.vec_b = aligned16(B) && (p->ldb % 4 == 0) && (p->batch_stride_b % 4 == 0) && (p->tap_stride_b % 4 == 0);
    dim3 grid(a.tiles_m * a.tiles_n, 1, p->batch);
    if (p->gather)
        hipLaunchKernelGGL((gemm_rows_kernel<1, true>), grid, dim3(NTHREADS), SMEM_ROWS_NN, glf::S(stream), a);
    else
        hipLaunchKernelGGL((gemm_rows_kernel<1, false>), grid, dim3(NTHREADS), SMEM_ROWS_NN, glf::S(stream), a);
    return glf::check_launch("gemm_nn");
}

extern "C" int glf_gemm_tn(const float* A, const float* B, float* C,
                           const glf_gemm_params* p, glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    if (int rc = validate(p, A, B, C)) return rc;
    GLF_REQUIRE(p->gather != 2, GLF_ERR_UNSUPPORTED, "gemm_tn: transposed gather is not defined for the reduction form");
    if (p->gather) GLF_REQUIRE((long long)p->n_img * p->hd * p->wd == p->K, GLF_ERR_BAD_SHAPE,
                               "gemm_tn: K (%d rows) != n_img*hd*wd", p->K);
    GemmArgs a = make_args(A, B, nullptr, C, p);
    const int ntap = __builtin_popcount(p->tap_mask);
    if (ntap == 0) return GLF_OK;       // every tap in the padding: dW stays as the caller left it
    GLF_REQUIRE((long long)p->batch * a.split <= 65535, GLF_ERR_BAD_SHAPE, "gemm_tn: batch*split too large");
    a.vec_a = aligned16(A) && (p->lda % 4 == 0) && (p->batch_stride_a % 4 == 0);
    a.vec_b = aligned16(B) && (p->ldb % 4 == 0) && (p->batch_stride_b % 4 == 0);
    dim3 grid(a.tiles_m * a.tiles_n, ntap, p->batch * a.split);
    if (p->gather)
        hipLaunchKernelGGL((gemm_tn_kernel<true>), grid, dim3(NTHREADS), SMEM_TN, glf::S(stream), a);
    else
        hipLaunchKernelGGL((gemm_tn_kernel<false>), grid, dim3(NTHREADS), SMEM_TN, glf::S(stream), a);
    return glf::check_launch("gemm_tn");
}
