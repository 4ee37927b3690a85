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
#include "gemm_common.h"

namespace {

// 64x64 per wave, K = 32 from LDS tiles laid out [k][m] / [k][n].
//  * Two-level accumulation: the 32-deep K-tile is summed by the MFMA chain into fresh accumulators
//    (C = 0), which are then added to the running sums on the VALU.  A single f32 fma chain over
//    K = 9*2048 products has a ~sqrt(K) rounding walk; chains of 32 + K/32 keep the contraction at the
//    accuracy of a blocked CPU GEMM.
//  * The MFMA pipe (64 cycles per instruction) is the roof, so everything else is issued in its shadow:
//    operand ds_reads run one k-step ahead (double-buffered registers); `mid(p)`, p = 0..7, is called
//    between k-steps 4..11 and stages 1/8 of the NEXT K-tile into the other LDS buffer; the last four
//    k-steps run accumulator-major so that the c += t adds of three accumulators overlap the MFMAs of
//    the next one.
// Written as a macro (not a function taking a closure): nested lambdas that capture the staging registers by
// reference made hipcc materialise the closures -- and everything they point to -- in scratch memory.
// MID(pc) is a statement macro of the enclosing kernel; pc is a compile-time constant after unrolling.
#define GLF_LD(x) (x)
// MID(q), q = 0..11, is called after the MFMAs of head k-step q (HEAD_ = 12).
#define GLF_MMA_KTILE(LDA_, LDB_, a_s, b_s, MID)                                                              \
    {                                                                                                         \
        constexpr int NS_ = BK / 2, HEAD_ = NS_ - 4;                                                          \
        f32x16 t00 = {0}, t01 = {0}, t10 = {0}, t11 = {0};                                                    \
        float a0_[2], a1_[2], b0_[2], b1_[2];                                                                 \
        float ta0_[4], ta1_[4], tb0_[4], tb1_[4];                                                             \
        a0_[0] = GLF_LD((a_s)[0]); a1_[0] = GLF_LD((a_s)[32]); b0_[0] = GLF_LD((b_s)[0]); b1_[0] = GLF_LD((b_s)[32]); \
        _Pragma("unroll") for (int st = 0; st < HEAD_; ++st) {                                                \
            const int cur = st & 1, nxt = cur ^ 1;                                                            \
            if (st + 1 < HEAD_) {                                                                             \
                a0_[nxt] = GLF_LD((a_s)[(2 * st + 2) * LDA_]); a1_[nxt] = GLF_LD((a_s)[(2 * st + 2) * LDA_ + 32]); \
                b0_[nxt] = GLF_LD((b_s)[(2 * st + 2) * LDB_]); b1_[nxt] = GLF_LD((b_s)[(2 * st + 2) * LDB_ + 32]); \
            } else {                                                                                          \
                _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                               \
                    ta0_[q] = GLF_LD((a_s)[(2 * (HEAD_ + q)) * LDA_]); ta1_[q] = GLF_LD((a_s)[(2 * (HEAD_ + q)) * LDA_ + 32]); \
                    tb0_[q] = GLF_LD((b_s)[(2 * (HEAD_ + q)) * LDB_]); tb1_[q] = GLF_LD((b_s)[(2 * (HEAD_ + q)) * LDB_ + 32]); \
                }                                                                                             \
            }                                                                                                 \
            __builtin_amdgcn_sched_barrier(0); /* keep the prefetch ahead of this step's MFMAs */             \
            t00 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_[cur], b0_[cur], t00, 0, 0, 0);                     \
            t01 = __builtin_amdgcn_mfma_f32_32x32x2f32(a0_[cur], b1_[cur], t01, 0, 0, 0);                     \
            t10 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_[cur], b0_[cur], t10, 0, 0, 0);                     \
            t11 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1_[cur], b1_[cur], t11, 0, 0, 0);                     \
            { MID(st) }                                                                                       \
        }                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) t00 = __builtin_amdgcn_mfma_f32_32x32x2f32(ta0_[q], tb0_[q], t00, 0, 0, 0); \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) t01 = __builtin_amdgcn_mfma_f32_32x32x2f32(ta0_[q], tb1_[q], t01, 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        c00 += t00;                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) t10 = __builtin_amdgcn_mfma_f32_32x32x2f32(ta1_[q], tb0_[q], t10, 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        c01 += t01;                                                                                           \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) t11 = __builtin_amdgcn_mfma_f32_32x32x2f32(ta1_[q], tb1_[q], t11, 0, 0, 0); \
        __builtin_amdgcn_sched_barrier(0);                                                                    \
        c10 += t10;                                                                                           \
        c11 += t11;                                                                                           \
    }

__device__ __forceinline__ void scatter4(float* dst, int ld, const float4& v) {
    dst[0] = v.x; dst[ld] = v.y; dst[2 * ld] = v.z; dst[3 * ld] = v.w;
}

// ----------------------------------------------------------------------------------------
// rows kernel: C[m][n] = alpha * sum_tap sum_k A[src(m,tap)][k] * B_tap(k,n) + bias[n]
// BMODE 0: B_tap[n][k]; BMODE 1: B_tap[k][n]
// ----------------------------------------------------------------------------------------
template <int BMODE, bool GATHER>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_rows_kernel(const GemmArgs args) {
    // ---- scalar copies of the launch arguments (see GeoS) ----
    const int pM = args.M, pN = args.N, pK = args.K, p_lda = args.lda, p_ldb = args.ldb, p_ldc = args.ldc;
    const int p_taps = args.taps, p_gather = args.gather, p_accumulate = args.accumulate;
    const int p_tiles_n = args.tiles_n, p_vec_a = args.vec_a, p_vec_b = args.vec_b;
    const unsigned p_tap_mask = args.tap_mask;
    const long long p_tsb = args.tap_stride_b, p_bsa = args.bsa, p_bsb = args.bsb, p_bsc = args.bsc;
    const float p_alpha = args.alpha;
    const float* __restrict__ p_A = args.A; const float* __restrict__ p_B = args.B; const float* __restrict__ p_bias = args.bias;
    float* __restrict__ p_C = args.C;
    const int g_hs = args.g.hs, g_ws = args.g.ws, g_hd = args.g.hd, g_wd = args.g.wd, g_kw = args.g.kw;
    const int g_stride = args.g.stride, g_pad = args.g.pad, g_dil = args.g.dil;
    const int g_nimg = args.g.n_img, p_rect = GATHER ? args.rect : 0;

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
    const int tn = bid % p_tiles_n;
    int tm = bid / p_tiles_n;
    int pMe = pM;                                   // rows of this block's GEMM (rect mode: its tap's rectangle)
    int r_y0 = 0, r_x0 = 0, r_h = g_hd, r_w = g_wd;
    unsigned mask = p_tap_mask;
    if (p_rect) {                                   // tiles are laid out tap after tap
        for (unsigned mm = p_tap_mask; mm; mm &= mm - 1) {
            const int t = __ffs(mm) - 1;
            int y0, y1, x0, x1;
            tap_rect(p_gather, t, g_kw, g_pad, g_dil, g_hs, g_ws, g_hd, g_wd, y0, y1, x0, x1);
            const int mt = g_nimg * (y1 - y0) * (x1 - x0);
            const int tiles = (mt + BM - 1) / BM;
            if (tm < tiles || (mm & (mm - 1)) == 0) { mask = 1u << t; pMe = mt; r_y0 = y0; r_x0 = x0; r_h = y1 - y0; r_w = x1 - x0; break; }
            tm -= tiles;
        }
    }
    const int bz = blockIdx.z;
    const float* __restrict__ A = p_A + (long long)bz * p_bsa;
    const float* __restrict__ B = p_B + (long long)bz * p_bsb;
    float* __restrict__ C = p_C + (long long)bz * p_bsc;

    const int ac = tid & 7, ar = tid >> 3;          // k-contiguous staging: 8 float4 per 32-deep row
    const int bc = tid & 31, br = tid >> 5;         // row-contiguous staging (BMODE 1)

    int a_n[4], a_y[4], a_x[4];
    long long a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = tm * BM + ar + 32 * j;
        if (GATHER) {
            if (m < pMe) {
                const int hw = r_h * r_w;
                const int n = m / hw, rem = m - n * hw;
                const int yy = rem / r_w;
                a_n[j] = n; a_y[j] = r_y0 + yy; a_x[j] = r_x0 + rem - yy * r_w;
            } else { a_n[j] = -1; a_y[j] = 0; a_x[j] = 0; }
            a_off[j] = -1;
        } else {
            a_n[j] = 0; a_y[j] = 0; a_x[j] = 0;
            a_off[j] = (m < pM) ? (long long)m * p_lda : -1;
        }
    }

    if (GATHER && p_taps > 1 && !p_rect) {
        if (tid == 0) *s_mask = 0u;
        __syncthreads();
        if (ac == 0) {
            unsigned local = 0;
            for (unsigned mm = mask; mm; mm &= mm - 1) {
                const int t = __ffs(mm) - 1;
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (a_n[j] >= 0 && map_src(g_hs, g_ws, g_kw, g_stride, g_pad, g_dil, p_gather, a_n[j], a_y[j], a_x[j], t) >= 0) local |= 1u << t;
            }
            if (local) atomicOr(s_mask, local);
        }
        __syncthreads();
        mask &= *s_mask;
    }

    const int nkc = (pK + BK - 1) / BK;
    const int ntiles = __popc(mask) * nkc;

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};

    float4 ra[4], rb[4];
    unsigned rem_mask = mask;
    int tap = -1, kc = nkc;          // "before the first tile"
    const bool vec_a = p_vec_a, vec_b = p_vec_b;

    auto advance = [&]() __attribute__((always_inline)) {
        if (++kc >= nkc) {
            kc = 0;
            tap = __ffs(rem_mask) - 1;
            rem_mask &= rem_mask - 1;
            if (GATHER) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    int s = (a_n[j] >= 0) ? map_src(g_hs, g_ws, g_kw, g_stride, g_pad, g_dil, p_gather, a_n[j], a_y[j], a_x[j], tap) : -1;
                    a_off[j] = (s >= 0) ? (long long)s * p_lda : -1;
                }
            }
        }
    };
    // Fast path (block-uniform): every float4 is in range and 16-byte aligned, so the eight loads of a tile are
    // issued unconditionally from clamped addresses; rows that are padding / out of range are zeroed by a
    // select when the registers are written to LDS (store_piece), i.e. AFTER the loads' latency has been
    // hidden under MFMAs -- a select at load time would make the wave wait for the data right there.  Loads inside per-lane
    // branches make hipcc wait vmcnt(0) at every merge -- eight serialised L2 round trips per K-tile, measured
    // as a 20 % loss on the whole kernel.
    const bool fast = vec_a && vec_b && (pK % BK) == 0 && (BMODE == 0 || (pN % 4) == 0);
    auto load_tile = [&]() __attribute__((always_inline)) {
        const int kbase = kc * BK + 4 * ac;
        const float* Bt = B + (long long)tap * p_tsb;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            ra[j] = (a_off[j] >= 0) ? ld4(A + a_off[j] + kbase, pK - kbase, vec_a) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (BMODE == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = tn * BN + ar + 32 * j;
                rb[j] = (n < pN) ? ld4(Bt + (long long)n * p_ldb + kbase, pK - kbase, vec_b) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        } else {
            const int n0 = tn * BN + 4 * bc;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kc * BK + br + 8 * j;
                rb[j] = (k < pK) ? ld4(Bt + (long long)k * p_ldb + n0, pN - n0, vec_b) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    // piece p of the staged tile -> LDS (p < 4: A float4 #p, else B float4 #(p-4)); static indices only
    // constant array indices only (a lambda parameter index sends ra/rb/a_off to scratch: 3x slower)
#define GLF_SP_A(J)                                                                          \
    {                                                                                        \
        const float4 v = ra[J];                                                              \
        scatter4(As + buf * A_SZ + (4 * ac) * LDA + ar + 32 * J, LDA, v);                    \
    }
#define GLF_SP_B(J)                                                                          \
    {                                                                                        \
        if (BMODE == 0) {                                                                    \
            const float4 v = rb[J];                                                          \
            scatter4(Bs + buf * B_SZ + (4 * ac) * LDB + ar + 32 * J, LDB, v);                \
        } else {                                                                             \
            const float4 v = rb[J];                                                          \
            *reinterpret_cast<float4*>(Bs + buf * B_SZ + (br + 8 * J) * LDB + 4 * bc) = v;   \
        }                                                                                    \
    }
#define GLF_SP(pc)                       \
    switch (pc) {                        \
        case 0: GLF_SP_A(0) break;       \
        case 1: GLF_SP_A(1) break;       \
        case 2: GLF_SP_A(2) break;       \
        case 3: GLF_SP_A(3) break;       \
        case 4: GLF_SP_B(0) break;       \
        case 5: GLF_SP_B(1) break;       \
        case 6: GLF_SP_B(2) break;       \
        default: GLF_SP_B(3) break;      \
    }
#define GLF_SP_NEXT(q) if (has_next && (q) >= 4) { GLF_SP((q) - 4) }

    // ---- fast path state: per-thread source pointers advanced incrementally (no per-tile address maths) ----
    const float* pa[4];
    const float* pb[4];
    unsigned a_ok = 0;
    const long long b_step = (BMODE == 0) ? (long long)BK : (long long)BK * p_ldb;
    auto advance_fast = [&]() __attribute__((always_inline)) {
        if (++kc >= nkc) {
            kc = 0;
            tap = __ffs(rem_mask) - 1;
            rem_mask &= rem_mask - 1;
            a_ok = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                long long off;
                if (GATHER) {
                    const int sr = (a_n[j] >= 0) ? map_src(g_hs, g_ws, g_kw, g_stride, g_pad, g_dil, p_gather, a_n[j], a_y[j], a_x[j], tap) : -1;
                    off = (sr >= 0) ? (long long)sr * p_lda : -1;
                } else {
                    off = a_off[j];
                }
                a_ok |= (off >= 0 ? 1u : 0u) << j;
                pa[j] = A + (off >= 0 ? off : 0) + 4 * ac;
            }
            const float* Bt = B + (long long)tap * p_tsb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (BMODE == 0) pb[j] = Bt + (long long)min(tn * BN + ar + 32 * j, pN - 1) * p_ldb + 4 * ac;
                else pb[j] = Bt + (long long)(br + 8 * j) * p_ldb + min(tn * BN + 4 * bc, pN - 4);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) { pa[j] += BK; pb[j] += b_step; }
        }
    };
#define GLF_FLP(pc)                                                                         \
    switch (pc) {                                                                           \
        case 0: ra[0] = *reinterpret_cast<const float4*>(pa[0]); break;                     \
        case 1: ra[1] = *reinterpret_cast<const float4*>(pa[1]); break;                     \
        case 2: ra[2] = *reinterpret_cast<const float4*>(pa[2]); break;                     \
        case 3: ra[3] = *reinterpret_cast<const float4*>(pa[3]); break;                     \
        case 4: rb[0] = *reinterpret_cast<const float4*>(pb[0]); break;                     \
        case 5: rb[1] = *reinterpret_cast<const float4*>(pb[1]); break;                     \
        case 6: rb[2] = *reinterpret_cast<const float4*>(pb[2]); break;                     \
        default: rb[3] = *reinterpret_cast<const float4*>(pb[3]); break;                    \
    }
#define GLF_FSP_A(J) scatter4(As + buf * A_SZ + (4 * ac) * LDA + ar + 32 * J, LDA, keep_if((a_ok >> J) & 1u, ra[J]));
#define GLF_FSP_B(J)                                                                                          \
    {                                                                                                         \
        if (BMODE == 0) scatter4(Bs + buf * B_SZ + (4 * ac) * LDB + ar + 32 * J, LDB, keep_if(tn * BN + ar + 32 * J < pN, rb[J])); \
        else *reinterpret_cast<float4*>(Bs + buf * B_SZ + (br + 8 * J) * LDB + 4 * bc) = keep_if(tn * BN + 4 * bc < pN, rb[J]);    \
    }
#define GLF_FSP(pc)                      \
    switch (pc) {                        \
        case 0: GLF_FSP_A(0) break;      \
        case 1: GLF_FSP_A(1) break;      \
        case 2: GLF_FSP_A(2) break;      \
        case 3: GLF_FSP_A(3) break;      \
        case 4: GLF_FSP_B(0) break;      \
        case 5: GLF_FSP_B(1) break;      \
        case 6: GLF_FSP_B(2) break;      \
        default: GLF_FSP_B(3) break;     \
    }
    // k-steps 0..3 issue the eight loads of the next tile (two per step), k-steps 4..11 write them to LDS
#define GLF_FAST_MID(q)                                                       \
    if (has_next) {                                                           \
        if ((q) < 4) { GLF_FLP(2 * (q)) GLF_FLP(2 * (q) + 1) }                \
        else { GLF_FSP((q) - 4) }                                             \
    }

    if (ntiles > 0) {
        const int a_lane = (lane >> 5) * LDA + wm + (lane & 31);
        const int b_lane = (lane >> 5) * LDB + wn + (lane & 31);
        if (fast) {
            advance_fast();
            {
                const int buf = 0;
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) { GLF_FLP(pc) }
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) { GLF_FSP(pc) }
            }
            __syncthreads();
            for (int it = 0; it < ntiles; ++it) {
                const bool has_next = (it + 1) < ntiles;
                if (has_next) advance_fast();
                const float* a_sp = As + (it & 1) * A_SZ + a_lane;
                const float* b_sp = Bs + (it & 1) * B_SZ + b_lane;
                const int buf = (it & 1) ^ 1;      // the staging macros write the OTHER buffer (free since the last barrier)
                GLF_MMA_KTILE(LDA, LDB, a_sp, b_sp, GLF_FAST_MID)
                __syncthreads();
            }
        } else {
            advance();
            load_tile();
            {
                const int buf = 0;
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) { GLF_SP(pc) }
            }
            __syncthreads();
            for (int it = 0; it < ntiles; ++it) {
                const bool has_next = (it + 1) < ntiles;
                if (has_next) { advance(); load_tile(); }
                const float* a_sp = As + (it & 1) * A_SZ + a_lane;
                const float* b_sp = Bs + (it & 1) * B_SZ + b_lane;
                const int buf = (it & 1) ^ 1;
                GLF_MMA_KTILE(LDA, LDB, a_sp, b_sp, GLF_SP_NEXT)
                __syncthreads();
            }
        }
    }

    // epilogue: lane holds column (lane&31), rows (r&3) + 8*(r>>2) + 4*(lane>>5)
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    auto emit = [&](const f32x16& acc, int ti, int tj) {
        const int col = tn * BN + wn + 32 * tj + col_l;
        if (col >= pN) return;
        const float bv = p_bias ? p_bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * BM + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + row_l;
            if (row < pMe) {
                if (p_rect) {                               // rectangle row -> output pixel; taps meet in atomics
                    const int hw = r_h * r_w;
                    const int n = row / hw, rem = row - n * hw;
                    const int yy = rem / r_w;
                    const long long orow = ((long long)n * g_hd + r_y0 + yy) * g_wd + r_x0 + (rem - yy * r_w);
                    atomicAdd(C + orow * p_ldc + col, p_alpha * acc[r]);
                } else {
                    float* dst = C + (long long)row * p_ldc + col;
                    float v = p_alpha * acc[r] + bv;
                    if (p_accumulate) v += *dst;
                    *dst = v;
                }
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
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_kernel(const GemmArgs args) {
    // ---- scalar copies of the launch arguments (see GeoS) ----
    const int pM = args.M, pN = args.N, pK = args.K, p_lda = args.lda, p_ldb = args.ldb, p_ldc = args.ldc;
    const int p_accumulate = args.accumulate, p_split = args.split;
    const int p_tiles_n = args.tiles_n, p_vec_a = args.vec_a, p_vec_b = args.vec_b;
    const unsigned p_tap_mask = args.tap_mask;
    const long long p_tsb = args.tap_stride_b, p_bsa = args.bsa, p_bsb = args.bsb, p_bsc = args.bsc;
    const float p_alpha = args.alpha;
    const float* __restrict__ p_A = args.A; const float* __restrict__ p_B = args.B;
    float* __restrict__ p_C = args.C;
    const int g_hs = args.g.hs, g_ws = args.g.ws, g_hd = args.g.hd, g_wd = args.g.wd, g_kw = args.g.kw;
    const int g_stride = args.g.stride, g_pad = args.g.pad, g_dil = args.g.dil;
    const int g_nimg = args.g.n_img, p_rect = GATHER ? args.rect : 0;

    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int LD = LD_V;
    constexpr int T_SZ = BK * LD;
    float* As = smem;
    float* Bs = smem + 2 * T_SZ;
    int* vflag = reinterpret_cast<int*>(smem + 4 * T_SZ);      // [2][32] row-valid flags

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p_tiles_n, tm = bid / p_tiles_n;
    // nth set bit of the tap mask
    int tap;
    {
        unsigned mm = p_tap_mask;
        for (int i = 0; i < (int)blockIdx.y; ++i) mm &= mm - 1;
        tap = __ffs(mm) - 1;
    }
    const int bz = blockIdx.z / p_split, sl = blockIdx.z - bz * p_split;
    const float* __restrict__ A = p_A + (long long)bz * p_bsa;
    const float* __restrict__ B = p_B + (long long)bz * p_bsb;
    // two-stage split-K: with a partial-sum workspace every (batch, slice, tap) block row stores its own [M][N] slab
    // (plain stores, no atomics, no zero fill); tn_reduce_kernel sums the slabs in a fixed order
    float* __restrict__ C = args.partial ? args.partial + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * ((long long)pM * pN)
                                         : p_C + (long long)bz * p_bsc + (long long)tap * p_tsb;
    const int ldc_e = args.partial ? pN : p_ldc;

    // rect mode: the reduction runs over this tap's rectangle of output pixels only (see tap_rect)
    int r_y0 = 0, r_x0 = 0, r_h = g_hd, r_w = g_wd, pKe = pK;
    if (p_rect) {
        int y0, y1, x0, x1;
        tap_rect(1, tap, g_kw, g_pad, g_dil, g_hs, g_ws, g_hd, g_wd, y0, y1, x0, x1);
        r_y0 = y0; r_x0 = x0; r_h = y1 - y0; r_w = x1 - x0;
        pKe = g_nimg * r_h * r_w;
    }
    int chunk = (pK + p_split - 1) / p_split;       // slices are cut from the FULL reduction length: a tap with a short rectangle uses fewer of them
    chunk = ((chunk + BK - 1) / BK) * BK;
    const int r0 = sl * chunk;
    const int r1 = min(pKe, r0 + chunk);
    if (r0 >= r1) return;                                      // block-uniform

    const int c4 = tid & 31, rr = tid >> 5;
    const int m0 = tm * BM + 4 * c4, n0 = tn * BN + 4 * c4;
    const bool vec_a = p_vec_a, vec_b = p_vec_b;
    const int hw = GATHER ? r_h * r_w : 1;

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    float4 ra[4], rb[4];
    int rvalid[4];

    const bool fast = vec_a && vec_b && (pM % 4) == 0 && (pN % 4) == 0;
    const int m0c = min(m0, pM - 4), n0c = min(n0, pN - 4);
    auto load_tile = [&](int rbase) __attribute__((always_inline)) {
        long long src[4], arow[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = rbase + rr + 8 * j;
            src[j] = -1;
            arow[j] = min(r, r1 - 1);
            if (GATHER) {
                const int rc = min(r, r1 - 1);
                const int n = rc / hw, rem = rc - n * hw;
                const int yy = rem / r_w;
                const int y = r_y0 + yy, x = r_x0 + rem - yy * r_w;
                arow[j] = ((long long)n * g_hd + y) * g_wd + x;
                if (r < r1) src[j] = map_src(g_hs, g_ws, g_kw, g_stride, g_pad, g_dil, 1, n, y, x, tap);
            } else if (r < r1) {
                src[j] = r;
            }
            rvalid[j] = src[j] >= 0;
        }
        if (fast) {          // unconditional loads from clamped addresses + select (see gemm_rows_kernel)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float4 va = *reinterpret_cast<const float4*>(A + arow[j] * p_lda + m0c);
                const float4 vb = *reinterpret_cast<const float4*>(B + (src[j] >= 0 ? src[j] : 0) * p_ldb + n0c);
                ra[j] = va;                  // zeroed at store time (rvalid / column range), not here
                rb[j] = vb;
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (src[j] >= 0) {
                ra[j] = ld4(A + arow[j] * p_lda + m0, pM - m0, vec_a);
                rb[j] = ld4(B + src[j] * p_ldb + n0, pN - n0, vec_b);
            } else {
                ra[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                rb[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
#define GLF_TP_A(J)                                                                          \
    {                                                                                        \
        const float4 v = keep_if(!fast || (rvalid[J] && m0 < pM), ra[J]);                 \
        *reinterpret_cast<float4*>(As + buf * T_SZ + (rr + 8 * J) * LD + 4 * c4) = v;        \
        if (c4 == 0) vflag[buf * 32 + rr + 8 * J] = rvalid[J];                               \
    }
#define GLF_TP_B(J)                                                                          \
    {                                                                                        \
        const float4 v = keep_if(!fast || (rvalid[J] && n0 < pN), rb[J]);                 \
        *reinterpret_cast<float4*>(Bs + buf * T_SZ + (rr + 8 * J) * LD + 4 * c4) = v;        \
    }
#define GLF_TP(pc)                       \
    switch (pc) {                        \
        case 0: GLF_TP_A(0) break;       \
        case 1: GLF_TP_A(1) break;       \
        case 2: GLF_TP_A(2) break;       \
        case 3: GLF_TP_A(3) break;       \
        case 4: GLF_TP_B(0) break;       \
        case 5: GLF_TP_B(1) break;       \
        case 6: GLF_TP_B(2) break;       \
        default: GLF_TP_B(3) break;      \
    }
#define GLF_TP_NEXT(q) if (has_next && (q) >= 4) { GLF_TP((q) - 4) }

    load_tile(r0);
    {
        const int buf = 0;
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) { GLF_TP(pc) }
    }
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
        {
            const float* a_sp = As + buf * T_SZ + a_lane;
            const float* b_sp = Bs + buf * T_SZ + b_lane;
            const int cbuf = buf;
            const int buf = cbuf ^ 1;              // GLF_TP writes the OTHER buffer
            if (any) {
                GLF_MMA_KTILE(LD, LD, a_sp, b_sp, GLF_TP_NEXT)
            } else if (has_next) {
#pragma unroll
                for (int pc = 0; pc < 8; ++pc) { GLF_TP(pc) }
            }
        }
        __syncthreads();
    }

    const bool atomic = !args.partial && ((p_split > 1) || p_accumulate);
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    auto emit = [&](const f32x16& acc, int ti, int tj) {
        const int col = tn * BN + wn + 32 * tj + col_l;
        if (col >= pN) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * BM + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + row_l;
            if (row < pM) {
                float* dst = C + (long long)row * ldc_e + col;
                const float v = p_alpha * acc[r];
                if (atomic) atomicAdd(dst, v); else *dst = v;
            }
        }
    };
    emit(c00, 0, 0); emit(c01, 0, 1); emit(c10, 1, 0); emit(c11, 1, 1);
}

// ----------------------------------------------------------------------------------------
// second stage of the two-stage split-K reduction of glf_gemm_tn: C_tap = (accumulate ? C_tap : 0) + sum over the valid
// slices of their partial slabs, in slice order (bitwise reproducible).  grid: x over M*N, y = kept-tap ordinal, z = batch.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tn_reduce_kernel(const float* __restrict__ partial, float* __restrict__ Cb, int M, int N, int ldc,
                                                        long long tsb, long long bsc, int split, unsigned tap_mask, int accumulate,
                                                        int K, int rect, Geo g, int vec, float* __restrict__ amax_c) {
    unsigned mm = tap_mask;
    for (int i = 0; i < (int)blockIdx.y; ++i) mm &= mm - 1;
    const int tap = __ffs(mm) - 1;
    const int ntap = gridDim.y, bz = blockIdx.z;
    int pKe = K;
    if (rect) {
        int y0, y1, x0, x1;
        tap_rect(1, tap, g.kw, g.pad, g.dil, g.hs, g.ws, g.hd, g.wd, y0, y1, x0, x1);
        pKe = g.n_img * (y1 - y0) * (x1 - x0);
    }
    int chunk = (K + split - 1) / split;                // as in the kernels: from the full reduction length
    chunk = ((chunk + BK - 1) / BK) * BK;
    const int nvalid = chunk > 0 ? min(split, (pKe + chunk - 1) / chunk) : 0;       // slices that ran (the others returned early)
    const long long mn = (long long)M * N;
    const float* __restrict__ src = partial + ((long long)bz * split * ntap + blockIdx.y) * mn;
    const long long slice_stride = (long long)ntap * mn;
    float* __restrict__ C = Cb + (long long)bz * bsc + (long long)tap * tsb;
    float cmax = 0.f;
    if (vec) {
        const int n4 = N >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)M * n4; i += (long long)gridDim.x * blockDim.x) {
            const long long row = i / n4;
            const int c4 = (int)(i - row * n4) * 4;
            float* dst = C + row * ldc + c4;
            // (the slabs are added in double: the second stage must not add a rounding walk of its own to the slices' fp32 chains)
            double ax = 0, ay = 0, az = 0, aw = 0;
            if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); ax = o.x; ay = o.y; az = o.z; aw = o.w; }
            const float* sp = src + row * N + c4;
            for (int sl = 0; sl < nvalid; ++sl) {
                const float4 v = *reinterpret_cast<const float4*>(sp + sl * slice_stride);
                ax += v.x; ay += v.y; az += v.z; aw += v.w;
            }
            const float4 acc = make_float4((float)ax, (float)ay, (float)az, (float)aw);
            *reinterpret_cast<float4*>(dst) = acc;
            cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < mn; i += (long long)gridDim.x * blockDim.x) {
            const long long row = i / N;
            const int col = (int)(i - row * N);
            float* dst = C + row * ldc + col;
            double accd = accumulate ? (double)*dst : 0.0;
            for (int sl = 0; sl < nvalid; ++sl) accd += src[sl * slice_stride + i];
            const float acc = (float)accd;
            *dst = acc;
            cmax = fmaxf(cmax, fabsf(acc));
        }
    }
    if (amax_c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
        if ((threadIdx.x & 63) == 0 && cmax > 0.f) atomicMax(reinterpret_cast<unsigned*>(amax_c), __float_as_uint(cmax));
    }
}

// the vector form with SL slice lanes per output (lane l adds slices l, l + SL, ... in double; the lanes meet in LDS in lane order):
// with one thread per float4 a layer-1 gradient (64 x 256 outputs, hundreds of slices) was a few thousand threads walking hundreds
// of loads one after the other -- ~100 us for 24 MB
template <int SL>
__global__ __launch_bounds__(256) void tn_reduce_lanes_kernel(const float* __restrict__ partial, float* __restrict__ Cb, int M, int N, int ldc,
                                                              long long tsb, long long bsc, int split, unsigned tap_mask, int accumulate,
                                                              int K, int rect, Geo g, float* __restrict__ amax_c) {
    constexpr int OG = 256 / SL;
    __shared__ double sh[4][256];
    unsigned mm = tap_mask;
    for (int i = 0; i < (int)blockIdx.y; ++i) mm &= mm - 1;
    const int tap = __ffs(mm) - 1;
    const int ntap = gridDim.y, bz = blockIdx.z;
    int pKe = K;
    if (rect) {
        int y0, y1, x0, x1;
        tap_rect(1, tap, g.kw, g.pad, g.dil, g.hs, g.ws, g.hd, g.wd, y0, y1, x0, x1);
        pKe = g.n_img * (y1 - y0) * (x1 - x0);
    }
    int chunk = (K + split - 1) / split;
    chunk = ((chunk + BK - 1) / BK) * BK;
    const int nvalid = chunk > 0 ? min(split, (pKe + chunk - 1) / chunk) : 0;
    const long long mn = (long long)M * N;
    const float* __restrict__ src = partial + ((long long)bz * split * ntap + blockIdx.y) * mn;
    const long long slice_stride = (long long)ntap * mn;
    float* __restrict__ C = Cb + (long long)bz * bsc + (long long)tap * tsb;
    const int n4 = N >> 2;
    const long long total = (long long)M * n4;
    const int ol = threadIdx.x % OG, sl = threadIdx.x / OG;
    float cmax = 0.f;
    for (long long base = (long long)blockIdx.x * OG; base < total; base += (long long)gridDim.x * OG) {
        const long long i = base + ol;
        const bool live = i < total;
        const long long row = live ? i / n4 : 0;
        const int c4 = (int)((live ? i : 0) - row * n4) * 4;
        double ax = 0, ay = 0, az = 0, aw = 0;
        if (live) {
            const float* sp = src + row * N + c4;
            int s_ = sl;
            for (; s_ + SL < nvalid; s_ += 2 * SL) {
                const float4 v0 = *reinterpret_cast<const float4*>(sp + (long long)s_ * slice_stride);
                const float4 v1 = *reinterpret_cast<const float4*>(sp + (long long)(s_ + SL) * slice_stride);
                ax += (double)v0.x + (double)v1.x; ay += (double)v0.y + (double)v1.y;
                az += (double)v0.z + (double)v1.z; aw += (double)v0.w + (double)v1.w;
            }
            for (; s_ < nvalid; s_ += SL) {
                const float4 v = *reinterpret_cast<const float4*>(sp + (long long)s_ * slice_stride);
                ax += v.x; ay += v.y; az += v.z; aw += v.w;
            }
        }
        sh[0][threadIdx.x] = ax; sh[1][threadIdx.x] = ay; sh[2][threadIdx.x] = az; sh[3][threadIdx.x] = aw;
        __syncthreads();
        if (sl == 0 && live) {
#pragma unroll
            for (int l = 1; l < SL; ++l) { ax += sh[0][l * OG + ol]; ay += sh[1][l * OG + ol]; az += sh[2][l * OG + ol]; aw += sh[3][l * OG + ol]; }
            float* dst = C + row * ldc + c4;
            if (accumulate) { const float4 o = *reinterpret_cast<const float4*>(dst); ax += o.x; ay += o.y; az += o.z; aw += o.w; }
            const float4 acc = make_float4((float)ax, (float)ay, (float)az, (float)aw);
            *reinterpret_cast<float4*>(dst) = acc;
            cmax = fmaxf(fmaxf(cmax, fmaxf(fabsf(acc.x), fabsf(acc.y))), fmaxf(fabsf(acc.z), fabsf(acc.w)));
        }
        __syncthreads();
    }
    if (amax_c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
        if ((threadIdx.x & 63) == 0 && cmax > 0.f) atomicMax(reinterpret_cast<unsigned*>(amax_c), __float_as_uint(cmax));
    }
}

constexpr size_t SMEM_ROWS_NT = (2 * BK * LD_T + 2 * BK * LD_T) * sizeof(float) + 16;
constexpr size_t SMEM_ROWS_NN = (2 * BK * LD_T + 2 * BK * LD_V) * sizeof(float) + 16;
constexpr size_t SMEM_TN = (4 * BK * LD_V) * sizeof(float) + 2 * 32 * sizeof(int);

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
    if (int rc = init_gemm_bf16s_attrs()) return rc;
    return init_gemm_f16s_attrs();
}
}  // namespace glf

namespace {
// f16x3 with amax_a / amax_b == NULL: measure the operands (exactly the elements the call can read) first
int self_amax(GemmArgs& a, const glf_gemm_params* p, bool tn, hipStream_t s) {
    if (a.amax_a && a.amax_b) return GLF_OK;
    float* slots = glf::amax_scratch(2, s);
    GLF_REQUIRE(slots != nullptr, GLF_ERR_WORKSPACE, "gemm(f16x3): no amax scratch");
    hipError_t e = hipMemsetAsync(slots, 0, 2 * sizeof(float), s);
    if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "hipMemsetAsync(amax): %s", hipGetErrorString(e));
    const long long src_rows = p->gather ? (long long)p->n_img * p->hs * p->ws : 0;
    for (int b = 0; b < p->batch; ++b) {
        if (!a.amax_a) {
            const long long rows = tn ? p->K : (p->gather ? src_rows : p->M);
            const int cols = tn ? p->M : p->K;
            if (int rc = glf::launch_amax(a.A + b * a.bsa, rows, cols, p->lda, a.vec_a, slots, s)) return rc;
        }
        if (!a.amax_b) {
            if (tn) {
                const long long rows = p->gather ? src_rows : p->K;
                if (int rc = glf::launch_amax(a.B + b * a.bsb, rows, p->N, p->ldb, a.vec_b, slots + 1, s)) return rc;
            } else {
                for (unsigned mm = p->tap_mask; mm; mm &= mm - 1) {
                    const int t = __builtin_ctz(mm);
                    if (int rc = glf::launch_amax(a.B + b * a.bsb + t * a.tap_stride_b, p->N, p->K, p->ldb, a.vec_b, slots + 1, s)) return rc;
                }
            }
        }
    }
    if (!a.amax_a) a.amax_a = slots;
    if (!a.amax_b) a.amax_b = slots + 1;
    return GLF_OK;
}
}  // namespace

extern "C" int glf_amax(const float* x, int64_t rows, int cols, int64_t ld, float* out, glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && out, GLF_ERR_NULL, "amax: null argument");
    GLF_REQUIRE(rows > 0 && cols > 0 && ld >= cols, GLF_ERR_BAD_SHAPE, "amax: bad extents");
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), glf::S(stream));
    if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "hipMemsetAsync(amax): %s", hipGetErrorString(e));
    return glf::launch_amax(x, rows, cols, ld, aligned16(x) && (ld % 4 == 0), out, glf::S(stream));
}

extern "C" int glf_split_f16_packed(const float* x, int64_t rows, int cols, int64_t ld, const float* amax, float* out, int64_t ldo,
                                    glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(x && amax && out, GLF_ERR_NULL, "split_f16_packed: null argument");
    GLF_REQUIRE(rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && ldo % 4 == 0 && ld >= cols && ldo >= cols, GLF_ERR_BAD_SHAPE,
                "split_f16_packed: cols and the row strides must be multiples of 4");
    GLF_REQUIRE(aligned16(x) && aligned16(out), GLF_ERR_BAD_SHAPE, "split_f16_packed: x and out must be 16-byte aligned");
    return glf::launch_split_packed(x, rows, cols, ld, amax, out, ldo, glf::S(stream));
}

// ----------------------------------------------------------------------------------------------------------
// Skinny NT: M <= 64 rows (the ASPP pooled branch, deeplabv3.py:123-135: ONE row per frame).  The tile kernels put such a
// shape on one or two workgroups that walk the whole K alone -- 64 dependent global-memory round trips, 0.2 ms for 34 MFLOP
// at 64 x 256 x 2048.  Here a workgroup owns SK_COLS output columns for all rows: thread = (row, one of four k-lanes), the
// B rows are the same address across the rows of a wave (broadcast out of L1), four independent A loads in flight per thread,
// the k-lanes folded with two shuffles.  fp32 FMA arithmetic (exact-fp32 grade under every precision setting).
// ----------------------------------------------------------------------------------------------------------
constexpr int SK_COLS = 2;
__global__ __launch_bounds__(256) void gemm_skinny_nt_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ bias,
                                                            float* __restrict__ C, int M, int N, int K, int lda, int ldb, int ldc, float alpha,
                                                            int accumulate, float* __restrict__ amax_c) {
    const int tid = threadIdx.x, kl = tid & 3, m = tid >> 2;
    const int n0 = blockIdx.x * SK_COLS;
    const float* a = A + (long long)min(m, M - 1) * lda + 4 * kl;
    const float* b[SK_COLS];
#pragma unroll
    for (int j = 0; j < SK_COLS; ++j) b[j] = B + (long long)min(n0 + j, N - 1) * ldb + 4 * kl;
    float acc[SK_COLS] = {};
    int k = 0;
    for (; k + 64 <= K; k += 64) {
        float4 av[4], bv[SK_COLS][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) av[u] = *reinterpret_cast<const float4*>(a + k + 16 * u);
#pragma unroll
        for (int j = 0; j < SK_COLS; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u) bv[j][u] = *reinterpret_cast<const float4*>(b[j] + k + 16 * u);
#pragma unroll
        for (int j = 0; j < SK_COLS; ++j)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                acc[j] += av[u].x * bv[j][u].x + av[u].y * bv[j][u].y + av[u].z * bv[j][u].z + av[u].w * bv[j][u].w;
    }
    for (; k < K; k += 16) {
        const float4 av = *reinterpret_cast<const float4*>(a + k);
#pragma unroll
        for (int j = 0; j < SK_COLS; ++j) {
            const float4 bv = *reinterpret_cast<const float4*>(b[j] + k);
            acc[j] += av.x * bv.x + av.y * bv.y + av.z * bv.z + av.w * bv.w;
        }
    }
    float cmax = 0.f;
#pragma unroll
    for (int j = 0; j < SK_COLS; ++j) {
        float v = acc[j];
        v += __shfl_xor(v, 1);
        v += __shfl_xor(v, 2);
        if (kl == 0 && m < M && n0 + j < N) {
            v = v * alpha + (bias ? bias[n0 + j] : 0.f);
            float* c = C + (long long)m * ldc + n0 + j;
            if (accumulate) v += *c;
            *c = v;
            cmax = fmaxf(cmax, fabsf(v));
        }
    }
    if (amax_c) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o));
        if ((tid & 63) == 0 && cmax > 0.f) atomicMax(reinterpret_cast<unsigned*>(amax_c), __float_as_uint(cmax));
    }
}

extern "C" int glf_gemm_nt(const float* A, const float* B, const float* bias, float* C,
                           const glf_gemm_params* p, glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    if (int rc = validate(p, A, B, C)) return rc;
    if (p->gather) GLF_REQUIRE((long long)p->n_img * p->hd * p->wd == p->M, GLF_ERR_BAD_SHAPE,
                               "gemm_nt: M (%d) != n_img*hd*wd", p->M);
    if (p->tap_mask == 0) return glf::fail(GLF_ERR_BAD_SHAPE, "gemm_nt: empty tap_mask");
    GemmArgs a = make_args(A, B, bias, C, p);
    const int prec = call_precision(p);
    a.vec_a = aligned16(A) && (p->lda % 4 == 0) && (p->batch_stride_a % 4 == 0);
    a.vec_b = aligned16(B) && (p->ldb % 4 == 0) && (p->batch_stride_b % 4 == 0) && (p->tap_stride_b % 4 == 0);
    dim3 grid(a.tiles_m * a.tiles_n, 1, p->batch);
    if (p->M <= 64 && !p->gather && p->taps == 1 && p->batch == 1 && !p->rect && !p->colstats && !p->a_presplit && !p->b_presplit && a.vec_a &&
        a.vec_b && p->K % 16 == 0 && p->N >= 64) {
        hipLaunchKernelGGL(gemm_skinny_nt_kernel, dim3((p->N + SK_COLS - 1) / SK_COLS), dim3(256), 0, glf::S(stream), A, B, bias, C, p->M, p->N, p->K,
                           p->lda, p->ldb, p->ldc, p->alpha, p->accumulate, p->amax_c);
        return glf::check_launch("gemm_nt(skinny)");
    }
    if (p->rect == 2) {                 // region mode: f16x3 kernels only (see region_of in gemm_common.h)
        GLF_REQUIRE(prec >= 2 && glf::f16s_rows_ok(a), GLF_ERR_UNSUPPORTED,
                    "glf_gemm_nt: rect = 2 (region mode) exists on the f16x3 kernels only (precision 2, K %% 32 == 0, aligned operands)");
        GLF_REQUIRE(p->gather != 0 && p->kh == 3 && p->kw == 3 && p->stride == 1 && p->pad == p->dil && p->hs == p->hd && p->ws == p->wd &&
                    p->batch == 1, GLF_ERR_UNSUPPORTED, "glf_gemm_nt: region mode needs a 3x3 stride-1 conv with pad == dil on equal maps, batch 1");
        a.rect = 2;
    } else if (p->rect) {
        if (int rc = setup_rect(p, bias, a, grid, "glf_gemm_nt")) return rc;
    }
    GLF_REQUIRE(!p->colstats || (prec >= 2 && glf::f16s_rows_ok(a) && p->rect != 1 && p->batch == 1), GLF_ERR_UNSUPPORTED,
                "glf_gemm_nt: colstats is honoured by the f16x3 kernels only (precision 2, K %% 32 == 0, aligned operands, no rect = 1, batch 1)");
    // colmax is produced by the wide-store (LDS-parked) epilogue of the split-fp16 kernels only: refuse it wherever a call would run
    // another epilogue (a zero colmax would make glf_bn_apply_from_sums bound max|y| too low and the packed image overflow its scale)
    GLF_REQUIRE(!p->colmax || (p->colstats && prec >= 2 && glf::f16s_rows_ok(a) && p->rect != 1 && p->ldc % 4 == 0 && p->N % 4 == 0 && aligned16(C) &&
                               p->batch_stride_c % 4 == 0),
                GLF_ERR_UNSUPPORTED, "glf_gemm_nt: colmax needs colstats, precision 3 / 4 on the aligned fast path, no rect = 1, and a 16-byte aligned C with "
                "ldc, N and batch_stride_c multiples of 4 (the wide-store epilogue is the one that folds it)");
    if (prec == 1 && glf::bf16s_rows_ok(a)) return glf::launch_rows_bf16s(a, grid, p->gather != 0, glf::S(stream));
    if (p->a_presplit || p->b_presplit)
        GLF_REQUIRE(prec >= 2 && glf::f16s_rows_ok(a) && (!p->a_presplit || p->amax_a) && (!p->b_presplit || p->amax_b), GLF_ERR_UNSUPPORTED,
                    "glf_gemm_nt: a pre-split operand needs precision 3 / 4, the aligned fast path and the amax it was split with");
    if (prec >= 2 && glf::f16s_rows_ok(a)) {
        if (int rc = self_amax(a, p, false, glf::S(stream))) return rc;
        return glf::launch_rows_f16s(a, grid, p->gather != 0, prec == 2 ? 3 : 1, glf::S(stream));
    }
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
    GLF_REQUIRE(!p->colstats, GLF_ERR_UNSUPPORTED, "glf_gemm_nn: colstats is honoured by glf_gemm_nt on the f16x3 kernels only");
    GLF_REQUIRE(p->rect != 2, GLF_ERR_UNSUPPORTED, "glf_gemm_nn: rect = 2 (region mode) is built for glf_gemm_nt on the f16x3 kernels only");
    if (p->rect) {
        if (int rc = setup_rect(p, bias, a, grid, "glf_gemm_nn")) return rc;
    }
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
    GLF_REQUIRE(!p->colstats, GLF_ERR_UNSUPPORTED, "gemm_tn: colstats is honoured by glf_gemm_nt on the f16x3 kernels only");
    if (p->gather) GLF_REQUIRE((long long)p->n_img * p->hd * p->wd == p->K, GLF_ERR_BAD_SHAPE,
                               "gemm_tn: K (%d rows) != n_img*hd*wd", p->K);
    GemmArgs a = make_args(A, B, nullptr, C, p);
    const int ntap = __builtin_popcount(p->tap_mask);
    if (ntap == 0) return GLF_OK;       // every tap in the padding: dW stays as the caller left it
    GLF_REQUIRE((long long)p->batch * a.split <= 65535, GLF_ERR_BAD_SHAPE, "gemm_tn: batch*split too large");
    a.vec_a = aligned16(A) && (p->lda % 4 == 0) && (p->batch_stride_a % 4 == 0);
    a.vec_b = aligned16(B) && (p->ldb % 4 == 0) && (p->batch_stride_b % 4 == 0);
    if (p->rect) {
        GLF_REQUIRE(p->gather == 1 && p->stride == 1 && p->batch == 1, GLF_ERR_UNSUPPORTED,
                    "gemm_tn: rect mode needs a stride-1 forward conv gather and batch 1");
        a.rect = 1;
    }
    dim3 grid(a.tiles_m * a.tiles_n, ntap, p->batch * a.split);
    const int prec = call_precision(p);
    const bool two_stage = a.split > 1 && p->workspace != nullptr;
    if (two_stage) {
        GLF_REQUIRE(p->workspace_bytes >= (int64_t)glf_gemm_tn_workspace_bytes(p), GLF_ERR_WORKSPACE,
                    "gemm_tn: workspace of %lld bytes, glf_gemm_tn_workspace_bytes() asks for %zu", (long long)p->workspace_bytes,
                    glf_gemm_tn_workspace_bytes(p));
        GLF_REQUIRE(aligned16(p->workspace), GLF_ERR_WORKSPACE, "gemm_tn: workspace must be 16-byte aligned");
        a.partial = p->workspace;
    }
    if (p->a_presplit || p->b_presplit)
        GLF_REQUIRE(prec >= 2 && glf::f16s_tn_ok(a) && (!p->a_presplit || p->amax_a) && (!p->b_presplit || p->amax_b), GLF_ERR_UNSUPPORTED,
                    "glf_gemm_tn: a pre-split operand needs precision 3 / 4, the aligned fast path and the amax it was split with");
    int rc;
    if (prec == 1 && glf::bf16s_tn_ok(a)) {
        rc = glf::launch_tn_bf16s(a, grid, p->gather != 0, glf::S(stream));
    } else if (prec >= 2 && glf::f16s_tn_ok(a)) {
        if (int rc2 = self_amax(a, p, true, glf::S(stream))) return rc2;
        rc = glf::launch_tn_f16s(a, grid, p->gather != 0, prec == 2 ? 3 : 1, glf::S(stream));
    } else {
        if (p->gather)
            hipLaunchKernelGGL((gemm_tn_kernel<true>), grid, dim3(NTHREADS), SMEM_TN, glf::S(stream), a);
        else
            hipLaunchKernelGGL((gemm_tn_kernel<false>), grid, dim3(NTHREADS), SMEM_TN, glf::S(stream), a);
        rc = glf::check_launch("gemm_tn");
    }
    if (rc != GLF_OK || !two_stage) return rc;
    const int vec = (p->N % 4 == 0) && (p->ldc % 4 == 0) && aligned16(C) && (p->tap_stride_b % 4 == 0) && (p->batch_stride_c % 4 == 0);
    const long long work = vec ? (long long)p->M * (p->N / 4) : (long long)p->M * p->N;
    long long bx = (work + 255) / 256;
    if (bx > 2048) bx = 2048;
    int lanes = 1;
    while (vec && lanes < 16 && lanes * 4 <= a.split && work * lanes < 65536) lanes *= 4;
    if (lanes > 1) {
        long long bl = (work * lanes + 255) / 256;
        if (bl > 4096) bl = 4096;
        if (lanes == 16)
            hipLaunchKernelGGL((tn_reduce_lanes_kernel<16>), dim3((unsigned)bl, ntap, p->batch), dim3(256), 0, glf::S(stream), a.partial, C, p->M, p->N,
                               p->ldc, (long long)p->tap_stride_b, (long long)p->batch_stride_c, a.split, p->tap_mask, p->accumulate, p->K, a.rect, a.g, p->amax_c);
        else
            hipLaunchKernelGGL((tn_reduce_lanes_kernel<4>), dim3((unsigned)bl, ntap, p->batch), dim3(256), 0, glf::S(stream), a.partial, C, p->M, p->N,
                               p->ldc, (long long)p->tap_stride_b, (long long)p->batch_stride_c, a.split, p->tap_mask, p->accumulate, p->K, a.rect, a.g, p->amax_c);
        return glf::check_launch("gemm_tn(reduce)");
    }
    hipLaunchKernelGGL(tn_reduce_kernel, dim3((unsigned)bx, ntap, p->batch), dim3(256), 0, glf::S(stream), a.partial, C, p->M, p->N, p->ldc,
                       (long long)p->tap_stride_b, (long long)p->batch_stride_c, a.split, p->tap_mask, p->accumulate, p->K, a.rect, a.g, vec,
                       p->amax_c);
    return glf::check_launch("gemm_tn(reduce)");
}

extern "C" size_t glf_gemm_tn_workspace_bytes(const glf_gemm_params* p) {
    if (!p || p->split <= 1) return 0;
    const size_t ntap = (size_t)__builtin_popcount(p->tap_mask);
    return (size_t)p->batch * (size_t)p->split * ntap * (size_t)p->M * (size_t)p->N * sizeof(float);
}

