// Split-bf16 ("bf16x6") contraction kernels: fp32-equivalent GEMM on the bf16 matrix cores of gfx950.
//
// Every fp32 operand x is split exactly into three bf16 pieces x = x1 + x2 + x3 (|x - (x1+x2+x3)| <= 2^-24 |x|)
// and the product is evaluated as
//      a*b ~= a1*b1 + (a1*b2 + a2*b1) + (a1*b3 + a2*b2 + a3*b1)          (dropped terms <= 2^-23 |a*b|)
// with six v_mfma_f32_32x32x16_bf16 (fp32 accumulate).  One 32x32x16 bf16 MFMA retires 16 k in 32 cycles, the exact
// v_mfma_f32_32x32x2_f32 2 k in 64: six of them cost 192 cycles per 16 k against 512 -- 2.67x the fp32-MFMA roof at
// fp32 accuracy (measured vs fp64: same error as the exact-fp32 kernels, see tests/test_gpu_ops.py).
//
// Layout: operands are split while they are written to LDS (one fp32 global read, three bf16 planes in LDS).
//   rows kernel (NT: A[m][k], B[n][k], both k-contiguous): planes [row][32 k] with swizzled 64-byte rows; a
//     fragment (row = lane&31, k = 8*(lane>>5) .. +7) is ONE conflict-free ds_read_b128.
//   tn kernel (reduction over rows r: A[r][m], B[r][n]): planes [r][128 m] with 320-byte rows; fragments are
//     transposed on the fly by ds_read_b64_tr_b16 (two per fragment), conflict-free for that stride.
// Same gather / tap_mask / rect semantics, epilogues and XCD-aware tile order as gemm_f32.hip.  Only the aligned
// fast path is built (K % 32 == 0, 16-byte aligned rows); everything else stays on the exact-fp32 kernels.
#include "gemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

constexpr int RST = 320;                 // bytes per LDS row of the tn kernel (128 bf16 + 64 B pad)
constexpr int PLANE_T = 32 * RST;
constexpr int OPER_T = 3 * PLANE_T;
constexpr size_t SMEM_TN_S = 2 * OPER_T + 32 * sizeof(int);

struct Split4 { bf16x4 h, m, l; };

// two floats -> one dword of two bf16 (round to nearest even): a single v_cvt_pk_bf16_f32
__device__ __forceinline__ unsigned cvt_pk_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }           // element 0 as float
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }   // element 1 as float

// x = h + m + l exactly to 2^-24 |x|, pairwise on packed words: 3 cvt_pk + 4 unpack + 4 sub per two elements
// (the element-wise form compiled to 2.3x as many VALU instructions and made the kernel issue-bound)
__device__ __forceinline__ Split4 split4(const float4 v) {
    const unsigned h0 = cvt_pk_bf16(v.x, v.y), h1 = cvt_pk_bf16(v.z, v.w);
    const float r0 = v.x - bf_lo(h0), r1 = v.y - bf_hi(h0), r2 = v.z - bf_lo(h1), r3 = v.w - bf_hi(h1);
    const unsigned m0 = cvt_pk_bf16(r0, r1), m1 = cvt_pk_bf16(r2, r3);
    const float q0 = r0 - bf_lo(m0), q1 = r1 - bf_hi(m0), q2 = r2 - bf_lo(m1), q3 = r3 - bf_hi(m1);
    const unsigned l0 = cvt_pk_bf16(q0, q1), l1 = cvt_pk_bf16(q2, q3);
    Split4 s;
    s.h = __builtin_bit_cast(bf16x4, make_uint2(h0, h1));
    s.m = __builtin_bit_cast(bf16x4, make_uint2(m0, m1));
    s.l = __builtin_bit_cast(bf16x4, make_uint2(l0, l1));
    return s;
}

#define GLF_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// six products of one (A tile-row, B tile-col) pair for one 16-deep k-step; low-order terms first
#define GLF_SIX(t, ah, am, al, bh, bm, bl)   \
    t = GLF_MFMA_BF16(al, bh, t);            \
    t = GLF_MFMA_BF16(am, bm, t);            \
    t = GLF_MFMA_BF16(ah, bl, t);            \
    t = GLF_MFMA_BF16(am, bh, t);            \
    t = GLF_MFMA_BF16(ah, bm, t);            \
    t = GLF_MFMA_BF16(ah, bh, t);

// same, starting a fresh accumulator: C = 0 is an inline constant of the first MFMA (zero-initialising the
// four f32x16 tile accumulators with v_mov cost 64 VALU instructions per K-tile)
#define GLF_SIX0(t, ah, am, al, bh, bm, bl)  \
    {                                        \
        const f32x16 z_ = {0};               \
        t = GLF_MFMA_BF16(al, bh, z_);       \
    }                                        \
    t = GLF_MFMA_BF16(am, bm, t);            \
    t = GLF_MFMA_BF16(ah, bl, t);            \
    t = GLF_MFMA_BF16(am, bh, t);            \
    t = GLF_MFMA_BF16(ah, bm, t);            \
    t = GLF_MFMA_BF16(ah, bh, t);
#define GLF_SIXS(s_, t, ah, am, al, bh, bm, bl) \
    if ((s_) == 0) { GLF_SIX0(t, ah, am, al, bh, bm, bl) } else { GLF_SIX(t, ah, am, al, bh, bm, bl) }

// ----------------------------------------------------------------------------------------------------------
// rows kernel, NT: C[m][n] = alpha * sum_tap sum_k A[src(m,tap)][k] * B_tap[n][k] (+ bias).  512 threads = 8 waves (4 x 2), tile 256 x 128 x 32, two LDS buffers.
//   * LDS planes have unpadded 64-byte rows ([row][32 bf16]) with the 16-byte chunk index XOR-swizzled by
//     (row >> 2) & 3, so both the staging writes (ds_write_b64) and the fragment reads (ds_read_b128) are
//     conflict-free: 2 buffers x (3 x 256 + 3 x 128 rows) x 64 B = 144 KB of the CU's 160 KB.
//   * software pipeline, one barrier per K-tile: while tile t is multiplied out of buffer t&1, the raw fp32
//     registers of tile t+1 (loaded one iteration earlier) are split to bf16 and written to the other buffer in
//     six pieces placed between groups of eight MFMAs, and each piece's registers are immediately re-loaded with
//     tile t+2 -- global latency has a whole K-tile (>= 1536 MFMA cycles) to hide.
// ----------------------------------------------------------------------------------------------------------
constexpr int BM8 = 256;
constexpr int NT8 = 512;
constexpr int PL_A8 = BM8 * 64, PL_B8 = BN * 64;
constexpr int BUF8 = 3 * PL_A8 + 3 * PL_B8;
constexpr size_t SMEM_ROWS_S8 = 2 * BUF8 + 16;

template <bool GATHER>
__global__ __launch_bounds__(NT8, 2) void gemm_rows_bf16s8_kernel(const GemmArgs args) {
    const int pM = args.M, pN = args.N, pK = args.K, p_lda = args.lda, p_ldb = args.ldb, p_ldc = args.ldc;
    const int p_taps = args.taps, p_gather = args.gather, p_accumulate = args.accumulate;
    const int p_tiles_n = args.tiles_n;
    const unsigned p_tap_mask = args.tap_mask;
    const long long p_tsb = args.tap_stride_b, p_bsa = args.bsa, p_bsb = args.bsb, p_bsc = args.bsc;
    const float p_alpha = args.alpha;
    const float* __restrict__ p_A = args.A; const float* __restrict__ p_B = args.B; const float* __restrict__ p_bias = args.bias;
    float* __restrict__ p_C = args.C;
    const int g_hs = args.g.hs, g_ws = args.g.ws, g_hd = args.g.hd, g_wd = args.g.wd, g_kw = args.g.kw;
    const int g_stride = args.g.stride, g_pad = args.g.pad, g_dil = args.g.dil;
    const int g_nimg = args.g.n_img, p_rect = GATHER ? args.rect : 0;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    unsigned* s_mask = reinterpret_cast<unsigned*>(smem_s + 2 * BUF8);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p_tiles_n;
    int tm = bid / p_tiles_n;
    int pMe = pM;
    int r_y0 = 0, r_x0 = 0, r_h = g_hd, r_w = g_wd;
    unsigned mask = p_tap_mask;
    if (p_rect) {
        for (unsigned mm = p_tap_mask; mm; mm &= mm - 1) {
            const int t = __ffs(mm) - 1;
            int y0, y1, x0, x1;
            tap_rect(p_gather, t, g_kw, g_pad, g_dil, g_hs, g_ws, g_hd, g_wd, y0, y1, x0, x1);
            const int mt = g_nimg * (y1 - y0) * (x1 - x0);
            const int tiles = (mt + BM8 - 1) / BM8;
            if (tm < tiles || (mm & (mm - 1)) == 0) { mask = 1u << t; pMe = mt; r_y0 = y0; r_x0 = x0; r_h = y1 - y0; r_w = x1 - x0; break; }
            tm -= tiles;
        }
    }
    const int bz = blockIdx.z;
    const float* __restrict__ A = p_A + (long long)bz * p_bsa;
    const float* __restrict__ B = p_B + (long long)bz * p_bsb;
    float* __restrict__ C = p_C + (long long)bz * p_bsc;

    const int ac = tid & 7, ar = tid >> 3;              // 8 float4 per 32-deep row; rows ar + 64 j

    int a_n[4], a_y[4], a_x[4];
    long long a_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int m = tm * BM8 + ar + 64 * j;
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

    const int nkc = pK / BK;
    const int ntiles = __popc(mask) * nkc;
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    float4 ra[4], rb[2];
    unsigned rem_mask = mask;
    int tap = -1, kc = nkc;
    const float* pa[4];
    const float* pb[2];
    unsigned a_ok = 0, a_ok_c = 0;                       // validity of the rows being LOADED / being CONVERTED

    auto advance = [&]() __attribute__((always_inline)) {
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
            for (int j = 0; j < 2; ++j) pb[j] = Bt + (long long)min(tn * BN + ar + 64 * j, pN - 1) * p_ldb + 4 * ac;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) pa[j] += BK;
#pragma unroll
            for (int j = 0; j < 2; ++j) pb[j] += BK;
        }
    };
    // swizzled staging offset of this thread inside a 64-byte row: 16-byte chunk (ac>>1) ^ ((row>>2)&3), half ac&1;
    // rows are ar + 64 j, so (row>>2)&3 == (ar>>2)&3 for every j
    const int st_off = ar * 64 + ((((ac >> 1) ^ ((ar >> 2) & 3)) << 4) | ((ac & 1) << 3));
#define GLF_S8_CONV_A(J, buf_)                                                                               \
    {                                                                                                        \
        const Split4 s = split4(keep_if((a_ok_c >> J) & 1u, ra[J]));                                         \
        unsigned char* d = smem_s + (buf_) * BUF8 + st_off + J * 64 * 64;                                    \
        *reinterpret_cast<bf16x4*>(d) = s.h; *reinterpret_cast<bf16x4*>(d + PL_A8) = s.m; *reinterpret_cast<bf16x4*>(d + 2 * PL_A8) = s.l; \
    }
#define GLF_S8_CONV_B(J, buf_)                                                                               \
    {                                                                                                        \
        const Split4 s = split4(keep_if(tn * BN + ar + 64 * J < pN, rb[J]));                                 \
        unsigned char* d = smem_s + (buf_) * BUF8 + 3 * PL_A8 + st_off + J * 64 * 64;                        \
        *reinterpret_cast<bf16x4*>(d) = s.h; *reinterpret_cast<bf16x4*>(d + PL_B8) = s.m; *reinterpret_cast<bf16x4*>(d + 2 * PL_B8) = s.l; \
    }
    // piece pc (0..5): convert + store registers of tile t+1, then refill them with tile t+2
#define GLF_S8_PIECE(pc, buf_, conv_, load_)                                                                 \
    switch (pc) {                                                                                            \
        case 0: if (conv_) GLF_S8_CONV_A(0, buf_) if (load_) ra[0] = *reinterpret_cast<const float4*>(pa[0]); break; \
        case 1: if (conv_) GLF_S8_CONV_A(1, buf_) if (load_) ra[1] = *reinterpret_cast<const float4*>(pa[1]); break; \
        case 2: if (conv_) GLF_S8_CONV_A(2, buf_) if (load_) ra[2] = *reinterpret_cast<const float4*>(pa[2]); break; \
        case 3: if (conv_) GLF_S8_CONV_A(3, buf_) if (load_) ra[3] = *reinterpret_cast<const float4*>(pa[3]); break; \
        case 4: if (conv_) GLF_S8_CONV_B(0, buf_) if (load_) rb[0] = *reinterpret_cast<const float4*>(pb[0]); break; \
        default: if (conv_) GLF_S8_CONV_B(1, buf_) if (load_) rb[1] = *reinterpret_cast<const float4*>(pb[1]); break; \
    }

    if (ntiles > 0) {
        // fragment offsets (swizzled) of this lane for the two 16-deep k-steps of a tile
        const int sw = (lane >> 2) & 3, hh = lane >> 5;
        const int fo0 = (lane & 31) * 64 + (((0 + hh) ^ sw) << 4);
        const int fo1 = (lane & 31) * 64 + (((2 + hh) ^ sw) << 4);
        // prologue: tile 0 -> buffer 0, tile 1 raw in registers
        advance();
        a_ok_c = a_ok;
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) { GLF_S8_PIECE(pc, 0, false, true) }
        {
            const bool more = ntiles > 1;
            if (more) advance();
#pragma unroll
            for (int pc = 0; pc < 6; ++pc) { GLF_S8_PIECE(pc, 0, true, more) }
        }
        __syncthreads();
        // One K-tile of work.  CONV_/LOAD_ are compile-time constants: the steady-state loop body is straight-line
        // code (per-piece branches made hipcc fall back to s_waitcnt vmcnt(0) in front of every piece, i.e. six
        // serialised global round trips per tile); the last two tiles run peeled copies.
#define GLF_S8_BODY(CONV_, LOAD_)                                                                             \
        {                                                                                                     \
            const int buf = it & 1;                                                                           \
            a_ok_c = a_ok;                      /* validity of tile it+1 (pointers still describe it) */      \
            if (LOAD_) advance();                                                                             \
            const unsigned char* ab = smem_s + buf * BUF8 + wm * 64;                                          \
            const unsigned char* bb = smem_s + buf * BUF8 + 3 * PL_A8 + wn * 64;                              \
            f32x16 t00, t01, t10, t11;                                                                        \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                   \
                const int fo = s ? fo1 : fo0;                                                                 \
                const bf16x8 b0h = *reinterpret_cast<const bf16x8*>(bb + fo);                                 \
                const bf16x8 b0m = *reinterpret_cast<const bf16x8*>(bb + fo + PL_B8);                         \
                const bf16x8 b0l = *reinterpret_cast<const bf16x8*>(bb + fo + 2 * PL_B8);                     \
                const bf16x8 b1h = *reinterpret_cast<const bf16x8*>(bb + 32 * 64 + fo);                       \
                const bf16x8 b1m = *reinterpret_cast<const bf16x8*>(bb + 32 * 64 + fo + PL_B8);               \
                const bf16x8 b1l = *reinterpret_cast<const bf16x8*>(bb + 32 * 64 + fo + 2 * PL_B8);           \
                {                                                                                             \
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ab + fo);                              \
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(ab + fo + PL_A8);                      \
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(ab + fo + 2 * PL_A8);                  \
                    GLF_SIXS(s, t00, ah, am, al, b0h, b0m, b0l)                                               \
                    GLF_S8_PIECE(3 * s + 0, buf ^ 1, CONV_, LOAD_)                                            \
                    GLF_SIXS(s, t01, ah, am, al, b1h, b1m, b1l)                                               \
                    GLF_S8_PIECE(3 * s + 1, buf ^ 1, CONV_, LOAD_)                                            \
                }                                                                                             \
                {                                                                                             \
                    const bf16x8 ah = *reinterpret_cast<const bf16x8*>(ab + 32 * 64 + fo);                    \
                    const bf16x8 am = *reinterpret_cast<const bf16x8*>(ab + 32 * 64 + fo + PL_A8);            \
                    const bf16x8 al = *reinterpret_cast<const bf16x8*>(ab + 32 * 64 + fo + 2 * PL_A8);        \
                    GLF_SIXS(s, t10, ah, am, al, b0h, b0m, b0l)                                               \
                    GLF_S8_PIECE(3 * s + 2, buf ^ 1, CONV_, LOAD_)                                            \
                    GLF_SIXS(s, t11, ah, am, al, b1h, b1m, b1l)                                               \
                }                                                                                             \
            }                                                                                                 \
            c00 += t00; c01 += t01; c10 += t10; c11 += t11;   /* two-level accumulation (see gemm_f32.hip) */ \
            __syncthreads();                                                                                  \
        }
        int it = 0;
        for (; it + 2 < ntiles; ++it) GLF_S8_BODY(true, true)
        if (it + 1 < ntiles) { GLF_S8_BODY(true, false) ++it; }
        GLF_S8_BODY(false, false)
    }

    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    auto emit = [&](const f32x16& acc, int ti, int tj) {
        const int col = tn * BN + wn + 32 * tj + col_l;
        if (col >= pN) return;
        const float bv = p_bias ? p_bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * BM8 + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + row_l;
            if (row < pMe) {
                if (p_rect) {
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

// ----------------------------------------------------------------------------------------------------------
// tn kernel: C_tap[m][n] (+)= alpha * sum_{r in slice} A[arow(r)][m] * B[src(r,tap)][n]
// ----------------------------------------------------------------------------------------------------------
template <bool GATHER>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_bf16s_kernel(const GemmArgs args) {
    const int pM = args.M, pN = args.N, pK = args.K, p_lda = args.lda, p_ldb = args.ldb, p_ldc = args.ldc;
    const int p_accumulate = args.accumulate, p_split = args.split;
    const int p_tiles_n = args.tiles_n;
    const unsigned p_tap_mask = args.tap_mask;
    const long long p_tsb = args.tap_stride_b, p_bsa = args.bsa, p_bsb = args.bsb, p_bsc = args.bsc;
    const float p_alpha = args.alpha;
    const float* __restrict__ p_A = args.A; const float* __restrict__ p_B = args.B;
    float* __restrict__ p_C = args.C;
    const int g_hs = args.g.hs, g_ws = args.g.ws, g_hd = args.g.hd, g_wd = args.g.wd, g_kw = args.g.kw;
    const int g_stride = args.g.stride, g_pad = args.g.pad, g_dil = args.g.dil;
    const int g_nimg = args.g.n_img, p_rect = GATHER ? args.rect : 0;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    unsigned char* As = smem_s;
    unsigned char* Bs = smem_s + OPER_T;
    int* vflag = reinterpret_cast<int*>(smem_s + 2 * OPER_T);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p_tiles_n, tm = bid / p_tiles_n;
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
    if (r0 >= r1) return;

    const int c4 = tid & 31, rr = tid >> 5;
    const int m0 = tm * BM + 4 * c4, n0 = tn * BN + 4 * c4;
    const int m0c = min(m0, pM - 4), n0c = min(n0, pN - 4);
    const int hw = GATHER ? r_h * r_w : 1;

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    float4 ra[4], rb[4];
    int rvalid[4];

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
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ra[j] = *reinterpret_cast<const float4*>(A + arow[j] * p_lda + m0c);
            rb[j] = *reinterpret_cast<const float4*>(B + (src[j] >= 0 ? src[j] : 0) * p_ldb + n0c);
        }
    };
#define GLF_ST_STORE(J)                                                                                    \
    {                                                                                                      \
        const Split4 sa = split4(keep_if(rvalid[J] && m0 < pM, ra[J]));                                    \
        const Split4 sb = split4(keep_if(rvalid[J] && n0 < pN, rb[J]));                                    \
        unsigned char* da = As + (rr + 8 * J) * RST + c4 * 8;                                              \
        unsigned char* db = Bs + (rr + 8 * J) * RST + c4 * 8;                                              \
        *reinterpret_cast<bf16x4*>(da) = sa.h; *reinterpret_cast<bf16x4*>(da + PLANE_T) = sa.m; *reinterpret_cast<bf16x4*>(da + 2 * PLANE_T) = sa.l; \
        *reinterpret_cast<bf16x4*>(db) = sb.h; *reinterpret_cast<bf16x4*>(db + PLANE_T) = sb.m; *reinterpret_cast<bf16x4*>(db + 2 * PLANE_T) = sb.l; \
        if (c4 == 0) vflag[rr + 8 * J] = rvalid[J];                                                        \
    }

    // transposing fragment read: lanes 16g..16g+15 fetch a block of 4 k-rows x 16 columns; lane 4q+p of the group
    // addresses row q, columns 4p..4p+3 and receives column (lane&15), rows 0..3.  Group g: columns 16*(g&1).., k half g>>1.
    const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int tr_off = (8 * (grp >> 1) + q) * RST + (16 * (grp & 1) + 4 * pp) * 2;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
#define GLF_TR_FRAG(base, dst)                                                                              \
    {                                                                                                       \
        const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));                       \
        const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)((base) + 4 * RST));          \
        typedef short s16x8_ __attribute__((ext_vector_type(8)));                                           \
        const s16x8_ both_ = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                     \
        dst = __builtin_bit_cast(bf16x8, both_);                                                            \
    }

    load_tile(r0);
    for (int rbase = r0; rbase < r1; rbase += BK) {
        GLF_ST_STORE(0) GLF_ST_STORE(1) GLF_ST_STORE(2) GLF_ST_STORE(3)
        __syncthreads();
        if (rbase + BK < r1) load_tile(rbase + BK);
        const bool any = __ballot(vflag[lane & 31] != 0) != 0ull;
        if (any) {
            f32x16 t00, t01, t10, t11;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned char* ab = As + s * 16 * RST + wm * 2 + tr_off;
                const unsigned char* bb = Bs + s * 16 * RST + wn * 2 + tr_off;
                bf16x8 b0h, b0m, b0l, b1h, b1m, b1l;
                GLF_TR_FRAG(bb, b0h) GLF_TR_FRAG(bb + PLANE_T, b0m) GLF_TR_FRAG(bb + 2 * PLANE_T, b0l)
                GLF_TR_FRAG(bb + 64, b1h) GLF_TR_FRAG(bb + 64 + PLANE_T, b1m) GLF_TR_FRAG(bb + 64 + 2 * PLANE_T, b1l)
                {
                    bf16x8 ah, am, al;
                    GLF_TR_FRAG(ab, ah) GLF_TR_FRAG(ab + PLANE_T, am) GLF_TR_FRAG(ab + 2 * PLANE_T, al)
                    GLF_SIXS(s, t00, ah, am, al, b0h, b0m, b0l)
                    GLF_SIXS(s, t01, ah, am, al, b1h, b1m, b1l)
                }
                {
                    bf16x8 ah, am, al;
                    GLF_TR_FRAG(ab + 64, ah) GLF_TR_FRAG(ab + 64 + PLANE_T, am) GLF_TR_FRAG(ab + 64 + 2 * PLANE_T, al)
                    GLF_SIXS(s, t10, ah, am, al, b0h, b0m, b0l)
                    GLF_SIXS(s, t11, ah, am, al, b1h, b1m, b1l)
                }
            }
            c00 += t00; c01 += t01; c10 += t10; c11 += t11;
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

}  // namespace

namespace glf {

int init_gemm_bf16s_attrs() {
    hipError_t e;
#define SET_ATTR(fn, bytes)                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
    if (e != hipSuccess) return fail(GLF_ERR_LAUNCH, "hipFuncSetAttribute(" #fn "): %s", hipGetErrorString(e));
    SET_ATTR((gemm_rows_bf16s8_kernel<false>), SMEM_ROWS_S8)
    SET_ATTR((gemm_rows_bf16s8_kernel<true>), SMEM_ROWS_S8)
    SET_ATTR((gemm_tn_bf16s_kernel<false>), SMEM_TN_S)
    SET_ATTR((gemm_tn_bf16s_kernel<true>), SMEM_TN_S)
#undef SET_ATTR
    return GLF_OK;
}

// Eligibility: the aligned fast path only.
bool bf16s_rows_ok(const GemmArgs& a) {
    return a.vec_a && a.vec_b && (a.K % BK) == 0 && a.K >= BK;
}
bool bf16s_tn_ok(const GemmArgs& a) {
    return a.vec_a && a.vec_b && (a.M % 4) == 0 && (a.N % 4) == 0 && a.M >= 4 && a.N >= 4;
}

// 256-row tiles: the grid is re-derived for BM8; rect mode sums its tiles per tap as setup_rect does
int launch_rows_bf16s8(const GemmArgs& a0, bool gather, int batch, hipStream_t s) {
    GemmArgs a = a0;
    long long tiles_m = (a.M + BM8 - 1) / BM8;
    if (a.rect) {
        tiles_m = 0;
        for (unsigned mm = a.tap_mask; mm; mm &= mm - 1) {
            const int t = __builtin_ctz(mm);
            int y0, y1, x0, x1;
            tap_rect(a.gather, t, a.g.kw, a.g.pad, a.g.dil, a.g.hs, a.g.ws, a.g.hd, a.g.wd, y0, y1, x0, x1);
            tiles_m += ((long long)a.g.n_img * (y1 - y0) * (x1 - x0) + BM8 - 1) / BM8;
        }
    }
    a.tiles_m = (int)tiles_m;
    dim3 grid((unsigned)(tiles_m * a.tiles_n), 1, batch);
    if (gather) hipLaunchKernelGGL((gemm_rows_bf16s8_kernel<true>), grid, dim3(NT8), SMEM_ROWS_S8, s, a);
    else hipLaunchKernelGGL((gemm_rows_bf16s8_kernel<false>), grid, dim3(NT8), SMEM_ROWS_S8, s, a);
    return check_launch("gemm_nt(bf16x6, 256x128)");
}

int launch_rows_bf16s(const GemmArgs& a, dim3 grid, bool gather, hipStream_t s) {
    return launch_rows_bf16s8(a, gather, (int)grid.z, s);
}
int launch_tn_bf16s(const GemmArgs& a, dim3 grid, bool gather, hipStream_t s) {
    if (gather) hipLaunchKernelGGL((gemm_tn_bf16s_kernel<true>), grid, dim3(NTHREADS), SMEM_TN_S, s, a);
    else hipLaunchKernelGGL((gemm_tn_bf16s_kernel<false>), grid, dim3(NTHREADS), SMEM_TN_S, s, a);
    return check_launch("gemm_tn(bf16x6)");
}

}  // namespace glf
