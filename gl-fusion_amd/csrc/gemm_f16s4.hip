// Split-fp16 NT contraction, SECOND tile configuration: 256 threads = 4 waves (2 x 2), tile 128 x 128 x 32, two LDS buffers
// (64 KB) and <= 256 registers per wave -- TWO workgroups per CU.
//
// The 8-wave 256 x 128 kernel of gemm_f16s.hip owns its CU (144 KB of LDS): nothing runs on the matrix pipe while it
// stages its first tiles or stores its results.  Per workgroup that is 10 % of the time at K = 2048 but 45 % at K = 256 and
// 78 % at K = 64 (in-kernel stamps, profiles/r02_stamps_nt_f16x3.txt), and the C2 step spends 60 ms in contractions of that
// kind (profiles/r02_bench_c2_f16x3_per_shape.csv: layer-1 shapes at 30-64 TF, K = 256 ... 512 at 140-280 TF against 375 TF for
// the projections).  With two co-resident workgroups one multiplies while the other stages or stores.  The price is a third
// more operand traffic per FLOP (128-row instead of 256-row A tiles), so the launcher picks this configuration by shape
// (glf::use_f16s4) and the large-K contractions stay on the 8-wave kernel.
//
// Same operand arithmetic (split_f16.h), LDS row swizzle, gather / tap-mask / rectangle / region semantics, wide-store
// epilogue, column statistics and maxima as gemm_f16s.hip -- results of the two configurations agree to the last few ulps
// (different summation grouping is the only difference: the K loop is the same, tile by tile).
#include "gemm_common.h"
#include "split_f16.h"
#include <cstdlib>

namespace {

#define GLF_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
#define GLF_ROW3(c0, c1, m0, m1, ah, al, b0h, b0l, b1h, b1l)  \
    c0 = GLF_MFMA_F16(ah, b0h, c0);                         \
    c1 = GLF_MFMA_F16(ah, b1h, c1);                         \
    if (NP == 3) {                                          \
        m0 = GLF_MFMA_F16(al, b0h, m0);                     \
        m1 = GLF_MFMA_F16(al, b1h, m1);                     \
        m0 = GLF_MFMA_F16(ah, b0l, m0);                     \
        m1 = GLF_MFMA_F16(ah, b1l, m1);                     \
    }

constexpr int BM4 = 128;                 // rows of a tile (BN = 128 columns, BK = 32 deep: gemm_common.h)
constexpr int NT4 = 256;                 // threads
constexpr int WAVE_ROWS = 2;             // wave rows of the workgroup (2 x 2 waves of 64 x 64)
constexpr int RS4 = NT4 / 8;             // rows one staging pass covers (8 threads x float4 per 32-deep row)
constexpr int PL4 = 128 * 64;            // one fp16 plane of an operand tile: 128 rows x 64 B
constexpr int BUF4 = 4 * PL4;            // A high, A low, B high, B low
constexpr size_t SMEM_ROWS_H4 = 2 * BUF4 + 16;

template <bool GATHER, int NP, bool PA, bool BP>
__global__ __launch_bounds__(NT4, 2) void gemm_rows_f16s4_kernel(const GemmArgs args) {
    const int pM = args.M, pN = args.N, pK = args.K, p_lda = args.lda, p_ldb = args.ldb, p_ldc = args.ldc;
    const int p_taps = args.taps, p_gather = args.gather, p_accumulate = args.accumulate;
    const int p_tiles_n = args.tiles_n;
    const unsigned p_tap_mask = args.tap_mask;
    const long long p_tsb = args.tap_stride_b, p_bsa = args.bsa, p_bsb = args.bsb, p_bsc = args.bsc;
    const float* __restrict__ p_A = args.A; const float* __restrict__ p_B = args.B; const float* __restrict__ p_bias = args.bias;
    float* __restrict__ p_C = args.C;
    const float* __restrict__ p_zero = args.zeros;
    const int g_hs = args.g.hs, g_ws = args.g.ws, g_hd = args.g.hd, g_wd = args.g.wd, g_kw = args.g.kw;
    const int g_stride = args.g.stride, g_pad = args.g.pad, g_dil = args.g.dil;
    const int g_nimg = args.g.n_img, p_rect = GATHER ? args.rect : 0;
    float sc_a, sc_b, inv_a, inv_b;
    pow2_scale(args.amax_a, sc_a, inv_a);
    pow2_scale(args.amax_b, sc_b, inv_b);
    const float p_alpha = args.alpha * inv_a * inv_b;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem_s[];
    unsigned* s_mask = reinterpret_cast<unsigned*>(smem_s + 2 * BUF4);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    // tile order inside an XCD's contiguous range: groups of `gm` row tiles x all column tiles, walked down the rows first, so
    // that the ~32 workgroups an XCD runs at a time form a gm x (32 / gm) block of tiles: each A row tile is shared by 32 / gm
    // of them and each B column tile by gm of them through that XCD's L2 (gm = 0: the plain row-major order, every B tile
    // fetched from the Infinity Cache once per row tile)
    int tn, tm;
    {
        const int gm = (args.flags >> 8) & 0xff;
        if (gm > 1) {
            const int tiles_m_all = gridDim.x / p_tiles_n;
            const int per_group = gm * p_tiles_n;
            const int grp = bid / per_group, in_grp = bid - grp * per_group;
            const int first_m = grp * gm;
            const int gsz = min(tiles_m_all - first_m, gm);
            tn = in_grp / gsz;
            tm = first_m + (in_grp - tn * gsz);
        } else {
            tn = bid % p_tiles_n;
            tm = bid / p_tiles_n;
        }
    }
    int pMe = pM;
    int r_y0 = 0, r_x0 = 0, r_h = g_hd, r_w = g_wd;
    unsigned mask = p_tap_mask;
    if (p_rect == 2) {                              // region mode: tiles laid out region after region (see region_of)
        bool found = false;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            int y0, y1, x0, x1;
            unsigned rm;
            region_of(p_gather, r, g_dil, g_hd, g_wd, y0, y1, x0, x1, rm);
            const int mt = g_nimg * (y1 - y0) * (x1 - x0);
            const int tiles = (mt + BM4 - 1) / BM4;
            if (!found) {
                if (tm < tiles) { found = true; mask = rm & p_tap_mask; pMe = mt; r_y0 = y0; r_x0 = x0; r_h = y1 - y0; r_w = x1 - x0; }
                else tm -= tiles;
            }
        }
        if (!found) return;
    } else if (p_rect) {
        for (unsigned mm = p_tap_mask; mm; mm &= mm - 1) {
            const int t = __ffs(mm) - 1;
            int y0, y1, x0, x1;
            tap_rect(p_gather, t, g_kw, g_pad, g_dil, g_hs, g_ws, g_hd, g_wd, y0, y1, x0, x1);
            const int mt = g_nimg * (y1 - y0) * (x1 - x0);
            const int tiles = (mt + BM4 - 1) / BM4;
            if (tm < tiles || (mm & (mm - 1)) == 0) { mask = 1u << t; pMe = mt; r_y0 = y0; r_x0 = x0; r_h = y1 - y0; r_w = x1 - x0; break; }
            tm -= tiles;
        }
    }
    const int bz = blockIdx.z;
    const float* __restrict__ A = p_A + (long long)bz * p_bsa;
    const float* __restrict__ B = p_B + (long long)bz * p_bsb;
    float* __restrict__ C = p_C + (long long)bz * p_bsc;

    const int ac = tid & 7, ar = tid >> 3;              // 8 float4 per 32-deep row; rows ar + RS4 j

    int a_n[4], a_y[4], a_x[4];
    long long a_off[4];
    {
        // pixel coordinates of the thread's first row by division, of the other three (RS4 rows further each) by carrying
        int cn = 0, cy = 0, cx = 0;
        if (GATHER) {
            const int m0 = tm * BM4 + ar, hw = r_h * r_w;
            cn = m0 / hw;
            const int rem = m0 - cn * hw;
            cy = rem / r_w; cx = rem - cy * r_w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = tm * BM4 + ar + RS4 * j;
            if (GATHER) {
                if (m < pMe) { a_n[j] = cn; a_y[j] = r_y0 + cy; a_x[j] = r_x0 + cx; }
                else { a_n[j] = -1; a_y[j] = 0; a_x[j] = 0; }
                a_off[j] = -1;
                if (j < 3 && m + RS4 < pMe) {
                    cx += RS4;
                    while (cx >= r_w) { cx -= r_w; ++cy; }
                    while (cy >= r_h) { cy -= r_h; ++cn; }
                }
            } else {
                a_n[j] = 0; a_y[j] = 0; a_x[j] = 0;
                a_off[j] = (m < pM) ? (long long)m * p_lda : -1;
            }
        }
    }
    // Fast gather form (forward / wgrad-style gathers, and dgrad gathers of stride-1 convs -- everything but the dgrad of a
    // strided conv): a row keeps (rb, y0, x0) = its source pixel index and coordinates for tap offset (0, 0); a tap then is
    // one uniform offset pair (oy, ox): source = rb + oy * ws + ox, in range iff 0 <= y0 + oy < hs and 0 <= x0 + ox < ws.  The
    // general map_src() (an integer division per call, two more for a strided dgrad, divergent branches around each) cost
    // ~3.7 k cycles per tap change and ~20 k in the tap census below: 29 % of a 256 -> 256 3x3 conv's workgroup time
    // (in-kernel stamps: prologue 26 k, 2 333 cycles per iteration against 1 869 for a plain GEMM).
    const bool fastg = GATHER && (p_gather == 1 || g_stride == 1);
    if (fastg) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (a_n[j] >= 0) {
                const int y0 = (p_gather == 1) ? a_y[j] * g_stride - g_pad : a_y[j] + g_pad;
                const int x0 = (p_gather == 1) ? a_x[j] * g_stride - g_pad : a_x[j] + g_pad;
                a_n[j] = (a_n[j] * g_hs + y0) * g_ws + x0; a_y[j] = y0; a_x[j] = x0;
            } else { a_n[j] = 0; a_y[j] = -(1 << 30); a_x[j] = 0; }          // never in range
        }
    }
    // tap -> its uniform offsets (oy, ox) in source pixels
    auto tap_offsets = [&](int t, int& oy, int& ox) __attribute__((always_inline)) {
        int ky = 0, kx = __builtin_amdgcn_readfirstlane(t);
        while (kx >= g_kw) { kx -= g_kw; ++ky; }
        oy = (p_gather == 1 ? ky : -ky) * g_dil;
        ox = (p_gather == 1 ? kx : -kx) * g_dil;
    };
    // tap census: drop the taps that fall into the padding for EVERY row of this tile.  It can only find one when the tile
    // covers fewer than dil + 1 full image rows (otherwise each tap has a row it reaches): skipped for the small dilations.
    if (GATHER && p_taps > 1 && !p_rect && BM4 < g_wd * (g_dil + 1)) {
        if (tid == 0) *s_mask = 0u;
        __syncthreads();
        if (ac == 0) {
            unsigned local = 0;
            for (unsigned mm = mask; mm; mm &= mm - 1) {
                const int t = __ffs(mm) - 1;
                if (fastg) {
                    int oy, ox;
                    tap_offsets(t, oy, ox);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if ((unsigned)(a_y[j] + oy) < (unsigned)g_hs && (unsigned)(a_x[j] + ox) < (unsigned)g_ws) local |= 1u << t;
                } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (a_n[j] >= 0 && map_src(g_hs, g_ws, g_kw, g_stride, g_pad, g_dil, p_gather, a_n[j], a_y[j], a_x[j], t) >= 0) local |= 1u << t;
                }
            }
            if (local) atomicOr(s_mask, local);
        }
        __syncthreads();
        mask &= *s_mask;
    }

    const int nkc = pK / BK;
    const int ntiles = __popc(mask) * nkc;
    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};      // main products
    f32x16 m00 = {0}, m01 = {0}, m10 = {0}, m11 = {0};      // mixed products (x 2^11)
    float4 ra[4], rb[4];
    unsigned rem_mask = mask;
    int tap = -1, kc = nkc;
    const float* pa[4];
    const float* pb[4];

    auto advance = [&]() __attribute__((always_inline)) {
        if (++kc >= nkc) {
            kc = 0;
            tap = __ffs(rem_mask) - 1;
            rem_mask &= rem_mask - 1;
            if (fastg) {
                int oy, ox;
                tap_offsets(tap, oy, ox);
                const int d = oy * g_ws + ox;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool ok = (unsigned)(a_y[j] + oy) < (unsigned)g_hs && (unsigned)(a_x[j] + ox) < (unsigned)g_ws;
                    pa[j] = (ok ? A + (long long)(a_n[j] + d) * p_lda : p_zero) + 4 * ac;          // padding rows read the zero page
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    long long off;
                    if (GATHER) {
                        const int sr = (a_n[j] >= 0) ? map_src(g_hs, g_ws, g_kw, g_stride, g_pad, g_dil, p_gather, a_n[j], a_y[j], a_x[j], tap) : -1;
                        off = (sr >= 0) ? (long long)sr * p_lda : -1;
                    } else {
                        off = a_off[j];
                    }
                    pa[j] = (off >= 0 ? A + off : p_zero) + 4 * ac;          // padding / overhang rows read the zero page
                }
            }
            const float* Bt = B + (long long)tap * p_tsb;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = tn * BN + ar + RS4 * j;
                pb[j] = (n < pN ? Bt + (long long)n * p_ldb : p_zero) + 4 * ac;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) pa[j] += BK;
#pragma unroll
            for (int j = 0; j < 4; ++j) pb[j] += BK;
        }
    };
    // swizzled staging offset of this thread inside a 64-byte row: 16-byte chunk (ac>>1) ^ ((row>>2)&3), half ac&1
    const int st_chunk = ac >> 1;
    const int st_off = ar * 64 + (((st_chunk ^ ((ar >> 2) & 3)) << 4) | ((ac & 1) << 3));
#define GLF_H4_LOAD() \
    { _Pragma("unroll") for (int j = 0; j < 4; ++j) ra[j] = *reinterpret_cast<const float4*>(pa[j]); \
      _Pragma("unroll") for (int j = 0; j < 4; ++j) rb[j] = *reinterpret_cast<const float4*>(pb[j]); }
#define GLF_H4_STORE_A(J, buf_)                                                                              \
    {                                                                                                        \
        unsigned char* d = smem_s + (buf_) * BUF4 + st_off + J * RS4 * 64;                                   \
        if (PA) {                                                                                            \
            *reinterpret_cast<float2*>(d) = make_float2(ra[J].x, ra[J].y);                                   \
            if (NP == 3) *reinterpret_cast<float2*>(d + PL4) = make_float2(ra[J].z, ra[J].w);                \
        } else {                                                                                             \
            const SplitH s = split4h(ra[J], sc_a);                                                           \
            *reinterpret_cast<f16x4*>(d) = s.h; if (NP == 3) *reinterpret_cast<f16x4*>(d + PL4) = s.l;       \
        }                                                                                                    \
    }
#define GLF_H4_STORE_B(J, buf_)                                                                              \
    {                                                                                                        \
        unsigned char* d = smem_s + (buf_) * BUF4 + 2 * PL4 + st_off + J * RS4 * 64;                         \
        if (BP) {                                                                                            \
            *reinterpret_cast<float2*>(d) = make_float2(rb[J].x, rb[J].y);                                   \
            if (NP == 3) *reinterpret_cast<float2*>(d + PL4) = make_float2(rb[J].z, rb[J].w);                \
        } else {                                                                                             \
            const SplitH s = split4h(rb[J], sc_b);                                                           \
            *reinterpret_cast<f16x4*>(d) = s.h; if (NP == 3) *reinterpret_cast<f16x4*>(d + PL4) = s.l;       \
        }                                                                                                    \
    }
#define GLF_H4_STORE(buf_) \
    { GLF_H4_STORE_A(0, buf_) GLF_H4_STORE_A(1, buf_) GLF_H4_STORE_A(2, buf_) GLF_H4_STORE_A(3, buf_) \
      GLF_H4_STORE_B(0, buf_) GLF_H4_STORE_B(1, buf_) GLF_H4_STORE_B(2, buf_) GLF_H4_STORE_B(3, buf_) }
#define GLF_H4_FRAGS(P, buf_, fo_)                                                                            \
        {                                                                                                     \
            const unsigned char* ab_ = smem_s + (buf_) * BUF4 + wm * 64 + (fo_);                              \
            const unsigned char* bb_ = smem_s + (buf_) * BUF4 + 2 * PL4 + wn * 64 + (fo_);                    \
            P##b0h = *reinterpret_cast<const f16x8*>(bb_);                                                    \
            P##b1h = *reinterpret_cast<const f16x8*>(bb_ + 32 * 64);                                          \
            P##a0h = *reinterpret_cast<const f16x8*>(ab_);                                                    \
            P##a1h = *reinterpret_cast<const f16x8*>(ab_ + 32 * 64);                                          \
            if (NP == 3) {                                                                                    \
                P##a0l = *reinterpret_cast<const f16x8*>(ab_ + PL4);                                          \
                P##b0l = *reinterpret_cast<const f16x8*>(bb_ + PL4);                                          \
                P##b1l = *reinterpret_cast<const f16x8*>(bb_ + 32 * 64 + PL4);                                \
                P##a1l = *reinterpret_cast<const f16x8*>(ab_ + 32 * 64 + PL4);                                \
            } else { P##a0l = P##a0h; P##b0l = P##b0h; P##b1l = P##b1h; P##a1l = P##a1h; }                    \
        }
    // Main loop: the classic two-buffer pipeline, one barrier per K-tile, no hand-placed issue slots -- the SECOND workgroup of
    // the CU (64 KB of LDS and 256 registers per wave leave room for exactly two) is what fills this workgroup's barrier waits,
    // LDS latencies, prologue and epilogue with MFMA work.  Tile it is multiplied out of buffer it & 1 while tile it + 1 (in
    // registers since the previous iteration) is staged into the other buffer and tile it + 2 is requested from global memory.
    if (ntiles > 0) {
        const int sw = (lane >> 2) & 3, hh = lane >> 5;
        const int fo0 = (lane & 31) * 64 + (((0 + hh) ^ sw) << 4);
        const int fo1 = (lane & 31) * 64 + (((2 + hh) ^ sw) << 4);
        advance();
        GLF_H4_LOAD()
        GLF_H4_STORE(0)
        if (ntiles > 1) { advance(); GLF_H4_LOAD() }
        __syncthreads();
        for (int it = 0; it < ntiles; ++it) {
            const int cur = it & 1;
            f16x8 fb0h, fb1h, fb0l, fb1l, fa0h, fa0l, fa1h, fa1l;
            f16x8 gb0h, gb1h, gb0l, gb1l, ga0h, ga0l, ga1h, ga1l;
            GLF_H4_FRAGS(f, cur, fo0)
            GLF_H4_FRAGS(g, cur, fo1)
            GLF_ROW3(c00, c01, m00, m01, fa0h, fa0l, fb0h, fb0l, fb1h, fb1l)
            GLF_ROW3(c10, c11, m10, m11, fa1h, fa1l, fb0h, fb0l, fb1h, fb1l)
            if (it + 1 < ntiles) {
                GLF_H4_STORE(cur ^ 1)
                if (it + 2 < ntiles) { advance(); GLF_H4_LOAD() }
            }
            GLF_ROW3(c00, c01, m00, m01, ga0h, ga0l, gb0h, gb0l, gb1h, gb1l)
            GLF_ROW3(c10, c11, m10, m11, ga1h, ga1l, gb0h, gb0l, gb1h, gb1l)
            __syncthreads();
        }
    }

    float cmax = 0.f;
    const bool p_colstats = args.colstats != nullptr;
    // one result element -> C (plain / accumulate / region store, or the atomic of per-tap rectangles); cs / cq: the calling
    // lane's column sum and sum of squares over the elements it stores (colstats)
    auto put = [&](float a, int row, int col, float bv, double& cs, double& cq) __attribute__((always_inline)) {
        if (row >= pMe) return;
        long long orow = row;
        if (p_rect) {
            const int hw = r_h * r_w;
            const int n = row / hw, rem = row - n * hw;
            const int yy = rem / r_w;
            orow = ((long long)n * g_hd + r_y0 + yy) * g_wd + r_x0 + (rem - yy * r_w);
            if (p_rect == 1) { atomicAdd(C + orow * p_ldc + col, p_alpha * a); return; }
        }
        float* dst = C + orow * p_ldc + col;
        float v = p_alpha * a + bv;
        if (p_accumulate) v += *dst;
        *dst = v;
        cmax = fmaxf(cmax, fabsf(v));
        if (p_colstats) { const double vd = (double)v; cs += vd; cq = fma(vd, vd, cq); }
    };
    // column statistics in double from the first product on: E[x^2] - E[x]^2 cancels badly when a channel's values are close
    // together (the ASPP pooled branch: N nearly equal frame averages), fp32 partial sums cost 4e-4 on its BatchNorm output
    // Fast epilogue (everything but the atomics of per-tap rectangles and unaligned outputs): every wave parks its 64 x 64
    // results in LDS (free once the main loop's last barrier is passed: 8 x 16 KB) and stores them as whole 16-byte pieces
    // of rows -- 16 global_store_dwordx4 per lane instead of 64 one-dword stores.  The one-dword form took ~19 k cycles per
    // workgroup (in-kernel stamps, profiles/r02_stamps_*.txt): 13 % of a K = 2048 tile's time, 40 % of a K = 512 tile's,
    // with the matrix pipe idle (one workgroup per CU: nothing else runs meanwhile).
    const bool wide_store = p_rect != 1 && (p_ldc % 4) == 0 && (pN % 4) == 0 && (reinterpret_cast<size_t>(C) % 16) == 0 &&
                            (p_bsc % 4) == 0;
    if (wide_store) {
        float* tile = reinterpret_cast<float*>(smem_s) + wave * (64 * 64);
        {
            const int col_l = lane & 31, row_l = 4 * (lane >> 5);
            auto park = [&](const f32x16& acc, int ti, int tj) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) tile[(32 * ti + (r & 3) + 8 * (r >> 2) + row_l) * 64 + 32 * tj + col_l] = acc[r];
            };
            park(c00 + m00 * 0x1p-11f, 0, 0); park(c01 + m01 * 0x1p-11f, 0, 1);
            park(c10 + m10 * 0x1p-11f, 1, 0); park(c11 + m11 * 0x1p-11f, 1, 1);
        }
        __syncthreads();
        const int c4 = 4 * (lane & 15), r0 = lane >> 4;
        const int col = tn * BN + wn + c4;
        double cs[4] = {0.0, 0.0, 0.0, 0.0}, cq[4] = {0.0, 0.0, 0.0, 0.0};
        float cx[4] = {0.f, 0.f, 0.f, 0.f};          // column maxima of |C| (args.colmax)
        if (col < pN) {
            float bv[4] = {0.f, 0.f, 0.f, 0.f};
            if (p_bias) {
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = p_bias[col + j];
            }
            // region / rectangle stores: pixel coordinates of the lane's first row by division, of the next ones (4 rows on) by carrying
            int en = 0, ey = 0, ex = 0;
            if (p_rect) {
                const int row0 = tm * BM4 + wm + r0, hw = r_h * r_w;
                en = row0 / hw;
                const int rem = row0 - en * hw;
                ey = rem / r_w; ex = rem - ey * r_w;
            }
            if (p_accumulate && !p_rect && !p_colstats) {
                // C += result (a dgrad landing on the shortcut's gradient): all 16 old values are requested before the first is
                // needed -- fetched inside the store loop, each of its 4-deep batches waited out a full memory round trip
                float4 prev[16];
                const int rowb = tm * BM4 + wm + r0;
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    prev[i] = rowb + 4 * i < pMe ? *reinterpret_cast<const float4*>(C + (long long)(rowb + 4 * i) * p_ldc + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    if (rowb + 4 * i < pMe) {
                        const float4 a = *reinterpret_cast<const float4*>(tile + (r0 + 4 * i) * 64 + c4);
                        float4 v = make_float4(p_alpha * a.x + bv[0], p_alpha * a.y + bv[1], p_alpha * a.z + bv[2], p_alpha * a.w + bv[3]);
                        v.x += prev[i].x; v.y += prev[i].y; v.z += prev[i].z; v.w += prev[i].w;
                        *reinterpret_cast<float4*>(C + (long long)(rowb + 4 * i) * p_ldc + col) = v;
                        cmax = fmaxf(cmax, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
                    }
                }
            } else
#pragma unroll 4
            for (int i = 0; i < 16; ++i) {
                const int rin = r0 + 4 * i;
                const int row = tm * BM4 + wm + rin;
                if (row >= pMe) break;
                long long orow = row;
                if (p_rect) {
                    orow = ((long long)en * g_hd + r_y0 + ey) * g_wd + r_x0 + ex;
                    ex += 4;
                    while (ex >= r_w) { ex -= r_w; ++ey; }
                    while (ey >= r_h) { ey -= r_h; ++en; }
                }
                const float4 a = *reinterpret_cast<const float4*>(tile + rin * 64 + c4);
                float* dst = C + orow * p_ldc + col;
                float v[4] = {p_alpha * a.x + bv[0], p_alpha * a.y + bv[1], p_alpha * a.z + bv[2], p_alpha * a.w + bv[3]};
                if (p_accumulate) {
                    const float4 o = *reinterpret_cast<const float4*>(dst);
                    v[0] += o.x; v[1] += o.y; v[2] += o.z; v[3] += o.w;
                }
                *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                cmax = fmaxf(cmax, fmaxf(fmaxf(fabsf(v[0]), fabsf(v[1])), fmaxf(fabsf(v[2]), fabsf(v[3]))));
                if (p_colstats) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) { const double vd = (double)v[j]; cs[j] += vd; cq[j] = fma(vd, vd, cq[j]); cx[j] = fmaxf(cx[j], fabsf(v[j])); }
                }
            }
        }
        if (p_colstats) {
            // lanes l, l + 16, l + 32, l + 48 hold the four row groups of the same four columns; the wave rows of the workgroup
            // are then folded through LDS (each wave's own parking area is free once its stores are issued), so that ONE f64
            // atomic per column, statistic and WORKGROUP reaches memory.  Per wave it was M / 64 atomics on each of the N
            // addresses: 3 025 per address on a 193 600-row layer-1 conv output -- ~150 us of serialised atomics behind a 60 us
            // contraction (profiles/r03_colstats_atomics.txt).
            double* st = args.colstats;
            double* fold = reinterpret_cast<double*>(tile);              // [64 columns][3]: sum, sum of squares, maximum
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                cs[j] += __shfl_xor(cs[j], 16, 64); cq[j] += __shfl_xor(cq[j], 16, 64);
                cs[j] += __shfl_xor(cs[j], 32, 64); cq[j] += __shfl_xor(cq[j], 32, 64);
                cx[j] = fmaxf(cx[j], __shfl_xor(cx[j], 16, 64)); cx[j] = fmaxf(cx[j], __shfl_xor(cx[j], 32, 64));
                if (lane < 16) { fold[3 * (c4 + j)] = cs[j]; fold[3 * (c4 + j) + 1] = cq[j]; fold[3 * (c4 + j) + 2] = (double)cx[j]; }
            }
            __syncthreads();
            if (wm == 0) {                                               // waves 0 / 1: the two column halves of the tile
                const int cl = lane;                                     // column within the wave's 64
                double s = 0.0, q = 0.0;
                float mx = 0.f;
#pragma unroll
                for (int w = 0; w < WAVE_ROWS; ++w) {
                    const double* f = reinterpret_cast<const double*>(reinterpret_cast<const float*>(smem_s) + (2 * w + (wave & 1)) * (64 * 64));
                    s += f[3 * cl]; q += f[3 * cl + 1]; mx = fmaxf(mx, (float)f[3 * cl + 2]);
                }
                const int cg = tn * BN + wn + cl;
                if (cg < pN) {
                    atomicAdd(st + cg, s); atomicAdd(st + pN + cg, q);
                    if (args.colmax && mx > 0.f) atomicMax(reinterpret_cast<unsigned*>(args.colmax + cg), __float_as_uint(mx));
                }
            }
        }
    } else {
        const int col_l = lane & 31, row_l = 4 * (lane >> 5);
        auto emit = [&](const f32x16& acc, int ti, int tj, double& cs, double& cq) {
            const int col = tn * BN + wn + 32 * tj + col_l;
            if (col >= pN) return;
            const float bv = p_bias ? p_bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) put(acc[r], tm * BM4 + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + row_l, col, bv, cs, cq);
        };
        double cs0 = 0.0, cq0 = 0.0, cs1 = 0.0, cq1 = 0.0;
        emit(c00 + m00 * 0x1p-11f, 0, 0, cs0, cq0); emit(c01 + m01 * 0x1p-11f, 0, 1, cs1, cq1);
        emit(c10 + m10 * 0x1p-11f, 1, 0, cs0, cq0); emit(c11 + m11 * 0x1p-11f, 1, 1, cs1, cq1);
        if (args.colstats && p_rect != 1) {
            // lanes l and l + 32 hold the two row groups of the same column: fold them, then one f64 atomic per column
            // and statistic from this wave's 64 rows
            double* st = args.colstats;
            cs0 += __shfl_xor(cs0, 32, 64); cq0 += __shfl_xor(cq0, 32, 64);
            cs1 += __shfl_xor(cs1, 32, 64); cq1 += __shfl_xor(cq1, 32, 64);
            if (lane < 32) {
                const int col0 = tn * BN + wn + col_l, col1 = col0 + 32;
                if (col0 < pN) { atomicAdd(st + col0, cs0); atomicAdd(st + pN + col0, cq0); }
                if (col1 < pN) { atomicAdd(st + col1, cs1); atomicAdd(st + pN + col1, cq1); }
            }
        }
    }
    if (args.amax_c && p_rect != 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
        if (lane == 0 && cmax > *reinterpret_cast<volatile float*>(args.amax_c))      // thousands of waves, ONE address: only a wave that raises it
            atomicMax(reinterpret_cast<unsigned*>(args.amax_c), __float_as_uint(cmax));
    }
}
}  // namespace

namespace glf {

int init_gemm_f16s4_attrs() {
    hipError_t e;
#define SET_ATTR(fn)                                                                                     \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_ROWS_H4); \
    if (e != hipSuccess) return fail(GLF_ERR_LAUNCH, "hipFuncSetAttribute(" #fn "): %s", hipGetErrorString(e));
#define SET_P(G, NP_) SET_ATTR((gemm_rows_f16s4_kernel<G, NP_, false, false>)) SET_ATTR((gemm_rows_f16s4_kernel<G, NP_, true, false>)) \
                      SET_ATTR((gemm_rows_f16s4_kernel<G, NP_, false, true>)) SET_ATTR((gemm_rows_f16s4_kernel<G, NP_, true, true>))
    SET_P(false, 3) SET_P(true, 3) SET_P(false, 1) SET_P(true, 1)
#undef SET_P
#undef SET_ATTR
    return GLF_OK;
}

// Which NT contractions take the 4-wave / two-workgroups-per-CU configuration.  GLF_F16S4 = 0: never, 2: always (A/B runs),
// default: by shape -- the reduction per output tile (K x kept taps) is short enough that the 8-wave kernel's un-overlapped
// prologue + epilogue would be a large share of its workgroup time.
bool use_f16s4(const GemmArgs& a) {
    static const int mode = [] { const char* e = getenv("GLF_F16S4"); return e ? atoi(e) : 1; }();
    if (mode == 0) return false;
    if (mode == 2) return true;
    static const int kmax = [] { const char* e = getenv("GLF_F16S4_KMAX"); return e ? atoi(e) : 256; }();
    const long long kred = (long long)a.K * __builtin_popcount(a.tap_mask);
    return kred <= kmax;
}

int launch_rows_f16s4(const GemmArgs& a0, dim3 grid, bool gather, int nprod, hipStream_t s) {
    GemmArgs a = a0;
    a.zeros = zero_page();
    long long tiles_m = (a.M + BM4 - 1) / BM4;
    if (a.rect == 2) {
        tiles_m = 0;
        for (int r = 0; r < 9; ++r) {
            int y0, y1, x0, x1;
            unsigned rm;
            region_of(a.gather, r, a.g.dil, a.g.hd, a.g.wd, y0, y1, x0, x1, rm);
            tiles_m += ((long long)a.g.n_img * (y1 - y0) * (x1 - x0) + BM4 - 1) / BM4;
        }
    } else if (a.rect) {
        tiles_m = 0;
        for (unsigned mm = a.tap_mask; mm; mm &= mm - 1) {
            const int t = __builtin_ctz(mm);
            int y0, y1, x0, x1;
            tap_rect(a.gather, t, a.g.kw, a.g.pad, a.g.dil, a.g.hs, a.g.ws, a.g.hd, a.g.wd, y0, y1, x0, x1);
            tiles_m += ((long long)a.g.n_img * (y1 - y0) * (x1 - x0) + BM4 - 1) / BM4;
        }
    }
    a.tiles_m = (int)tiles_m;
    dim3 g2((unsigned)(tiles_m * a.tiles_n), 1, grid.z);
    a.flags = (a.rect == 0 && a.tiles_n >= 8 ? 4 : 0) << 8;          // grouped tile order, as the 8-wave kernel
    const bool pa = a.a_presplit != 0, pb = a.b_presplit != 0;
#define GLF_L4(G, NP_, PA_, PB_) hipLaunchKernelGGL((gemm_rows_f16s4_kernel<G, NP_, PA_, PB_>), g2, dim3(NT4), SMEM_ROWS_H4, s, a)
#define GLF_L4P(G, NP_)                                                                   \
    { if (pa && pb) GLF_L4(G, NP_, true, true); else if (pa) GLF_L4(G, NP_, true, false); \
      else if (pb) GLF_L4(G, NP_, false, true); else GLF_L4(G, NP_, false, false); }
    if (nprod == 3) { if (gather) GLF_L4P(true, 3) else GLF_L4P(false, 3) }
    else { if (gather) GLF_L4P(true, 1) else GLF_L4P(false, 1) }
#undef GLF_L4P
#undef GLF_L4
    return check_launch("gemm_nt(f16x3, 128x128, 2 workgroups / CU)");
}

}  // namespace glf
