// Split-fp16 ("f16x3") contraction kernels: fp32-class GEMM on the fp16 matrix cores of gfx950 with THREE MFMAs per
// product instead of the six of gemm_bf16s.hip.
//
// Each operand is first multiplied by a power of two s = 2^(13 - floor(log2 amax)) derived from the operand's max
// magnitude (a device scalar: glf_gemm_params.amax_a / amax_b, produced by glf_amax or by the kernel that wrote the
// operand), so that |x s| < 2^14 sits at the top of the fp16 range.  x s is then split into two fp16 pieces
//      x s = h + 2^-11 l + e,      h = rne16(x s),   l = rne16((x s - h) 2^11),   |e| <= 2^-22 |x s|
// (full 22-bit precision for every element within 2^-27 of amax, an absolute error floor of 2^-49 amax below), and
//      a*b ~= ha*hb + 2^-11 (ha*lb + la*hb)                                    (dropped la*lb <= 2^-22 |a*b|)
// is evaluated with three v_mfma_f32_32x32x16_f16 per 16 k: the main product accumulates in one fp32 accumulator,
// the two mixed products in a second one that is folded in (times 2^-11) by the epilogue, where the operand
// powers of two are undone exactly as well.  That is 96 MFMA cycles per 16 k against 192 (bf16x6) and 512 (exact
// fp32 MFMA).  The MFMA sums 16 k per instruction, so the accumulation chain is K/16 long -- no per-tile
// two-level fold is needed to stay at the fp32 kernels' error level (tests/test_gpu_ops.py measure it vs fp64).
// Rows that must read as zero (conv padding, tile overhang) are LOADED from a zero page (args.zeros) instead of
// being masked after the load: no per-element selects in the staging path.
//
// Same tiling, LDS swizzle, software pipeline, gather / tap_mask / rect semantics and XCD-aware tile order as
// gemm_bf16s.hip, with two planes per operand instead of three (96 KB of LDS).  Only the aligned fast path is
// built; everything else stays on the exact-fp32 kernels.
#include "gemm_common.h"
#include "split_f16.h"
#include <cstdlib>

namespace {

#define GLF_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0)
// one A tile-row against both B tile-cols for one 16-deep k-step: main products into c, mixed products into m
// (dependent MFMAs are two apart)
// NP (a compile-time constant of the enclosing kernel) = MFMAs per product: 3 = split-fp16 (fp32-grade results),
// 1 = the high halves only (precision 3, "f16": plain fp16 operands with an amax scale, fp32 accumulate)
#define GLF_ROW3(c0, c1, m0, m1, ah, al, b0h, b0l, b1h, b1l)  \
    c0 = GLF_MFMA_F16(ah, b0h, c0);                         \
    c1 = GLF_MFMA_F16(ah, b1h, c1);                         \
    if (NP == 3) {                                          \
        m0 = GLF_MFMA_F16(al, b0h, m0);                     \
        m1 = GLF_MFMA_F16(al, b1h, m1);                     \
        m0 = GLF_MFMA_F16(ah, b0l, m0);                     \
        m1 = GLF_MFMA_F16(ah, b1l, m1);                     \
    }

// ----------------------------------------------------------------------------------------------------------
// rows kernel, NT.  512 threads = 8 waves (4 x 2), tile 256 x 128 x 32, THREE LDS buffers of 2 x (256 + 128) rows
// (144 KB).
// ----------------------------------------------------------------------------------------------------------
constexpr int BM8 = 256;
constexpr int NT8 = 512;
constexpr int WAVE_ROWS = 4;            // wave rows of the 8-wave NT workgroup (4 x 2 waves of 64 x 64)
#ifndef GLF_IL_ALL          // 1: the slot-interleaved iteration also when neither operand is pre-split
#define GLF_IL_ALL 1
#endif
constexpr int PL_A8 = BM8 * 64, PL_B8 = BN * 64;
constexpr int BUF8 = 2 * PL_A8 + 2 * PL_B8;
#ifdef GLF_STAMPS       // diagnostic build (profiles/ubench/stamps.sh): per-wave s_memtime stamps of 16 main-loop iterations of one workgroup
constexpr size_t SMEM_ROWS_H8 = 3 * BUF8 + 16 + 8192;
#else
constexpr size_t SMEM_ROWS_H8 = 3 * BUF8 + 16;
#endif

// M16: the products run on v_mfma_f32_16x16x32_f16 (16 x 16 output tiles, the whole 32-deep K-tile per instruction)
// instead of v_mfma_f32_32x32x16_f16: the same MFMA cycles per FLOP, the same LDS fragment traffic, but under load the
// chip holds a higher clock on the smaller shape (MI355X_MICROARCH.md, DVFS item 7).  LDS rows keep their 64-byte
// layout; only the 16-byte chunk swizzle differs (chunk g(c) ^ (row >> 2 & 3), g = 0,3,1,2: conflict-free ds_read_b128 for
// the lane -> (row = lane & 15, chunk = lane >> 4) fragment map).
// PA / BP: the A / B operand arrives PRE-SPLIT in the packed image glf_split_f16_packed writes: every aligned group of four
// consecutive elements (16 bytes of fp32) is replaced IN PLACE by {h0 h1 h2 h3 l0 l1 l2 l3} (fp16), x * s = h + 2^-11 l with the
// scale of args.amax_a / amax_b.  Same byte size, same strides, same addressing as the fp32 operand -- the staging path keeps its
// loads, pointers, zero page and gather logic and only drops the conversion (a bit-cast instead of ~18 VALU per float4): the
// operand is split ONCE (by glf_split_f16_packed, per tensor) instead of in every tile of every launch that reads it.
// (The variants that were measured and dropped -- deeper register prefetch, ping-pong segments, sched_group_barrier interleave,
// static wave priorities, other tile-group sizes -- are described with their numbers in DESIGN.md section 8; their code is gone.)
template <bool GATHER, int NP, bool M16, bool BP, bool PA = false>
__global__ __launch_bounds__(NT8, 2) void gemm_rows_f16s8_kernel(const GemmArgs args) {
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
    unsigned* s_mask = reinterpret_cast<unsigned*>(smem_s + 3 * BUF8);
#if defined(GLF_STAMPS) && GLF_STAMPS == 2     // workgroup-level stamps only: entry, main loop start / end, exit (no per-iteration cost)
    unsigned long long wg_t0 = __builtin_amdgcn_s_memtime(), wg_t1 = 0, wg_t2 = 0;
#endif

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
            const int tiles = (mt + BM8 - 1) / BM8;
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
    {
        // pixel coordinates of the thread's first row by division, of the other three (64 rows further each) by carrying
        int cn = 0, cy = 0, cx = 0;
        if (GATHER) {
            const int m0 = tm * BM8 + ar, hw = r_h * r_w;
            cn = m0 / hw;
            const int rem = m0 - cn * hw;
            cy = rem / r_w; cx = rem - cy * r_w;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = tm * BM8 + ar + 64 * j;
            if (GATHER) {
                if (m < pMe) { a_n[j] = cn; a_y[j] = r_y0 + cy; a_x[j] = r_x0 + cx; }
                else { a_n[j] = -1; a_y[j] = 0; a_x[j] = 0; }
                a_off[j] = -1;
                if (j < 3 && m + 64 < pMe) {
                    cx += 64;
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
    if (GATHER && p_taps > 1 && !p_rect && BM8 < g_wd * (g_dil + 1)) {
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
    float4 ra[4], rb[2];
    unsigned rem_mask = mask;
    int tap = -1, kc = nkc;
    const float* pa[4];
    const float* pb[2];

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
            for (int j = 0; j < 2; ++j) {
                const int n = tn * BN + ar + 64 * j;
                pb[j] = (n < pN ? Bt + (long long)n * p_ldb : p_zero) + 4 * ac;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) pa[j] += BK;
#pragma unroll
            for (int j = 0; j < 2; ++j) pb[j] += BK;
        }
    };
    // swizzled staging offset of this thread inside a 64-byte row: 16-byte chunk (ac>>1) ^ ((row>>2)&3), half ac&1
    const int st_chunk = M16 ? (int)((0x2130u >> (4 * (ac >> 1))) & 3u) : (ac >> 1);      // g = 0,3,1,2 for the 16x16x32 fragment map
    const int st_off = ar * 64 + (((st_chunk ^ ((ar >> 2) & 3)) << 4) | ((ac & 1) << 3));
#define GLF_H8_CONV_A(J, buf_)                                                                               \
    {                                                                                                        \
        unsigned char* d = smem_s + (buf_) * BUF8 + st_off + J * 64 * 64;                                    \
        if (PA) {                                                                                            \
            *reinterpret_cast<float2*>(d) = make_float2(ra[J].x, ra[J].y);                                   \
            if (NP == 3) *reinterpret_cast<float2*>(d + PL_A8) = make_float2(ra[J].z, ra[J].w);              \
        } else {                                                                                             \
            const SplitH s = split4h(ra[J], sc_a);                                                           \
            *reinterpret_cast<f16x4*>(d) = s.h; if (NP == 3) *reinterpret_cast<f16x4*>(d + PL_A8) = s.l;     \
        }                                                                                                    \
    }
#define GLF_H8_CONV_B(J, buf_)                                                                               \
    {                                                                                                        \
        unsigned char* d = smem_s + (buf_) * BUF8 + 2 * PL_A8 + st_off + J * 64 * 64;                        \
        if (BP) {                                                                                            \
            *reinterpret_cast<float2*>(d) = make_float2(rb[J].x, rb[J].y);                                   \
            if (NP == 3) *reinterpret_cast<float2*>(d + PL_B8) = make_float2(rb[J].z, rb[J].w);              \
        } else {                                                                                             \
            const SplitH s = split4h(rb[J], sc_b);                                                           \
            *reinterpret_cast<f16x4*>(d) = s.h; if (NP == 3) *reinterpret_cast<f16x4*>(d + PL_B8) = s.l;     \
        }                                                                                                    \
    }
#define GLF_H8_LOAD_B(J) rb[J] = *reinterpret_cast<const float4*>(pb[J]);
    // piece pc (0..5): convert + store registers of tile t+1, then refill them with tile t+2
// A pre-split operand goes from its load registers straight into LDS: the store must stay AHEAD of the load that refills the
// same registers (GLF_H8_PIN).  Left to itself the scheduler hoists the load above the store, the two values then need two
// register sets, and the copy it adds at the loop's back edge waits for loads issued a few hundred cycles earlier --
// s_waitcnt vmcnt(1) in every iteration, the whole global-load latency exposed (measured: the pre-split kernel 4 % faster
// than the splitting one instead of the ~1.4x its instruction count promises).
#define GLF_H8_PIN(P_) if (P_) __builtin_amdgcn_sched_barrier(0);
#define GLF_H8_PIECE(pc, buf_, conv_, load_)                                                                 \
    switch (pc) {                                                                                            \
        case 0: if (conv_) GLF_H8_CONV_A(0, buf_) GLF_H8_PIN(PA) if (load_) ra[0] = *reinterpret_cast<const float4*>(pa[0]); GLF_H8_PIN(PA) break; \
        case 1: if (conv_) GLF_H8_CONV_A(1, buf_) GLF_H8_PIN(PA) if (load_) ra[1] = *reinterpret_cast<const float4*>(pa[1]); GLF_H8_PIN(PA) break; \
        case 2: if (conv_) GLF_H8_CONV_A(2, buf_) GLF_H8_PIN(PA) if (load_) ra[2] = *reinterpret_cast<const float4*>(pa[2]); GLF_H8_PIN(PA) break; \
        case 3: if (conv_) GLF_H8_CONV_A(3, buf_) GLF_H8_PIN(PA) if (load_) ra[3] = *reinterpret_cast<const float4*>(pa[3]); GLF_H8_PIN(PA) break; \
        case 4: if (conv_) GLF_H8_CONV_B(0, buf_) GLF_H8_PIN(BP) if (load_) GLF_H8_LOAD_B(0) GLF_H8_PIN(BP) break;      \
        default: if (conv_) GLF_H8_CONV_B(1, buf_) GLF_H8_PIN(BP) if (load_) GLF_H8_LOAD_B(1) GLF_H8_PIN(BP) break;     \
    }

    if (!M16 && ntiles > 0) {
        const int sw = (lane >> 2) & 3, hh = lane >> 5;
        const int fo0 = (lane & 31) * 64 + (((0 + hh) ^ sw) << 4);
        const int fo1 = (lane & 31) * 64 + (((2 + hh) ^ sw) << 4);
        // Pipeline state at the top of iteration `it`: tiles it and it+1 sit converted in LDS buffers it%3 and
        // (it+1)%3, tile it+2 raw in registers, the k-step-0 fragments of tile it in f*.  The iteration multiplies
        // tile it, converts tile it+2 into buffer (it+2)%3, loads tile it+3 and -- after its first k-step --
        // prefetches the k-step-0 fragments of tile it+1, so no LDS latency is exposed behind the barrier.
        // Prologue: the loads of the first THREE tiles are issued back to back (the accumulators are not live yet, registers are
        // free), then tiles 0 and 1 are converted: one global-memory latency instead of three in a row (5.7 k of a K = 2048
        // workgroup's 132 k cycles went here, a quarter of a K = 256 one's).
        {
            float4 ta[4], tb[2], ua[4], ub[2];
            const bool more = ntiles > 1, more2 = ntiles > 2;
            advance();
#pragma unroll
            for (int pc = 0; pc < 6; ++pc) { GLF_H8_PIECE(pc, 0, false, true) }
            if (more) {
                advance();
#pragma unroll
                for (int j = 0; j < 4; ++j) ta[j] = *reinterpret_cast<const float4*>(pa[j]);
#pragma unroll
                for (int j = 0; j < 2; ++j) tb[j] = *reinterpret_cast<const float4*>(pb[j]);
            }
            if (more2) {
                advance();
#pragma unroll
                for (int j = 0; j < 4; ++j) ua[j] = *reinterpret_cast<const float4*>(pa[j]);
#pragma unroll
                for (int j = 0; j < 2; ++j) ub[j] = *reinterpret_cast<const float4*>(pb[j]);
            }
#pragma unroll
            for (int pc = 0; pc < 6; ++pc) { GLF_H8_PIECE(pc, 0, true, false) }
            if (more) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[j] = ta[j];
#pragma unroll
                for (int j = 0; j < 2; ++j) rb[j] = tb[j];
#pragma unroll
                for (int pc = 0; pc < 6; ++pc) { GLF_H8_PIECE(pc, 1, true, false) }
            }
            if (more2) {
#pragma unroll
                for (int j = 0; j < 4; ++j) ra[j] = ua[j];
#pragma unroll
                for (int j = 0; j < 2; ++j) rb[j] = ub[j];
            }
        }
        __syncthreads();
        f16x8 fb0h, fb1h, fb0l, fb1l, fa0h, fa0l, fa1h, fa1l;
#define GLF_H8_FRAGS(P, buf_, fo_)                                                                            \
        {                                                                                                     \
            const unsigned char* ab_ = smem_s + (buf_) * BUF8 + wm * 64 + (fo_);                              \
            const unsigned char* bb_ = smem_s + (buf_) * BUF8 + 2 * PL_A8 + wn * 64 + (fo_);                  \
            P##b0h = *reinterpret_cast<const f16x8*>(bb_);                                                    \
            P##b1h = *reinterpret_cast<const f16x8*>(bb_ + 32 * 64);                                          \
            P##a0h = *reinterpret_cast<const f16x8*>(ab_);                                                    \
            P##a1h = *reinterpret_cast<const f16x8*>(ab_ + 32 * 64);                                          \
            if (NP == 3) {                                                                                    \
                P##a0l = *reinterpret_cast<const f16x8*>(ab_ + PL_A8);                                        \
                P##b0l = *reinterpret_cast<const f16x8*>(bb_ + PL_B8);                                        \
                P##b1l = *reinterpret_cast<const f16x8*>(bb_ + 32 * 64 + PL_B8);                              \
                P##a1l = *reinterpret_cast<const f16x8*>(ab_ + 32 * 64 + PL_A8);                              \
            } else { P##a0l = P##a0h; P##b0l = P##b0h; P##b1l = P##b1h; P##a1l = P##a1h; }                    \
        }
        GLF_H8_FRAGS(f, 0, fo0)
        int cur = 0, nxt = 1, wr = 2;           // LDS buffers of tile it, it+1, it+2
        // CONV_/LOAD_/NEXT_ are compile-time constants: the steady-state body is straight-line code
#if defined(GLF_STAMPS) && GLF_STAMPS == 1
        unsigned long long st_[5] = {0, 0, 0, 0, 0};
        const bool stamp_on = args.partial != nullptr && blockIdx.x == gridDim.x / 2;
        unsigned long long* stamp_lds = reinterpret_cast<unsigned long long*>(smem_s + 3 * BUF8 + 16);
#define GLF_STAMP(k_) { __builtin_amdgcn_sched_barrier(0); st_[k_] = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define GLF_STAMP_FLUSH()                                                                                     \
            if (stamp_on && it >= 16 && it < 32 && lane == 0) {                                               \
                unsigned long long* d_ = stamp_lds + (wave * 16 + (it - 16)) * 8;                             \
                d_[0] = st_[0]; d_[1] = st_[1]; d_[2] = st_[2]; d_[3] = st_[3]; d_[4] = st_[4];               \
            }
#else
#define GLF_STAMP(k_)
#define GLF_STAMP_FLUSH()
#endif
#define GLF_H8_BODY(CONV_, LOAD_, NEXT_)                                                                      \
        {                                                                                                     \
            GLF_STAMP(0)                                                                                      \
            if (LOAD_) advance();                                                                             \
            f16x8 gb0h, gb1h, gb0l, gb1l, ga0h, ga0l, ga1h, ga1l;                                             \
            GLF_H8_FRAGS(g, cur, fo1)                                                                         \
            __builtin_amdgcn_sched_barrier(0);    /* keep the fragment reads up here (hipcc sinks them to their use) */ \
            GLF_H8_PIECE(0, wr, CONV_, LOAD_)                                                                 \
            GLF_ROW3(c00, c01, m00, m01, fa0h, fa0l, fb0h, fb0l, fb1h, fb1l)                                  \
            GLF_STAMP(1)                                                                                      \
            GLF_H8_PIECE(1, wr, CONV_, LOAD_)                                                                 \
            GLF_ROW3(c10, c11, m10, m11, fa1h, fa1l, fb0h, fb0l, fb1h, fb1l)                                  \
            GLF_H8_PIECE(2, wr, CONV_, LOAD_)                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            GLF_STAMP(2)                                                                                      \
            if (NEXT_) GLF_H8_FRAGS(f, nxt, fo0)                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            GLF_H8_PIECE(3, wr, CONV_, LOAD_)                                                                 \
            GLF_ROW3(c00, c01, m00, m01, ga0h, ga0l, gb0h, gb0l, gb1h, gb1l)                                  \
            GLF_STAMP(3)                                                                                      \
            GLF_H8_PIECE(4, wr, CONV_, LOAD_)                                                                 \
            GLF_ROW3(c10, c11, m10, m11, ga1h, ga1l, gb0h, gb0l, gb1h, gb1l)                                  \
            GLF_H8_PIECE(5, wr, CONV_, LOAD_)                                                                 \
            { const int t_ = cur; cur = nxt; nxt = wr; wr = t_; }                                             \
            GLF_STAMP(4)                                                                                      \
            __syncthreads();                                                                                  \
            GLF_STAMP_FLUSH()                                                                                 \
        }
// The iteration, instruction by instruction (GLF_IL_*: a slot = what is issued behind one MFMA, pinned by sched_barriers).
// A wave issues in order, so whatever sits between two of ITS MFMAs delays the second one unless it fits the ~24 issue cycles
// the first leaves free; in bursts (the compiler's choice: all fragment reads, then six MFMAs, then a whole conversion
// piece ...) the matrix pipe idles behind every burst -- in-kernel stamps showed 2 575 cycles per iteration for 1 536 cycles
// of MFMA work, and the eight waves' LDS / global bursts colliding.  Per wave and iteration: 24 MFMAs, 16 fragment reads,
// 6 staging pieces.  Slots 1-4: the k-step-1 fragments (two reads each); 5: pointer advance; 6-19: the staging pieces -- a
// pre-split operand's piece is one slot (LDS store of the registers loaded an iteration ago + their refill), a piece that
// still has to be split is cut into three (scale + high halves | residual | low halves + store + refill: 4-6 VALU each);
// 12-18 also carry the next tile's k-step-0 fragments, each into registers whose last reader has issued; the last MFMAs run
// with nothing behind them, so every LDS operation has landed when the barrier is reached.
#define GLF_IL_RD(P, w_, base_, off_) P##w_ = *reinterpret_cast<const f16x8*>((base_) + (off_));
#define GLF_IL_PIN() __builtin_amdgcn_sched_barrier(0);
#define GLF_IL_MM(c_, a_, b_) c_ = GLF_MFMA_F16(a_, b_, c_);
#define GLF_IL_ROW(c0, c1, m0, m1, ah, al, b0h, b0l, b1h, b1l, S1_, S2_, S3_, S4_, S5_, S6_)                   \
            GLF_IL_MM(c0, ah, b0h) GLF_IL_PIN() S1_ GLF_IL_PIN()                                              \
            GLF_IL_MM(c1, ah, b1h) GLF_IL_PIN() S2_ GLF_IL_PIN()                                              \
            if (NP == 3) { GLF_IL_MM(m0, al, b0h) GLF_IL_PIN() } S3_ GLF_IL_PIN()                             \
            if (NP == 3) { GLF_IL_MM(m1, al, b1h) GLF_IL_PIN() } S4_ GLF_IL_PIN()                             \
            if (NP == 3) { GLF_IL_MM(m0, ah, b0l) GLF_IL_PIN() } S5_ GLF_IL_PIN()                             \
            if (NP == 3) { GLF_IL_MM(m1, ah, b1l) GLF_IL_PIN() } S6_ GLF_IL_PIN()
        // the three stages of a piece that is split here (SRC_ = its float4, SC_ = the operand's scale)
        f32x2 sx01_, sx23_, sr01_, sr23_;
        f16x2 sh01_, sh23_;
#define GLF_IL_ST1(SRC_, SC_)                                                                                 \
            { const f32x2 s2_ = {SC_, SC_}; const f32x2 v01_ = {SRC_.x, SRC_.y}, v23_ = {SRC_.z, SRC_.w};     \
              sx01_ = v01_ * s2_; sx23_ = v23_ * s2_;                                                         \
              sh01_ = __builtin_convertvector(sx01_, f16x2); sh23_ = __builtin_convertvector(sx23_, f16x2); }
#define GLF_IL_ST2()                                                                                          \
            { const f32x2 k2_ = {2048.f, 2048.f};                                                             \
              sr01_ = (sx01_ - __builtin_convertvector(sh01_, f32x2)) * k2_;                                  \
              sr23_ = (sx23_ - __builtin_convertvector(sh23_, f32x2)) * k2_; }
#define GLF_IL_ST3(DST_, PL_)                                                                                 \
            { *reinterpret_cast<f16x4*>(DST_) = __builtin_shufflevector(sh01_, sh23_, 0, 1, 2, 3);            \
              if (NP == 3) *reinterpret_cast<f16x4*>((DST_) + (PL_)) =                                        \
                  __builtin_shufflevector(__builtin_convertvector(sr01_, f16x2), __builtin_convertvector(sr23_, f16x2), 0, 1, 2, 3); }
        // stage ST_ (1..3) of A piece J / B piece J of the tile being staged into buffer `wr`
#define GLF_IL_A(J, ST_, CONV_, LOAD_)                                                                        \
            if (PA) { if (ST_ == 3) { GLF_H8_PIECE(J, wr, CONV_, LOAD_) } }                                   \
            else if (CONV_) {                                                                                 \
                if (ST_ == 1) GLF_IL_ST1(ra[J], sc_a)                                                         \
                else if (ST_ == 2) GLF_IL_ST2()                                                               \
                else { GLF_IL_ST3(smem_s + wr * BUF8 + st_off + J * 64 * 64, PL_A8)                           \
                       if (LOAD_) ra[J] = *reinterpret_cast<const float4*>(pa[J]); }                          \
            }
#define GLF_IL_B(J, ST_, CONV_, LOAD_)                                                                        \
            if (BP) { if (ST_ == 3) { GLF_H8_PIECE(4 + J, wr, CONV_, LOAD_) } }                               \
            else if (CONV_) {                                                                                 \
                if (ST_ == 1) GLF_IL_ST1(rb[J], sc_b)                                                         \
                else if (ST_ == 2) GLF_IL_ST2()                                                               \
                else { GLF_IL_ST3(smem_s + wr * BUF8 + 2 * PL_A8 + st_off + J * 64 * 64, PL_B8)               \
                       if (LOAD_) GLF_H8_LOAD_B(J) }                                                          \
            }
// staging work item i_ of the iteration: the A pieces' stages, then the B pieces' (one item per pre-split piece, three per
// piece split here) -- items sit in consecutive slots from slot 6 on, so pre-split operands finish their LDS stores early
#define GLF_IL_ITEM(I_, CONV_, LOAD_)                                                                         \
            {                                                                                                 \
                constexpr int nA_ = PA ? 4 : 12, nB_ = BP ? 2 : 6, i_ = (I_);                                 \
                if constexpr (i_ < nA_) {                                                                     \
                    constexpr int j_ = PA ? i_ : i_ / 3, st_ = PA ? 3 : i_ % 3 + 1;                           \
                    GLF_IL_A(j_, st_, CONV_, LOAD_)                                                           \
                } else if constexpr (i_ < nA_ + nB_) {                                                        \
                    constexpr int k_ = i_ - nA_, j_ = BP ? k_ : k_ / 3, st_ = BP ? 3 : k_ % 3 + 1;            \
                    GLF_IL_B(j_, st_, CONV_, LOAD_)                                                           \
                }                                                                                             \
            }
#define GLF_H8_BODY_IL(CONV_, LOAD_, NEXT_)                                                                   \
        {                                                                                                     \
            GLF_STAMP(0)                                                                                      \
            f16x8 gb0h, gb1h, gb0l, gb1l, ga0h, ga0l, ga1h, ga1l;                                             \
            const unsigned char* ga_ = smem_s + cur * BUF8 + wm * 64 + fo1;                                   \
            const unsigned char* gb_ = smem_s + cur * BUF8 + 2 * PL_A8 + wn * 64 + fo1;                       \
            const unsigned char* na_ = smem_s + nxt * BUF8 + wm * 64 + fo0;                                   \
            const unsigned char* nb_ = smem_s + nxt * BUF8 + 2 * PL_A8 + wn * 64 + fo0;                       \
            GLF_IL_ROW(c00, c01, m00, m01, fa0h, fa0l, fb0h, fb0l, fb1h, fb1l,                                \
                { GLF_IL_RD(g, b0h, gb_, 0) GLF_IL_RD(g, a0h, ga_, 0) },                                      \
                { GLF_IL_RD(g, b1h, gb_, 32 * 64) if (NP == 3) { GLF_IL_RD(g, a0l, ga_, PL_A8) } },           \
                { if (NP == 3) { GLF_IL_RD(g, b0l, gb_, PL_B8) GLF_IL_RD(g, b1l, gb_, 32 * 64 + PL_B8) } },   \
                { GLF_IL_RD(g, a1h, ga_, 32 * 64) if (NP == 3) { GLF_IL_RD(g, a1l, ga_, 32 * 64 + PL_A8) } }, \
                { if (LOAD_) advance(); },                                                                    \
                { GLF_IL_ITEM(0, CONV_, LOAD_) })                                                             \
            GLF_STAMP(1)                                                                                      \
            GLF_IL_ROW(c10, c11, m10, m11, fa1h, fa1l, fb0h, fb0l, fb1h, fb1l,                                \
                { GLF_IL_ITEM(1, CONV_, LOAD_) },                                                             \
                { GLF_IL_ITEM(2, CONV_, LOAD_) },                                                             \
                { GLF_IL_ITEM(3, CONV_, LOAD_) },                                                             \
                { GLF_IL_ITEM(4, CONV_, LOAD_) },                                                             \
                { GLF_IL_ITEM(5, CONV_, LOAD_) },                                                             \
                { GLF_IL_ITEM(6, CONV_, LOAD_) if (NEXT_) { GLF_IL_RD(f, a0h, na_, 0) if (NP == 3) { GLF_IL_RD(f, a0l, na_, PL_A8) } } }) \
            if (NP != 3) { ga0l = ga0h; gb0l = gb0h; gb1l = gb1h; ga1l = ga1h; }                              \
            GLF_STAMP(2)                                                                                      \
            GLF_IL_ROW(c00, c01, m00, m01, ga0h, ga0l, gb0h, gb0l, gb1h, gb1l,                                \
                { GLF_IL_ITEM(7, CONV_, LOAD_) if (NEXT_) { GLF_IL_RD(f, b0h, nb_, 0) GLF_IL_RD(f, b1h, nb_, 32 * 64) } }, \
                { GLF_IL_ITEM(8, CONV_, LOAD_) if (NEXT_ && NP == 3) { GLF_IL_RD(f, b0l, nb_, PL_B8) GLF_IL_RD(f, b1l, nb_, 32 * 64 + PL_B8) } }, \
                { GLF_IL_ITEM(9, CONV_, LOAD_) if (NEXT_) { GLF_IL_RD(f, a1h, na_, 32 * 64) if (NP == 3) { GLF_IL_RD(f, a1l, na_, 32 * 64 + PL_A8) } } }, \
                { GLF_IL_ITEM(10, CONV_, LOAD_) },                                                            \
                { GLF_IL_ITEM(11, CONV_, LOAD_) },                                                            \
                { GLF_IL_ITEM(12, CONV_, LOAD_) })                                                            \
            GLF_STAMP(3)                                                                                      \
            GLF_IL_ROW(c10, c11, m10, m11, ga1h, ga1l, gb0h, gb0l, gb1h, gb1l,                                \
                { GLF_IL_ITEM(13, CONV_, LOAD_) },                                                            \
                { GLF_IL_ITEM(14, CONV_, LOAD_) },                                                            \
                { GLF_IL_ITEM(15, CONV_, LOAD_) },                                                            \
                { GLF_IL_ITEM(16, CONV_, LOAD_) },                                                            \
                { GLF_IL_ITEM(17, CONV_, LOAD_) },                                                            \
                {})                                                                                           \
            if (NP != 3 && NEXT_) { fa0l = fa0h; fb0l = fb0h; fb1l = fb1h; fa1l = fa1h; }                     \
            { const int t_ = cur; cur = nxt; nxt = wr; wr = t_; }                                             \
            GLF_STAMP(4)                                                                                      \
            __syncthreads();                                                                                  \
            GLF_STAMP_FLUSH()                                                                                 \
        }
        int it = 0;
#if defined(GLF_STAMPS) && GLF_STAMPS == 2
        wg_t1 = __builtin_amdgcn_s_memtime();
#endif
        if (PA || BP || GLF_IL_ALL) {
            for (; it + 3 < ntiles; ++it) GLF_H8_BODY_IL(true, true, true)
            if (it + 2 < ntiles) { GLF_H8_BODY_IL(true, false, true) ++it; }
            if (it + 1 < ntiles) { GLF_H8_BODY_IL(false, false, true) ++it; }
            GLF_H8_BODY_IL(false, false, false)
        } else {
            for (; it + 3 < ntiles; ++it) GLF_H8_BODY(true, true, true)
            if (it + 2 < ntiles) { GLF_H8_BODY(true, false, true) ++it; }
            if (it + 1 < ntiles) { GLF_H8_BODY(false, false, true) ++it; }
            GLF_H8_BODY(false, false, false)
        }
#if defined(GLF_STAMPS) && GLF_STAMPS == 1
        if (stamp_on) {
            __syncthreads();
            for (int i = tid; i < 8 * 16 * 8; i += NT8) reinterpret_cast<unsigned long long*>(args.partial)[i] = stamp_lds[i];
        }
#endif
#if defined(GLF_STAMPS) && GLF_STAMPS == 2
        wg_t2 = __builtin_amdgcn_s_memtime();
#endif
    }

    // ---- 16x16x32 variant: accumulators t[i][j] (main) / u[i][j] (mixed), i = 16-row slab, j = 16-column slab of the
    //      wave's 64 x 64 tile ----
    f32x4 t00 = {0}, t01 = {0}, t02 = {0}, t03 = {0}, t10 = {0}, t11 = {0}, t12 = {0}, t13 = {0};
    f32x4 t20 = {0}, t21 = {0}, t22 = {0}, t23 = {0}, t30 = {0}, t31 = {0}, t32 = {0}, t33 = {0};
    f32x4 u00 = {0}, u01 = {0}, u02 = {0}, u03 = {0}, u10 = {0}, u11 = {0}, u12 = {0}, u13 = {0};
    f32x4 u20 = {0}, u21 = {0}, u22 = {0}, u23 = {0}, u30 = {0}, u31 = {0}, u32 = {0}, u33 = {0};
    if (M16 && ntiles > 0) {
        const int lr = lane & 15;
        const int fo = lr * 64 + (((int)((0x2130u >> (4 * (lane >> 4))) & 3u) ^ ((lr >> 2) & 3)) << 4);
        advance();
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) { GLF_H8_PIECE(pc, 0, false, true) }
        {
            const bool more = ntiles > 1;
            if (more) advance();
#pragma unroll
            for (int pc = 0; pc < 6; ++pc) { GLF_H8_PIECE(pc, 0, true, more) }
            if (more) {
                const bool more2 = ntiles > 2;
                if (more2) advance();
#pragma unroll
                for (int pc = 0; pc < 6; ++pc) { GLF_H8_PIECE(pc, 1, true, more2) }
            }
        }
        __syncthreads();
#define GLF_MFMA_16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0)
        // one 16-row slab of A against one 16-column slab of B: main product into t, the two mixed ones into u
#define GLF_M16_TILE(t, u, ah, al, bh, bl)                          \
        t = GLF_MFMA_16(ah, bh, t);                                 \
        if (NP == 3) { u = GLF_MFMA_16(al, bh, u); u = GLF_MFMA_16(ah, bl, u); }
#define GLF_M16_A(buf_, s_, dh, dl)                                                                             \
        {                                                                                                       \
            const unsigned char* p_ = smem_s + (buf_) * BUF8 + (wm + 16 * (s_)) * 64 + fo;                      \
            dh = *reinterpret_cast<const f16x8*>(p_);                                                           \
            if (NP == 3) dl = *reinterpret_cast<const f16x8*>(p_ + PL_A8); else dl = dh;                        \
        }
#define GLF_M16_B(buf_, s_, dh, dl)                                                                             \
        {                                                                                                       \
            const unsigned char* p_ = smem_s + (buf_) * BUF8 + 2 * PL_A8 + (wn + 16 * (s_)) * 64 + fo;          \
            dh = *reinterpret_cast<const f16x8*>(p_);                                                           \
            if (NP == 3) dl = *reinterpret_cast<const f16x8*>(p_ + PL_B8); else dl = dh;                        \
        }
        f16x8 a0h, a0l, a1h, a1l, a2h, a2l, a3h, a3l, b0h, b0l, b1h, b1l, b2h, b2l, b3h, b3l;
        GLF_M16_B(0, 0, b0h, b0l) GLF_M16_B(0, 1, b1h, b1l) GLF_M16_B(0, 2, b2h, b2l)
        if (!(PA && BP)) GLF_M16_B(0, 3, b3h, b3l)
        GLF_M16_A(0, 0, a0h, a0l) GLF_M16_A(0, 1, a1h, a1l)
        int cur = 0, nxt = 1, wr = 2;
        // Per iteration: slabs 0, 1 of A (loaded one iteration ahead) against all of B while slabs 2, 3 arrive; then slabs 2, 3
        // column by column, each B slab being replaced by the next tile's as soon as its last product is issued, and slabs
        // 0, 1 of the next tile are fetched at the half-way point: 64 fragment registers in all, nothing exposed behind the barrier.
#define GLF_M16_BODY(CONV_, LOAD_, NEXT_)                                                                     \
        {                                                                                                     \
            if (LOAD_) advance();                                                                             \
            GLF_M16_A(cur, 2, a2h, a2l) GLF_M16_A(cur, 3, a3h, a3l)                                           \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            GLF_H8_PIECE(0, wr, CONV_, LOAD_)                                                                 \
            GLF_M16_TILE(t00, u00, a0h, a0l, b0h, b0l) GLF_M16_TILE(t01, u01, a0h, a0l, b1h, b1l)             \
            GLF_M16_TILE(t02, u02, a0h, a0l, b2h, b2l) GLF_M16_TILE(t03, u03, a0h, a0l, b3h, b3l)             \
            GLF_H8_PIECE(1, wr, CONV_, LOAD_)                                                                 \
            GLF_M16_TILE(t10, u10, a1h, a1l, b0h, b0l) GLF_M16_TILE(t11, u11, a1h, a1l, b1h, b1l)             \
            GLF_M16_TILE(t12, u12, a1h, a1l, b2h, b2l) GLF_M16_TILE(t13, u13, a1h, a1l, b3h, b3l)             \
            GLF_H8_PIECE(2, wr, CONV_, LOAD_)                                                                 \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            if (NEXT_) { GLF_M16_A(nxt, 0, a0h, a0l) GLF_M16_A(nxt, 1, a1h, a1l) }                            \
            __builtin_amdgcn_sched_barrier(0);                                                                \
            GLF_M16_TILE(t20, u20, a2h, a2l, b0h, b0l) GLF_M16_TILE(t30, u30, a3h, a3l, b0h, b0l)             \
            if (NEXT_) GLF_M16_B(nxt, 0, b0h, b0l)                                                            \
            GLF_H8_PIECE(3, wr, CONV_, LOAD_)                                                                 \
            GLF_M16_TILE(t21, u21, a2h, a2l, b1h, b1l) GLF_M16_TILE(t31, u31, a3h, a3l, b1h, b1l)             \
            if (NEXT_) GLF_M16_B(nxt, 1, b1h, b1l)                                                            \
            GLF_H8_PIECE(4, wr, CONV_, LOAD_)                                                                 \
            GLF_M16_TILE(t22, u22, a2h, a2l, b2h, b2l) GLF_M16_TILE(t32, u32, a3h, a3l, b2h, b2l)             \
            if (NEXT_) GLF_M16_B(nxt, 2, b2h, b2l)                                                            \
            GLF_H8_PIECE(5, wr, CONV_, LOAD_)                                                                 \
            GLF_M16_TILE(t23, u23, a2h, a2l, b3h, b3l) GLF_M16_TILE(t33, u33, a3h, a3l, b3h, b3l)             \
            if (NEXT_) GLF_M16_B(nxt, 3, b3h, b3l)                                                            \
            { const int t_ = cur; cur = nxt; nxt = wr; wr = t_; }                                             \
            __syncthreads();                                                                                  \
        }
        // Pre-split operands: the same products with everything else placed tile by tile (a slot = what is issued behind the
        // three MFMAs of one 16 x 16 tile; sched_barriers pin it).  B slab 3 of the NEXT tile is not fetched before the barrier
        // (its registers are busy until the last product) but in the first slot of the next iteration, three tiles before its
        // first use: no LDS operation is younger than two tiles when the barrier is reached.
#define GLF_M16_T(t, u, ah, al, bh, bl, SLOT_) GLF_M16_TILE(t, u, ah, al, bh, bl) GLF_IL_PIN() SLOT_ GLF_IL_PIN()
#define GLF_M16_BODY_IL(CONV_, LOAD_, NEXT_)                                                                  \
        {                                                                                                     \
            GLF_M16_T(t00, u00, a0h, a0l, b0h, b0l, { GLF_M16_B(cur, 3, b3h, b3l) GLF_M16_A(cur, 2, a2h, a2l) }) \
            GLF_M16_T(t01, u01, a0h, a0l, b1h, b1l, { GLF_M16_A(cur, 3, a3h, a3l) })                          \
            GLF_M16_T(t02, u02, a0h, a0l, b2h, b2l, { if (LOAD_) advance(); })                                \
            GLF_M16_T(t03, u03, a0h, a0l, b3h, b3l, { if (NEXT_) GLF_M16_A(nxt, 0, a0h, a0l) })               \
            GLF_M16_T(t10, u10, a1h, a1l, b0h, b0l, { GLF_H8_PIECE(0, wr, CONV_, LOAD_) })                    \
            GLF_M16_T(t11, u11, a1h, a1l, b1h, b1l, { GLF_H8_PIECE(1, wr, CONV_, LOAD_) })                    \
            GLF_M16_T(t12, u12, a1h, a1l, b2h, b2l, { GLF_H8_PIECE(2, wr, CONV_, LOAD_) })                    \
            GLF_M16_T(t13, u13, a1h, a1l, b3h, b3l, { GLF_H8_PIECE(3, wr, CONV_, LOAD_) })                    \
            GLF_M16_T(t20, u20, a2h, a2l, b0h, b0l, { if (NEXT_) GLF_M16_A(nxt, 1, a1h, a1l) })               \
            GLF_M16_T(t30, u30, a3h, a3l, b0h, b0l, { if (NEXT_) GLF_M16_B(nxt, 0, b0h, b0l) GLF_H8_PIECE(4, wr, CONV_, LOAD_) }) \
            GLF_M16_T(t21, u21, a2h, a2l, b1h, b1l, { GLF_H8_PIECE(5, wr, CONV_, LOAD_) })                    \
            GLF_M16_T(t31, u31, a3h, a3l, b1h, b1l, { if (NEXT_) GLF_M16_B(nxt, 1, b1h, b1l) })               \
            GLF_M16_T(t22, u22, a2h, a2l, b2h, b2l, {})                                                       \
            GLF_M16_T(t32, u32, a3h, a3l, b2h, b2l, { if (NEXT_) GLF_M16_B(nxt, 2, b2h, b2l) })               \
            GLF_M16_T(t23, u23, a2h, a2l, b3h, b3l, {})                                                       \
            GLF_M16_T(t33, u33, a3h, a3l, b3h, b3l, {})                                                       \
            { const int t_ = cur; cur = nxt; nxt = wr; wr = t_; }                                             \
            __syncthreads();                                                                                  \
        }
        int it = 0;
        if (PA && BP) {
            for (; it + 3 < ntiles; ++it) GLF_M16_BODY_IL(true, true, true)
            if (it + 2 < ntiles) { GLF_M16_BODY_IL(true, false, true) ++it; }
            if (it + 1 < ntiles) { GLF_M16_BODY_IL(false, false, true) ++it; }
            GLF_M16_BODY_IL(false, false, false)
        } else {
            for (; it + 3 < ntiles; ++it) GLF_M16_BODY(true, true, true)
            if (it + 2 < ntiles) { GLF_M16_BODY(true, false, true) ++it; }
            if (it + 1 < ntiles) { GLF_M16_BODY(false, false, true) ++it; }
            GLF_M16_BODY(false, false, false)
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
        if (M16) {        // 16 x 16 tiles: element r of tile (i, j) is row 16 i + 4 (lane >> 4) + r, column 16 j + (lane & 15)
            const int col_l = lane & 15, row_l = 4 * (lane >> 4);
            auto park16 = [&](const f32x4& acc, int ti, int tj) __attribute__((always_inline)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[(16 * ti + row_l + r) * 64 + 16 * tj + col_l] = acc[r];
            };
            park16(t00 + u00 * 0x1p-11f, 0, 0); park16(t01 + u01 * 0x1p-11f, 0, 1); park16(t02 + u02 * 0x1p-11f, 0, 2); park16(t03 + u03 * 0x1p-11f, 0, 3);
            park16(t10 + u10 * 0x1p-11f, 1, 0); park16(t11 + u11 * 0x1p-11f, 1, 1); park16(t12 + u12 * 0x1p-11f, 1, 2); park16(t13 + u13 * 0x1p-11f, 1, 3);
            park16(t20 + u20 * 0x1p-11f, 2, 0); park16(t21 + u21 * 0x1p-11f, 2, 1); park16(t22 + u22 * 0x1p-11f, 2, 2); park16(t23 + u23 * 0x1p-11f, 2, 3);
            park16(t30 + u30 * 0x1p-11f, 3, 0); park16(t31 + u31 * 0x1p-11f, 3, 1); park16(t32 + u32 * 0x1p-11f, 3, 2); park16(t33 + u33 * 0x1p-11f, 3, 3);
        } else {
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
                const int row0 = tm * BM8 + wm + r0, hw = r_h * r_w;
                en = row0 / hw;
                const int rem = row0 - en * hw;
                ey = rem / r_w; ex = rem - ey * r_w;
            }
            if (p_accumulate && !p_rect && !p_colstats) {
                // C += result (a dgrad landing on the shortcut's gradient): all 16 old values are requested before the first is
                // needed -- fetched inside the store loop, each of its 4-deep batches waited out a full memory round trip
                float4 prev[16];
                const int rowb = tm * BM8 + wm + r0;
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
                const int row = tm * BM8 + wm + rin;
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
    } else if (!M16) {
        const int col_l = lane & 31, row_l = 4 * (lane >> 5);
        auto emit = [&](const f32x16& acc, int ti, int tj, double& cs, double& cq) {
            const int col = tn * BN + wn + 32 * tj + col_l;
            if (col >= pN) return;
            const float bv = p_bias ? p_bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) put(acc[r], tm * BM8 + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + row_l, col, bv, cs, cq);
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
    } else {
        // 16 x 16 tiles: element r of tile (i, j) is row 16 i + 4 (lane >> 4) + r, column 16 j + (lane & 15)
        const int col_l = lane & 15, row_l = 4 * (lane >> 4);
        double cs[4] = {0.0, 0.0, 0.0, 0.0}, cq[4] = {0.0, 0.0, 0.0, 0.0};
        auto emit16 = [&](const f32x4& acc, int ti, int tj) {
            const int col = tn * BN + wn + 16 * tj + col_l;
            if (col >= pN) return;
            const float bv = p_bias ? p_bias[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) put(acc[r], tm * BM8 + wm + 16 * ti + row_l + r, col, bv, cs[tj], cq[tj]);
        };
        emit16(t00 + u00 * 0x1p-11f, 0, 0); emit16(t01 + u01 * 0x1p-11f, 0, 1); emit16(t02 + u02 * 0x1p-11f, 0, 2); emit16(t03 + u03 * 0x1p-11f, 0, 3);
        emit16(t10 + u10 * 0x1p-11f, 1, 0); emit16(t11 + u11 * 0x1p-11f, 1, 1); emit16(t12 + u12 * 0x1p-11f, 1, 2); emit16(t13 + u13 * 0x1p-11f, 1, 3);
        emit16(t20 + u20 * 0x1p-11f, 2, 0); emit16(t21 + u21 * 0x1p-11f, 2, 1); emit16(t22 + u22 * 0x1p-11f, 2, 2); emit16(t23 + u23 * 0x1p-11f, 2, 3);
        emit16(t30 + u30 * 0x1p-11f, 3, 0); emit16(t31 + u31 * 0x1p-11f, 3, 1); emit16(t32 + u32 * 0x1p-11f, 3, 2); emit16(t33 + u33 * 0x1p-11f, 3, 3);
        if (args.colstats && p_rect != 1) {
            double* st = args.colstats;
#pragma unroll
            for (int j = 0; j < 4; ++j) {               // the four row groups of a column sit in lanes l, l + 16, l + 32, l + 48
                cs[j] += __shfl_xor(cs[j], 16, 64); cq[j] += __shfl_xor(cq[j], 16, 64);
                cs[j] += __shfl_xor(cs[j], 32, 64); cq[j] += __shfl_xor(cq[j], 32, 64);
                const int col = tn * BN + wn + 16 * j + col_l;
                if (lane < 16 && col < pN) { atomicAdd(st + col, cs[j]); atomicAdd(st + pN + col, cq[j]); }
            }
        }
    }
    if (args.amax_c && p_rect != 1) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
        if (lane == 0 && cmax > *reinterpret_cast<volatile float*>(args.amax_c))      // thousands of waves, ONE address: only a wave that raises it
            atomicMax(reinterpret_cast<unsigned*>(args.amax_c), __float_as_uint(cmax));
    }
#if defined(GLF_STAMPS) && GLF_STAMPS == 2
    if (args.partial != nullptr && blockIdx.x == gridDim.x / 2) {
        __builtin_amdgcn_s_waitcnt(0);          // the stores of this wave's part of C have been accepted
        const unsigned long long wg_t3 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* d_ = reinterpret_cast<unsigned long long*>(args.partial) + wave * 4;
            d_[0] = wg_t0; d_[1] = wg_t1; d_[2] = wg_t2; d_[3] = wg_t3;
        }
    }
#endif
}

// ----------------------------------------------------------------------------------------------------------
// tn kernel: C_tap[m][n] (+)= alpha * sum_{r in slice} A[arow(r)][m] * B[src(r,tap)][n]
// planes [r][128 cols] with 320-byte rows, fragments transposed on the fly by ds_read_b64_tr_b16.
// ----------------------------------------------------------------------------------------------------------
constexpr int RST = 320;
constexpr int PLANE_T = 32 * RST;
constexpr int OPER_T = 2 * PLANE_T;
constexpr size_t SMEM_TN_H = 2 * OPER_T + 32 * sizeof(int);

// SMALL: tile 64 x 64 -- every wave owns ONE 32 x 32 accumulator pair instead of a 64 x 64 quadrant, and a staging pass loads
// 64 columns per operand row instead of 128.  For the layer-1 weight gradients (M, N = 64 ... 256 with one of them 64, K = 193 600
// rows): on the 128 x 128 tile three of four MFMAs multiplied zero-page columns (64 x 64 x 193 600, nine taps: 0.32 ms at 45 TF).
template <bool GATHER, int NP, bool PA = false, bool PB = false, bool SMALL = false>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_tn_f16s_kernel(const GemmArgs args) {
    constexpr int TBM = SMALL ? 64 : BM, TBN = SMALL ? 64 : BN, WT = SMALL ? 32 : 64;
    const int pM = args.M, pN = args.N, pK = args.K, p_lda = args.lda, p_ldb = args.ldb, p_ldc = args.ldc;
    const int p_accumulate = args.accumulate, p_split = args.split;
    const int p_tiles_n = args.tiles_n;
    const unsigned p_tap_mask = args.tap_mask;
    const long long p_tsb = args.tap_stride_b, p_bsa = args.bsa, p_bsb = args.bsb, p_bsc = args.bsc;
    const float* __restrict__ p_A = args.A; const float* __restrict__ p_B = args.B;
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
    unsigned char* As = smem_s;
    unsigned char* Bs = smem_s + OPER_T;
    int* vflag = reinterpret_cast<int*>(smem_s + 2 * OPER_T);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = (wave >> 1) * WT, wn = (wave & 1) * WT;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p_tiles_n, tm = bid / p_tiles_n;
    int tap;
    {
        unsigned mm = p_tap_mask;
        for (int i = 0; i < (int)blockIdx.y; ++i) mm &= mm - 1;
        tap = __ffs(mm) - 1;
    }
    int tap_cy = 0, tap_cx = 0;  // gathered B rows: source pixel = (y * stride + tap_cy, x * stride + tap_cx)
    if (GATHER) {
        int ky = 0, kx = tap;
        while (kx >= args.g.kw) { kx -= args.g.kw; ++ky; }
        tap_cy = ky * args.g.dil - args.g.pad; tap_cx = kx * args.g.dil - args.g.pad;
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

    // staging map: a row of the tile is TBM (= TBN) floats = TBM / 4 threads; NTHREADS / (TBM / 4) rows per pass, 32 rows per tile
    constexpr int TPR = TBM / 4, RPP = NTHREADS / TPR, NPASS = 32 / RPP;          // 32 threads x 8 rows x 4 passes | 16 x 16 x 2
    const int c4 = tid % TPR, rr = tid / TPR;
    const int m0 = tm * TBM + 4 * c4, n0 = tn * TBN + 4 * c4;
    const int m0c = min(m0, pM - 4), n0c = min(n0, pN - 4);
    const int hw = GATHER ? r_h * r_w : 1;

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    f32x16 m00 = {0}, m01 = {0}, m10 = {0}, m11 = {0};
    float4 ra[NPASS], rb[NPASS];
    int rvalid[NPASS];

    // pixel coordinates (inside the tap's rectangle) of this thread's four rows of the NEXT tile to load: set once by
    // division, advanced by 32 rows per tile by carrying (tiles are loaded in order)
    int gn[NPASS] = {0}, gy[NPASS] = {0}, gx[NPASS] = {0};
    if (GATHER) {
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            const int r = r0 + rr + RPP * j;
            gn[j] = r / hw;
            const int rem = r - gn[j] * hw;
            gy[j] = rem / r_w; gx[j] = rem - gy[j] * r_w;
        }
    }
    auto load_tile = [&](int rbase) __attribute__((always_inline)) {
        long long src[NPASS], arow[NPASS];
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            const int r = rbase + rr + RPP * j;
            src[j] = -1;
            arow[j] = min(r, r1 - 1);
            if (GATHER) {
                if (r < r1) {          // (the tap is fixed for the workgroup: constant source offsets, no division for it)
                    const int y = r_y0 + gy[j], x = r_x0 + gx[j];
                    arow[j] = ((long long)gn[j] * g_hd + y) * g_wd + x;
                    const int sy = y * g_stride + tap_cy, sx = x * g_stride + tap_cx;
                    if ((unsigned)sy < (unsigned)g_hs && (unsigned)sx < (unsigned)g_ws) src[j] = (gn[j] * g_hs + sy) * g_ws + sx;
                }
                gx[j] += BK;
                while (gx[j] >= r_w) { gx[j] -= r_w; ++gy[j]; }
                while (gy[j] >= r_h) { gy[j] -= r_h; ++gn[j]; }
            } else if (r < r1) {
                src[j] = r;
            }
            rvalid[j] = src[j] >= 0;
        }
#pragma unroll
        for (int j = 0; j < NPASS; ++j) {
            // rows beyond the slice / in the conv padding and the column overhang read the zero page
            ra[j] = *reinterpret_cast<const float4*>((src[j] >= 0 && m0 < pM ? A + arow[j] * p_lda : p_zero) + m0c);
            rb[j] = *reinterpret_cast<const float4*>((src[j] >= 0 && n0 < pN ? B + src[j] * p_ldb : p_zero) + n0c);
        }
    };
#define GLF_HT_STORE(J)                                                                                    \
    {                                                                                                      \
        const SplitH sa = PA ? unpack4h(ra[J]) : split4h(ra[J], sc_a);                                     \
        const SplitH sb = PB ? unpack4h(rb[J]) : split4h(rb[J], sc_b);                                     \
        unsigned char* da = As + (rr + RPP * J) * RST + c4 * 8;                                            \
        unsigned char* db = Bs + (rr + RPP * J) * RST + c4 * 8;                                            \
        *reinterpret_cast<f16x4*>(da) = sa.h; if (NP == 3) *reinterpret_cast<f16x4*>(da + PLANE_T) = sa.l; \
        *reinterpret_cast<f16x4*>(db) = sb.h; if (NP == 3) *reinterpret_cast<f16x4*>(db + PLANE_T) = sb.l; \
        if (c4 == 0) vflag[rr + RPP * J] = rvalid[J];                                                      \
    }

    // transposing fragment read (see gemm_bf16s.hip): group g of 16 lanes: columns 16*(g&1).., k half g>>1
    const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int tr_off = (8 * (grp >> 1) + q) * RST + (16 * (grp & 1) + 4 * pp) * 2;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
#define GLF_HTR_FRAG(base, dst)                                                                             \
    {                                                                                                       \
        const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));                       \
        const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)((base) + 4 * RST));          \
        typedef short s16x8_ __attribute__((ext_vector_type(8)));                                           \
        const s16x8_ both_ = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                     \
        dst = __builtin_bit_cast(f16x8, both_);                                                             \
    }

    load_tile(r0);
    for (int rbase = r0; rbase < r1; rbase += BK) {
        GLF_HT_STORE(0) GLF_HT_STORE(1)
        if constexpr (NPASS == 4) { GLF_HT_STORE(2) GLF_HT_STORE(3) }
        __syncthreads();
        if (rbase + BK < r1) load_tile(rbase + BK);
        const bool any = __ballot(vflag[lane & 31] != 0) != 0ull;
        if (any) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const unsigned char* ab = As + s * 16 * RST + wm * 2 + tr_off;
                const unsigned char* bb = Bs + s * 16 * RST + wn * 2 + tr_off;
                if constexpr (SMALL) {
                    f16x8 ah, al, bh, bl;
                    GLF_HTR_FRAG(ab, ah) GLF_HTR_FRAG(ab + PLANE_T, al) GLF_HTR_FRAG(bb, bh) GLF_HTR_FRAG(bb + PLANE_T, bl)
                    c00 = GLF_MFMA_F16(ah, bh, c00);
                    if (NP == 3) { m00 = GLF_MFMA_F16(al, bh, m00); m00 = GLF_MFMA_F16(ah, bl, m00); }
                } else {
                    f16x8 b0h, b0l, b1h, b1l;
                    GLF_HTR_FRAG(bb, b0h) GLF_HTR_FRAG(bb + 64, b1h) GLF_HTR_FRAG(bb + PLANE_T, b0l) GLF_HTR_FRAG(bb + 64 + PLANE_T, b1l)
                    {
                        f16x8 ah, al;
                        GLF_HTR_FRAG(ab, ah) GLF_HTR_FRAG(ab + PLANE_T, al)
                        GLF_ROW3(c00, c01, m00, m01, ah, al, b0h, b0l, b1h, b1l)
                    }
                    {
                        f16x8 ah, al;
                        GLF_HTR_FRAG(ab + 64, ah) GLF_HTR_FRAG(ab + 64 + PLANE_T, al)
                        GLF_ROW3(c10, c11, m10, m11, ah, al, b0h, b0l, b1h, b1l)
                    }
                }
            }
        }
        __syncthreads();
    }

    const bool atomic = !args.partial && ((p_split > 1) || p_accumulate);
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    float cmax = 0.f;
    auto emit = [&](const f32x16& acc, int ti, int tj) {
        const int col = tn * TBN + wn + 32 * tj + col_l;
        if (col >= pN) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * TBM + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + row_l;
            if (row < pM) {
                float* dst = C + (long long)row * ldc_e + col;
                const float v = p_alpha * acc[r];
                if (atomic) atomicAdd(dst, v); else { *dst = v; cmax = fmaxf(cmax, fabsf(v)); }
            }
        }
    };
    emit(c00 + m00 * 0x1p-11f, 0, 0);
    if constexpr (!SMALL) {
        emit(c01 + m01 * 0x1p-11f, 0, 1);
        emit(c10 + m10 * 0x1p-11f, 1, 0); emit(c11 + m11 * 0x1p-11f, 1, 1);
    }
    if (args.amax_c && !atomic && !args.partial) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
        if (lane == 0 && cmax > *reinterpret_cast<volatile float*>(args.amax_c))      // thousands of waves, ONE address: only a wave that raises it
            atomicMax(reinterpret_cast<unsigned*>(args.amax_c), __float_as_uint(cmax));
    }
}

// ----------------------------------------------------------------------------------------------------------
// tn kernel, 8 waves: tile 256 (A columns) x 128 (B columns) x 32 rows, THREE LDS buffers and the slot-interleaved software
// pipeline of the rows kernel (tile t multiplied out of buffer t%3 with its k-step-0 fragments read before the barrier,
// tile t+2 converted into buffer (t+2)%3, tile t+3 in flight from global memory).  Planes are [r][cols] with dense rows
// (512 bytes for A, 256 for B: 48 KB per buffer) and the 64-byte chunks of a row XOR-swizzled with (row & 3): the
// transposing ds_read_b64_tr_b16 fragment reads (four rows x 64 bytes per half wave) and the 8-byte staging writes are
// conflict-free without padding.  Row coordinates (n, y, x) advance incrementally by 32 per tile; rows outside the slice,
// in the conv padding or in the column overhang are loaded from the zero page.
// Used when M > 128 (otherwise half of the 256-wide tile would be idle and the 128 x 128 kernel above runs).
// ----------------------------------------------------------------------------------------------------------
constexpr int TM8 = 256;
constexpr int RSA = 512, RSB = 256;
constexpr int PA8 = 32 * RSA, PB8 = 32 * RSB;
constexpr int TBUF8 = 2 * PA8 + 2 * PB8;
constexpr size_t SMEM_TN_H8 = 3 * TBUF8;

template <bool GATHER, int NP, bool PA = false, bool PB = false>
__global__ __launch_bounds__(NT8, 2) void gemm_tn_f16s8_kernel(const GemmArgs args) {
    const int pM = args.M, pN = args.N, pK = args.K, p_lda = args.lda, p_ldb = args.ldb, p_ldc = args.ldc;
    const int p_accumulate = args.accumulate, p_split = args.split;
    const int p_tiles_n = args.tiles_n;
    const unsigned p_tap_mask = args.tap_mask;
    const long long p_tsb = args.tap_stride_b, p_bsa = args.bsa, p_bsb = args.bsb, p_bsc = args.bsc;
    const float* __restrict__ p_A = args.A; const float* __restrict__ p_B = args.B;
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
#if defined(GLF_STAMPS) && GLF_STAMPS == 2
    unsigned long long wg_t0 = __builtin_amdgcn_s_memtime(), wg_t1 = 0, wg_t2 = 0;
#endif

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform, and known to be: everything derived from it stays in SGPRs
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
    const int ntiles = (r1 - r0 + BK - 1) / BK;

    // staging map: A tile 32 x 256 floats = 4 float4 per thread (rows ra0 + 8 j, column ca), B tile 32 x 128 = 2 (rows rb0 + 16 j).
    // ra0 is the wave index: an A row, its pixel coordinates and its base pointer are WAVE-UNIFORM and live in SGPRs (the
    // kernel sits at the 256-register limit: per-lane copies of that state spilled into the main loop, and a scratch reload
    // between the global loads turns every s_waitcnt vmcnt(5) into vmcnt(0)); a lane adds only its column offset.  Lanes in the
    // column overhang (m0 >= M) read column 0 instead: they feed accumulator columns that are never stored.
    const int ca = tid & 63, ra0 = wave;
    const int cb = tid & 31, rb0 = tid >> 5;
    const int m0 = tm * TM8 + 4 * ca, n0 = tn * BN + 4 * cb;
    const bool a_col_ok = m0 < pM, b_col_ok = n0 < pN;
    const unsigned a_col = a_col_ok ? (unsigned)m0 : 0u;
    const float* __restrict__ Bc = B + (b_col_ok ? n0 : 0);
    const float* __restrict__ zb = p_zero + (b_col_ok ? n0 : 0);

    // current row index and (for conv gathers) its pixel coordinates inside the rectangle, per staged row
    int ar[4], an[4], ay[4], ax[4];
    int br[2], bn[2], by[2], bx[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        ar[j] = r0 + ra0 + 8 * j;
        an[j] = 0; ay[j] = 0; ax[j] = 0;
        if (GATHER && p_rect) {
            const int hw = r_h * r_w;
            const int n = ar[j] / hw, rem = ar[j] - n * hw;
            const int yy = rem / r_w;
            an[j] = n; ay[j] = yy; ax[j] = rem - yy * r_w;
        }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        br[j] = r0 + rb0 + 16 * j;
        bn[j] = 0; by[j] = 0; bx[j] = 0;
        if (GATHER) {
            const int hw = r_h * r_w;
            const int n = br[j] / hw, rem = br[j] - n * hw;
            const int yy = rem / r_w;
            bn[j] = n; by[j] = yy; bx[j] = rem - yy * r_w;
        }
    }
    const float* pa[4];          // row bases (uniform); the lane's column offset a_col is added at the load
    const float* pb[2];
    int tap_cy = 0, tap_cx = 0;  // gathered B rows: source pixel = (y * stride + tap_cy, x * stride + tap_cx)
    if (GATHER) {
        int ky = 0, kx = tap;
        while (kx >= g_kw) { kx -= g_kw; ++ky; }
        tap_cy = ky * g_dil - g_pad; tap_cx = kx * g_dil - g_pad;
    }
    // pointers of the NEXT tile to load, then step every row by 32
    auto advance = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool ok = ar[j] < r1;
            long long arow = ar[j];
            if (GATHER && p_rect) arow = ((long long)an[j] * g_hd + r_y0 + ay[j]) * g_wd + r_x0 + ax[j];
            pa[j] = ok ? A + arow * p_lda : p_zero;
            ar[j] += BK;
            if (GATHER && p_rect) {
                ax[j] += BK;
                while (ax[j] >= r_w) { ax[j] -= r_w; ++ay[j]; }
                while (ay[j] >= r_h) { ay[j] -= r_h; ++an[j]; }
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            long long src = br[j];
            bool ok = br[j] < r1 && b_col_ok;
            if (GATHER) {
                // (the tap is fixed for the workgroup: its source offsets are two constants -- no per-row division, see map_src)
                const int sy = (r_y0 + by[j]) * g_stride + tap_cy, sx = (r_x0 + bx[j]) * g_stride + tap_cx;
                ok = ok && (unsigned)sy < (unsigned)g_hs && (unsigned)sx < (unsigned)g_ws;
                src = (bn[j] * g_hs + sy) * g_ws + sx;
            }
            pb[j] = ok ? Bc + src * p_ldb : zb;
            br[j] += BK;
            if (GATHER) {
                bx[j] += BK;
                while (bx[j] >= r_w) { bx[j] -= r_w; ++by[j]; }
                while (by[j] >= r_h) { by[j] -= r_h; ++bn[j]; }
            }
        }
    };

    f32x16 c00 = {0}, c01 = {0}, c10 = {0}, c11 = {0};
    f32x16 m00 = {0}, m01 = {0}, m10 = {0}, m11 = {0};
    float4 ra[4], rb[2];
    // swizzled staging offsets: 8-byte unit ca of row ra0 + 8 j (A) / unit cb of row rb0 + 16 j (B): (row & 3) is the thread's own
    const int st_a = ra0 * RSA + ((((ca >> 3) ^ (ra0 & 3)) << 6) | ((ca & 7) << 3));
    const int st_b = rb0 * RSB + ((((cb >> 3) ^ (rb0 & 3)) << 6) | ((cb & 7) << 3));
#define GLF_T8_CONV_A(J, buf_)                                                                               \
    {                                                                                                        \
        const SplitH s = PA ? unpack4h(ra[J]) : split4h(ra[J], sc_a);                                        \
        unsigned char* d = smem_s + (buf_) * TBUF8 + st_a + J * 8 * RSA;                                     \
        *reinterpret_cast<f16x4*>(d) = s.h; if (NP == 3) *reinterpret_cast<f16x4*>(d + PA8) = s.l;           \
    }
#define GLF_T8_CONV_B(J, buf_)                                                                               \
    {                                                                                                        \
        const SplitH s = PB ? unpack4h(rb[J]) : split4h(rb[J], sc_b);                                        \
        unsigned char* d = smem_s + (buf_) * TBUF8 + 2 * PA8 + st_b + J * 16 * RSB;                          \
        *reinterpret_cast<f16x4*>(d) = s.h; if (NP == 3) *reinterpret_cast<f16x4*>(d + PB8) = s.l;           \
    }
    // (a pre-split piece: the store stays ahead of the load that refills its registers -- see GLF_H8_PIN)
#define GLF_T8_PIECE(pc, buf_, conv_, load_)                                                                 \
    switch (pc) {                                                                                            \
        case 0: if (conv_) GLF_T8_CONV_A(0, buf_) GLF_H8_PIN(PA) if (load_) ra[0] = *reinterpret_cast<const float4*>(pa[0] + a_col); GLF_H8_PIN(PA) break; \
        case 1: if (conv_) GLF_T8_CONV_A(1, buf_) GLF_H8_PIN(PA) if (load_) ra[1] = *reinterpret_cast<const float4*>(pa[1] + a_col); GLF_H8_PIN(PA) break; \
        case 2: if (conv_) GLF_T8_CONV_A(2, buf_) GLF_H8_PIN(PA) if (load_) ra[2] = *reinterpret_cast<const float4*>(pa[2] + a_col); GLF_H8_PIN(PA) break; \
        case 3: if (conv_) GLF_T8_CONV_A(3, buf_) GLF_H8_PIN(PA) if (load_) ra[3] = *reinterpret_cast<const float4*>(pa[3] + a_col); GLF_H8_PIN(PA) break; \
        case 4: if (conv_) GLF_T8_CONV_B(0, buf_) GLF_H8_PIN(PB) if (load_) rb[0] = *reinterpret_cast<const float4*>(pb[0]); GLF_H8_PIN(PB) break; \
        default: if (conv_) GLF_T8_CONV_B(1, buf_) GLF_H8_PIN(PB) if (load_) rb[1] = *reinterpret_cast<const float4*>(pb[1]); GLF_H8_PIN(PB) break; \
    }

    // transposing fragment reads (see gemm_bf16s.hip): group g of 16 lanes: columns 16*(g&1).., k half g>>1; the lane's rows
    // are q and q + 4 (mod 8), so its swizzle key is q
    const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int tr_in = 32 * (grp & 1) + 8 * pp;
    const int tr_a0 = (8 * (grp >> 1) + q) * RSA + ((((wm >> 5) + 0) ^ q) << 6) + tr_in;       // columns wm .. wm + 31
    const int tr_a1 = (8 * (grp >> 1) + q) * RSA + ((((wm >> 5) + 1) ^ q) << 6) + tr_in;       // columns wm + 32 .. wm + 63
    const int tr_b0 = (8 * (grp >> 1) + q) * RSB + ((((wn >> 5) + 0) ^ q) << 6) + tr_in;
    const int tr_b1 = (8 * (grp >> 1) + q) * RSB + ((((wn >> 5) + 1) ^ q) << 6) + tr_in;
    typedef __attribute__((address_space(3))) s16x4* lds_s16x4;
#define GLF_T8_FRAG(base, RS_, dst)                                                                         \
    {                                                                                                       \
        const s16x4 lo_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(base));                       \
        const s16x4 hi_ = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)((base) + 4 * RS_));          \
        typedef short s16x8_ __attribute__((ext_vector_type(8)));                                           \
        const s16x8_ both_ = __builtin_shufflevector(lo_, hi_, 0, 1, 2, 3, 4, 5, 6, 7);                     \
        dst = __builtin_bit_cast(f16x8, both_);                                                             \
    }
    // fragment w_ of set P (f = k-step 0, g = k-step 1) of the tile in buffer buf_
#define GLF_TL_A(P, w_, buf_, ks_, off_, pl_) GLF_T8_FRAG(smem_s + (buf_) * TBUF8 + (ks_) * 16 * RSA + (off_) + (pl_), RSA, P##w_)
#define GLF_TL_B(P, w_, buf_, ks_, off_, pl_) GLF_T8_FRAG(smem_s + (buf_) * TBUF8 + 2 * PA8 + (ks_) * 16 * RSB + (off_) + (pl_), RSB, P##w_)

    if (ntiles <= 0) return;
    // prologue: tiles 0 and 1 -> buffers 0 and 1, tile 2 raw in registers, the k-step-0 fragments of tile 0 in f*
    advance();
#pragma unroll
    for (int pc = 0; pc < 6; ++pc) { GLF_T8_PIECE(pc, 0, false, true) }
    {
        const bool more = ntiles > 1;
        if (more) advance();
#pragma unroll
        for (int pc = 0; pc < 6; ++pc) { GLF_T8_PIECE(pc, 0, true, more) }
        if (more) {
            const bool more2 = ntiles > 2;
            if (more2) advance();
#pragma unroll
            for (int pc = 0; pc < 6; ++pc) { GLF_T8_PIECE(pc, 1, true, more2) }
        }
    }
    __syncthreads();
    f16x8 fb0h, fb1h, fb0l, fb1l, fa0h, fa0l, fa1h, fa1l;
    GLF_TL_B(f, b0h, 0, 0, tr_b0, 0) GLF_TL_B(f, b1h, 0, 0, tr_b1, 0) GLF_TL_A(f, a0h, 0, 0, tr_a0, 0) GLF_TL_A(f, a1h, 0, 0, tr_a1, 0)
    if (NP == 3) {
        GLF_TL_A(f, a0l, 0, 0, tr_a0, PA8) GLF_TL_B(f, b0l, 0, 0, tr_b0, PB8) GLF_TL_B(f, b1l, 0, 0, tr_b1, PB8) GLF_TL_A(f, a1l, 0, 0, tr_a1, PA8)
    } else { fa0l = fa0h; fb0l = fb0h; fb1l = fb1h; fa1l = fa1h; }
    int cur = 0, nxt = 1, wr = 2;           // LDS buffers of tile it, it+1, it+2
    // staging work items as in the rows kernel (GLF_IL_ITEM): one slot per pre-split piece, three per piece split here
    f32x2 sx01_, sx23_, sr01_, sr23_;
    f16x2 sh01_, sh23_;
#define GLF_TL_SA(J, ST_, CONV_, LOAD_)                                                                       \
            if (PA) { if (ST_ == 3) { GLF_T8_PIECE(J, wr, CONV_, LOAD_) } }                                   \
            else if (CONV_) {                                                                                 \
                if (ST_ == 1) GLF_IL_ST1(ra[J], sc_a)                                                         \
                else if (ST_ == 2) GLF_IL_ST2()                                                               \
                else { GLF_IL_ST3(smem_s + wr * TBUF8 + st_a + J * 8 * RSA, PA8)                              \
                       if (LOAD_) ra[J] = *reinterpret_cast<const float4*>(pa[J] + a_col); }                  \
            }
#define GLF_TL_SB(J, ST_, CONV_, LOAD_)                                                                       \
            if (PB) { if (ST_ == 3) { GLF_T8_PIECE(4 + J, wr, CONV_, LOAD_) } }                               \
            else if (CONV_) {                                                                                 \
                if (ST_ == 1) GLF_IL_ST1(rb[J], sc_b)                                                         \
                else if (ST_ == 2) GLF_IL_ST2()                                                               \
                else { GLF_IL_ST3(smem_s + wr * TBUF8 + 2 * PA8 + st_b + J * 16 * RSB, PB8)                   \
                       if (LOAD_) rb[J] = *reinterpret_cast<const float4*>(pb[J]); }                          \
            }
#define GLF_TL_ITEM(I_, CONV_, LOAD_)                                                                         \
            {                                                                                                 \
                constexpr int nA_ = PA ? 4 : 12, nB_ = PB ? 2 : 6, i_ = (I_);                                 \
                if constexpr (i_ < nA_) {                                                                     \
                    constexpr int j_ = PA ? i_ : i_ / 3, st_ = PA ? 3 : i_ % 3 + 1;                           \
                    GLF_TL_SA(j_, st_, CONV_, LOAD_)                                                          \
                } else if constexpr (i_ < nA_ + nB_) {                                                        \
                    constexpr int k_ = i_ - nA_, j_ = PB ? k_ : k_ / 3, st_ = PB ? 3 : k_ % 3 + 1;            \
                    GLF_TL_SB(j_, st_, CONV_, LOAD_)                                                          \
                }                                                                                             \
            }
#define GLF_T8_BODY(CONV_, LOAD_, NEXT_)                                                                      \
    {                                                                                                         \
        f16x8 gb0h, gb1h, gb0l, gb1l, ga0h, ga0l, ga1h, ga1l;                                                 \
        GLF_IL_ROW(c00, c01, m00, m01, fa0h, fa0l, fb0h, fb0l, fb1h, fb1l,                                    \
            { GLF_TL_B(g, b0h, cur, 1, tr_b0, 0) GLF_TL_A(g, a0h, cur, 1, tr_a0, 0) },                        \
            { GLF_TL_B(g, b1h, cur, 1, tr_b1, 0) if (NP == 3) { GLF_TL_A(g, a0l, cur, 1, tr_a0, PA8) } },     \
            { if (NP == 3) { GLF_TL_B(g, b0l, cur, 1, tr_b0, PB8) GLF_TL_B(g, b1l, cur, 1, tr_b1, PB8) } },   \
            { GLF_TL_A(g, a1h, cur, 1, tr_a1, 0) if (NP == 3) { GLF_TL_A(g, a1l, cur, 1, tr_a1, PA8) } },     \
            { if (LOAD_) advance(); },                                                                        \
            { GLF_TL_ITEM(0, CONV_, LOAD_) })                                                                 \
        GLF_IL_ROW(c10, c11, m10, m11, fa1h, fa1l, fb0h, fb0l, fb1h, fb1l,                                    \
            { GLF_TL_ITEM(1, CONV_, LOAD_) },                                                                 \
            { GLF_TL_ITEM(2, CONV_, LOAD_) },                                                                 \
            { GLF_TL_ITEM(3, CONV_, LOAD_) },                                                                 \
            { GLF_TL_ITEM(4, CONV_, LOAD_) },                                                                 \
            { GLF_TL_ITEM(5, CONV_, LOAD_) },                                                                 \
            { GLF_TL_ITEM(6, CONV_, LOAD_) if (NEXT_) { GLF_TL_A(f, a0h, nxt, 0, tr_a0, 0) if (NP == 3) { GLF_TL_A(f, a0l, nxt, 0, tr_a0, PA8) } } }) \
        if (NP != 3) { ga0l = ga0h; gb0l = gb0h; gb1l = gb1h; ga1l = ga1h; }                                  \
        GLF_IL_ROW(c00, c01, m00, m01, ga0h, ga0l, gb0h, gb0l, gb1h, gb1l,                                    \
            { GLF_TL_ITEM(7, CONV_, LOAD_) if (NEXT_) { GLF_TL_B(f, b0h, nxt, 0, tr_b0, 0) GLF_TL_B(f, b1h, nxt, 0, tr_b1, 0) } }, \
            { GLF_TL_ITEM(8, CONV_, LOAD_) if (NEXT_ && NP == 3) { GLF_TL_B(f, b0l, nxt, 0, tr_b0, PB8) GLF_TL_B(f, b1l, nxt, 0, tr_b1, PB8) } }, \
            { GLF_TL_ITEM(9, CONV_, LOAD_) if (NEXT_) { GLF_TL_A(f, a1h, nxt, 0, tr_a1, 0) if (NP == 3) { GLF_TL_A(f, a1l, nxt, 0, tr_a1, PA8) } } }, \
            { GLF_TL_ITEM(10, CONV_, LOAD_) },                                                                \
            { GLF_TL_ITEM(11, CONV_, LOAD_) },                                                                \
            { GLF_TL_ITEM(12, CONV_, LOAD_) })                                                                \
        GLF_IL_ROW(c10, c11, m10, m11, ga1h, ga1l, gb0h, gb0l, gb1h, gb1l,                                    \
            { GLF_TL_ITEM(13, CONV_, LOAD_) },                                                                \
            { GLF_TL_ITEM(14, CONV_, LOAD_) },                                                                \
            { GLF_TL_ITEM(15, CONV_, LOAD_) },                                                                \
            { GLF_TL_ITEM(16, CONV_, LOAD_) },                                                                \
            { GLF_TL_ITEM(17, CONV_, LOAD_) },                                                                \
            {})                                                                                               \
        if (NP != 3 && NEXT_) { fa0l = fa0h; fb0l = fb0h; fb1l = fb1h; fa1l = fa1h; }                         \
        { const int t_ = cur; cur = nxt; nxt = wr; wr = t_; }                                                 \
        __syncthreads();                                                                                      \
    }
    int it = 0;
#if defined(GLF_STAMPS) && GLF_STAMPS == 2
    wg_t1 = __builtin_amdgcn_s_memtime();
#endif
    for (; it + 3 < ntiles; ++it) GLF_T8_BODY(true, true, true)
    if (it + 2 < ntiles) { GLF_T8_BODY(true, false, true) ++it; }
    if (it + 1 < ntiles) { GLF_T8_BODY(false, false, true) ++it; }
    GLF_T8_BODY(false, false, false)
#if defined(GLF_STAMPS) && GLF_STAMPS == 2
    wg_t2 = __builtin_amdgcn_s_memtime();
#endif

    const bool atomic = !args.partial && ((p_split > 1) || p_accumulate);
    const int col_l = lane & 31, row_l = 4 * (lane >> 5);
    float cmax = 0.f;
    auto emit = [&](const f32x16& acc, int ti, int tj) {
        const int col = tn * BN + wn + 32 * tj + col_l;
        if (col >= pN) return;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = tm * TM8 + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + row_l;
            if (row < pM) {
                float* dst = C + (long long)row * ldc_e + col;
                const float v = p_alpha * acc[r];
                if (atomic) atomicAdd(dst, v); else { *dst = v; cmax = fmaxf(cmax, fabsf(v)); }
            }
        }
    };
    emit(c00 + m00 * 0x1p-11f, 0, 0); emit(c01 + m01 * 0x1p-11f, 0, 1);
    emit(c10 + m10 * 0x1p-11f, 1, 0); emit(c11 + m11 * 0x1p-11f, 1, 1);
    if (args.amax_c && !atomic && !args.partial) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) cmax = fmaxf(cmax, __shfl_xor(cmax, o, 64));
        if (lane == 0 && cmax > *reinterpret_cast<volatile float*>(args.amax_c))      // thousands of waves, ONE address: only a wave that raises it
            atomicMax(reinterpret_cast<unsigned*>(args.amax_c), __float_as_uint(cmax));
    }
#if defined(GLF_STAMPS) && GLF_STAMPS == 2
    if (args.stamps != nullptr && blockIdx.x == gridDim.x / 2 && blockIdx.z == 0 && blockIdx.y == 0) {
        __builtin_amdgcn_s_waitcnt(0);
        const unsigned long long wg_t3 = __builtin_amdgcn_s_memtime();
        if (lane == 0) {
            unsigned long long* d_ = args.stamps + 64 + wave * 4;
            d_[0] = wg_t0; d_[1] = wg_t1; d_[2] = wg_t2; d_[3] = wg_t3; if (wave == 0) args.stamps[63] = (unsigned long long)ntiles;
        }
    }
#endif
}

// max |x| over a [rows, cols] view (row stride ld) -> *out (non-negative floats order like their bit patterns)
__global__ __launch_bounds__(256) void amax_kernel(const float* __restrict__ x, long long rows, int cols, long long ld,
                                                   int vec, unsigned* __restrict__ out) {
    const int c4n = vec ? cols >> 2 : 0;
    const long long total4 = rows * c4n;
    float m = 0.f;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / c4n;
        const int c = (int)(i - r * c4n) * 4;
        const float4 v = *reinterpret_cast<const float4*>(x + r * ld + c);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    const int tail = cols - 4 * c4n;
    if (tail) {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < rows * tail; i += (long long)gridDim.x * blockDim.x) {
            const long long r = i / tail;
            m = fmaxf(m, fabsf(x[r * ld + (cols - tail) + (int)(i - r * tail)]));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        if (m > 0.f) atomicMax(out, __float_as_uint(m));
    }
}

// x [rows][cols] (row stride ld, cols % 4 == 0) -> the packed pre-split image out (row stride ldo): every float4 of x
// becomes {h0 h1 h2 h3 l0 l1 l2 l3}, x * s = h + 2^-11 l, s = the power of two pow2_scale() derives from *amax -- exactly
// what the contraction kernels compute in their staging path
__global__ __launch_bounds__(256) void split_packed_kernel(const float* __restrict__ x, long long rows, int cols4, long long ld,
                                                           const float* __restrict__ amax, float* __restrict__ out, long long ldo) {
    float sc, inv;
    pow2_scale(amax, sc, inv);
    const long long total = rows * cols4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / cols4;
        const int c = (int)(i - r * cols4) * 4;
        const SplitH s = split4h(*reinterpret_cast<const float4*>(x + r * ld + c), sc);
        const float2 h = __builtin_bit_cast(float2, s.h), l = __builtin_bit_cast(float2, s.l);
        *reinterpret_cast<float4*>(out + r * ldo + c) = make_float4(h.x, h.y, l.x, l.y);
    }
}

}  // namespace

namespace glf {

int launch_split_packed(const float* x, long long rows, int cols, long long ld, const float* amax, float* out, long long ldo, hipStream_t s) {
    const long long total = rows * (cols / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(split_packed_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, rows, cols / 4, ld, amax, out, ldo);
    return check_launch("split_f16_packed");
}

int init_gemm_f16s_attrs() {
    hipError_t e;
#define SET_ATTR(fn, bytes)                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
    if (e != hipSuccess) return fail(GLF_ERR_LAUNCH, "hipFuncSetAttribute(" #fn "): %s", hipGetErrorString(e));
#define SET_ROWS(G, NP_, PA_, PB_) SET_ATTR((gemm_rows_f16s8_kernel<G, NP_, false, PB_, PA_>), SMEM_ROWS_H8)
#define SET_TN(G, NP_, PA_, PB_) SET_ATTR((gemm_tn_f16s_kernel<G, NP_, PA_, PB_>), SMEM_TN_H) SET_ATTR((gemm_tn_f16s_kernel<G, NP_, PA_, PB_, true>), SMEM_TN_H) \
                                 SET_ATTR((gemm_tn_f16s8_kernel<G, NP_, PA_, PB_>), SMEM_TN_H8)
#define SET_ALL(G, NP_) SET_ATTR((gemm_rows_f16s8_kernel<G, 3, true, true, true>), SMEM_ROWS_H8) SET_ROWS(G, NP_, false, false) SET_ROWS(G, NP_, true, false) SET_ROWS(G, NP_, false, true) SET_ROWS(G, NP_, true, true) \
                        SET_TN(G, NP_, false, false) SET_TN(G, NP_, true, false) SET_TN(G, NP_, false, true) SET_TN(G, NP_, true, true)
    SET_ALL(false, 3) SET_ALL(true, 3) SET_ALL(false, 1) SET_ALL(true, 1)
#undef SET_ALL
#undef SET_TN
#undef SET_ROWS
#undef SET_ATTR
    return init_gemm_f16s4_attrs();
}

// Eligibility: the aligned fast path, and every row that can be redirected to the zero page must fit into it.
bool f16s_rows_ok(const GemmArgs& a) {
    return a.vec_a && a.vec_b && (a.K % BK) == 0 && a.K >= BK && a.K <= ZERO_PAGE_FLOATS && zero_page() != nullptr;
}
bool f16s_tn_ok(const GemmArgs& a) {
    return a.vec_a && a.vec_b && (a.M % 4) == 0 && (a.N % 4) == 0 && a.M >= 4 && a.N >= 4 &&
           a.M <= ZERO_PAGE_FLOATS && a.N <= ZERO_PAGE_FLOATS && zero_page() != nullptr;
}

#ifdef GLF_STAMPS
static void* stamps_buffer() {
    static void* p = [] { void* q = nullptr; (void)hipMalloc(&q, 8192); (void)hipMemset(q, 0, 8192); return q; }();
    return p;
}
extern "C" int glf_debug_stamps(void* host_out) {          // [8 waves][16 iterations][8] s_memtime values of the LAST NT launch
    (void)hipDeviceSynchronize();
    return (int)hipMemcpy(host_out, stamps_buffer(), 8192, hipMemcpyDeviceToHost);
}
#endif

int launch_rows_f16s(const GemmArgs& a0, dim3 grid, bool gather, int nprod, hipStream_t s) {
    if (use_f16s4(a0)) return launch_rows_f16s4(a0, grid, gather, nprod, s);      // short reductions: two workgroups per CU
    GemmArgs a = a0;
    a.zeros = zero_page();
    long long tiles_m = (a.M + BM8 - 1) / BM8;
    if (a.rect == 2) {
        tiles_m = 0;
        for (int r = 0; r < 9; ++r) {
            int y0, y1, x0, x1;
            unsigned rm;
            region_of(a.gather, r, a.g.dil, a.g.hd, a.g.wd, y0, y1, x0, x1, rm);
            tiles_m += ((long long)a.g.n_img * (y1 - y0) * (x1 - x0) + BM8 - 1) / BM8;
        }
    } else if (a.rect) {
        tiles_m = 0;
        for (unsigned mm = a.tap_mask; mm; mm &= mm - 1) {
            const int t = __builtin_ctz(mm);
            int y0, y1, x0, x1;
            tap_rect(a.gather, t, a.g.kw, a.g.pad, a.g.dil, a.g.hs, a.g.ws, a.g.hd, a.g.wd, y0, y1, x0, x1);
            tiles_m += ((long long)a.g.n_img * (y1 - y0) * (x1 - x0) + BM8 - 1) / BM8;
        }
    }
    a.tiles_m = (int)tiles_m;
    dim3 g2((unsigned)(tiles_m * a.tiles_n), 1, grid.z);
    // tile order: groups of 4 row tiles x all column tiles once there are >= 8 column tiles (more of the streamed operands
    // served from the XCD's L2: +2 % on the N >= 2048 shapes now that the loop is MFMA-bound; -2 % at 4 column tiles)
    const int group_m = (a.rect == 0 && a.tiles_n >= 8) ? 4 : 0;
    a.flags = (group_m & 0xff) << 8;
    const bool pa = a.a_presplit != 0, pb = a.b_presplit != 0;
#ifdef GLF_STAMPS
    a.partial = reinterpret_cast<float*>(stamps_buffer());
#endif
#define GLF_LAUNCH_ROWS(G, NP_, PA_, PB_) hipLaunchKernelGGL((gemm_rows_f16s8_kernel<G, NP_, false, PB_, PA_>), g2, dim3(NT8), SMEM_ROWS_H8, s, a)
#define GLF_ROWS_P(G, NP_)                                                                     \
    { if (pa && pb) GLF_LAUNCH_ROWS(G, NP_, true, true); else if (pa) GLF_LAUNCH_ROWS(G, NP_, true, false); \
      else if (pb) GLF_LAUNCH_ROWS(G, NP_, false, true); else GLF_LAUNCH_ROWS(G, NP_, false, false); }
    // both operands pre-split, three products: the 16x16x32 MFMA form of the loop (same cycles per FLOP as 32x32x16; the chip
    // holds a higher clock on it -- MI355X_MICROARCH.md, DVFS give-back item 7)
    static const bool m16p = [] { const char* e = getenv("GLF_MFMA16_PRESPLIT"); return e ? e[0] != '0' : false; }();
    if (nprod == 3 && pa && pb && m16p) {
        if (gather) hipLaunchKernelGGL((gemm_rows_f16s8_kernel<true, 3, true, true, true>), g2, dim3(NT8), SMEM_ROWS_H8, s, a);
        else hipLaunchKernelGGL((gemm_rows_f16s8_kernel<false, 3, true, true, true>), g2, dim3(NT8), SMEM_ROWS_H8, s, a);
        return check_launch("gemm_nt(f16x3, 16x16x32, pre-split)");
    }
    if (nprod == 3) { if (gather) GLF_ROWS_P(true, 3) else GLF_ROWS_P(false, 3) }
    else { if (gather) GLF_ROWS_P(true, 1) else GLF_ROWS_P(false, 1) }
#undef GLF_ROWS_P
#undef GLF_LAUNCH_ROWS
    return check_launch("gemm_nt(f16x3, 256x128)");
}

int launch_tn_f16s(const GemmArgs& a0, dim3 grid, bool gather, int nprod, hipStream_t s) {
    GemmArgs a = a0;
    a.zeros = zero_page();
#ifdef GLF_STAMPS
    a.stamps = reinterpret_cast<unsigned long long*>(stamps_buffer());
#endif
    const bool pa = a.a_presplit != 0, pb = a.b_presplit != 0;
    // 64 x 64 tiles when one extent is <= 64 and the other small too (layer 1: 64 x 64, 256 x 64, 64 x 256): no zero-page columns
    static const bool small_on = [] { const char* e = getenv("GLF_TN_SMALL"); return e ? e[0] != '0' : true; }();
    const bool small = small_on && (a.M <= 64 || a.N <= 64) && a.M <= 256 && a.N <= 256;
    const bool wide = !small && a.M > BM;   // 256-wide tiles: re-derive the grid
    dim3 g2 = grid;
    if (wide) {
        a.tiles_m = (a.M + TM8 - 1) / TM8;
        g2 = dim3((unsigned)(a.tiles_m * a.tiles_n), grid.y, grid.z);
    } else if (small) {
        a.tiles_m = (a.M + 63) / 64; a.tiles_n = (a.N + 63) / 64;
        g2 = dim3((unsigned)(a.tiles_m * a.tiles_n), grid.y, grid.z);
    }
#define GLF_LAUNCH_TN(G, NP_, PA_, PB_)                                                                                        \
    { if (wide) hipLaunchKernelGGL((gemm_tn_f16s8_kernel<G, NP_, PA_, PB_>), g2, dim3(NT8), SMEM_TN_H8, s, a);                 \
      else if (small) hipLaunchKernelGGL((gemm_tn_f16s_kernel<G, NP_, PA_, PB_, true>), g2, dim3(NTHREADS), SMEM_TN_H, s, a);  \
      else hipLaunchKernelGGL((gemm_tn_f16s_kernel<G, NP_, PA_, PB_>), g2, dim3(NTHREADS), SMEM_TN_H, s, a); }
#define GLF_TN_P(G, NP_)                                                                       \
    { if (pa && pb) GLF_LAUNCH_TN(G, NP_, true, true) else if (pa) GLF_LAUNCH_TN(G, NP_, true, false) \
      else if (pb) GLF_LAUNCH_TN(G, NP_, false, true) else GLF_LAUNCH_TN(G, NP_, false, false) }
    if (nprod == 3) { if (gather) GLF_TN_P(true, 3) else GLF_TN_P(false, 3) }
    else { if (gather) GLF_TN_P(true, 1) else GLF_TN_P(false, 1) }
#undef GLF_TN_P
#undef GLF_LAUNCH_TN
    return check_launch("gemm_tn(f16x3)");
}

// *out = max(*out, max |x|): out must hold a non-negative float (0 to start a fresh maximum)
int launch_amax(const float* x, long long rows, int cols, long long ld, int vec, float* out, hipStream_t s) {
    const long long total4 = rows * (long long)(vec && (cols >> 2) > 0 ? (cols >> 2) : cols);
    long long blocks = (total4 + 256 * 8 - 1) / (256 * 8);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(amax_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, rows, cols, ld, vec, reinterpret_cast<unsigned*>(out));
    return check_launch("amax");
}

}  // namespace glf
