// 16-bit-storage ("S16", BASELINE configs 3 / 5) contraction kernels: bf16 operands in HBM, ONE v_mfma_f32_32x32x16_bf16 per
// product, fp32 accumulate, bf16 (activations / their gradients) or fp32 (weight gradients) results.
//
// What is different from the fp32-storage kernels (gemm_f16s.hip): the operands need no conversion, so nothing passes
// through registers on the way in -- every tile goes global -> LDS by LDS-DMA (global_load_lds_dwordx4: 1 KiB per wave
// instruction, destination = wave-uniform base + 16 * lane).  The LDS image is therefore lane-linear and the bank swizzle
// sits on the SOURCE side: lane i of an instruction fetches the 16-byte chunk that belongs at position i.  Conv padding,
// tile overhang and out-of-range gather rows are loads from the library's zero page (no selects, no branches).
//   rows kernel (NT): C[m][n] = alpha * sum_tap sum_k A[src(m,tap)][k] * B_tap[n][k] (+ bias)
//     256 x BN x 64 tiles (BN = 128 | 64), 8 waves (4 x 2), three LDS stages; per K-tile: one counted s_waitcnt vmcnt, one raw
//     s_barrier, the loads of tile t+2 issued, 4 k-steps of ds_read_b128 fragments + MFMAs.  LDS rows are 128 bytes with
//     chunk ^= (row >> 1) & 7: conflict-free ds_read_b128 for the 32x32x16 operand map (row = lane & 31, chunk = 2 s + lane >> 5).
//     Epilogue: accumulators parked in LDS (fp32), rows written back as whole 16-byte pieces (8 bf16 / 4 floats); optional
//     per-column sum / sum of squares (BatchNorm statistics) folded through LDS to one f64 atomic per column and workgroup.
//     Gather modes as in gemm_common.h: forward conv (1), transposed conv (2), region mode (rect = 2) for 3x3 "same" convs.
//   tn kernel: C_tap[m][n] = alpha * sum_r A[r][m] * B[src(r,tap)][n]      (weight gradients; attention M = phi^T g)
//     256 x 128 tiles over 64 reduction rows per stage; LDS images stay [r][m] / [r][n] (dense 512 / 256-byte rows, 64-byte
//     chunks XORed with r & 3) and fragments are transposed on the way out by ds_read_b64_tr_b16; split-K slices store
//     partial slabs that s16_tn_reduce_kernel folds in slice order (no atomics: bitwise reproducible gradients).
#include "gemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16;

struct S16Args {
    const u16* A; const u16* B; const float* bias; void* C;
    int c_bf16;                                   // 1: C holds bf16, 0: fp32
    int M, N, K, lda, ldb, ldc, taps;
    unsigned tap_mask;
    long long tap_stride_b;
    int gather;
    Geo g;
    long long bsa, bsb, bsc;
    float alpha;
    int split, tiles_n, rect;
    const u16* zeros;                             // zero page (>= 512 KiB of zeros)
    double* colstats;                             // NT: [2][N] += column sums of C and C^2 (null = not wanted)
    float* partial;                               // TN: partial slabs [batch*split][kept taps][M][N] (null = direct store)
    int accumulate;                               // NT, fp32 or bf16 C: C += result
};

#define GLF_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* glb_ptr_t;

// one LDS-DMA wave instruction: lane i's 16 bytes at `src` land at lds_base + 16 i (lds_base wave-uniform)
__device__ __forceinline__ void glds16(const u16* src, unsigned char* lds_base) {
    __builtin_amdgcn_global_load_lds((glb_ptr_t)src, (lds_ptr_t)lds_base, 16, 0, 0);
}

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const f32x2_ v = {a, b};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}

constexpr int TM = 256;                  // rows of a workgroup tile
constexpr int TK = 64;                   // reduction depth of a stage (tn kernel; rows kernel: template parameter TK_)
constexpr int NTH = 512;
constexpr int NSTAGE = 3;

// TK_ = 64: 144 KiB of LDS, one workgroup per CU.  TK_ = 32: 72 KiB, TWO workgroups per CU (<= 128 registers): the second one's
// main loop covers the first one's prologue and epilogue -- what the short reductions (K <= 512: 4-16 K-tiles between a cold
// start and a 64 KiB store phase) are made of.
template <int BN_, int TK_> struct RowsCfg {
    static constexpr int A_STAGE = TM * TK_ * 2;
    static constexpr int B_STAGE = BN_ * TK_ * 2;
    static constexpr int STAGE = A_STAGE + B_STAGE;
    static constexpr int WNC = BN_ / 2;               // columns of a wave's output tile
    static constexpr int NJ = WNC / 32;               // 32-column blocks per wave
    static constexpr int PS = WNC + 4;                // parked row stride in floats
    static constexpr int PARK_ROWS = TK_ == 64 ? 64 : 32;   // rows of its 64-row tile a wave parks at a time
    static constexpr int PARK = PARK_ROWS * PS * 4;   // bytes per wave
    static constexpr int STATS_OFF = 8 * PARK;        // [4][BN][2] floats behind the parked tiles
    static constexpr size_t SMEM = (size_t)(NSTAGE * STAGE > STATS_OFF + 4 * BN_ * 2 * 4 ? NSTAGE * STAGE : STATS_OFF + 4 * BN_ * 2 * 4) + 16;
};

// ----------------------------------------------------------------------------------------------------------------------
// rows kernel (NT)
// ----------------------------------------------------------------------------------------------------------------------
template <bool GATHER, int BN_, int TK_>
__global__ __launch_bounds__(NTH, TK_ == 64 ? 2 : 4) void s16_rows_kernel(const S16Args args) {
    using Cfg = RowsCfg<BN_, TK_>;
    constexpr int NJ = Cfg::NJ, WNC = Cfg::WNC, PS = Cfg::PS, A_STAGE = Cfg::A_STAGE;
    constexpr int ROWB = TK_ * 2;                       // bytes of an LDS row
    constexpr int CPR = TK_ / 8;                        // 16-byte chunks per row (8 | 4)
    constexpr int RPI = 64 / CPR;                       // rows per LDS-DMA wave instruction (8 | 16)
    constexpr int AJ = TM / (8 * RPI);                  // A instructions per wave and stage (4 | 2)
    constexpr int BJ = BN_ / (8 * RPI);                 // B instructions per wave and stage
    static_assert(BJ >= 1, "BN_ x TK_ tile too small for eight loading waves");
    constexpr int PER = AJ + BJ;                        // LDS-DMA instructions a wave issues per stage
    constexpr int KS = TK_ / 16;                        // 16-deep k-steps per stage
    const int pM = args.M, pN = args.N, pK = args.K, p_lda = args.lda, p_ldb = args.ldb, p_ldc = args.ldc;
    const int p_taps = args.taps, p_gather = args.gather;
    const int p_tiles_n = args.tiles_n;
    const unsigned p_tap_mask = args.tap_mask;
    const long long p_tsb = args.tap_stride_b;
    const u16* __restrict__ p_zero = args.zeros;
    const int g_hs = args.g.hs, g_ws = args.g.ws, g_hd = args.g.hd, g_wd = args.g.wd, g_kw = args.g.kw;
    const int g_stride = args.g.stride, g_pad = args.g.pad, g_dil = args.g.dil;
    const int g_nimg = args.g.n_img, p_rect = GATHER ? args.rect : 0;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* s_mask = reinterpret_cast<unsigned*>(smem + Cfg::SMEM - 16);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = (wave >> 1) * 64, wn = (wave & 1) * WNC;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int tn = bid % p_tiles_n;
    int tm = bid / p_tiles_n;
    int pMe = pM;
    int r_y0 = 0, r_x0 = 0, r_h = g_hd, r_w = g_wd;
    unsigned mask = p_tap_mask;
    if (p_rect == 2) {                                  // region mode: tiles laid out region after region (region_of)
        bool found = false;
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            int y0, y1, x0, x1;
            unsigned rm;
            region_of(p_gather, r, g_dil, g_hd, g_wd, y0, y1, x0, x1, rm);
            const int mt = g_nimg * (y1 - y0) * (x1 - x0);
            const int tiles = (mt + TM - 1) / TM;
            if (!found) {
                if (tm < tiles) { found = true; mask = rm & p_tap_mask; pMe = mt; r_y0 = y0; r_x0 = x0; r_h = y1 - y0; r_w = x1 - x0; }
                else tm -= tiles;
            }
        }
        if (!found) return;
    }
    const int bz = blockIdx.z;
    const u16* __restrict__ A = args.A + (long long)bz * args.bsa;
    const u16* __restrict__ B = args.B + (long long)bz * args.bsb;

    // ---- staging state: this lane's rows and its (source-side) swizzled chunk -----------------------------------------
    const int lrow = lane / CPR, lch = lane % CPR;       // RPI rows x CPR chunks per wave instruction
    // chunk swizzle of LDS row r: TK_ = 64 (128-byte rows): (r >> 1) & 7;  TK_ = 32 (64-byte rows): (r >> 2) & 3 -- conflict-free
    // ds_read_b128 for the 32x32x16 operand map (row = lane & 31, chunk = 2 s + (lane >> 5)) in both cases
    auto swz = [](int r) __attribute__((always_inline)) { return CPR == 8 ? ((r >> 1) & 7) : ((r >> 2) & 3); };
    // A rows of this lane: wave * (AJ * RPI) + RPI j + lrow
    int a_rb[AJ], a_y[AJ], a_x[AJ];                      // gather: source pixel index / coordinates for tap offset (0, 0)
    long long a_off[AJ];                                 // plain: element offset of the row, -1 = overhang
    int a_sw[AJ];
    {
        int cn = 0, cy = 0, cx = 0;
        if (GATHER) {
            const int m0 = tm * TM + wave * (AJ * RPI) + lrow, hw = r_h * r_w;
            cn = m0 / hw;
            const int rem = m0 - cn * hw;
            cy = rem / r_w; cx = rem - cy * r_w;
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int m = tm * TM + wave * (AJ * RPI) + RPI * j + lrow;
            a_sw[j] = (lch ^ swz(wave * (AJ * RPI) + RPI * j + lrow)) * 8;
            if (GATHER) {
                if (m < pMe) { a_rb[j] = cn; a_y[j] = r_y0 + cy; a_x[j] = r_x0 + cx; }
                else { a_rb[j] = -1; a_y[j] = 0; a_x[j] = 0; }
                a_off[j] = -1;
                if (j < AJ - 1) {
                    cx += RPI;
                    while (cx >= r_w) { cx -= r_w; ++cy; }
                    while (cy >= r_h) { cy -= r_h; ++cn; }
                }
            } else {
                a_rb[j] = 0; a_y[j] = 0; a_x[j] = 0;
                a_off[j] = (m < pM) ? (long long)m * p_lda : -1;
            }
        }
    }
    // fast gather form: everything but the dgrad of a strided conv keeps (source index, y, x) for tap offset (0, 0)
    const bool fastg = GATHER && (p_gather == 1 || g_stride == 1);
    if (fastg) {
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            if (a_rb[j] >= 0) {
                const int y0 = (p_gather == 1) ? a_y[j] * g_stride - g_pad : a_y[j] + g_pad;
                const int x0 = (p_gather == 1) ? a_x[j] * g_stride - g_pad : a_x[j] + g_pad;
                a_rb[j] = (a_rb[j] * g_hs + y0) * g_ws + x0; a_y[j] = y0; a_x[j] = x0;
            } else { a_rb[j] = 0; a_y[j] = -(1 << 30); a_x[j] = 0; }
        }
    }
    auto tap_offsets = [&](int t, int& oy, int& ox) __attribute__((always_inline)) {
        int ky = 0, kx = __builtin_amdgcn_readfirstlane(t);
        while (kx >= g_kw) { kx -= g_kw; ++ky; }
        oy = (p_gather == 1 ? ky : -ky) * g_dil;
        ox = (p_gather == 1 ? kx : -kx) * g_dil;
    };
    // tap census: drop the taps that are padding for EVERY row of this tile (only possible when the tile covers fewer than
    // dil + 1 image rows)
    if (GATHER && p_taps > 1 && !p_rect && TM < g_wd * (g_dil + 1)) {
        if (tid == 0) *s_mask = 0u;
        __syncthreads();
        if (lch == 0) {
            unsigned local = 0;
            for (unsigned mm = mask; mm; mm &= mm - 1) {
                const int t = __ffs(mm) - 1;
                if (fastg) {
                    int oy, ox;
                    tap_offsets(t, oy, ox);
#pragma unroll
                    for (int j = 0; j < AJ; ++j)
                        if ((unsigned)(a_y[j] + oy) < (unsigned)g_hs && (unsigned)(a_x[j] + ox) < (unsigned)g_ws) local |= 1u << t;
                } else {
#pragma unroll
                    for (int j = 0; j < AJ; ++j)
                        if (a_rb[j] >= 0 && map_src(g_hs, g_ws, g_kw, g_stride, g_pad, g_dil, p_gather, a_rb[j], a_y[j], a_x[j], t) >= 0) local |= 1u << t;
                }
            }
            if (local) atomicOr(s_mask, local);
        }
        __syncthreads();
        mask &= *s_mask;
        __syncthreads();                                  // s_mask shares LDS with nothing else, but keep the read ahead of any reuse
    }

    const int nkc = pK / TK_;
    const int ntiles = __popc(mask) * nkc;
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = f32x16{0};

    unsigned rem_mask = mask;
    int tap = -1, kc = nkc;
    const u16* pa[AJ];
    const u16* pb[BJ];
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
                for (int j = 0; j < AJ; ++j) {
                    const bool ok = (unsigned)(a_y[j] + oy) < (unsigned)g_hs && (unsigned)(a_x[j] + ox) < (unsigned)g_ws;
                    pa[j] = (ok ? A + (long long)(a_rb[j] + d) * p_lda : p_zero) + a_sw[j];
                }
            } else {
#pragma unroll
                for (int j = 0; j < AJ; ++j) {
                    long long off;
                    if (GATHER) {
                        const int sr = (a_rb[j] >= 0) ? map_src(g_hs, g_ws, g_kw, g_stride, g_pad, g_dil, p_gather, a_rb[j], a_y[j], a_x[j], tap) : -1;
                        off = (sr >= 0) ? (long long)sr * p_lda : -1;
                    } else {
                        off = a_off[j];
                    }
                    pa[j] = (off >= 0 ? A + off : p_zero) + a_sw[j];
                }
            }
            const u16* Bt = B + (long long)tap * p_tsb;
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                // B rows of this lane: wave * (RPI BJ) + RPI j + lrow
                const int rb = wave * (RPI * BJ) + RPI * j + lrow;
                const int n = tn * BN_ + rb;
                pb[j] = (n < pN ? Bt + (long long)n * p_ldb : p_zero) + (lch ^ swz(rb)) * 8;
            }
        } else {
#pragma unroll
            for (int j = 0; j < AJ; ++j) pa[j] += TK_;
#pragma unroll
            for (int j = 0; j < BJ; ++j) pb[j] += TK_;
        }
    };
    auto issue = [&](int stage) __attribute__((always_inline)) {
        unsigned char* sa = smem + stage * Cfg::STAGE + (wave * AJ) * 1024;
        unsigned char* sb = smem + stage * Cfg::STAGE + A_STAGE + (wave * BJ) * 1024;
#pragma unroll
        for (int j = 0; j < AJ; ++j) glds16(pa[j], sa + j * 1024);
#pragma unroll
        for (int j = 0; j < BJ; ++j) glds16(pb[j], sb + j * 1024);
    };

    // fragment offsets of this lane: row = lane & 31, 16-byte chunk (2 s + (lane >> 5)) ^ swz(row), s = 0 .. KS-1
    // (wm, wn and the 32-row block offsets are multiples of 32 rows: they do not change swz)
    const int l31 = lane & 31, hh = lane >> 5;
    const int sw = swz(l31);
    int fo[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) fo[s] = l31 * ROWB + (((2 * s + hh) ^ sw) << 4);

    auto compute = [&](int stage) __attribute__((always_inline)) {
        const unsigned char* ab = smem + stage * Cfg::STAGE + wm * ROWB;
        const unsigned char* bb = smem + stage * Cfg::STAGE + A_STAGE + wn * ROWB;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            bf16x8 bf[NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bf[j] = *reinterpret_cast<const bf16x8*>(bb + j * 32 * ROWB + fo[s]);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const bf16x8 af = *reinterpret_cast<const bf16x8*>(ab + i * 32 * ROWB + fo[s]);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = GLF_MFMA_BF16(af, bf[j], acc[i][j]);
            }
        }
    };

    if (ntiles > 0) {
        advance(); issue(0);
        if (ntiles > 1) { advance(); issue(1); }
        int st = 0;                                     // stage of tile t
        int t = 0;
        for (; t + 2 < ntiles; ++t) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");      // tile t landed (this wave's part); tile t+1 may fly
            __builtin_amdgcn_s_barrier();                                    // ... everyone's part; and everyone is done reading tile t-1
            advance();
            issue(st == 0 ? 2 : st - 1);                                     // tile t+2 -> the stage tile t-1 used
            compute(st);
            st = st == 2 ? 0 : st + 1;
        }
        if (t + 1 < ntiles) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            __builtin_amdgcn_s_barrier();
            compute(st);
            st = st == 2 ? 0 : st + 1;
            ++t;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        compute(st);
    }
    __syncthreads();                                    // every wave is done with the stage buffers: they become parking space

    // ---- epilogue ------------------------------------------------------------------------------------------------------
    // A wave parks PARK_ROWS rows of its 64 x WNC tile in LDS (fp32, its own area) and writes them back out as whole 16-byte
    // pieces of rows; with TK_ = 32 (72 KiB of LDS) that takes two rounds of 32 rows.
    constexpr int PARK_ROWS = Cfg::PARK_ROWS, NH = 64 / PARK_ROWS, IB = PARK_ROWS / 32;
    float* park = reinterpret_cast<float*>(smem + wave * Cfg::PARK);
    const float p_alpha = args.alpha;
    const float* __restrict__ p_bias = args.bias;
    const bool stats = args.colstats != nullptr;
    float cs[NJ], cq[NJ], bv[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        cs[j] = 0.f; cq[j] = 0.f;
        const int col = tn * BN_ + wn + 32 * j + l31;
        bv[j] = (p_bias && col < pN) ? p_bias[col] : 0.f;
    }
    const int row_l = 4 * hh;
    unsigned char* __restrict__ Cb = reinterpret_cast<unsigned char*>(args.C);
    const int esz = args.c_bf16 ? 2 : 4;
    Cb += (long long)bz * args.bsc * esz;
    const bool wide = (p_ldc % 8) == 0 && (pN % 8) == 0 && (reinterpret_cast<size_t>(args.C) % 16) == 0 && (args.bsc % 8) == 0;
    auto out_row = [&](int row) __attribute__((always_inline)) -> long long {
        if (p_rect != 2) return row;
        const int hw = r_h * r_w;
        const int n = row / hw, rem = row - n * hw;
        const int yy = rem / r_w;
        return ((long long)n * g_hd + r_y0 + yy) * g_wd + r_x0 + (rem - yy * r_w);
    };
#pragma unroll
    for (int h = 0; h < NH; ++h) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
            for (int ii = 0; ii < IB; ++ii) {
                const int i = h * IB + ii;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl = 32 * ii + (r & 3) + 8 * (r >> 2) + row_l;
                    const float v = p_alpha * acc[i][j][r] + bv[j];
                    park[rl * PS + 32 * j + l31] = v;
                    if (stats) {
                        const bool ok = tm * TM + wm + h * PARK_ROWS + rl < pMe;
                        cs[j] += ok ? v : 0.f;
                        cq[j] += ok ? v * v : 0.f;
                    }
                }
            }
        }
        const int row_base = tm * TM + wm + h * PARK_ROWS;
        if (wide) {
            // bf16: 8 columns per lane (two ds_read_b128 -> one 16-byte store), WNC / 8 lanes per row; fp32: 4 columns per lane
            const int cpl = args.c_bf16 ? 8 : 4;
            const int lpr = WNC / cpl;                       // lanes per row
            const int rpi = 64 / lpr;                        // rows per iteration
            const int c0 = (lane % lpr) * cpl;
            const int col = tn * BN_ + wn + c0;
            int rl = lane / lpr;
            for (int it = 0; it < PARK_ROWS / rpi; ++it, rl += rpi) {
                const int row = row_base + rl;
                if (row < pMe && col < pN) {
                    const long long orow = out_row(row);
                    const float4 v0 = *reinterpret_cast<const float4*>(park + rl * PS + c0);
                    if (args.c_bf16) {
                        const float4 v1 = *reinterpret_cast<const float4*>(park + rl * PS + c0 + 4);
                        uint4 o;
                        u16* dst = reinterpret_cast<u16*>(Cb) + orow * p_ldc + col;
                        if (args.accumulate) {
                            const uint4 old = *reinterpret_cast<const uint4*>(dst);
                            o.x = pack_bf16(v0.x + __uint_as_float(old.x << 16), v0.y + __uint_as_float(old.x & 0xffff0000u));
                            o.y = pack_bf16(v0.z + __uint_as_float(old.y << 16), v0.w + __uint_as_float(old.y & 0xffff0000u));
                            o.z = pack_bf16(v1.x + __uint_as_float(old.z << 16), v1.y + __uint_as_float(old.z & 0xffff0000u));
                            o.w = pack_bf16(v1.z + __uint_as_float(old.w << 16), v1.w + __uint_as_float(old.w & 0xffff0000u));
                        } else {
                            o.x = pack_bf16(v0.x, v0.y); o.y = pack_bf16(v0.z, v0.w); o.z = pack_bf16(v1.x, v1.y); o.w = pack_bf16(v1.z, v1.w);
                        }
                        *reinterpret_cast<uint4*>(dst) = o;
                    } else {
                        float* dst = reinterpret_cast<float*>(Cb) + orow * p_ldc + col;
                        float4 o = v0;
                        if (args.accumulate) {
                            const float4 old = *reinterpret_cast<const float4*>(dst);
                            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
                        }
                        *reinterpret_cast<float4*>(dst) = o;
                    }
                }
            }
        } else {
            // narrow / unaligned outputs: one element per lane and store
            for (int rl = 0; rl < PARK_ROWS; ++rl) {
                const int row = row_base + rl;
                if (row >= pMe) break;
                const long long orow = out_row(row);
                for (int c = lane; c < WNC; c += 64) {
                    const int col = tn * BN_ + wn + c;
                    if (col >= pN) continue;
                    float v = park[rl * PS + c];
                    if (args.c_bf16) {
                        u16* dst = reinterpret_cast<u16*>(Cb) + orow * p_ldc + col;
                        if (args.accumulate) v += __uint_as_float((unsigned)*dst << 16);
                        *dst = (u16)(pack_bf16(v, 0.f) & 0xffffu);
                    } else {
                        float* dst = reinterpret_cast<float*>(Cb) + orow * p_ldc + col;
                        if (args.accumulate) v += *dst;
                        *dst = v;
                    }
                }
            }
        }
    }
    if (stats) {
        float* sst = reinterpret_cast<float*>(smem + Cfg::STATS_OFF);       // [4 wave rows][BN][2]
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            cs[j] += __shfl_xor(cs[j], 32, 64);
            cq[j] += __shfl_xor(cq[j], 32, 64);
            if (hh == 0) {
                float* d = sst + (((wave >> 1) * BN_) + wn + 32 * j + l31) * 2;
                d[0] = cs[j]; d[1] = cq[j];
            }
        }
        __syncthreads();
        if (tid < BN_) {
            const int col = tn * BN_ + tid;
            if (col < pN) {
                double s = 0, q = 0;
#pragma unroll
                for (int w = 0; w < 4; ++w) { s += sst[(w * BN_ + tid) * 2]; q += sst[(w * BN_ + tid) * 2 + 1]; }
                atomicAdd(args.colstats + col, s);
                atomicAdd(args.colstats + pN + col, q);
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------------
// tn kernel
// ----------------------------------------------------------------------------------------------------------------------
// output pixels y (of hd) whose source pixel y * stride + o lies inside [0, hs): the band [lo, hi)
__host__ __device__ inline void tap_band(int o, int stride, int hs, int hd, int& lo, int& hi) {
    lo = o < 0 ? (-o + stride - 1) / stride : 0;
    const int top = hs - 1 - o;
    hi = top < 0 ? 0 : (top / stride + 1 < hd ? top / stride + 1 : hd);
    if (lo > hi) lo = hi;
}
// reduction rows of tap `tap` of a gathered TN contraction in rectangle mode (rect = 1): only the output pixels whose source pixel
// is inside the map, enumerated image by image, row by row: n_img * (yhi - ylo) * (xhi - xlo)
__host__ __device__ inline int tn_rect_rows(const Geo& g, int tap, int K) {
    const int ky = tap / g.kw, kx = tap - ky * g.kw;
    int ylo, yhi, xlo, xhi;
    tap_band(ky * g.dil - g.pad, g.stride, g.hs, g.hd, ylo, yhi);
    tap_band(kx * g.dil - g.pad, g.stride, g.ws, g.wd, xlo, xhi);
    return (K / (g.hd * g.wd)) * (yhi - ylo) * (xhi - xlo);
}
// rectangle mode keeps only the slabs of slices that run, tap after tap: first slab of `tap` (its number of slices in nvalid);
// tap = -1: the total number of slabs
__host__ __device__ inline int tn_rect_slab(const Geo& g, unsigned tap_mask, int tap, int K, int split, int& nvalid) {
    int chunk = (K + split - 1) / split;
    chunk = ((chunk + TK - 1) / TK) * TK;
    int prefix = 0;
    nvalid = 0;
    for (unsigned mm = tap_mask; mm; mm &= mm - 1) {
        const int t = __builtin_ctz(mm);
        const int rows = tn_rect_rows(g, t, K);
        int nv = (rows + chunk - 1) / chunk;
        if (nv > split) nv = split;
        if (t == tap) { nvalid = nv; return prefix; }
        prefix += nv;
    }
    return prefix;
}
// r / d for 0 <= r < 2^24, d >= 1, inv = 1.f / d: one multiply, one conversion and two corrections instead of an integer division
__device__ __forceinline__ int fdiv(int r, int d, float inv) {
    int q = (int)((float)r * inv);
    if (q * d > r) --q;
    if ((q + 1) * d <= r) ++q;
    return q;
}

constexpr int TN_BN = 128;
constexpr int TNA_STAGE = TK * TM * 2;           // [64 r][256 m]: 32 KiB
constexpr int TNB_STAGE = TK * TN_BN * 2;        // [64 r][128 n]: 16 KiB
constexpr int TN_STAGE = TNA_STAGE + TNB_STAGE;
constexpr size_t SMEM_TN16 = (size_t)NSTAGE * TN_STAGE + 16;

template <bool GATHER>
__global__ __launch_bounds__(NTH, 2) void s16_tn_kernel(const S16Args args) {
    constexpr int AJ = 4;                               // A: 2 rows of 512 bytes per instruction, 32 per stage
    constexpr int BJ = 2;                               // B: 4 rows of 256 bytes per instruction, 16 per stage
    constexpr int PER = AJ + BJ;
    const int pM = args.M, pN = args.N, pK = args.K, p_lda = args.lda, p_ldb = args.ldb, p_ldc = args.ldc;
    const int p_split = args.split, p_tiles_n = args.tiles_n;
    const unsigned p_tap_mask = args.tap_mask;
    const long long p_tsb = args.tap_stride_b;
    const u16* __restrict__ p_zero = args.zeros;
    const int g_hs = args.g.hs, g_ws = args.g.ws, g_hd = args.g.hd, g_wd = args.g.wd, g_kw = args.g.kw;
    const int g_stride = args.g.stride, g_pad = args.g.pad, g_dil = args.g.dil;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned* s_any = reinterpret_cast<unsigned*>(smem + SMEM_TN16 - 16);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const u16* __restrict__ A = args.A + (long long)bz * args.bsa;
    const u16* __restrict__ B = args.B + (long long)bz * args.bsb;

    // rectangle mode (gathered taps that fall mostly into the padding: the ASPP rates): the reduction of a tap runs over ITS in-range
    // output pixels only, enumerated compactly; a slice is the same number of rows for every tap, so a tap with a small rectangle uses
    // fewer slices (the others return here) and all workgroups reduce equally long
    const bool RECT = GATHER && args.rect != 0;
    int q_ylo = 0, q_xlo = 0, q_h = g_hd, q_w = g_wd;
    if (RECT) {
        const int ky = tap / g_kw, kx = tap - ky * g_kw;
        int yhi_, xhi_;
        tap_band(ky * g_dil - g_pad, g_stride, g_hs, g_hd, q_ylo, yhi_);
        tap_band(kx * g_dil - g_pad, g_stride, g_ws, g_wd, q_xlo, xhi_);
        q_h = yhi_ - q_ylo; q_w = xhi_ - q_xlo;
    }
    const int q_hw = q_h * q_w;
    const int k_rows = RECT ? (pK / (g_hd * g_wd)) * q_hw : pK;
    const float q_inv_hw = 1.0f / (float)(q_hw > 0 ? q_hw : 1), q_inv_w = 1.0f / (float)(q_w > 0 ? q_w : 1);
    int chunk = (pK + p_split - 1) / p_split;
    chunk = ((chunk + TK - 1) / TK) * TK;
    const int r0 = sl * chunk;
    const int r1 = min(k_rows, r0 + chunk);
    if (r0 >= r1) return;

    // staging: A instruction j of this wave holds reduction rows wave * 8 + 2 j + (lane >> 5), 16-byte piece lane & 31 of the
    // 512-byte row; B instruction j rows wave * 8 + 4 j + (lane >> 4), piece lane & 15.  Source-side swizzle: the 64-byte chunk
    // (piece >> 2) is XORed with (row & 3).
    const int a_rl = lane >> 5, a_pc = lane & 31;
    const int b_rl = lane >> 4, b_pc = lane & 15;
    int a_col[AJ], b_col[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int rr = wave * 8 + 2 * j + a_rl;
        const int m = tm * TM + ((((a_pc >> 2) ^ (rr & 3)) << 2) | (a_pc & 3)) * 8;
        a_col[j] = m < pM ? m : -1;                     // whole 8-column pieces: M % 8 == 0
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
        const int rr = wave * 8 + 4 * j + b_rl;
        const int n = tn * TN_BN + ((((b_pc >> 2) ^ (rr & 3)) << 2) | (b_pc & 3)) * 8;
        b_col[j] = n < pN ? n : -1;
    }
    const int hw = GATHER ? g_hd * g_wd : 1;
    int oy = 0, ox = 0;
    if (GATHER) {
        const int ky = tap / g_kw, kx = tap - ky * g_kw;
        oy = ky * g_dil - g_pad; ox = kx * g_dil - g_pad;
    }
    // per-lane element offsets of its rows relative to the tile's first row (the tile base is wave-uniform: scalar arithmetic)
    long long a_base[AJ], b_base[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j) a_base[j] = (long long)(wave * 8 + 2 * j + a_rl) * p_lda + (a_col[j] >= 0 ? a_col[j] : 0);
#pragma unroll
    for (int j = 0; j < BJ; ++j) b_base[j] = (long long)(wave * 8 + 4 * j + b_rl) * p_ldb + (b_col[j] >= 0 ? b_col[j] : 0);
    auto issue = [&](int stage, int rbase) __attribute__((always_inline)) {
        unsigned char* sa = smem + stage * TN_STAGE + (wave * AJ) * 1024;
        unsigned char* sb = smem + stage * TN_STAGE + TNA_STAGE + (wave * BJ) * 1024;
        const u16* At = A + (long long)rbase * p_lda;
        if (RECT) {
#pragma unroll
            for (int j = 0; j < AJ; ++j) {
                const int r = rbase + wave * 8 + 2 * j + a_rl;
                const u16* src = p_zero;
                if (r < r1 && a_col[j] >= 0) {
                    const int n = fdiv(r, q_hw, q_inv_hw), rem = r - n * q_hw;
                    const int yy = fdiv(rem, q_w, q_inv_w), xx = rem - yy * q_w;
                    src = A + ((long long)(n * g_hd + q_ylo + yy) * g_wd + q_xlo + xx) * p_lda + a_col[j];
                }
                glds16(src, sa + j * 1024);
            }
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int r = rbase + wave * 8 + 4 * j + b_rl;
                const u16* src = p_zero;
                if (r < r1 && b_col[j] >= 0) {
                    const int n = fdiv(r, q_hw, q_inv_hw), rem = r - n * q_hw;
                    const int yy = fdiv(rem, q_w, q_inv_w), xx = rem - yy * q_w;
                    const int sy = (q_ylo + yy) * g_stride + oy, sx = (q_xlo + xx) * g_stride + ox;       // in range by construction
                    src = B + ((long long)(n * g_hs + sy) * g_ws + sx) * p_ldb + b_col[j];
                }
                glds16(src, sb + j * 1024);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < AJ; ++j) {
            const int r = rbase + wave * 8 + 2 * j + a_rl;
            glds16((r < r1 && a_col[j] >= 0) ? At + a_base[j] : p_zero, sa + j * 1024);
        }
        if (!GATHER) {
            const u16* Bt = B + (long long)rbase * p_ldb;
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int r = rbase + wave * 8 + 4 * j + b_rl;
                glds16((r < r1 && b_col[j] >= 0) ? Bt + b_base[j] : p_zero, sb + j * 1024);
            }
        } else {
#pragma unroll
            for (int j = 0; j < BJ; ++j) {
                const int r = rbase + wave * 8 + 4 * j + b_rl;
                long long src = -1;
                if (r < r1 && b_col[j] >= 0) {
                    const int n = r / hw, rem = r - n * hw;
                    const int y = rem / g_wd, x = rem - y * g_wd;
                    const int sy = y * g_stride + oy, sx = x * g_stride + ox;
                    if ((unsigned)sy < (unsigned)g_hs && (unsigned)sx < (unsigned)g_ws) src = ((long long)n * g_hs + sy) * g_ws + sx;
                }
                glds16(src >= 0 ? B + src * p_ldb + b_col[j] : p_zero, sb + j * 1024);
            }
        }
    };

    // transposing fragment read (ds_read_b64_tr_b16): lanes 16 g .. 16 g + 15 fetch a block of 4 reduction rows x 16 columns;
    // lane 4 q + p of the group addresses row q, columns 4 p .. 4 p + 3 and receives column (lane & 15), rows 0..3.
    // Group g: columns 16 (g & 1) .., reduction half g >> 1 (rows 8 (g >> 1) + q, and + 4 for the second read).
    // Offsets: a k-step (16 rows) and the second read (4 rows) further on keep row & 3, hence the chunk swizzle: ONE base per
    // fragment and lane, everything else is an immediate.
    const int grp = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int trow = 8 * (grp >> 1) + q;                 // row inside a 16-deep k-step (second read: + 4)
    const int tcolb = (16 * (grp & 1) + 4 * pp) * 2;     // byte offset of the 4 columns inside a 32-column block
#define GLF_TR_OFF(rr, cb, ROWB) ((rr) * (ROWB) + ((((cb) >> 6) ^ ((rr) & 3)) << 6) + ((cb) & 63))
    int fa[2], fb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        fa[i] = GLF_TR_OFF(trow, (wm + 32 * i) * 2 + tcolb, 512);
        fb[i] = TNA_STAGE + GLF_TR_OFF(trow, (wn + 32 * i) * 2 + tcolb, 256);
    }
    // The transposing reads are issued as inline asm: through the builtin, hipcc orders every ds_read_b64_tr_b16 behind ALL
    // outstanding LDS-DMA writes (s_waitcnt vmcnt(0) in front of the first read of each K-tile -- the loads of tile t+2, issued a
    // moment earlier, were waited for on the spot and the three-stage pipeline ran as a serial load -> compute loop: 60 % of the
    // wave time in waits, 580 TF where the rows kernel reaches 880).  What orders a read behind the DMA that filled ITS stage is the
    // counted vmcnt + barrier at the top of the iteration.
    typedef int v2i_ __attribute__((ext_vector_type(2)));
#define GLF_TR_READ(dst, addr, OFF) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF))
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x16{0};

    auto compute = [&](int stage) __attribute__((always_inline)) {
        const unsigned sbase = lds0 + stage * TN_STAGE;
        const unsigned aa0 = sbase + fa[0], aa1 = sbase + fa[1], ab0 = sbase + fb[0], ab1 = sbase + fb[1];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            v2i_ r[8];
            GLF_TR_READ(r[0], ab0, s * 16 * 256);          GLF_TR_READ(r[1], ab0, s * 16 * 256 + 4 * 256);
            GLF_TR_READ(r[2], ab1, s * 16 * 256);          GLF_TR_READ(r[3], ab1, s * 16 * 256 + 4 * 256);
            GLF_TR_READ(r[4], aa0, s * 16 * 512);          GLF_TR_READ(r[5], aa0, s * 16 * 512 + 4 * 512);
            GLF_TR_READ(r[6], aa1, s * 16 * 512);          GLF_TR_READ(r[7], aa1, s * 16 * 512 + 4 * 512);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            typedef int v4i_ __attribute__((ext_vector_type(4)));
            bf16x8 bf[2], af[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                bf[k] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(r[2 * k], r[2 * k + 1], 0, 1, 2, 3));
                af[k] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(r[4 + 2 * k], r[5 + 2 * k], 0, 1, 2, 3));
            }
            (void)sizeof(v4i_);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = GLF_MFMA_BF16(af[i], bf[j], acc[i][j]);
        }
    };
    (void)s_any;

    // The tile sequence of this slice.  A gathered tap reads only output rows y whose source row y * stride + oy lies inside the
    // map: [ylo, yhi) per image.  For the ASPP rates on a 28 x 28 map that band is 4 (rate 24) or 16 (rate 12) of 28 rows for six
    // of the nine taps -- K-tiles outside it would multiply zero-page rows, so the cursor jumps over them (wave-uniform scalar
    // arithmetic, two divisions per tile).
    int ylo = 0, yhi = g_hd;
    if (GATHER) {
        ylo = oy < 0 ? (-oy + g_stride - 1) / g_stride : 0;
        const int ymax = (g_hs - 1 - oy) / g_stride;           // largest y with y * stride + oy <= hs - 1 (oy <= hs - 1 for a kept tap)
        yhi = (g_hs - 1 - oy) < 0 ? 0 : (ymax + 1 < g_hd ? ymax + 1 : g_hd);
        if (ylo > yhi) ylo = yhi;
    }
    const bool banded = GATHER && !RECT && (ylo > 0 || yhi < g_hd);
    auto next_valid = [&](int r) __attribute__((always_inline)) -> int {
        if (!banded || r >= r1) return r;
        if (ylo >= yhi) return r1;
        const int n = r / hw, rem = r - n * hw;
        const int y = rem / g_wd;
        if (y < ylo) return n * hw + ylo * g_wd;
        if (y >= yhi) return (n + 1) * hw + ylo * g_wd;
        return r;
    };
    int cursor = __builtin_amdgcn_readfirstlane(next_valid(r0));
    auto take = [&]() __attribute__((always_inline)) -> int {
        const int r = cursor;
        cursor = r < r1 ? __builtin_amdgcn_readfirstlane(next_valid(r + TK)) : r1;
        return r;
    };
    int t0 = take();
    if (t0 < r1) {
        issue(0, t0);
        int t1 = take();
        bool has_next = t1 < r1;
        if (has_next) issue(1, t1);
        int st = 0;
        for (;;) {
            if (has_next) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            bool has_next2 = false;
            if (has_next) {
                const int t2 = take();
                has_next2 = t2 < r1;
                if (has_next2) issue(st == 0 ? 2 : st - 1, t2);
            }
            compute(st);
            if (!has_next) break;
            has_next = has_next2;
            st = st == 2 ? 0 : st + 1;
        }
    }

    // ---- epilogue: direct stores (32 consecutive columns per half wave) ----------------------------------------------------
    const float p_alpha = args.alpha;
    const int l31 = lane & 31, row_l = 4 * (lane >> 5);
    float* __restrict__ Cf;
    u16* __restrict__ Ch = nullptr;
    int ldc_e;
    if (args.partial) {
        if (RECT) {
            int nv;
            Cf = args.partial + (long long)(tn_rect_slab(args.g, p_tap_mask, tap, pK, p_split, nv) + sl) * ((long long)pM * pN);
        } else {
            Cf = args.partial + ((long long)blockIdx.z * gridDim.y + blockIdx.y) * ((long long)pM * pN);
        }
        ldc_e = pN;
    } else {
        Cf = reinterpret_cast<float*>(args.C) + (long long)bz * args.bsc + (long long)tap * p_tsb;
        Ch = reinterpret_cast<u16*>(args.C) + (long long)bz * args.bsc + (long long)tap * p_tsb;
        ldc_e = p_ldc;
    }
    const bool out16 = args.c_bf16 && !args.partial;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = tn * TN_BN + wn + 32 * j + l31;
            if (col < pN) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = tm * TM + wm + 32 * i + (r & 3) + 8 * (r >> 2) + row_l;
                    if (row < pM) {
                        const float v = p_alpha * acc[i][j][r];
                        if (out16) Ch[(long long)row * ldc_e + col] = (u16)(pack_bf16(v, 0.f) & 0xffffu);
                        else Cf[(long long)row * ldc_e + col] = v;
                    }
                }
            }
        }
}

// second stage of the split-K reduction: C_tap = sum over the slices of their partial slabs, in a fixed order.  SL slice lanes share
// an output (lane l adds slices l, l + SL, ...; the lanes meet in LDS, lane order): with one thread per output a 378-slice layer-1
// gradient was 378 dependent-latency loads on 4 096 threads -- 100 us for 24 MB.
template <int SL>
__global__ __launch_bounds__(256) void s16_tn_reduce_kernel(const float* __restrict__ partial, void* __restrict__ Cv, int c_bf16, int M, int N, int ldc,
                                                            long long tsb, long long bsc, int split, unsigned tap_mask, int K, int rect, Geo g) {
    constexpr int OG = 256 / SL;                                      // outputs (float4) per block
    __shared__ float4 sh[SL > 1 ? 256 : 1];
    unsigned mm = tap_mask;
    for (int i = 0; i < (int)blockIdx.y; ++i) mm &= mm - 1;
    const int tap = __ffs(mm) - 1;
    const int ntap = gridDim.y, bz = blockIdx.z;
    int chunk = (K + split - 1) / split;
    chunk = ((chunk + TK - 1) / TK) * TK;
    const long long mn = (long long)M * N;
    int nvalid = min(split, (K + chunk - 1) / chunk);                 // slices that ran (the others returned early)
    const float* __restrict__ src = partial + ((long long)bz * split * ntap + blockIdx.y) * mn;
    long long slice_stride = (long long)ntap * mn;
    if (rect) {                                                       // compact slabs, tap after tap (batch 1)
        src = partial + (long long)tn_rect_slab(g, tap_mask, tap, K, split, nvalid) * mn;
        slice_stride = mn;
    }
    const int n4 = N >> 2;
    const long long total = (long long)M * n4;
    const int ol = threadIdx.x % OG, sl = threadIdx.x / OG;
    for (long long base = (long long)blockIdx.x * OG; base < total; base += (long long)gridDim.x * OG) {
        const long long i = base + ol;
        const bool live = i < total;
        const long long row = live ? i / n4 : 0;
        const int c4 = (int)((live ? i : 0) - row * n4) * 4;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        if (live) {
            const float* sp = src + row * N + c4;
            int sidx = sl;
            for (; sidx + 3 * SL < nvalid; sidx += 4 * SL) {
                const float4 v0 = *reinterpret_cast<const float4*>(sp + (long long)sidx * slice_stride);
                const float4 v1 = *reinterpret_cast<const float4*>(sp + (long long)(sidx + SL) * slice_stride);
                const float4 v2 = *reinterpret_cast<const float4*>(sp + (long long)(sidx + 2 * SL) * slice_stride);
                const float4 v3 = *reinterpret_cast<const float4*>(sp + (long long)(sidx + 3 * SL) * slice_stride);
                a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
                a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
            }
            for (; sidx < nvalid; sidx += SL) {
                const float4 v = *reinterpret_cast<const float4*>(sp + (long long)sidx * slice_stride);
                a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
            }
        }
        if (SL > 1) {
            sh[threadIdx.x] = a;
            __syncthreads();
            if (sl == 0) {
#pragma unroll
                for (int l = 1; l < SL; ++l) { const float4 v = sh[l * OG + ol]; a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w; }
            }
            __syncthreads();
        }
        if (sl == 0 && live) {
            const long long o = (long long)bz * bsc + (long long)tap * tsb + row * ldc + c4;
            if (c_bf16) *reinterpret_cast<uint2*>(reinterpret_cast<u16*>(Cv) + o) = make_uint2(pack_bf16(a.x, a.y), pack_bf16(a.z, a.w));
            else *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + o) = a;
        }
    }
}

int validate16(const glf_gemm_params* p, const void* A, const void* B, const void* C, const char* who) {
    GLF_REQUIRE(p && A && B && C, GLF_ERR_NULL, "%s: null argument", who);
    GLF_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, GLF_ERR_BAD_SHAPE, "%s: M,N,K must be > 0 (got %d,%d,%d)", who, p->M, p->N, p->K);
    GLF_REQUIRE(p->taps >= 1 && p->taps <= 32, GLF_ERR_BAD_SHAPE, "%s: taps must be in [1,32] (got %d)", who, p->taps);
    GLF_REQUIRE(p->batch >= 1 && p->batch <= 65535, GLF_ERR_BAD_SHAPE, "%s: batch out of range (%d)", who, p->batch);
    GLF_REQUIRE(p->gather >= 0 && p->gather <= 2, GLF_ERR_BAD_SHAPE, "%s: gather must be 0,1,2", who);
    GLF_REQUIRE(p->c_dtype == GLF_DT_F32 || p->c_dtype == GLF_DT_BF16, GLF_ERR_UNSUPPORTED, "%s: c_dtype must be GLF_DT_F32 or GLF_DT_BF16", who);
    const unsigned full = p->taps == 32 ? 0xffffffffu : ((1u << p->taps) - 1u);
    GLF_REQUIRE((p->tap_mask & ~full) == 0, GLF_ERR_BAD_SHAPE, "%s: tap_mask has bits beyond taps", who);
    if (p->gather) {
        GLF_REQUIRE(p->kh * p->kw == p->taps, GLF_ERR_BAD_SHAPE, "%s: kh*kw != taps", who);
        GLF_REQUIRE(p->stride >= 1 && p->dil >= 1 && p->hs > 0 && p->ws > 0 && p->hd > 0 && p->wd > 0 && p->n_img > 0,
                    GLF_ERR_BAD_SHAPE, "%s: bad conv geometry", who);
    } else {
        GLF_REQUIRE(p->taps == 1, GLF_ERR_BAD_SHAPE, "%s: taps > 1 needs a gather mapping", who);
    }
    GLF_REQUIRE(aligned16(A) && aligned16(B), GLF_ERR_BAD_SHAPE, "%s: A and B must be 16-byte aligned", who);
    GLF_REQUIRE(p->lda % 8 == 0 && p->ldb % 8 == 0 && p->batch_stride_a % 8 == 0 && p->batch_stride_b % 8 == 0 && p->tap_stride_b % 8 == 0,
                GLF_ERR_BAD_SHAPE, "%s: row / batch / tap strides of the 16-bit operands must be multiples of 8 elements", who);
    return GLF_OK;
}

S16Args make_args16(const void* A, const void* B, const float* bias, void* C, const glf_gemm_params* p) {
    S16Args a;
    a.A = static_cast<const u16*>(A); a.B = static_cast<const u16*>(B); a.bias = bias; a.C = C;
    a.c_bf16 = p->c_dtype == GLF_DT_BF16;
    a.M = p->M; a.N = p->N; a.K = p->K; a.lda = p->lda; a.ldb = p->ldb; a.ldc = p->ldc; a.taps = p->taps;
    a.tap_mask = p->tap_mask; a.tap_stride_b = p->tap_stride_b; a.gather = p->gather;
    a.g = Geo{p->n_img, p->hs, p->ws, p->hd, p->wd, p->kh, p->kw, p->stride, p->pad, p->dil};
    a.bsa = p->batch_stride_a; a.bsb = p->batch_stride_b; a.bsc = p->batch_stride_c;
    a.alpha = p->alpha; a.split = p->split < 1 ? 1 : p->split; a.tiles_n = 1; a.rect = 0;
    a.zeros = reinterpret_cast<const u16*>(glf::zero_page());
    a.colstats = p->colstats; a.partial = nullptr; a.accumulate = p->accumulate;
    return a;
}

}  // namespace

namespace glf {
int init_gemm_s16_attrs() {
    hipError_t e;
#define SET_ATTR(fn, bytes)                                                                              \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); \
    if (e != hipSuccess) return fail(GLF_ERR_LAUNCH, "hipFuncSetAttribute(" #fn "): %s", hipGetErrorString(e));
    SET_ATTR((s16_rows_kernel<false, 128, 64>), (RowsCfg<128, 64>::SMEM))
    SET_ATTR((s16_rows_kernel<true, 128, 64>), (RowsCfg<128, 64>::SMEM))
    SET_ATTR((s16_rows_kernel<false, 128, 32>), (RowsCfg<128, 32>::SMEM))
    SET_ATTR((s16_rows_kernel<true, 128, 32>), (RowsCfg<128, 32>::SMEM))
    SET_ATTR((s16_rows_kernel<false, 64, 64>), (RowsCfg<64, 64>::SMEM))
    SET_ATTR((s16_rows_kernel<true, 64, 64>), (RowsCfg<64, 64>::SMEM))
    SET_ATTR((s16_tn_kernel<false>), SMEM_TN16)
    SET_ATTR((s16_tn_kernel<true>), SMEM_TN16)
#undef SET_ATTR
    return GLF_OK;
}
}  // namespace glf

extern "C" int glf_s16_gemm_nt(const void* A, const void* B, const float* bias, void* C, const glf_gemm_params* p, glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    if (int rc = validate16(p, A, B, C, "s16_gemm_nt")) return rc;
    GLF_REQUIRE(p->K % TK == 0 && p->K <= (1 << 18), GLF_ERR_UNSUPPORTED, "s16_gemm_nt: K must be a multiple of %d (got %d)", TK, p->K);
    if (p->gather) GLF_REQUIRE((long long)p->n_img * p->hd * p->wd == p->M, GLF_ERR_BAD_SHAPE, "s16_gemm_nt: M (%d) != n_img*hd*wd", p->M);
    if (p->tap_mask == 0) return glf::fail(GLF_ERR_BAD_SHAPE, "s16_gemm_nt: empty tap_mask");
    GLF_REQUIRE(p->rect == 0 || p->rect == 2, GLF_ERR_UNSUPPORTED, "s16_gemm_nt: rect must be 0 or 2 (region mode); per-tap rectangles need float atomics");
    GLF_REQUIRE(!p->colstats || p->batch == 1, GLF_ERR_UNSUPPORTED, "s16_gemm_nt: colstats needs batch 1");
    S16Args a = make_args16(A, B, bias, C, p);
    // (halving the column tile of the region-mode ASPP forward launches, whose 392 workgroups leave slots empty while the longest tap
    // chains run, was measured: +1.4 ms per step -- profiles/r04_ab_s16.txt)
    const int bn = p->N <= 64 ? 64 : 128;
    a.tiles_n = (p->N + bn - 1) / bn;
    long long tiles_m = (p->M + TM - 1) / TM;
    if (p->rect == 2) {
        GLF_REQUIRE(p->gather != 0 && p->kh == 3 && p->kw == 3 && p->stride == 1 && p->pad == p->dil && p->hs == p->hd && p->ws == p->wd && p->batch == 1,
                    GLF_ERR_UNSUPPORTED, "s16_gemm_nt: region mode needs a 3x3 stride-1 conv with pad == dil on equal maps, batch 1");
        a.rect = 2;
        tiles_m = 0;
        for (int r = 0; r < 9; ++r) {
            int y0, y1, x0, x1;
            unsigned rm;
            region_of(p->gather, r, p->dil, p->hd, p->wd, y0, y1, x0, x1, rm);
            tiles_m += ((long long)p->n_img * (y1 - y0) * (x1 - x0) + TM - 1) / TM;
        }
    }
    GLF_REQUIRE(tiles_m * a.tiles_n < 2147483647LL, GLF_ERR_BAD_SHAPE, "s16_gemm_nt: grid out of range");
    dim3 grid((unsigned)(tiles_m * a.tiles_n), 1, p->batch);
    const bool gather = p->gather != 0;
    // stage depth: 32 = two workgroups per CU (the default), 64 = one per CU with half the barriers; GLF_S16_TK=32|64 forces one
    static const int tk_force = []() { const char* e = getenv("GLF_S16_TK"); return e ? atoi(e) : 0; }();
    // measured (profiles/r04_s16_tk_ab.txt): two workgroups per CU win on every shape of the model (K = 256: +29 %, the 3x3 convs
    // +13-18 %, the M = 150 528 projections +3 %) except the region-mode launches with many short region blocks (rate 12: -10 %)
    const bool tk32 = bn == 128 && (tk_force ? tk_force == 32 : p->rect != 2);
    if (bn == 64) {
        if (gather) hipLaunchKernelGGL((s16_rows_kernel<true, 64, 64>), grid, dim3(NTH), (RowsCfg<64, 64>::SMEM), glf::S(stream), a);
        else hipLaunchKernelGGL((s16_rows_kernel<false, 64, 64>), grid, dim3(NTH), (RowsCfg<64, 64>::SMEM), glf::S(stream), a);
    } else if (tk32) {
        if (gather) hipLaunchKernelGGL((s16_rows_kernel<true, 128, 32>), grid, dim3(NTH), (RowsCfg<128, 32>::SMEM), glf::S(stream), a);
        else hipLaunchKernelGGL((s16_rows_kernel<false, 128, 32>), grid, dim3(NTH), (RowsCfg<128, 32>::SMEM), glf::S(stream), a);
    } else {
        if (gather) hipLaunchKernelGGL((s16_rows_kernel<true, 128, 64>), grid, dim3(NTH), (RowsCfg<128, 64>::SMEM), glf::S(stream), a);
        else hipLaunchKernelGGL((s16_rows_kernel<false, 128, 64>), grid, dim3(NTH), (RowsCfg<128, 64>::SMEM), glf::S(stream), a);
    }
    return glf::check_launch("s16_gemm_nt");
}

extern "C" size_t glf_s16_gemm_tn_workspace_bytes(const glf_gemm_params* p) {
    if (!p || p->split <= 1) return 0;
    const size_t ntap = (size_t)__builtin_popcount(p->tap_mask);
    if (p->rect == 1 && p->gather == 1 && p->batch == 1 && p->hd > 0 && p->wd > 0 && p->kw > 0 && p->stride > 0) {
        const Geo g{p->n_img, p->hs, p->ws, p->hd, p->wd, p->kh, p->kw, p->stride, p->pad, p->dil};
        int nv;
        return (size_t)tn_rect_slab(g, p->tap_mask, -1, p->K, p->split, nv) * (size_t)p->M * (size_t)p->N * sizeof(float);
    }
    return (size_t)p->batch * (size_t)p->split * ntap * (size_t)p->M * (size_t)p->N * sizeof(float);
}

extern "C" int glf_s16_gemm_tn(const void* A, const void* B, void* C, const glf_gemm_params* p, glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    if (int rc = validate16(p, A, B, C, "s16_gemm_tn")) return rc;
    GLF_REQUIRE(p->gather != 2, GLF_ERR_UNSUPPORTED, "s16_gemm_tn: transposed gather is not defined for the reduction form");
    GLF_REQUIRE(!p->colstats && !p->accumulate, GLF_ERR_UNSUPPORTED, "s16_gemm_tn: colstats / accumulate are not built");
    GLF_REQUIRE(p->rect == 0 || (p->rect == 1 && p->gather == 1 && p->split > 1 && p->batch == 1), GLF_ERR_UNSUPPORTED,
                "s16_gemm_tn: rect = 1 (per-tap rectangles) needs a forward gather, batch 1 and split > 1 (only the slabs of slices that run are kept)");
    GLF_REQUIRE(p->M % 8 == 0 && p->N % 8 == 0, GLF_ERR_UNSUPPORTED, "s16_gemm_tn: M and N must be multiples of 8 (got %d, %d)", p->M, p->N);
    if (p->gather) GLF_REQUIRE((long long)p->n_img * p->hd * p->wd == p->K, GLF_ERR_BAD_SHAPE, "s16_gemm_tn: K (%d rows) != n_img*hd*wd", p->K);
    S16Args a = make_args16(A, B, nullptr, C, p);
    const int ntap = __builtin_popcount(p->tap_mask);
    if (ntap == 0) return GLF_OK;
    a.rect = p->rect;
    GLF_REQUIRE((long long)p->batch * a.split <= 65535, GLF_ERR_BAD_SHAPE, "s16_gemm_tn: batch*split too large");
    a.tiles_n = (p->N + TN_BN - 1) / TN_BN;
    const int tiles_m = (p->M + TM - 1) / TM;
    const bool two_stage = a.split > 1;
    if (two_stage) {
        GLF_REQUIRE(p->workspace && p->workspace_bytes >= (int64_t)glf_s16_gemm_tn_workspace_bytes(p), GLF_ERR_WORKSPACE,
                    "s16_gemm_tn: split > 1 needs a workspace of glf_s16_gemm_tn_workspace_bytes() = %zu bytes", glf_s16_gemm_tn_workspace_bytes(p));
        GLF_REQUIRE(aligned16(p->workspace) && p->N % 4 == 0 && p->ldc % 4 == 0 && p->tap_stride_b % 4 == 0 && p->batch_stride_c % 4 == 0,
                    GLF_ERR_WORKSPACE, "s16_gemm_tn: split > 1 needs a 16-byte aligned workspace and 4-element aligned C strides");
        a.partial = p->workspace;
    }
    dim3 grid(tiles_m * a.tiles_n, ntap, p->batch * a.split);
    if (p->gather) hipLaunchKernelGGL((s16_tn_kernel<true>), grid, dim3(NTH), SMEM_TN16, glf::S(stream), a);
    else hipLaunchKernelGGL((s16_tn_kernel<false>), grid, dim3(NTH), SMEM_TN16, glf::S(stream), a);
    if (int rc = glf::check_launch("s16_gemm_tn")) return rc;
    if (!two_stage) return GLF_OK;
    const long long work = (long long)p->M * (p->N / 4);
    // slice lanes per output: enough threads to keep the slabs' loads in flight (work x lanes >= ~64 k threads), at most 16
    int lanes = 1;
    while (lanes < 16 && lanes * 4 <= a.split && work * lanes < 65536) lanes *= 4;
#define GLF_S16_REDUCE(SL_)                                                                                                          \
    {                                                                                                                                \
        long long bx = (work + (256 / SL_) - 1) / (256 / SL_);                                                                       \
        if (bx > 4096) bx = 4096;                                                                                                    \
        hipLaunchKernelGGL((s16_tn_reduce_kernel<SL_>), dim3((unsigned)bx, ntap, p->batch), dim3(256), 0, glf::S(stream), a.partial, C, a.c_bf16, \
                           p->M, p->N, p->ldc, (long long)p->tap_stride_b, (long long)p->batch_stride_c, a.split, p->tap_mask, p->K, a.rect, a.g); \
    }
    if (lanes == 16) GLF_S16_REDUCE(16) else if (lanes == 4) GLF_S16_REDUCE(4) else GLF_S16_REDUCE(1)
#undef GLF_S16_REDUCE
    return glf::check_launch("s16_gemm_tn(reduce)");
}
