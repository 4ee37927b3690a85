// Shared pieces of the contraction kernels (gemm_f32.hip: exact-fp32 MFMA; gemm_bf16s.hip: split-bf16 MFMA).
#pragma once
#include "glf_common.h"

namespace glf {
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
    int rect;       // tap-parallel rectangle mode (see tap_rect)
    const float* amax_a; const float* amax_b;   // f16x3 only: device scalars bounding max|A|, max|B| (null = no scaling)
    const float* zeros;                         // f16x3 only: device page of ZERO_PAGE_FLOATS zeros
    float* amax_c;                              // f16x3 only: receives max(*amax_c, max|C written|) (null = not wanted)
    double* colstats;                           // f16x3 NT only: [2][N] += column sums of C and of C^2 (null = not wanted)
    float* colmax;                              // f16x3 NT only, with colstats: [N] = max(colmax, column maxima of |C|) (null = not wanted)
    float* partial;                             // TN only: partial-sum slabs [batch*split][kept taps][M][N] (null = atomics into C)
    unsigned long long* stamps;                 // diagnostic builds (-DGLF_STAMPS) only
    int a_presplit, b_presplit;                 // f16x3 / f16 kernels: the operand pointer is the packed pre-split image (glf_split_f16_packed)
    int flags;                                  // f16x3 rows kernel: bit 0 = waves 4-7 run at s_setprio 1 (the younger half of an 8-wave workgroup)
};
constexpr int ZERO_PAGE_FLOATS = 1 << 18;


int precision();                               // 0 = exact fp32 MFMA, 1 = split-bf16 (bf16x6), 2 = split-fp16 (f16x3), 3 = fp16 (one MFMA per product); glf_api.hip
int init_gemm_bf16s_attrs();                   // gemm_bf16s.hip
bool bf16s_rows_ok(const GemmArgs& a);
bool bf16s_tn_ok(const GemmArgs& a);
int launch_rows_bf16s(const GemmArgs& a, dim3 grid, bool gather, hipStream_t s);
int launch_tn_bf16s(const GemmArgs& a, dim3 grid, bool gather, hipStream_t s);
int init_gemm_f16s_attrs();                    // gemm_f16s.hip
int launch_rows_f16s(const GemmArgs& a, dim3 grid, bool gather, int nprod, hipStream_t s);   // nprod: 3 = f16x3, 1 = f16
int launch_tn_f16s(const GemmArgs& a, dim3 grid, bool gather, int nprod, hipStream_t s);
int init_gemm_f16s4_attrs();                   // gemm_f16s4.hip: the 4-wave 128 x 128 NT configuration (two workgroups per CU)
bool use_f16s4(const GemmArgs& a);
int launch_rows_f16s4(const GemmArgs& a, dim3 grid, bool gather, int nprod, hipStream_t s);
int launch_amax(const float* x, long long rows, int cols, long long ld, int vec, float* out, hipStream_t s);
int launch_split_packed(const float* x, long long rows, int cols, long long ld, const float* amax, float* out, long long ldo, hipStream_t s);   // gemm_f16s.hip
float* amax_scratch(int n, hipStream_t s);     // n consecutive device floats from the ring of stream s (glf_api.hip)
const float* zero_page();                      // ZERO_PAGE_FLOATS zeros on the device (glf_api.hip)
bool f16s_rows_ok(const GemmArgs& a);
bool f16s_tn_ok(const GemmArgs& a);
}  // namespace glf

namespace {
using glf::Geo;
using glf::GemmArgs;

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LD_T = 129;   // LDS row stride of a tile filled by the transposing scatter
constexpr int LD_V = 132;   // LDS row stride of a tile filled with ds_write_b128
constexpr int NTHREADS = 256;

// Scalar copy of the conv geometry: kernels keep it (and every other GemmArgs field they use) in local
// scalars -- lambdas that capture the by-value kernel argument struct by reference made hipcc spill the whole
// struct to scratch and re-load fields from there (3x slower).
struct GeoS { int hs, ws, hd, wd, kw, stride, pad, dil; };

// source row of GEMM row (n,y,x) for `tap`, or -1 when the tap falls into the padding
__device__ __forceinline__ int map_src(const int hs, const int ws, const int kw, const int stride, const int pad,
                                       const int dil, const int gather, int n, int y, int x, int tap) {
    const int ky = tap / kw, kx = tap - ky * kw;
    int sy, sx;
    if (gather == 1) {
        sy = y * stride - pad + ky * dil;
        sx = x * stride - pad + kx * dil;
        if ((unsigned)sy >= (unsigned)hs || (unsigned)sx >= (unsigned)ws) return -1;
    } else {
        sy = y + pad - ky * dil;
        sx = x + pad - kx * dil;
        if (sy < 0 || sx < 0) return -1;
        if (stride > 1) {
            if ((sy % stride) != 0 || (sx % stride) != 0) return -1;
            sy /= stride; sx /= stride;
        }
        if (sy >= hs || sx >= ws) return -1;
    }
    return (n * hs + sy) * ws + sx;
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

// Rectangle of destination pixels for which `tap` reads inside the source map (stride 1 only):
//   gather 1: 0 <= y - pad + ky*dil < hs ;  gather 2: 0 <= y + pad - ky*dil < hs.
// ASPP's rate-12/24 3x3 convs on a 28x28 map spend most taps in the padding (in-bounds fractions 51 % / 18 %);
// in rect mode every tap becomes its own GEMM over exactly its rectangle (rows enumerate the rectangle, no
// padding work at all) and the per-tap results are summed with float atomics into a zero-filled output.
__host__ __device__ inline void tap_rect(int gather, int tap, int kw, int pad, int dil, int hs, int ws, int hd, int wd,
                                         int& y0, int& y1, int& x0, int& x1) {
    const int ky = tap / kw, kx = tap - ky * kw;
    const int oy = (gather == 1) ? pad - ky * dil : ky * dil - pad;
    const int ox = (gather == 1) ? pad - kx * dil : kx * dil - pad;
    y0 = oy > 0 ? oy : 0;           y1 = hs + oy < hd ? hs + oy : hd;
    x0 = ox > 0 ? ox : 0;           x1 = ws + ox < wd ? ws + ox : wd;
    if (y1 < y0) y1 = y0;
    if (x1 < x0) x1 = x0;
}

// Region mode (rect = 2; 3x3 "same" convs: stride 1, pad == dil, source and destination maps of equal size).
// Along one axis of length H the three taps have offsets (+d, 0, -d) (forward gather; mirrored for the transposed
// one), so the axis splits into at most three bands [0, e1), [e1, e2), [e2, H) with e1 = min(d, H-d), e2 = max(d, H-d)
// inside each of which the SET of in-range taps is constant:  band 0 = {centre, far side}, band 2 = {near side,
// centre}, band 1 = all three when d <= H-d, the centre tap alone otherwise.  The 3 x 3 products of bands are
// rectangles of output pixels that (a) partition the map -- every output element is stored exactly once, no atomics,
// no zero fill -- and (b) need exactly their own taps, every one of them in range for every pixel -- no padding work.
// band b of an axis: [lo, hi) and the 3-bit set of taps (bit t = tap index t along that axis).
__host__ __device__ inline void axis_band(int gather, int b, int d, int h, int& lo, int& hi, unsigned& taps) {
    int e1 = d < h - d ? d : h - d, e2 = d < h - d ? h - d : d;
    e1 = e1 < 0 ? 0 : (e1 > h ? h : e1);
    e2 = e2 < 0 ? 0 : (e2 > h ? h : e2);
    lo = b == 0 ? 0 : (b == 1 ? e1 : e2);
    hi = b == 0 ? e1 : (b == 1 ? e2 : h);
    // forward gather: tap 0 reads y - d (valid for y >= d), tap 2 reads y + d (valid for y < h - d); transposed: mirrored
    const unsigned first = gather == 1 ? 0b110u : 0b011u, last = gather == 1 ? 0b011u : 0b110u;
    taps = b == 0 ? first : (b == 2 ? last : (d <= h - d ? 0b111u : 0b010u));
}
// region number r (0..8) in DISPATCH order: the centre (all its taps in range: the longest blocks) first, then the four
// edges, then the corners, so that the last, partly filled round of workgroups consists of the shortest blocks.
// Rectangle and 9-bit tap mask (bit ky*3 + kx).
__host__ __device__ inline void region_of(int gather, int r, int d, int hd, int wd, int& y0, int& y1, int& x0, int& x1, unsigned& mask) {
    // (by, bx) packed as by*3 + bx for r = 0..8: centre, edges, corners
    const int cell = (int)((0x8620'7531'4ull >> (4 * r)) & 0xf);
    unsigned ty, tx;
    axis_band(gather, cell / 3, d, hd, y0, y1, ty);
    axis_band(gather, cell % 3, d, wd, x0, x1, tx);
    mask = 0;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
            if (((ty >> ky) & 1u) && ((tx >> kx) & 1u)) mask |= 1u << (ky * 3 + kx);
}

// value select (a `cond ? reg4 : make_float4(0.f, 0.f, 0.f, 0.f)` on two lvalues becomes a pointer select through scratch)
__device__ __forceinline__ float4 keep_if(bool c, const float4 v) {
    return make_float4(c ? v.x : 0.f, c ? v.y : 0.f, c ? v.z : 0.f, c ? v.w : 0.f);
}

__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int q = nblk >> 3, r = nblk & 7, xcd = bid & 7, idx = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

int validate(const glf_gemm_params* p, const void* A, const void* B, const void* C) {
    GLF_REQUIRE(p && A && B && C, GLF_ERR_NULL, "gemm: null argument");
    GLF_REQUIRE(p->M > 0 && p->N > 0 && p->K > 0, GLF_ERR_BAD_SHAPE, "gemm: M,N,K must be > 0 (got %d,%d,%d)", p->M, p->N, p->K);
    GLF_REQUIRE(p->taps >= 1 && p->taps <= 32, GLF_ERR_BAD_SHAPE, "gemm: taps must be in [1,32] (got %d)", p->taps);
    GLF_REQUIRE(p->batch >= 1 && p->batch <= 65535, GLF_ERR_BAD_SHAPE, "gemm: batch out of range (%d)", p->batch);
    GLF_REQUIRE(p->gather >= 0 && p->gather <= 2, GLF_ERR_BAD_SHAPE, "gemm: gather must be 0,1,2");
    GLF_REQUIRE(p->precision >= 0 && p->precision <= 4, GLF_ERR_UNSUPPORTED, "gemm: precision must be 0 (process default), 1 (fp32), 2 (bf16x6), 3 (f16x3) or 4 (f16)");
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
    a.vec_a = 0; a.vec_b = 0; a.rect = 0;
    a.amax_a = p->amax_a; a.amax_b = p->amax_b; a.zeros = nullptr; a.amax_c = p->amax_c; a.colstats = p->colstats; a.colmax = p->colstats ? p->colmax : nullptr; a.partial = nullptr; a.flags = 0; a.stamps = nullptr; a.a_presplit = p->a_presplit; a.b_presplit = p->b_presplit;
    return a;
}

// rect mode (tap-parallel rectangles): validates and sizes the grid as the sum over taps of their tiles
int setup_rect(const glf_gemm_params* p, const float* bias, GemmArgs& a, dim3& grid, const char* who) {
    GLF_REQUIRE(p->gather == 1 || p->gather == 2, GLF_ERR_BAD_SHAPE, "%s: rect mode needs a conv gather", who);
    GLF_REQUIRE(p->stride == 1, GLF_ERR_UNSUPPORTED, "%s: rect mode needs stride 1", who);
    GLF_REQUIRE(bias == nullptr && !p->accumulate && p->batch == 1, GLF_ERR_UNSUPPORTED,
                "%s: rect mode takes no bias / accumulate / batch (C must be zero-filled by the caller)", who);
    long long tiles = 0;
    for (unsigned mm = p->tap_mask; mm; mm &= mm - 1) {
        const int t = __builtin_ctz(mm);
        int y0, y1, x0, x1;
        tap_rect(p->gather, t, p->kw, p->pad, p->dil, p->hs, p->ws, p->hd, p->wd, y0, y1, x0, x1);
        const long long mt = (long long)p->n_img * (y1 - y0) * (x1 - x0);
        tiles += (mt + BM - 1) / BM;
    }
    GLF_REQUIRE(tiles > 0 && tiles * a.tiles_n < 2147483647LL, GLF_ERR_BAD_SHAPE, "%s: rect mode grid out of range", who);
    a.rect = 1;
    grid = dim3((unsigned)(tiles * a.tiles_n), 1, 1);
    return GLF_OK;
}

// contraction precision of this call: glf_gemm_params.precision (1 + mode) or, when 0, the process default
inline int call_precision(const glf_gemm_params* p) {
    return (p->precision >= 1 && p->precision <= 4) ? p->precision - 1 : glf::precision();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }


}  // namespace
