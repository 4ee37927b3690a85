/*
 * glfusion.h -- C ABI of libglfusion_hip.so: the MI355X (gfx950 / CDNA4) engine for the
 * GL-Fusion forward/backward hot path.
 *
 * The reference (xmed-lab/GL-Fusion) is pure PyTorch: it has no FFI of its own.  Each entry
 * point below replaces the ATen operator(s) the reference launches implicitly at the cited
 * call sites (file:line relative to GLfusion/ in the reference tree); the Python host side
 * (gl-fusion_amd/) binds them with ctypes from torch.autograd.Function wrappers.
 *
 * Conventions
 *   - plain pointers and sizes only; every tensor and workspace buffer is CALLER-OWNED (inputs,
 *     outputs, workspace, saved-for-backward) and no caller pointer is kept after return.  The
 *     library owns exactly two kinds of small device allocations, made lazily and kept until the
 *     process exits: a 1 MiB page of zeros per device (the split-fp16 kernels load conv-padding
 *     and tile-overhang rows from it) and a 16 KiB ring of floats per (device, stream) that is
 *     used only when a precision-3 contraction is called with amax_a / amax_b == NULL (the
 *     library then measures the operand maxima itself, on that stream).
 *   - every function only ENQUEUES work on `stream` and returns; no host synchronisation.
 *   - return 0 on success, a negative glf_status otherwise; text via glf_last_error()
 *     (thread-local).  No exception crosses this boundary.
 *   - activations are channels-last: a tensor [N,H,W,C] is the row-major matrix
 *     [rows = N*H*W][C]; `ld*` arguments are row strides in elements.
 *   - arithmetic type: fp32 operands and results; contractions run on v_mfma_f32_32x32x2_f32 or on the
 *     split 16-bit MFMA schemes selected per call / per process (see glf_set_precision).
 */
#ifndef GLFUSION_H
#define GLFUSION_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* glf_stream_t; /* hipStream_t */

typedef enum {
    GLF_OK = 0,
    GLF_ERR_BAD_SHAPE = -1,
    GLF_ERR_UNSUPPORTED = -2,
    GLF_ERR_WORKSPACE = -3,
    GLF_ERR_LAUNCH = -4,
    GLF_ERR_NULL = -5
} glf_status;

/* Element type tags of the typed ("_t") and 16-bit-storage ("glf_s16_") entry points. */
typedef enum { GLF_DT_F32 = 0, GLF_DT_BF16 = 1 } glf_dtype;

const char* glf_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int glf_abi_version(void);
/* Queries the current device once (CU count) and raises the dynamic-LDS limit of the MFMA
 * kernels.  Optional: every entry point calls it lazily. */
int glf_init(void);
/* DEFAULT contraction precision of glf_gemm_nt / glf_gemm_tn, used by calls whose glf_gemm_params.precision is 0:
 *   0 = exact fp32 on v_mfma_f32_32x32x2_f32 (default);
 *   1 = split-bf16 "bf16x6": each fp32 operand is split into three bf16 pieces and six
 *       v_mfma_f32_32x32x16_bf16 reproduce the fp32 product to 2^-23 (fp32 accumulate) -- fp32-equivalent
 *       results at 2.67x the fp32-MFMA roof.  Calls that miss the aligned fast path (and glf_gemm_nn) stay
 *       on the exact kernels;
 *   2 = split-fp16 "f16x3": each operand is scaled by a power of two taken from its max magnitude
 *       (glf_gemm_params.amax_a / amax_b) and split into two fp16 pieces; three
 *       v_mfma_f32_32x32x16_f16 reproduce the fp32 product to 2^-22 relative to the operand maxima
 *       (fp32 accumulate) at 5.3x the fp32-MFMA roof.  Elements more than 2^27 below their operand's
 *       maximum keep an absolute (not relative) error bound, 2^-49 of that maximum.  Same fast-path rule;
 *   3 = fp16 "f16": the same kernels keeping only the high fp16 half of each scaled operand -- ONE
 *       v_mfma_f32_32x32x16_f16 per product, fp32 accumulate, 11-bit operands (relative error ~2^-11 per
 *       element): the 16-bit-arithmetic configuration (BASELINE.json configs[2]), NOT fp32-equivalent. */
int glf_set_precision(int mode);
int glf_get_precision(void);
/* sizeof(glf_gemm_params) as compiled into the library (binding self-check). */
size_t glf_sizeof_gemm_params(void);

/* ---------------------------------------------------------------------------------------
 * Contractions (implicit GEMM, NHWC, im2col-free; v_mfma_f32_32x32x2_f32; 128x128x32 tiles
 * staged through LDS).  One parameter block serves conv forward / dgrad / wgrad and the
 * plain (batched) GEMMs of the fusion block.
 *
 *   C[b][m][n] (+)= alpha * sum_{tap} sum_{k} A_tap[b][m][k] * B_tap[b][k][n]  (+ bias[n])
 *
 * gather: 0 = A rows are taken as they are (plain GEMM / 1x1 stride-1 conv)
 *         1 = forward conv mapping : GEMM row m = output pixel (n,y,x) on the hd x wd grid reads
 *             source pixel (y*stride - pad + ky*dil, x*stride - pad + kx*dil) on the hs x ws grid
 *         2 = transposed mapping (dgrad): row m = input pixel (n,y,x) on hd x wd reads the
 *             output-gradient pixel ((y + pad - ky*dil)/stride, ...) on hs x ws when divisible
 * Out-of-range taps read zeros; taps invalid for a whole 128-row tile are skipped (or see `rect`).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int32_t M, N, K;            /* per-tap GEMM extents (see each entry point)                   */
    int32_t lda, ldb, ldc;      /* row strides (elements)                                        */
    int32_t taps;               /* kh*kw (1 for plain GEMM)                                      */
    uint32_t tap_mask;          /* bit t set = tap t can be valid for some row (host-computed)   */
    int64_t tap_stride_b;       /* elements between consecutive taps of the weight operand       */
    int32_t gather;             /* 0 / 1 / 2 as above                                            */
    int32_t n_img, hs, ws, hd, wd, kh, kw, stride, pad, dil;
    int32_t batch;              /* >= 1                                                          */
    int64_t batch_stride_a, batch_stride_b, batch_stride_c;
    float alpha;                /* scale applied to the accumulator                              */
    int32_t accumulate;         /* 1: C += result (fwd/dgrad: read-modify-write; wgrad: atomics) */
    int32_t split;              /* wgrad/TN only: number of reduction slices (>=1); >1 requires  */
                                /* C zero-filled (or holding the value to accumulate onto).  A   */
                                /* slice is ceil(K / split) rows (rounded up to 32) of EVERY      */
                                /* tap's reduction: in rect mode a tap whose rectangle is short   */
                                /* uses fewer slices, so all workgroups get reductions of equal   */
                                /* length whatever the rectangles' sizes.                        */
    int32_t rect;               /* 1: tap-parallel rectangle mode (stride-1 convs whose taps fall */
                                /* mostly into the padding, i.e. ASPP): each tap runs as its own  */
                                /* GEMM over exactly its in-range rectangle of pixels; nt/nn sum  */
                                /* the taps with float atomics into a ZERO-FILLED C (no bias); tn */
                                /* shortens each tap's reduction to its rectangle.                */
                                /* 2 (glf_gemm_nt, precision 2 only): region mode for 3x3 stride-1 */
                                /* convs with pad == dil on equal maps: the output map is cut into */
                                /* <= 9 rectangles inside each of which the set of in-range taps   */
                                /* is constant; each is a GEMM over exactly its taps, every output */
                                /* element is stored ONCE (no atomics, no zero fill, bias and      */
                                /* accumulate allowed, no padding work)                            */
    const float* amax_a;        /* precision 2 only: DEVICE scalars holding an upper bound of max|A| */
    const float* amax_b;        /* and max|B| over the elements the call reads (glf_amax, or the     */
                                /* kernel that produced the operand).  NULL: the library measures    */
                                /* the operand itself first (one extra read of it).                  */
    float* amax_c;              /* precision 2 only, may be NULL: device float that receives                 */
                                /* max(*amax_c, max|C|) over the elements this call stores -- the amax of    */
                                /* the NEXT contraction's operand for free.  Ignored (left untouched) by     */
                                /* calls that sum partial results with atomics (rect mode, split > 1).       */
    double* colstats;           /* glf_gemm_nt, precision 2 only, may be NULL: DEVICE doubles [2][N], zero-filled by   */
                                /* the caller; the epilogue adds the column sums of the stored C and of C^2 --   */
                                /* the BatchNorm batch statistics of a conv output without a pass over it        */
                                /* (finish with glf_bn_stats_from_sums).  A call that cannot honour it (exact    */
                                /* kernels, rect = 1) fails with GLF_ERR_UNSUPPORTED instead of ignoring it.     */
    float* workspace;           /* glf_gemm_tn with split > 1, may be NULL: caller-owned scratch of at least            */
    int64_t workspace_bytes;    /* glf_gemm_tn_workspace_bytes(p) bytes.  With it the reduction slices store partial     */
                                /* sums there and a second kernel adds them in a fixed order: no float atomics, C need   */
                                /* not be zero-filled, results are bitwise reproducible.  Without it (NULL) the slices   */
                                /* meet in float atomics and C must be zero-filled (or hold the value to add to).        */
    int32_t a_presplit;         /* precision 3 / 4 only: 1 = the A (B) pointer is not fp32 data but its packed pre-split   */
    int32_t b_presplit;         /* image written by glf_split_f16_packed with the SAME amax_a (amax_b): same byte size,     */
                                /* strides and addressing as the fp32 operand, every 16-byte group of four elements holding */
                                /* {h0..h3, l0..l3} (fp16).  The kernels then skip the split in their staging path: an       */
                                /* operand read by many tiles / launches (activations across column tiles, weights across   */
                                /* row tiles, a conv input again in its weight gradient) is split once.                     */
    int32_t precision;          /* contraction precision of THIS call: 0 = the process default (glf_set_precision), */
                                /* 1 = exact fp32, 2 = split-bf16 x6, 3 = split-fp16 x3, 4 = fp16 x1 (= 1 + the modes of */
                                /* glf_set_precision).  Two models with different precisions can share a process.   */
    float* colmax;              /* with colstats, may be NULL: DEVICE floats [N], zero-filled by the caller; the epilogue   */
                                /* raises colmax[n] to the largest |C[m][n]| it stores -- with the batch statistics it bounds */
                                /* the BatchNorm OUTPUT's maximum before that output exists (glf_bn_apply_from_sums         */
                                /* packed_y: the activation written once, as the packed image its consumer reads)           */
    int32_t c_dtype;            /* glf_s16_gemm_nt / _tn only: element type of C (GLF_DT_F32 or GLF_DT_BF16); the fp32-storage    */
    int32_t reserved0;          /* entry points ignore it (C is fp32).  reserved0: padding, leave 0.                              */
} glf_gemm_params;

/* *out = max |x| over the [rows, cols] view with row stride ld (elements); out is a device float. */
int glf_amax(const float* x, int64_t rows, int cols, int64_t ld, float* out, glf_stream_t stream);
/* out = packed pre-split image of x [rows][cols] (row strides ld / ldo in floats; cols, ld, ldo % 4 == 0, 16-byte
 * aligned): every float4 {x0..x3} -> {h0..h3, l0..l3} fp16 with x * s = h + 2^-11 l, s the power of two the precision-3
 * kernels derive from *amax (a device float >= max|x|).  For glf_gemm_params.a_presplit / b_presplit. */
int glf_split_f16_packed(const float* x, int64_t rows, int cols, int64_t ld, const float* amax, float* out, int64_t ldo,
                         glf_stream_t stream);

/* A[m][k] (k contiguous, rows gathered per `gather`), B_tap[n][k] (k contiguous: torch's
 * [Cout][Cin] weight layout per tap), C[m][n].  Replaces F.conv2d / nn.Conv3d(1x1x1) forward:
 * models/_utils.py:192 (stem is separate), torchvision Bottleneck convs (ours.py:1797-1800),
 * deeplabv3.py:104-165, ours.py:866,878-879,908.  bias may be NULL. */
int glf_gemm_nt(const float* A, const float* B, const float* bias, float* C,
                const glf_gemm_params* p, glf_stream_t stream);
/* A[m][k] (rows gathered), B_tap[k][n] (n contiguous), C[m][n].  Conv dgrad (A = dY, B = the
 * same [tap][Cout][Cin] weights, gather = 2) and y = theta @ M of the re-associated dot
 * attention (ours.py:881-902). */
int glf_gemm_nn(const float* A, const float* B, const float* bias, float* C,
                const glf_gemm_params* p, glf_stream_t stream);
/* Reduction over ROWS: C_tap[m][n] (+)= alpha * sum_r A[r][m] * B[src(r,tap)][n], r < K.
 * Here p->K is the number of rows r, p->M / p->N the two channel extents, `gather` applies
 * to B's rows.  Conv wgrad (A = dY [rows][Cout], B = X [rows][Cin], C = dW [tap][Cout][Cin])
 * and M = phi^T g of the dot attention.  Output taps are p->tap_stride_b apart in C. */
int glf_gemm_tn(const float* A, const float* B, float* C,
                const glf_gemm_params* p, glf_stream_t stream);
/* bytes of glf_gemm_params.workspace the two-stage reduction of this call needs (0 when p->split <= 1). */
size_t glf_gemm_tn_workspace_bytes(const glf_gemm_params* p);

/* ---------------------------------------------------------------------------------------
 * 16-bit storage ("S16": BASELINE.json configs[2] and [4] -- bf16 activations, saved tensors and
 * activation gradients in HBM, fp32 master weights, fp32 accumulate).  Same contraction
 * semantics and parameter block as glf_gemm_nt / glf_gemm_tn with these differences:
 *   - A and B hold bf16 (row / batch / tap strides multiples of 8 elements, 16-byte aligned);
 *     C holds p->c_dtype (activations: GLF_DT_BF16, weight gradients: GLF_DT_F32);
 *   - ONE v_mfma_f32_32x32x16_bf16 per product; operands go global -> LDS by LDS-DMA
 *     (global_load_lds_dwordx4), padding / overhang rows come from the library's zero page;
 *   - K % 64 == 0 (NT); M % 8 == 0 and N % 8 == 0 (TN); rect = 0 or 2 (NT region mode); no
 *     amax / presplit / precision fields (ignored); colstats is honoured by glf_s16_gemm_nt
 *     (sums of the fp32 results before they are rounded to bf16);
 *   - glf_s16_gemm_tn with split > 1 REQUIRES the workspace (glf_s16_gemm_tn_workspace_bytes):
 *     slices store partial slabs, a second kernel adds them in slice order (no atomics).
 * Replaces (reference call sites): the same convolutions / projections as glf_gemm_*, i.e.
 * torchvision Bottleneck convs (ours.py:1797-1800), ASPP / head convs (deeplabv3.py:102-166),
 * TPAVI theta/phi/g/W_z and the attention matmuls (ours.py:866-908), run under the bf16
 * configuration the reference itself never had.
 * ------------------------------------------------------------------------------------- */
int glf_s16_gemm_nt(const void* A, const void* B, const float* bias, void* C, const glf_gemm_params* p, glf_stream_t stream);
int glf_s16_gemm_tn(const void* A, const void* B, void* C, const glf_gemm_params* p, glf_stream_t stream);
size_t glf_s16_gemm_tn_workspace_bytes(const glf_gemm_params* p);

/* ---------------------------------------------------------------------------------------
 * Convolution entry points (F.conv2d forward / backward: models/_utils.py:192 excluded -- glf_stem7x7_* --,
 * torchvision Bottleneck convs at ours.py:1797-1800, deeplabv3.py:104-165).  x [n][h][w][cin] and y / dy
 * [n][ho][wo][cout] channels-last, weights tap-major [kh*kw][cout][cin] (glf_oihw_to_tap_major), groups = 1, zero
 * padding.  They sit on top of glf_gemm_* and embed its launch policy (host-side tap mask of the taps that can touch
 * the map at all, per-tap rectangle / region modes when most tap work would be padding -- the ASPP rates on a 28 x 28
 * map --, split-K of the weight gradient, zero fills), so a C caller needs nothing but this block.
 * glf_conv2d_plan reports what a pass will do (GEMM extents, kept taps, mode, slices, workspace) without launching.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n, h, w, cin, cout, kh, kw, stride, pad, dil;
    int32_t precision;          /* as glf_gemm_params.precision (0 = process default)                            */
    const float* amax_x;        /* precision 3 / 4, optional: device scalars bounding max|x|, max|w|, max|dy|     */
    const float* amax_w;        /* (NULL: measured by the library)                                               */
    const float* amax_dy;
    float* amax_out;            /* optional: receives max|output| of fwd / dgrad where the kernel stores directly */
    double* colstats;           /* fwd only, optional: [2][cout] zero-filled doubles, += column sums of y, y^2    */
} glf_conv_params;
typedef struct {
    int32_t ho, wo;             /* output map                                                                     */
    int32_t M, N, K;            /* per-tap GEMM extents of the pass                                               */
    int32_t taps, kept_taps;    /* kh*kw and how many of them can be in range                                     */
    uint32_t tap_mask;
    int32_t plain;              /* 1: a 1x1 stride-1 conv = plain GEMM                                            */
    int32_t rect;               /* 0 dense, 1 per-tap rectangles (atomics into a zero-filled output), 2 regions   */
    int32_t split;              /* wgrad: reduction slices                                                        */
    int32_t zero_fill;          /* the call zero-fills its output first                                           */
    int32_t colstats_ok;        /* fwd: glf_conv_params.colstats can be honoured                                  */
    int64_t workspace_bytes;    /* wgrad: scratch for the two-stage split-K reduction (0: none needed)            */
} glf_conv_plan;
int glf_conv2d_plan(const glf_conv_params* p, int pass /* 0 fwd, 1 dgrad, 2 wgrad */, glf_conv_plan* plan);
int glf_conv2d_fwd(const float* x, const float* w_tap, const float* bias, float* y, const glf_conv_params* p, glf_stream_t s);
/* w_tap_t: the same weights as [tap][cin][cout] (glf_oihw_to_tap_major_t); needed by the split-precision kernels,
 * may be NULL for exact fp32 (then w_tap is used). */
int glf_conv2d_dgrad(const float* dy, const float* w_tap, const float* w_tap_t, float* dx, const glf_conv_params* p, glf_stream_t s);
/* dw_tap [kh*kw][cout][cin].  workspace (>= plan.workspace_bytes, may be NULL: float atomics instead of the two-stage
 * reduction) is caller-owned scratch. */
int glf_conv2d_wgrad(const float* dy, const float* x, float* dw_tap, float* workspace, int64_t workspace_bytes,
                     const glf_conv_params* p, glf_stream_t s);

/* ---------------------------------------------------------------------------------------
 * Fused softmax attention of TPAVIModule's `embedded` mode (ours.py:881, 896-897, 902):
 *   y[n] = softmax(theta[n] phi[n]^T, dim = -1) g[n]      per frame n, theta / phi / g: [L][Ci] rows (row strides ld*).
 * One kernel per pass; the [L][L] score matrix is never written to memory (online row max / sum over 64-key tiles
 * staged through LDS, P g accumulated in MFMA accumulators; backward recomputes the tiles from theta, phi and the saved
 * row log-sum-exp).  Exact fp32 arithmetic (v_mfma_f32_32x32x2_f32).  Ci % 32 == 0, Ci <= 1024; frame n of every operand
 * starts L rows after frame n - 1 (frame stride = L * its row stride).
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int32_t frames, L, ci;
    int64_t ldq, ldk, ldv;      /* row strides of theta, phi, g (elements; column slices of one [rows][3 Ci] buffer: 3 Ci) */
    int64_t ldy;                /* row stride of y                                                                       */
    int64_t lddy;               /* backward: row stride of dy                                                            */
    int64_t ldd;                /* backward: common row stride of dtheta, dphi, dg                                        */
} glf_attn_params;
/* y [frames*L][Ci] and lse [frames*L] (row log-sum-exp, needed by the backward pass). */
int glf_attn_softmax_fwd(const float* theta, const float* phi, const float* g, float* y, float* lse,
                         const glf_attn_params* p, glf_stream_t stream);
/* dtheta, dphi, dg from dy (every output element is written exactly once: no zero fill, no atomics).
 * dsum_ws: caller-owned scratch of frames*L floats. */
int glf_attn_softmax_bwd(const float* theta, const float* phi, const float* g, const float* y, const float* dy, const float* lse,
                         float* dtheta, float* dphi, float* dg, float* dsum_ws, const glf_attn_params* p, glf_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Weight layout: torch OIHW [Cout][Cin][kh][kw] <-> tap-major [kh*kw][Cout][Cin].
 * ------------------------------------------------------------------------------------- */
int glf_oihw_to_tap_major(const float* w, float* out, int cout, int cin, int taps, glf_stream_t s);
int glf_tap_major_to_oihw(const float* w, float* out, int cout, int cin, int taps, glf_stream_t s);
/* [kh*kw][Cin][Cout] (k = Cout contiguous): the dgrad weight operand when dgrad runs as an NT contraction. */
int glf_oihw_to_tap_major_t(const float* w, float* out, int cout, int cin, int taps, glf_stream_t s);
/* batched 2-D transpose dst[b][c][r] = src[b][r][c] (operand re-layout for NT contractions). */
int glf_transpose2d(const float* src, float* dst, int rows, int cols, int batch, glf_stream_t s);
/* The same with strides: src[b][r][c] = src[b * batch_stride_src + r * ld_src + c]; dst[b][c][r] (row stride ld_dst) for r < rows,
 * ZERO for rows <= r < rows_pad: a column slice of a wider matrix transposed into an operand whose reduction dimension is padded to
 * the contraction kernels' K granule (the g^T / phi^T operands of the per-frame softmax attention, fusion.py). */
int glf_transpose2d_strided(const float* src, int64_t ld_src, int64_t batch_stride_src, float* dst, int64_t ld_dst,
                            int64_t batch_stride_dst, int rows, int cols, int rows_pad, int batch, glf_stream_t s);

/* ---------------------------------------------------------------------------------------
 * Multi-tensor weight refresh.  Everything the contraction kernels derive from the parameters -- tap-major re-layouts
 * (forward / wgrad order and the transposed dgrad order), transposed copies of 1x1 / linear weights, the stacked
 * theta | phi | g operand of a fusion block (ours.py:866,878-879), each parameter's max magnitude (precision 3 / 4) and
 * the packed pre-split images of all of those -- has to be rebuilt after EVERY optimizer step (main.py:243).  Through the
 * per-tensor entry points above that is one launch per image, ~1 000 launches per update for the 3-view model; this block
 * does it in one launch per dependency level.  The caller describes every derived image once as a glf_weight_job (pointers
 * are caller-owned and must stay valid), sorts the jobs by `pass`, lets glf_weights_plan fill in the grid bookkeeping,
 * copies the table to the device, and calls glf_weights_refresh after each update.  Results are bit-identical to the
 * per-tensor entry points.
 * ------------------------------------------------------------------------------------- */
enum {
    GLF_WJ_COPY = 0,        /* dst[i] = src[i], d0 elements                                   (pass 0) */
    GLF_WJ_AMAX = 1,        /* *amax = max(*amax, max|src|), d0 elements; dst unused           (pass 1) */
    GLF_WJ_TAP_MAJOR = 2,   /* src OIHW [d0=Cout][d1=Cin][d2=taps] -> dst [taps][Cout][Cin]     (pass 2) */
    GLF_WJ_TAP_MAJOR_T = 3, /* ... -> dst [taps][Cin][Cout]                                     (pass 2) */
    GLF_WJ_TRANSPOSE = 4,   /* src [d0=rows][d1=cols] -> dst [cols][rows]                       (pass 2) */
    GLF_WJ_PACK = 5,        /* dst = packed pre-split image of src (d0 elements, % 4 == 0, both 16-byte aligned)
                               scaled by *amax, as glf_split_f16_packed                          (pass 3) */
    GLF_WJ_ZERO = 6,        /* dst[i] = 0, d0 elements (src unused): the amax slots of a table that covers only SOME of an
                               arena's parameters -- such a caller passes amax_arena = NULL to glf_weights_refresh      (pass 0) */
    GLF_WJ_CVT_BF16 = 7,    /* dst (bf16 elements behind the float* field) = round-to-nearest-even(src), d0 elements (% 4 == 0): the operand
                               images of the 16-bit-storage contractions, from fp32 master weights or their layouts   (pass 3) */
    GLF_WJ_PASSES = 4
};
typedef struct {
    const float* src;
    float* dst;
    float* amax;            /* AMAX: the slot to raise (inside the arena glf_weights_refresh zero-fills); PACK: the scale source */
    int32_t kind;           /* GLF_WJ_*                                                        */
    int32_t pass;           /* dependency level, 0 .. GLF_WJ_PASSES-1; the table is sorted by it */
    int32_t d0, d1, d2;     /* extents, see the kinds                                          */
    int32_t reserved;
    int64_t first_wg;       /* filled by glf_weights_plan: first workgroup of this job in its pass */
} glf_weight_job;
/* Validates a HOST table sorted by pass, fills first_wg of every job and, per pass p (arrays of GLF_WJ_PASSES entries),
 * the index of its first job, its job count and its workgroup count.  No device work. */
int glf_weights_plan(glf_weight_job* jobs_host, int n_jobs, int* pass_first, int* pass_count, int64_t* pass_wgs);
/* Zero-fills amax_arena[0 .. amax_floats) (may be NULL / 0) and runs the passes of a planned table that has been copied
 * to device memory (jobs_dev), in order, on `s`. */
int glf_weights_refresh(const glf_weight_job* jobs_dev, const int* pass_first, const int* pass_count, const int64_t* pass_wgs,
                        float* amax_arena, int64_t amax_floats, glf_stream_t s);

/* ---------------------------------------------------------------------------------------
 * 16-bit storage ("S16"), streaming kernels: the counterparts of the normalisation / pointwise
 * entry points above for bf16 tensors (`void*` = bf16 elements unless named float; channel
 * counts and row strides multiples of 8, 16-byte aligned tensors).  fp32 arithmetic, one
 * rounding at the store.  BatchNorm sums follow the glf_gemm_params.colstats contract: DEVICE
 * doubles [2][C], zero-filled by the caller, filled by glf_s16_gemm_nt's epilogue or by
 * glf_s16_colstats; the apply kernels finish the statistics themselves.
 * ------------------------------------------------------------------------------------- */
/* sums[c] += sum_r x[r][c]; sums[C + c] += sum_r x[r][c]^2 (f64 atomics, one per column and workgroup). */
int glf_s16_colstats(const void* x, int ldx, int rows, int c, double* sums, glf_stream_t s);
/* y = [relu]( BN(x) [+ residual] ).  sums != NULL (train): batch statistics from sums, written to mean / invstd, running
 * statistics updated (as glf_bn_apply_from_sums).  sums == NULL (eval): mean / invstd are inputs (glf_bn_eval_coeffs).
 * relu_mask (may be NULL): one byte per 8 channels, bit j = sign of the pre-ReLU value of channel 8 i + j. */
int glf_s16_bn_apply(const void* x, int ldx, const void* residual, int ldr, void* y, int ldy, const double* sums,
                     int rows, int c, float eps, float momentum, const float* gamma, const float* beta,
                     float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     int relu, uint8_t* relu_mask, glf_stream_t s);
/* BatchNorm backward in two launches: column reduction (sum dy', sum dy' xhat; into `sums`, DEVICE doubles [2][C] zero-filled
 * by the caller) and apply (finishes the sums, writes dgamma / dbeta / dx / dres).  dy2 (may be NULL): second addend of the
 * incoming gradient; relu: relu_mask (bytes of glf_s16_bn_apply) or, without a residual, recomputed from x and beta. */
int glf_s16_bn_bwd(const void* dy, int lddy, const void* dy2, int lddy2, const void* x, int ldx,
                   const float* mean, const float* invstd, const float* gamma, const float* beta,
                   void* dx, int lddx, void* dres, int lddres, float* dgamma, float* dbeta,
                   int rows, int c, int relu, int training, double* sums, const uint8_t* relu_mask, glf_stream_t s);
/* db[c] = sum_r dy[r][c]; workspace: 2 C doubles (zero-filled by the call). */
int glf_s16_colsum(const void* dy, int lddy, float* db, int rows, int c, double* workspace, glf_stream_t s);
/* TPAVI tail (ours.py:908-915) on bf16 w, x, z, dz, du; statistics vectors fp32. */
int glf_s16_bn_res_ln_fwd(const void* w, const void* x, const float* bn_mean, const float* bn_invstd, const float* bn_gamma,
                          const float* bn_beta, const float* ln_gamma, const float* ln_beta, float ln_eps, void* z,
                          float* row_mean, float* row_rstd, int rows, int c, glf_stream_t s);
size_t glf_s16_bn_res_ln_workspace(int rows, int c);   /* bytes */
int glf_s16_bn_res_ln_bwd(const void* dz, const void* w, const void* x, const float* bn_mean, const float* bn_invstd,
                          const float* bn_gamma, const float* bn_beta, const float* ln_gamma, const float* row_mean,
                          const float* row_rstd, void* du, float* dln_gamma, float* dln_beta, int rows, int c,
                          float* workspace, glf_stream_t s);
/* Stem with a bf16 conv output / output gradient (x, w, bias, dw, db fp32). */
int glf_s16_stem7x7_fwd(const float* x, const float* w, const float* bias, void* y,
                        int n, int h, int wdt, int cout, int pad, glf_stream_t s);
int glf_s16_stem7x7_wgrad(const float* x, const void* dy, float* dw, float* db, float* partial,
                          int n, int h, int wdt, int cout, int pad, glf_stream_t s);
int glf_s16_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* idx, int n, int h, int w, int c, glf_stream_t s);
int glf_s16_maxpool3x3s2_bwd(const void* dy, const uint8_t* idx, void* dx, int n, int h, int w, int c, glf_stream_t s);
/* y[n][c] = scale * sum_p x[n][p][c] (x bf16, row stride ldx; y of y_dtype); y[n][p][c] (bf16, row stride ldy) = scale * x[n][c]
 * (x of x_dtype).  The ASPP pooled branch (deeplabv3.py:123-135) keeps its N per-frame vectors in fp32. */
int glf_s16_sum_rows(const void* x, int ldx, void* y, int y_dtype, float scale, int n, int p, int c, glf_stream_t s);
int glf_s16_bcast_rows(const void* x, int x_dtype, void* y, int ldy, float scale, int n, int p, int c, glf_stream_t s);
int glf_s16_dropout(const void* x, void* y, int64_t numel, float p, uint64_t seed, const uint64_t* step_counter, glf_stream_t s);
int glf_s16_relu_fwd(const void* x, void* y, int64_t numel, glf_stream_t s);
int glf_s16_relu_bwd(const void* dy, const void* y, void* dx, int64_t numel, glf_stream_t s);
int glf_s16_axpby(const void* x, const void* y, void* out, float a, float b, int64_t numel, glf_stream_t s);
/* Local gate (ours.py:1802-1816): cls / ctr logits, gate map, their gradients fp32; f, y, dy, df bf16. */
int glf_s16_gate_fwd(const float* cls, int ncls, const float* ctr, const void* f, void* y, float* a, int32_t* argmax, float weight,
                     int rows, int c, glf_stream_t s);
int glf_s16_gate_bwd(const void* dy, const void* f, const float* cls, int ncls, const float* ctr, const float* a, const int32_t* argmax,
                     float weight, void* df, float* dcls, float* dctr, int rows, int c, glf_stream_t s);
int glf_s16_add_frames(const void* a, int64_t a_fs, const void* b, int64_t b_fs, void* dst, int64_t dst_fs, int n, int64_t inner,
                       glf_stream_t s);
int glf_s16_add_n(const void* const* inputs, int k, void* out, int64_t numel, glf_stream_t s);
/* dst (dst_dtype) = src (src_dtype): GLF_DT_BF16 -> GLF_DT_F32 or GLF_DT_F32 -> GLF_DT_BF16; numel % 8 == 0. */
int glf_s16_cast(const void* src, int src_dtype, void* dst, int dst_dtype, int64_t numel, glf_stream_t s);
/* dst[b][c][r] = src[b][r][c] for 16-bit elements. */
int glf_s16_transpose2d(const void* src, void* dst, int rows, int cols, int batch, glf_stream_t s);

/* ---------------------------------------------------------------------------------------
 * Single-call fusion block, 16-bit storage (SURVEY 8b rows tpavi_proj + tpavi_attn_dot +
 * tpavi_out_bn_res_ln; TPAVIModule.forward, ours.py:845-917, mode 'dot' -- the shipped model).
 * One call per direction enqueues the whole block on `stream`; the library allocates nothing:
 * results, the tensors kept for backward and the workspace are caller-owned device buffers.
 *   x, z, dz, dx   [n*L][C]    bf16   (n clips of L = V*h*w positions, channels-last)
 *   w_qkv          [3 Ci][C]   bf16   theta | phi | g weights stacked;  b_qkv [3 Ci] fp32
 *   w_z            [C][Ci]     bf16;  b_z [C] fp32
 *   w_qkv_t / w_z_t             bf16   their transposes [C][3 Ci] / [Ci][C] (glf_s16_transpose2d)
 *   qkv [n*L][3 Ci], att_t [n][Ci][Ci], y [n*L][Ci], wz [n*L][C]  bf16, bn_mean / bn_invstd [C],
 *   row_mean / row_rstd [n*L] fp32: written by the forward, read by the backward.
 *   dw_* / db_* / d*_gamma / d*_beta: fp32, torch parameter layouts.
 * C, Ci multiples of 64, C <= 2048.  Workspace: glf_s16_tpavi_workspace_bytes(p, pass) bytes (pass 0 = forward,
 * 1 = backward), 256-byte aligned, contents undefined on entry and exit.
 * ------------------------------------------------------------------------------------- */
typedef struct {
    int32_t n, L, c, ci;
    int32_t training;            /* BatchNorm3d mode: batch statistics + running update (1) or running statistics (0) */
    float bn_eps, bn_momentum, ln_eps;
} glf_tpavi_params;
size_t glf_sizeof_tpavi_params(void);
size_t glf_s16_tpavi_workspace_bytes(const glf_tpavi_params* p, int pass);
int glf_s16_tpavi_fwd(const void* x, const void* w_qkv, const float* b_qkv, const void* w_z, const float* b_z,
                      const float* bn_gamma, const float* bn_beta, float* bn_running_mean, float* bn_running_var,
                      int64_t* num_batches_tracked, const float* ln_gamma, const float* ln_beta, void* z,
                      void* qkv, void* att_t, void* y, void* wz, float* bn_mean, float* bn_invstd, float* row_mean, float* row_rstd,
                      const glf_tpavi_params* p, void* workspace, size_t workspace_bytes, glf_stream_t stream);
int glf_s16_tpavi_bwd(const void* dz, const void* x, const void* qkv, const void* att_t, const void* y, const void* wz,
                      const float* bn_mean, const float* bn_invstd, const float* row_mean, const float* row_rstd,
                      const void* w_qkv_t, const void* w_z_t, const float* bn_gamma, const float* bn_beta, const float* ln_gamma,
                      void* dx, float* dw_qkv, float* db_qkv, float* dw_z, float* db_z, float* dbn_gamma, float* dbn_beta,
                      float* dln_gamma, float* dln_beta, const glf_tpavi_params* p, void* workspace, size_t workspace_bytes,
                      glf_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Stem (a2): Conv2d(1,64,7,stride 1,pad 2)+bias (models/_utils.py:192; used ours.py:1796).
 * x [N][H][W] (C=1), w [Cout][49], y [N][Ho][Wo][Cout] with Ho = H + 2*pad - 6.
 * ------------------------------------------------------------------------------------- */
int glf_stem7x7_fwd(const float* x, const float* w, const float* bias, float* y,
                    int n, int h, int wdt, int cout, int pad, glf_stream_t s);
/* Inference-mode stem in one launch (SURVEY 8b `stem7x7_bn_relu_pool`; ours.py:1725-1730, 1796: init_block = conv1, bn1, relu,
 * maxpool): y [N][Hp][Wp][Cout] = maxpool3x3s2p1(relu(bn(conv7x7(x) + bias))) with Hp = (Ho - 1) / 2 + 1 and the BatchNorm given by
 * mean / invstd (glf_bn_eval_coeffs of the running statistics), gamma, beta.  The conv output and its normalised copy are never
 * written.  Values equal the three-launch chain's bit for bit (same tap order, same BatchNorm expression).  amax_out (may be NULL):
 * as in glf_bn_apply.  Forward only: a train-mode step needs the batch statistics of the conv output before it can normalise, and
 * the conv output again in backward -- the three kernels stay. */
int glf_stem7x7_bn_relu_pool(const float* x, const float* w, const float* bias, const float* mean, const float* invstd,
                             const float* gamma, const float* beta, float* y, int n, int h, int wdt, int cout, int pad,
                             float* amax_out, glf_stream_t s);
/* dW [Cout][49] and db [Cout]; partial must hold glf_stem7x7_wgrad_workspace() floats. */
size_t glf_stem7x7_wgrad_workspace(int n, int h, int wdt, int cout, int pad);
int glf_stem7x7_wgrad(const float* x, const float* dy, float* dw, float* db, float* partial,
                      int n, int h, int wdt, int cout, int pad, glf_stream_t s);

/* ---------------------------------------------------------------------------------------
 * BatchNorm (nn.BatchNorm2d / BatchNorm3d in train and eval mode), channels-last [rows][C].
 * ------------------------------------------------------------------------------------- */
/* number of doubles of workspace for bn_stats / bn_bwd_reduce */
size_t glf_bn_workspace(int rows, int c);
/* Batch statistics over `rows`: writes mean[c], invstd[c] (biased var, eps) and, when
 * running_mean != NULL, updates running_mean/var with `momentum` (unbiased var), as
 * nn.BatchNorm does in train(). */
int glf_bn_stats(const float* x, int ldx, int rows, int c, float eps, float momentum,
                 float* mean, float* invstd, float* running_mean, float* running_var,
                 int64_t* num_batches_tracked /* may be NULL; += 1 */,
                 double* workspace, glf_stream_t s);
/* The same outputs / running-statistics update as glf_bn_stats from per-channel sums[2][c] = (sum x, sum x^2) over
 * `rows` rows -- the colstats a contraction epilogue accumulated. */
int glf_bn_stats_from_sums(const double* sums, int rows, int c, float eps, float momentum, float* mean, float* invstd,
                           float* running_mean, float* running_var, int64_t* num_batches_tracked, glf_stream_t s);
/* Replays the running-statistics update of a train-mode BatchNorm from its saved batch statistics (mean, invstd over
 * `rows` samples): running_* <- (1 - momentum) running_* + momentum {mean, unbiased var}, num_batches_tracked += 1.
 * The engine shares the ASPP trunk between the two classifier calls on the same f4 (ours.py:1806, 1840); the second
 * call's three-fold effect on the buffers is reproduced with this. */
int glf_bn_replay_running(const float* mean, const float* invstd, int rows, int c, float eps, float momentum,
                          float* running_mean, float* running_var, int64_t* num_batches_tracked, glf_stream_t s);
/* y = [relu]( (x - mean)*invstd*gamma + beta [+ residual] ).  For eval() pass running_mean and
 * 1/sqrt(running_var+eps) (glf_bn_eval_coeffs).  In-place (y == x) allowed.
 * amax_out (may be NULL): device float that must hold 0 (or any lower bound) before the call and
 * receives max(*amax_out, max|y|) -- the operand maximum the f16x3 contractions need, for free. */
int glf_bn_eval_coeffs(const float* running_mean, const float* running_var, float eps,
                       float* mean, float* invstd, int c, glf_stream_t s);
/* relu_mask (may be NULL): receives the sign of the pre-ReLU value, one byte per four consecutive channels (bit j = element
 * 4 i + j is positive; [rows][c / 4] dense) -- what glf_bn_bwd needs of the forward output when a residual was added: the
 * backward passes then read 1 byte where they read 16 (y is the largest tensor class of the path: the block outputs). */
int glf_bn_apply(const float* x, int ldx, const float* residual, int ldr, float* y, int ldy,
                 const float* mean, const float* invstd, const float* gamma, const float* beta,
                 int rows, int c, int relu, float* amax_out, uint8_t* relu_mask, glf_stream_t s);
/* glf_bn_stats_from_sums + glf_bn_apply in ONE launch (train mode, statistics from a contraction's colstats): every
 * workgroup finishes mean / invstd for all channels in LDS, workgroup 0 writes them to mean / invstd (for the backward pass)
 * and updates running_mean / running_var / num_batches_tracked (all three may be NULL).  Bit-identical to the two calls.
 * C <= 4096.
 * colmax (may be NULL; precision 3 / 4 callers): the per-channel maxima of |x| the same contraction epilogue left
 * (glf_gemm_params.colmax).  With it y is NOT written as fp32 but once, directly, as the packed pre-split image of
 * glf_split_f16_packed -- the form the next convolution reads in its forward AND its weight gradient (a_presplit /
 * b_presplit) -- scaled with the bound  max_c |gamma_c| invstd_c (max|x_c| + |mean_c|) + |beta_c|  >=  max |y|, which is
 * known before a single element of y is; *amax_out receives that bound (plain store).  Needs amax_out, no residual, y != x.
 * The fp32 y is then never materialised: the activation inside a bottleneck costs 4 B / element once instead of
 * 4 B (write) + 4 B (re-read by the split pass) + 4 B (the image). */
int glf_bn_apply_from_sums(const float* x, int ldx, const float* residual, int ldr, float* y, int ldy, const double* sums,
                           int rows, int c, float eps, float momentum, const float* gamma, const float* beta,
                           float* mean, float* invstd, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                           int relu, float* amax_out, uint8_t* relu_mask, const float* colmax, glf_stream_t s);
/* Backward.  y is the forward output (ReLU mask = y > 0); it may be NULL when relu == 0, and also when
 * relu != 0 and there was NO residual: the mask is then recomputed from x with beta (one tensor read less
 * in both passes; with a residual the sign of y depends on it, so y is required).
 * training != 0: full batch-stat backward; training == 0: dx = dy*mask*gamma*invstd.
 * dres (may be NULL) receives dy*mask (gradient of the residual input).  amax_out: as in
 * glf_bn_apply, for dx.
 * packed_dx != 0 (precision 3 / 4 callers): dx is NOT written as fp32 but directly as the packed pre-split image of
 * glf_split_f16_packed -- what the producing convolution's dgrad and wgrad read (glf_gemm_params.a_presplit) -- so the
 * gradient of a conv output exists once, in the form its consumers want, and the separate split pass (one read + one write
 * of the tensor) is gone.  The image's power-of-two scale must be known before the first element is written: the reduction
 * pass also takes max|dy'| and max|xhat| per channel, and *amax_out (required, zeroed by the caller) receives the upper
 * bound of max|dx| derived from them (typically within 2x of the true maximum); pass the same scalar as amax_a.
 * dy2 (may be NULL; row stride lddy2): a second addend of the incoming gradient -- the node sees dy + dy2.  The input of a
 * residual block feeds its shortcut and its first conv (models/resnet.py:59-79), so the gradient that reaches the previous
 * block's last BatchNorm is a sum of two tensors: both passes add them while reading instead of a separate add kernel
 * writing the sum (3 tensor passes) that both passes then read (2 more). */
int glf_bn_bwd(const float* dy, int lddy, const float* x, int ldx, const float* y, int ldy,
               const float* mean, const float* invstd, const float* gamma, const float* beta /* may be NULL with y */,
               float* dx, int lddx, float* dres, int lddres, float* dgamma, float* dbeta,
               int rows, int c, int relu, int training, double* workspace, float* amax_out, int packed_dx,
               const uint8_t* relu_mask /* glf_bn_apply's sign bytes: replaces y */, const float* dy2, int lddy2,
               double* fused_sums /* may be NULL.  Non-NULL (C <= 4096): 2 C doubles followed by 2 C floats, ALL ZERO on entry -- the
                                     backward then takes TWO launches: the reduction meets in one f64 atomic per column and workgroup
                                     there, the apply kernel finishes the sums itself (no finalize launch; `workspace` may be NULL).
                                     dgamma / dbeta agree with the three-launch form to fp32 rounding, not bit for bit. */,
               glf_stream_t s);

/* ---------------------------------------------------------------------------------------
 * Pooling / resampling / pointwise pieces of the path.
 * ------------------------------------------------------------------------------------- */
/* nn.MaxPool2d(3, stride 2, pad 1) on [N][H][W][C] -> [N][Ho][Wo][C]; idx (uint8, same shape as
 * y) records the winning tap (first maximum in scan order, as ATen). */
int glf_maxpool3x3s2_fwd(const float* x, float* y, uint8_t* idx, int n, int h, int w, int c, glf_stream_t s);
int glf_maxpool3x3s2_bwd(const float* dy, const uint8_t* idx, float* dx, int n, int h, int w, int c, glf_stream_t s);
/* AdaptiveAvgPool2d(1) (deeplabv3.py:126): x [N][P][C] -> y [N][C], and its backward. */
int glf_avgpool_fwd(const float* x, float* y, int n, int p, int c, glf_stream_t s);
/* broadcast rows: y[n][p][0..c) (row stride ldy) = x[n][0..c) -- the "bilinear from 1x1"
 * upsample of ASPPPooling (deeplabv3.py:135). */
int glf_bcast_rows_fwd(const float* x, float* y, int ldy, int n, int p, int c, glf_stream_t s);
/* dx[n][c] = scale * sum_p dy[n][p][c] (row stride lddy): backward of the broadcast (scale=1)
 * and, with scale = 1/p on a dense dy, nothing else; avgpool backward is bcast with scale 1/p. */
int glf_sum_rows_fwd(const float* dy, int lddy, float* dx, float scale, int n, int p, int c, glf_stream_t s);
int glf_bcast_rows_scaled(const float* x, float* y, int ldy, float scale, int n, int p, int c, glf_stream_t s);
/* Stand-alone nn.ReLU (the fused paths apply it inside glf_bn_apply). */
int glf_relu_fwd(const float* x, float* y, int64_t numel, glf_stream_t s);
int glf_relu_bwd(const float* dy, const float* y, float* dx, int64_t numel, glf_stream_t s);
/* Dropout with a counter-based generator: keep iff hash(seed', element) >= p.  y = x*keep/(1-p).
 * seed' = seed + *step_counter * odd constant when step_counter (a DEVICE uint64, may be NULL) is given: a launch recorded in
 * a hipGraph replays with the same `seed` argument, the counter -- advanced once per training step by glf_counter_add, itself
 * part of the graph -- gives every replay its own masks (deeplabv3.py:159 Dropout(0.5), a fresh mask per forward). */
int glf_dropout(const float* x, float* y, int64_t numel, float p, uint64_t seed, const uint64_t* step_counter, glf_stream_t s);
/* Operand-maximum bookkeeping for tensors whose maximum follows from their sources' (no pass over the data):
 * *out = max(*out, scale * (sum ? |*a| + |*b| : max(|*a|, |*b|))); b may be NULL.  A dropout output is bounded by its input's
 * maximum / (1 - p), a sum of two tensors by the sum of their maxima, a stack of tensors by the largest.  Upper bounds are
 * what glf_gemm_params.amax_a / amax_b ask for. */
int glf_amax_combine(const float* a, const float* b, float scale, int sum, float* out, glf_stream_t s);
/* *counter += inc (a device uint64), stream-ordered. */
int glf_counter_add(uint64_t* counter, uint64_t inc, glf_stream_t s);
/* Local gate (a5, ours.py:1802-1816): a[r] = sigmoid(w * max_c sigmoid(cls[r][c]) * sigmoid(ctr[r])),
 * y[r][:] = f[r][:] * a[r].  argmax (int32 per row) is saved for backward. */
int glf_gate_fwd(const float* cls, int ncls, const float* ctr, const float* f, float* y, float* a,
                 int32_t* argmax, float weight, int rows, int c, glf_stream_t s);
/* df[r][:] = dy[r][:]*a[r]; dcls/dctr from da[r] = sum_c dy[r][c]*f[r][c]. */
int glf_gate_bwd(const float* dy, const float* f, const float* cls, int ncls, const float* ctr,
                 const float* a, const int32_t* argmax, float weight,
                 float* df, float* dcls, float* dctr, int rows, int c, glf_stream_t s);
/* Stack / slice views for the fusion block (ours.py:1819-1834): frame-strided row copies.
 * dst[n][.] (frame stride dst_fs) = src[n][.] (frame stride src_fs), `inner` floats per frame. */
int glf_copy_frames(const float* src, int64_t src_fs, float* dst, int64_t dst_fs,
                    int n, int64_t inner, glf_stream_t s);
/* glf_copy_frames that ALSO writes the packed pre-split image (glf_split_f16_packed's format, same layout as dst) of what it
 * copies, scaled by *amax -- a device float that already holds an upper bound of max|src| (e.g. glf_amax_combine of the sources'
 * maxima).  ours.py:1819-1820: the stacked fusion-block input is read once, by the copy, instead of again by a split pass. */
int glf_copy_frames_split(const float* src, int64_t src_fs, float* dst, float* dst_packed, int64_t dst_fs, int n, int64_t inner,
                          const float* amax, glf_stream_t s);
/* dst = a + b with independent frame strides (f4_fusion = f4_global_fusion + f4_local_fusion). */
int glf_add_frames(const float* a, int64_t a_fs, const float* b, int64_t b_fs, float* dst, int64_t dst_fs,
                   int n, int64_t inner, glf_stream_t s);
/* out = sum of k (<= 8) same-sized tensors given as a HOST array of device pointers: the gradient fan-in of a
 * tensor that feeds several branches (f4 -> classifier / centerness / gate / fusion; ASPP input -> 5 branches). */
int glf_add_n(const float* const* inputs, int k, float* out, int64_t numel, glf_stream_t s);
/* W_z tail of TPAVIModule (ours.py:908-915): z = LayerNorm_C( BN(w) + x ) with the BN already
 * folded into per-channel (bn_mean, bn_invstd, gamma, beta).  Saves row mean / rstd.  amax_out (may be NULL): as in
 * glf_bn_apply, receives max(*amax_out, max|z|). */
int glf_bn_res_ln_fwd(const float* w, const float* x, const float* bn_mean, const float* bn_invstd,
                      const float* bn_gamma, const float* bn_beta, const float* ln_gamma,
                      const float* ln_beta, float ln_eps, float* z, float* row_mean, float* row_rstd,
                      int rows, int c, float* amax_out, glf_stream_t s);
/* LayerNorm backward: du (gradient w.r.t. u = BN(w)+x), dgamma/dbeta of the LayerNorm.
 * workspace: glf_bn_workspace(rows, c) doubles. */
int glf_bn_res_ln_bwd(const float* dz, const float* w, const float* x, const float* bn_mean,
                      const float* bn_invstd, const float* bn_gamma, const float* bn_beta,
                      const float* ln_gamma, const float* row_mean, const float* row_rstd,
                      float* du, float* dln_gamma, float* dln_beta, int rows, int c,
                      double* workspace, glf_stream_t s);
/* Row softmax in place over `cols` (TPAVI 'embedded' mode, ours.py:896-897) and its backward
 * ds = p * (dp - sum(dp*p)). */
int glf_softmax_rows(float* x, int64_t rows, int cols, glf_stream_t s);
int glf_softmax_rows_bwd(const float* p, float* dp_inout, int64_t rows, int cols, glf_stream_t s);
/* The same over rows of stride ld >= cols; the padding columns [cols, ld) are written as zeros (zero probabilities / zero score
 * gradients: the matrices then serve as contraction operands with K = ld). */
int glf_softmax_rows_ld(float* x, int64_t rows, int cols, int ld, glf_stream_t s);
int glf_softmax_rows_bwd_ld(const float* p, float* dp_inout, int64_t rows, int cols, int ld, glf_stream_t s);
/* F.interpolate(mode='bilinear', align_corners=False) (ours.py:1838,1841): x [N][h][w][C]
 * channels-last -> y [N][C][H][W] (NCHW, what the caller's loss consumes), and its adjoint. */
int glf_bilinear_up_fwd(const float* x, float* y, int n, int h, int w, int c, int ho, int wo, glf_stream_t s);
int glf_bilinear_up_bwd(const float* dy, float* dx, int n, int h, int w, int c, int ho, int wo, glf_stream_t s);
/* nn.BCEWithLogitsLoss(reduction='sum') (main.py:87,209-211): loss_out (double[1], zeroed by the
 * call) and, when dx != NULL, dx = (sigmoid(x) - t) * grad_scale * (*grad_scale_dev), the latter a
 * device scalar (autograd's upstream gradient) that may be NULL. */
int glf_bce_logits_sum(const float* x, const float* t, double* loss_out, float* dx, float grad_scale,
                       const float* grad_scale_dev, int64_t numel, glf_stream_t s);
/* Trainer._calculate_overlap_metrics (main.py:800-815) on pred = sigmoid(x) > 0.5:
 * counts[4] = {tp, fp, fn, tn} as int64 (zeroed by the call). */
int glf_overlap_counts(const float* logits, const float* target, int64_t* counts, int64_t numel, glf_stream_t s);
/* bias gradient: db[c] = sum_r dy[r][c]; workspace glf_bn_workspace(rows, c) doubles. */
int glf_colsum(const float* dy, int lddy, float* db, int rows, int c, double* workspace, glf_stream_t s);

/* ---------------------------------------------------------------------------------------
 * Optimizer: torch.optim.Adam as main.py:162-165 builds it (L2 weight decay folded into the
 * gradient, no amsgrad), every parameter of one step count in ONE launch.  `table` is a DEVICE array of
 * n_rows x 5 int64 { param ptr, grad ptr, exp_avg ptr, exp_avg_sq ptr, n elements }, one row per chunk of a
 * parameter; `step` = the parameters' step count t, counted from 1 (torch keeps one per parameter; parameters
 * with different counts go into separate calls).  The bias corrections, 1 - beta and lr / (1 - beta1^t) are
 * formed in double on the host as torch does.
 * Updates param, exp_avg and exp_avg_sq in place with the operation order of torch.optim._functional.adam.
 * ------------------------------------------------------------------------------------- */
int glf_adam_step(const int64_t* table, int n_rows, double lr, double beta1, double beta2, double eps,
                  double weight_decay, int64_t step, glf_stream_t s);

/* ---------------------------------------------------------------------------------------
 * Temporal cycle-consistency loss: Trainer.seg_cycle and Trainer.dense_seg_cycle
 * (main.py:650-718, 720-798) on the pooled fusion features feat [T][F] (main.py:226-231: T = 40,
 * target_region 16, cyc_off 2, chunk_size 3, temperature 10).  *loss_out = weight * sum over the start frames
 * start0, start0 + stride, ... (n_starts of them) of the mean BCE-with-logits; dfeat [T][F] (may be NULL)
 * receives d loss / d feat.  seg_cycle: n_starts = 1, weight = 1, start0 = the frame the reference draws with
 * np.random.choice (an explicit input here).  dense_seg_cycle: start0 = 0, stride = 1 (is_overlap) or
 * chunk_size, weight = 1 / (target_region - chunk_size - cyc_off + 1); soft_label as main.py:790-791.
 * ------------------------------------------------------------------------------------- */
int glf_seg_cycle(const float* feat, int T, int F, int target_region, int cyc_off, int chunk_size, float temperature,
                  int start0, int n_starts, int stride, float weight, int soft_label, float* loss_out, float* dfeat,
                  glf_stream_t s);
/* out = a*x + b*y (out may alias x or y): the background branch f4 * (1 - gate) = f4 - f4 * gate of
 * Foreground_and_Background (ours.py:2966) and its gradient. */
int glf_axpby(const float* x, const float* y, float* out, float a, float b, int64_t numel, glf_stream_t s);
/* y = x * scale * (*scale_dev) (scale_dev: device scalar, may be NULL): a saved gradient times autograd's
 * upstream gradient. */
int glf_scale(const float* x, float* y, int64_t numel, float scale, const float* scale_dev, glf_stream_t s);
/* ---------------------------------------------------------------------------------------
 * Data path and evaluation harness.
 * glf_prepare_frames: one raw volume [H0][W0][T] (T fastest; img = grey levels, lab = class ids 0..4, either may be
 * NULL with its output) -> frames out_img [T][1][out_h][out_w] = img / img_div (255 for labelled, 1 for unlabelled clips,
 * loader.py:325-327) and part masks out_mask [T][5][out_h][out_w]: nearest-neighbour resize to resize x resize
 * (Resized(mode='nearest'), loader.py:478, 488), crop window at (off_y, off_x) (RandSpatialCropd / CenterSpatialCropd,
 * loader.py:480, 489: centre = (resize - out) / 2), class id k = 1..4 -> channel class_to_channel[k-1] (HOST array of 4
 * ints, -1 = dropped; the per-view tables of loader.py:298-316 + mask_to_allclass, loader.py:358-414).
 * glf_overlap_counts_nchw: counts[c][4] = (tp, fp, fn, tn) of sigmoid(logits) > 0.5 vs target per class channel of
 * [n][c][hw] tensors (the per-part Dice of main.py:537-543); counts is zeroed by the call.
 * ------------------------------------------------------------------------------------- */
int glf_prepare_frames(const float* img, const float* lab, float* out_img, float* out_mask, int H0, int W0, int T,
                       int resize, int out_h, int out_w, int off_y, int off_x, const int* class_to_channel, float img_div,
                       glf_stream_t s);
int glf_overlap_counts_nchw(const float* logits, const float* target, int64_t* counts, int n, int c, int64_t hw, glf_stream_t s);
/* Zero `bytes` bytes at p on the stream (accumulation targets: atomically summed gradients, column statistics, maxima). */
int glf_zero(void* p, int64_t bytes, glf_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* GLFUSION_H */
