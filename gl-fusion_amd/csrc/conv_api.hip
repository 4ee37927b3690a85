// glf_conv2d_{fwd,dgrad,wgrad}: the convolution entry points of the C ABI.  They embed the launch policy a caller of the
// generic contraction interface (glf_gemm_*) would otherwise have to re-derive: which taps of a dilated kernel can touch
// the map at all (rate 36 on 28 x 28: the centre tap only), when most of the remaining tap work is padding and the taps
// should run as GEMMs over exactly their in-range rectangles (rect) or over regions with a constant tap set (region, no
// atomics), how many reduction slices the weight gradient is cut into, and which outputs must be zero-filled first.
// NHWC activations, tap-major weights [kh*kw][Cout][Cin] (glf_oihw_to_tap_major).  F.conv2d forward / backward at
// models/_utils.py:192 (stem excluded: glf_stem7x7_*), torchvision Bottleneck convs (ours.py:1797-1800), deeplabv3.py:104-165.
#include "gemm_common.h"
#include <cstring>

namespace {

int conv_out(int h, int k, int stride, int pad, int dil) { return (h + 2 * pad - dil * (k - 1) - 1) / stride + 1; }

// bit t set = tap t can be in range for some destination pixel.  gather 1: destination = output map (forward, wgrad);
// gather 2: destination = input map (dgrad)
unsigned tap_mask_of(int gather, int hd, int wd, int hs, int ws, int kh, int kw, int stride, int pad, int dil) {
    auto axis = [&](int nd, int ns, int k, bool* ok) {
        for (int t = 0; t < k; ++t) {
            ok[t] = false;
            for (int y = 0; y < nd && !ok[t]; ++y) {
                if (gather == 1) {
                    const int sy = y * stride - pad + t * dil;
                    ok[t] = sy >= 0 && sy < ns;
                } else {
                    const int v = y + pad - t * dil;
                    ok[t] = v >= 0 && v % stride == 0 && v / stride < ns;
                }
            }
        }
    };
    bool vy[32], vx[32];
    axis(hd, hs, kh, vy);
    axis(wd, ws, kw, vx);
    unsigned m = 0;
    for (int ky = 0; ky < kh; ++ky)
        for (int kx = 0; kx < kw; ++kx)
            if (vy[ky] && vx[kx]) m |= 1u << (ky * kw + kx);
    return m;
}

// fraction of the kept taps' work that is in range (stride 1): small = mostly padding
double rect_fraction(int gather, int hd, int wd, int hs, int ws, int kw, int pad, int dil, unsigned mask) {
    long long tot = 0;
    int n = 0;
    for (unsigned mm = mask; mm; mm &= mm - 1) {
        const int t = __builtin_ctz(mm);
        int y0, y1, x0, x1;
        tap_rect(gather, t, kw, pad, dil, hs, ws, hd, wd, y0, y1, x0, x1);
        tot += (long long)(y1 - y0) * (x1 - x0);
        ++n;
    }
    return n ? (double)tot / ((double)n * hd * wd) : 1.0;
}

// reduction slices of the weight gradient: ~1024 (split-fp16: 256-wide tiles, one workgroup per CU) or ~2048 workgroups,
// >= 512 rows per slice; for the split-fp16 kernels the count near the target whose workgroups fill whole rounds of the CUs best
int tn_split(long long rows, int m, int n, int ntaps, int prec) {
    const long long tiles = (long long)((m + 127) / 128) * ((n + 127) / 128) * (ntaps > 0 ? ntaps : 1);
    const long long target = prec >= 2 ? 1024 : 2048;
    long long want = (target + tiles - 1) / tiles;
    if (want < 1) want = 1;
    long long cap = rows / 512;
    if (cap < 1) cap = 1;
    const long long hi = 65535;
    if (prec >= 2) {
        const long long wg = (long long)((m + 255) / 256) * ((n + 127) / 128) * (ntaps > 0 ? ntaps : 1);
        long long best = want;
        double best_score = -1.0;
        long long lo = want / 2 > 1 ? want / 2 : 1, up = want * 2;
        if (up > cap) up = cap;
        if (up > hi) up = hi;
        if (up < 1) up = 1;
        for (long long s = lo; s <= up; ++s) {
            const long long b = wg * s;
            const double score = (double)b / ((double)((b + 255) / 256) * 256.0) - 0.004 * (double)s;
            if (score > best_score) { best = s; best_score = score; }
        }
        want = best;
    }
    if (want > cap) want = cap;
    if (want > hi) want = hi;
    return (int)(want < 1 ? 1 : want);
}

int check_params(const glf_conv_params* p, const char* who) {
    GLF_REQUIRE(p != nullptr, GLF_ERR_NULL, "%s: null parameter block", who);
    GLF_REQUIRE(p->n > 0 && p->h > 0 && p->w > 0 && p->cin > 0 && p->cout > 0, GLF_ERR_BAD_SHAPE, "%s: extents must be > 0", who);
    GLF_REQUIRE(p->kh >= 1 && p->kw >= 1 && p->kh * p->kw <= 32 && p->stride >= 1 && p->dil >= 1 && p->pad >= 0, GLF_ERR_BAD_SHAPE,
                "%s: bad kernel geometry (kh*kw must be <= 32)", who);
    GLF_REQUIRE(p->precision >= 0 && p->precision <= 4, GLF_ERR_UNSUPPORTED, "%s: precision must be 0..4", who);
    GLF_REQUIRE(conv_out(p->h, p->kh, p->stride, p->pad, p->dil) > 0 && conv_out(p->w, p->kw, p->stride, p->pad, p->dil) > 0, GLF_ERR_BAD_SHAPE,
                "%s: empty output", who);
    return GLF_OK;
}

int effective_precision(const glf_conv_params* p) { return p->precision >= 1 ? p->precision - 1 : glf::precision(); }

constexpr double THR_FWD = 0.8, THR_DGRAD = 0.8, THR_DGRAD_F16 = 0.35, THR_WGRAD = 0.8, THR_REGION = 0.8;

void fill_geo(glf_gemm_params& g, const glf_conv_params* p, bool to_input, int ho, int wo) {
    // destination grid = what GEMM rows enumerate; source grid = what is gathered
    g.n_img = p->n;
    g.kh = p->kh; g.kw = p->kw; g.stride = p->stride; g.pad = p->pad; g.dil = p->dil;
    if (to_input) { g.hs = ho; g.ws = wo; g.hd = p->h; g.wd = p->w; }
    else { g.hs = p->h; g.ws = p->w; g.hd = ho; g.wd = wo; }
}

}  // namespace

extern "C" int glf_conv2d_plan(const glf_conv_params* p, int pass, glf_conv_plan* plan) {
    if (int rc = check_params(p, "conv2d_plan")) return rc;
    GLF_REQUIRE(plan != nullptr && pass >= 0 && pass <= 2, GLF_ERR_NULL, "conv2d_plan: plan missing or pass not in {0 fwd, 1 dgrad, 2 wgrad}");
    std::memset(plan, 0, sizeof(*plan));
    const int prec = effective_precision(p);
    const int ho = conv_out(p->h, p->kh, p->stride, p->pad, p->dil), wo = conv_out(p->w, p->kw, p->stride, p->pad, p->dil);
    const int taps = p->kh * p->kw;
    const bool plain = taps == 1 && p->stride == 1 && p->pad == 0;
    plan->ho = ho; plan->wo = wo; plan->taps = taps; plan->plain = plain; plan->split = 1;
    if (pass == 0 || pass == 2) {
        const unsigned mask = plain ? 1u : tap_mask_of(1, ho, wo, p->h, p->w, p->kh, p->kw, p->stride, p->pad, p->dil);
        const int kept = __builtin_popcount(mask);
        const double frac = (plain || p->stride != 1) ? 1.0 : rect_fraction(1, ho, wo, p->h, p->w, p->kw, p->pad, p->dil, mask);
        plan->tap_mask = mask; plan->kept_taps = kept;
        if (pass == 0) {
            plan->M = p->n * ho * wo; plan->N = p->cout; plan->K = p->cin;
            plan->rect = (!plain && taps > 1 && p->stride == 1 && kept > 1 && frac < THR_FWD) ? 1 : 0;
            plan->zero_fill = plan->rect == 1;
            plan->colstats_ok = prec >= 2 && p->cin % 32 == 0 && p->cout % 4 == 0 && plan->rect == 0;
        } else {
            const bool rect = !plain && taps > 1 && p->stride == 1 && kept > 1 && frac < THR_WGRAD;
            const long long rows_o = (long long)p->n * ho * wo;
            long long eff = rect ? (long long)((double)rows_o * frac) : rows_o;
            if (eff < 512) eff = 512;
            plan->M = p->cout; plan->N = p->cin; plan->K = (int)rows_o;
            plan->rect = rect ? 1 : 0;
            plan->split = tn_split(eff, p->cout, p->cin, kept, prec);
            if (rect) {
                // slices are cut from the full reduction length (glf_gemm_params.split): a tap uses rows_tap / chunk of them, the
                // kept taps together ~ kept * frac * split -- scale up so that the launch still fills the chip
                long long s2 = (long long)((double)plan->split / (2.0 * (frac > 0.02 ? frac : 0.02)) + 0.999);      // half of split / frac: measured best
                if (s2 < plan->split) s2 = plan->split;
                const long long cap = rows_o / 512 > 1 ? rows_o / 512 : 1;
                if (s2 > cap) s2 = cap;
                if (s2 > 65535) s2 = 65535;
                plan->split = (int)(s2 < 1 ? 1 : s2);
            }
            plan->workspace_bytes = plan->split > 1 ? (int64_t)plan->split * kept * p->cout * p->cin * (int64_t)sizeof(float) : 0;
            plan->zero_fill = mask != ((taps == 32) ? 0xffffffffu : ((1u << taps) - 1u));     // taps outside the mask stay zero
        }
    } else {
        const unsigned mask = plain ? 1u : tap_mask_of(2, p->h, p->w, ho, wo, p->kh, p->kw, p->stride, p->pad, p->dil);
        const int kept = __builtin_popcount(mask);
        const double frac = (plain || p->stride != 1) ? 1.0 : rect_fraction(2, p->h, p->w, ho, wo, p->kw, p->pad, p->dil, mask);
        plan->tap_mask = mask; plan->kept_taps = kept;
        plan->M = p->n * p->h * p->w; plan->N = p->cin; plan->K = p->cout;
        const bool region = prec >= 2 && !plain && kept > 1 && taps == 9 && p->kh == 3 && p->stride == 1 && p->pad == p->dil && ho == p->h &&
                            wo == p->w && p->cout % 32 == 0 && frac < THR_REGION;
        const double thr = prec >= 2 ? THR_DGRAD_F16 : THR_DGRAD;
        plan->rect = region ? 2 : ((!plain && taps > 1 && p->stride == 1 && kept > 1 && frac < thr) ? 1 : 0);
        plan->zero_fill = plan->rect == 1 || mask == 0;
    }
    return GLF_OK;
}

extern "C" int glf_conv2d_fwd(const float* x, const float* w_tap, const float* bias, float* y, const glf_conv_params* p, glf_stream_t s) {
    glf_conv_plan pl;
    if (int rc = glf_conv2d_plan(p, 0, &pl)) return rc;
    GLF_REQUIRE(x && w_tap && y, GLF_ERR_NULL, "conv2d_fwd: null argument");
    GLF_REQUIRE(!(pl.rect && bias), GLF_ERR_UNSUPPORTED, "conv2d_fwd: a bias on a conv that runs as per-tap rectangles is not built (none on the path)");
    GLF_REQUIRE(!p->colstats || pl.colstats_ok, GLF_ERR_UNSUPPORTED, "conv2d_fwd: colstats cannot be honoured for this conv / precision (see glf_conv_plan.colstats_ok)");
    if (pl.zero_fill) {
        hipError_t e = hipMemsetAsync(y, 0, (size_t)pl.M * pl.N * sizeof(float), glf::S(s));
        if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "conv2d_fwd: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    glf_gemm_params g;
    std::memset(&g, 0, sizeof(g));
    g.M = pl.M; g.N = pl.N; g.K = pl.K; g.lda = p->cin; g.ldb = p->cin; g.ldc = p->cout;
    g.taps = pl.taps; g.tap_mask = pl.tap_mask; g.tap_stride_b = (int64_t)p->cout * p->cin;
    g.gather = pl.plain ? 0 : 1;
    fill_geo(g, p, false, pl.ho, pl.wo);
    if (pl.plain) { g.n_img = 1; g.hs = g.ws = g.hd = g.wd = 1; g.kh = g.kw = 1; g.stride = 1; g.pad = 0; g.dil = 1; }
    g.batch = 1; g.alpha = 1.f; g.split = 1; g.rect = pl.rect;
    g.amax_a = p->amax_x; g.amax_b = p->amax_w; g.amax_c = p->amax_out; g.colstats = p->colstats; g.precision = p->precision;
    if (pl.rect == 1 && effective_precision(p) == 0) {
        // exact fp32: the rectangles one tap per launch.  In one launch the taps of an output pixel meet in float atomics in whatever
        // order the workgroups finish; launched tap after tap every element receives its (at most nine) addends in tap order -- the
        // forward pass of the strict-precision leg is reproducible, with the rectangles' short (K-long) fp32 chains kept.  (The
        // order-dependent last bits were harmless in the forward pass and came back from the model's backward pass as 2-8e-3
        // run-to-run changes of whole gradient tensors: ops.Conv2dFn.forward, profiles/r04_exact_leg_spread.txt.)
        for (unsigned mm = pl.tap_mask; mm; mm &= mm - 1) {
            g.tap_mask = mm & (~mm + 1u);
            if (int rc = glf_gemm_nt(x, w_tap, bias, y, &g, s)) return rc;
        }
        return GLF_OK;
    }
    return glf_gemm_nt(x, w_tap, bias, y, &g, s);
}

extern "C" int glf_conv2d_dgrad(const float* dy, const float* w_tap, const float* w_tap_t, float* dx, const glf_conv_params* p, glf_stream_t s) {
    glf_conv_plan pl;
    if (int rc = glf_conv2d_plan(p, 1, &pl)) return rc;
    GLF_REQUIRE(dy && dx && (w_tap || w_tap_t), GLF_ERR_NULL, "conv2d_dgrad: null argument");
    if (pl.zero_fill) {
        hipError_t e = hipMemsetAsync(dx, 0, (size_t)pl.M * pl.N * sizeof(float), glf::S(s));
        if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "conv2d_dgrad: hipMemsetAsync: %s", hipGetErrorString(e));
        if (pl.tap_mask == 0) return GLF_OK;
    }
    const int prec = effective_precision(p);
    glf_gemm_params g;
    std::memset(&g, 0, sizeof(g));
    g.M = pl.M; g.N = pl.N; g.K = pl.K; g.lda = p->cout; g.ldc = p->cin;
    g.taps = pl.taps; g.tap_mask = pl.tap_mask; g.tap_stride_b = (int64_t)p->cout * p->cin;
    g.gather = pl.plain ? 0 : 2;
    fill_geo(g, p, true, pl.ho, pl.wo);
    if (pl.plain) { g.n_img = 1; g.hs = g.ws = g.hd = g.wd = 1; g.kh = g.kw = 1; g.stride = 1; g.pad = 0; g.dil = 1; }
    g.batch = 1; g.alpha = 1.f; g.split = 1; g.rect = pl.rect;
    g.amax_a = p->amax_dy; g.amax_b = p->amax_w; g.amax_c = p->amax_out; g.precision = p->precision;
    if (prec >= 1 && p->cout % 32 == 0 && w_tap_t) {          // split kernels are NT / TN only: B_tap[n = ci][k = co]
        g.ldb = p->cout;
        return glf_gemm_nt(dy, w_tap_t, nullptr, dx, &g, s);
    }
    GLF_REQUIRE(w_tap != nullptr, GLF_ERR_NULL, "conv2d_dgrad: the exact-fp32 path needs the [tap][Cout][Cin] weights");
    GLF_REQUIRE(pl.rect != 2, GLF_ERR_UNSUPPORTED, "conv2d_dgrad: region mode needs the transposed weights (w_tap_t)");
    g.ldb = p->cin;
    g.precision = 1;                                           // NN form exists on the exact kernels only
    g.amax_a = g.amax_b = nullptr; g.amax_c = nullptr;
    return glf_gemm_nn(dy, w_tap, nullptr, dx, &g, s);
}

extern "C" int glf_conv2d_wgrad(const float* dy, const float* x, float* dw_tap, float* workspace, int64_t workspace_bytes,
                                const glf_conv_params* p, glf_stream_t s) {
    glf_conv_plan pl;
    if (int rc = glf_conv2d_plan(p, 2, &pl)) return rc;
    GLF_REQUIRE(dy && x && dw_tap, GLF_ERR_NULL, "conv2d_wgrad: null argument");
    const bool two_stage = pl.split > 1 && workspace != nullptr;
    GLF_REQUIRE(!two_stage || workspace_bytes >= pl.workspace_bytes, GLF_ERR_WORKSPACE, "conv2d_wgrad: workspace of %lld bytes, plan asks for %lld",
                (long long)workspace_bytes, (long long)pl.workspace_bytes);
    if (pl.zero_fill || (pl.split > 1 && !two_stage)) {
        hipError_t e = hipMemsetAsync(dw_tap, 0, (size_t)pl.taps * p->cout * p->cin * sizeof(float), glf::S(s));
        if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "conv2d_wgrad: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    if (pl.tap_mask == 0) return GLF_OK;
    glf_gemm_params g;
    std::memset(&g, 0, sizeof(g));
    g.M = pl.M; g.N = pl.N; g.K = pl.K; g.lda = p->cout; g.ldb = p->cin; g.ldc = p->cin;
    g.taps = pl.taps; g.tap_mask = pl.tap_mask; g.tap_stride_b = (int64_t)p->cout * p->cin;
    g.gather = pl.plain ? 0 : 1;
    fill_geo(g, p, false, pl.ho, pl.wo);
    // the TN form gathers B's rows: source grid = the input map, destination grid = the output map (rows of dY)
    if (pl.plain) { g.n_img = 1; g.hs = g.ws = g.hd = g.wd = 1; g.kh = g.kw = 1; g.stride = 1; g.pad = 0; g.dil = 1; }
    g.batch = 1; g.alpha = 1.f; g.split = pl.split; g.rect = pl.rect;
    g.amax_a = p->amax_dy; g.amax_b = p->amax_x; g.precision = p->precision;
    if (two_stage) { g.workspace = workspace; g.workspace_bytes = workspace_bytes; }
    return glf_gemm_tn(dy, x, dw_tap, &g, s);
}
