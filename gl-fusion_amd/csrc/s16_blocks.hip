// Single-call block entry points of the 16-bit-storage path (`tpavi_proj` + `tpavi_attn_dot` + `tpavi_out_bn_res_ln` as
// ONE call per direction): a C host runs TPAVIModule.forward / backward (reference models/ours.py:845-917, mode 'dot') without
// composing the contractions itself.  Host code only -- the kernels are the ones of gemm_s16.hip / s16_ops.hip / norm.hip; every
// buffer (results, saved-for-backward tensors, workspace) is caller-owned, the calls only enqueue on `stream`.
#include "glf_common.h"
#include <cstring>

namespace {

inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// reduction slices of a plain weight gradient [M][N] over `rows` (mirror of glfusion_amd.ops16.tn_split16)
int tn_split16(long long rows, int m, int n) {
    const long long tiles = (long long)((m + 255) / 256) * ((n + 127) / 128);
    long long want = (512 + tiles - 1) / tiles;
    if (want < 1) want = 1;
    long long cap = rows / 512;
    if (cap < 1) cap = 1;
    if (cap > 65535) cap = 65535;
    long long best = want < cap ? want : cap;
    double best_score = -1.0;
    const long long lo = want / 2 > 1 ? want / 2 : 1, hi = 2 * want < cap ? 2 * want : cap;
    for (long long sp = lo; sp <= (hi > 1 ? hi : 1); ++sp) {
        const long long b = tiles * sp;
        const double score = (double)b / ((double)((b + 255) / 256) * 256.0) - 0.01 * (double)sp;
        if (score > best_score) { best = sp; best_score = score; }
    }
    if (best > cap) best = cap;
    return (int)(best < 1 ? 1 : best);
}

glf_gemm_params plain(int M, int N, int K, int lda, int ldb, int ldc, int c_dtype) {
    glf_gemm_params p;
    std::memset(&p, 0, sizeof(p));
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.taps = 1; p.tap_mask = 1; p.batch = 1; p.alpha = 1.0f; p.split = 1;
    p.n_img = p.hs = p.ws = p.hd = p.wd = p.kh = p.kw = p.stride = p.dil = 1;
    p.c_dtype = c_dtype;
    return p;
}

struct BwdLayout { size_t dwz, dy, dqkv, att, dM, dMT, slab, sums, colsum, split, total; int sp_z, sp_qkv; };

BwdLayout bwd_layout(const glf_tpavi_params* p) {
    BwdLayout w;
    const size_t rows = (size_t)p->n * p->L, c = p->c, ci = p->ci;
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += al256(bytes); return o; };
    w.dwz = take(rows * c * 2);
    w.dy = take(rows * ci * 2);
    w.dqkv = take(rows * 3 * ci * 2);
    w.att = take((size_t)p->n * ci * ci * 2);
    w.dM = take((size_t)p->n * ci * ci * 2);
    w.dMT = take((size_t)p->n * ci * ci * 2);
    w.slab = take(glf_s16_bn_res_ln_workspace((int)rows, (int)c));
    w.sums = take(2 * c * sizeof(double));
    w.colsum = take(2 * 3 * ci * sizeof(double));
    w.sp_z = tn_split16((long long)rows, (int)c, (int)ci);
    w.sp_qkv = tn_split16((long long)rows, (int)(3 * ci), (int)c);
    const size_t s1 = w.sp_z > 1 ? (size_t)w.sp_z * c * ci * 4 : 0, s2 = w.sp_qkv > 1 ? (size_t)w.sp_qkv * 3 * ci * c * 4 : 0;
    w.split = take(s1 > s2 ? s1 : s2);
    w.total = off;
    return w;
}

int check_tp(const glf_tpavi_params* p, const char* who) {
    GLF_REQUIRE(p != nullptr, GLF_ERR_NULL, "%s: null parameter block", who);
    GLF_REQUIRE(p->n > 0 && p->L > 0 && p->c > 0 && p->ci > 0, GLF_ERR_BAD_SHAPE, "%s: extents must be > 0", who);
    GLF_REQUIRE(p->c % 64 == 0 && p->ci % 64 == 0 && p->c <= 2048, GLF_ERR_UNSUPPORTED, "%s: C and Ci must be multiples of 64, C <= 2048 (got %d, %d)", who, p->c, p->ci);
    GLF_REQUIRE((long long)p->n * p->L < 2147483647LL / 4, GLF_ERR_BAD_SHAPE, "%s: n * L out of range", who);
    return GLF_OK;
}

#define TRY(call) do { if (int rc_ = (call)) return rc_; } while (0)

}  // namespace

extern "C" size_t glf_sizeof_tpavi_params(void) { return sizeof(glf_tpavi_params); }

extern "C" size_t glf_s16_tpavi_workspace_bytes(const glf_tpavi_params* p, int pass) {
    if (!p || p->n <= 0 || p->L <= 0 || p->c <= 0 || p->ci <= 0) return 0;
    if (pass == 0) return al256(2 * (size_t)p->c * sizeof(double));
    return bwd_layout(p).total;
}

extern "C" int glf_s16_tpavi_fwd(const void* x, const void* w_qkv, const float* b_qkv, const void* w_z, const float* b_z,
                                 const float* bn_gamma, const float* bn_beta, float* bn_running_mean, float* bn_running_var,
                                 int64_t* num_batches_tracked, const float* ln_gamma, const float* ln_beta, void* z,
                                 void* qkv, void* att_t, void* y, void* wz, float* bn_mean, float* bn_invstd, float* row_mean, float* row_rstd,
                                 const glf_tpavi_params* p, void* workspace, size_t workspace_bytes, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    TRY(check_tp(p, "s16_tpavi_fwd"));
    GLF_REQUIRE(x && w_qkv && b_qkv && w_z && b_z && bn_gamma && bn_beta && ln_gamma && ln_beta && z && qkv && att_t && y && wz && bn_mean && bn_invstd &&
                row_mean && row_rstd, GLF_ERR_NULL, "s16_tpavi_fwd: null argument");
    GLF_REQUIRE(workspace && workspace_bytes >= glf_s16_tpavi_workspace_bytes(p, 0), GLF_ERR_WORKSPACE,
                "s16_tpavi_fwd: workspace of %zu bytes, glf_s16_tpavi_workspace_bytes(p, 0) asks for %zu", workspace_bytes, glf_s16_tpavi_workspace_bytes(p, 0));
    GLF_REQUIRE(p->training || (bn_running_mean && bn_running_var), GLF_ERR_NULL, "s16_tpavi_fwd: eval mode needs the running statistics");
    const int rows = p->n * p->L, c = p->c, ci = p->ci, c3 = 3 * ci, L = p->L;
    const long long bq = (long long)L * c3;
    typedef unsigned short u16;
    const u16* q16 = static_cast<const u16*>(qkv);
    // theta | phi | g in ONE contraction over the shared input (ours.py:866-880): qkv[rows][3 Ci]
    glf_gemm_params g = plain(rows, c3, c, c, c, c3, GLF_DT_BF16);
    TRY(glf_s16_gemm_nt(x, w_qkv, b_qkv, qkv, &g, s));
    // dot attention re-associated (ours.py:881-902: y = (theta phi^T / L) g = theta (phi^T g / L)):
    // M_n^T[a][b] = sum_r g[r][a] phi[r][b] / L, the B operand of y_n = theta_n M_n as it stands
    g = plain(ci, ci, L, c3, c3, ci, GLF_DT_BF16);
    g.batch = p->n; g.batch_stride_a = bq; g.batch_stride_b = bq; g.batch_stride_c = (long long)ci * ci; g.alpha = 1.0f / (float)L;
    TRY(glf_s16_gemm_tn(q16 + 2 * ci, q16 + ci, att_t, &g, s));
    g = plain(L, ci, ci, c3, ci, ci, GLF_DT_BF16);
    g.batch = p->n; g.batch_stride_a = bq; g.batch_stride_b = (long long)ci * ci; g.batch_stride_c = (long long)L * ci;
    TRY(glf_s16_gemm_nt(q16, att_t, nullptr, y, &g, s));
    // W_z (ours.py:908) with the BatchNorm3d batch statistics out of its own epilogue
    double* sums = static_cast<double*>(workspace);
    g = plain(rows, c, ci, ci, ci, c, GLF_DT_BF16);
    if (p->training) {
        hipError_t e = hipMemsetAsync(sums, 0, 2 * (size_t)c * sizeof(double), glf::S(s));
        if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "s16_tpavi_fwd: hipMemsetAsync: %s", hipGetErrorString(e));
        g.colstats = sums;
    }
    TRY(glf_s16_gemm_nt(y, w_z, b_z, wz, &g, s));
    if (p->training)
        TRY(glf_bn_stats_from_sums(sums, rows, c, p->bn_eps, p->bn_momentum, bn_mean, bn_invstd, bn_running_mean, bn_running_var, num_batches_tracked, s));
    else
        TRY(glf_bn_eval_coeffs(bn_running_mean, bn_running_var, p->bn_eps, bn_mean, bn_invstd, c, s));
    // z = LayerNorm_C(BatchNorm(w) + x)  (ours.py:908-915)
    return glf_s16_bn_res_ln_fwd(wz, x, bn_mean, bn_invstd, bn_gamma, bn_beta, ln_gamma, ln_beta, p->ln_eps, z, row_mean, row_rstd, rows, c, s);
}

extern "C" int glf_s16_tpavi_bwd(const void* dz, const void* x, const void* qkv, const void* att_t, const void* y, const void* wz,
                                 const float* bn_mean, const float* bn_invstd, const float* row_mean, const float* row_rstd,
                                 const void* w_qkv_t, const void* w_z_t, const float* bn_gamma, const float* bn_beta, const float* ln_gamma,
                                 void* dx, float* dw_qkv, float* db_qkv, float* dw_z, float* db_z, float* dbn_gamma, float* dbn_beta,
                                 float* dln_gamma, float* dln_beta, const glf_tpavi_params* p, void* workspace, size_t workspace_bytes,
                                 glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    TRY(check_tp(p, "s16_tpavi_bwd"));
    GLF_REQUIRE(dz && x && qkv && att_t && y && wz && bn_mean && bn_invstd && row_mean && row_rstd && w_qkv_t && w_z_t && bn_gamma && bn_beta && ln_gamma &&
                dx && dw_qkv && db_qkv && dw_z && db_z && dbn_gamma && dbn_beta && dln_gamma && dln_beta, GLF_ERR_NULL, "s16_tpavi_bwd: null argument");
    const BwdLayout w = bwd_layout(p);
    GLF_REQUIRE(workspace && workspace_bytes >= w.total, GLF_ERR_WORKSPACE, "s16_tpavi_bwd: workspace of %zu bytes, glf_s16_tpavi_workspace_bytes(p, 1) asks for %zu",
                workspace_bytes, w.total);
    GLF_REQUIRE((reinterpret_cast<uintptr_t>(workspace) & 255u) == 0, GLF_ERR_WORKSPACE, "s16_tpavi_bwd: workspace must be 256-byte aligned");
    typedef unsigned short u16;
    unsigned char* ws = static_cast<unsigned char*>(workspace);
    const int rows = p->n * p->L, c = p->c, ci = p->ci, c3 = 3 * ci, L = p->L;
    const long long bq = (long long)L * c3, bs = (long long)L * ci, cc = (long long)ci * ci;
    const u16* q16 = static_cast<const u16*>(qkv);
    u16* dqkv = reinterpret_cast<u16*>(ws + w.dqkv);
    void* dwz = ws + w.dwz;
    void* dy = ws + w.dy;
    void* att = ws + w.att;
    void* dM = ws + w.dM;
    void* dMT = ws + w.dMT;
    float* split_ws = reinterpret_cast<float*>(ws + w.split);
    // LayerNorm backward -> du (the gradient of u = BN(w) + x, written into dx: it is also the residual's gradient)
    TRY(glf_s16_bn_res_ln_bwd(dz, wz, x, bn_mean, bn_invstd, bn_gamma, bn_beta, ln_gamma, row_mean, row_rstd, dx, dln_gamma, dln_beta, rows, c,
                              reinterpret_cast<float*>(ws + w.slab), s));
    // BatchNorm3d backward on w
    hipError_t e = hipMemsetAsync(ws + w.sums, 0, 2 * (size_t)c * sizeof(double), glf::S(s));
    if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "s16_tpavi_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
    TRY(glf_s16_bn_bwd(dx, c, nullptr, 0, wz, c, bn_mean, bn_invstd, bn_gamma, nullptr, dwz, c, nullptr, 0, dbn_gamma, dbn_beta, rows, c, 0, p->training,
                       reinterpret_cast<double*>(ws + w.sums), nullptr, s));
    // W_z: w = y zW^T + b
    glf_gemm_params g = plain(c, ci, rows, c, ci, ci, GLF_DT_F32);
    g.split = w.sp_z; g.workspace = split_ws; g.workspace_bytes = (int64_t)glf_s16_gemm_tn_workspace_bytes(&g);
    TRY(glf_s16_gemm_tn(dwz, y, dw_z, &g, s));
    if (p->training) {
        // the bias feeds a train-mode BatchNorm: its gradient is zero in exact arithmetic (sum_r dwz = -gamma invstd (sum_r xhat)(...) / n)
        e = hipMemsetAsync(db_z, 0, (size_t)c * sizeof(float), glf::S(s));
        if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "s16_tpavi_bwd: hipMemsetAsync: %s", hipGetErrorString(e));
    } else {
        TRY(glf_s16_colsum(dwz, c, db_z, rows, c, reinterpret_cast<double*>(ws + w.colsum), s));
    }
    g = plain(rows, ci, c, c, c, ci, GLF_DT_BF16);
    TRY(glf_s16_gemm_nt(dwz, w_z_t, nullptr, dy, &g, s));
    // y_n = theta_n M_n ; M_n = phi_n^T g_n / L
    TRY(glf_s16_transpose2d(att_t, att, ci, ci, p->n, s));                           // M_n
    g = plain(L, ci, ci, ci, ci, c3, GLF_DT_BF16);
    g.batch = p->n; g.batch_stride_a = bs; g.batch_stride_b = cc; g.batch_stride_c = bq;
    TRY(glf_s16_gemm_nt(dy, att, nullptr, dqkv, &g, s));                             // d theta
    g = plain(ci, ci, L, c3, ci, ci, GLF_DT_BF16);
    g.batch = p->n; g.batch_stride_a = bq; g.batch_stride_b = bs; g.batch_stride_c = cc;
    TRY(glf_s16_gemm_tn(q16, dy, dM, &g, s));                                        // dM_n = theta_n^T dy_n
    g = plain(L, ci, ci, c3, ci, c3, GLF_DT_BF16);
    g.batch = p->n; g.batch_stride_a = bq; g.batch_stride_b = cc; g.batch_stride_c = bq; g.alpha = 1.0f / (float)L;
    TRY(glf_s16_gemm_nt(q16 + 2 * ci, dM, nullptr, dqkv + ci, &g, s));               // d phi = g dM^T / L
    TRY(glf_s16_transpose2d(dM, dMT, ci, ci, p->n, s));
    TRY(glf_s16_gemm_nt(q16 + ci, dMT, nullptr, dqkv + 2 * ci, &g, s));              // d g = phi dM / L
    // the three projections as one: qkv = x Wcat^T + bcat
    g = plain(c3, c, rows, c3, c, c, GLF_DT_F32);
    g.split = w.sp_qkv; g.workspace = split_ws; g.workspace_bytes = (int64_t)glf_s16_gemm_tn_workspace_bytes(&g);
    TRY(glf_s16_gemm_tn(dqkv, x, dw_qkv, &g, s));
    TRY(glf_s16_colsum(dqkv, c3, db_qkv, rows, c3, reinterpret_cast<double*>(ws + w.colsum), s));
    g = plain(rows, c, c3, c3, c3, c, GLF_DT_BF16);
    g.accumulate = 1;                                                                // dx = du + dqkv Wcat
    return glf_s16_gemm_nt(dqkv, w_qkv_t, nullptr, dx, &g, s);
}
