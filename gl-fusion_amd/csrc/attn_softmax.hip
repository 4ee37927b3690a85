// Fused softmax attention of the fusion block's `embedded` mode (reference models/ours.py:881, 896-897, 902):
//      f = theta phi^T            [L, L]   per frame, L = V h w positions, contraction over Ci
//      P = softmax(f, dim = -1)
//      y = P g                    [L, Ci]
// as ONE kernel per pass that never writes the [L, L] score matrix (22 MB per frame at config 2, 983 MB per frame at
// the 5-view 224^2 config): a 64-row block of scores is formed tile by tile in LDS, normalised online (running row
// max / row sum, 4 lanes per row folded with __shfl_xor) and immediately contracted with the g rows into MFMA
// accumulators.  Backward recomputes the score tiles from theta / phi and the saved row log-sum-exp (three passes:
// d g, d phi, d theta), again without materialising anything of size L x L.
//
// Arithmetic: exact fp32 on v_mfma_f32_32x32x2_f32 (the softmax exponentiates the scores: they are kept at full fp32
// accuracy; `embedded` is not the shipped mode, `dot` is and runs on the split-fp16 contraction kernels).
//
// Work decomposition (all four passes share one skeleton).  A workgroup = 256 threads = 4 waves, one per SIMD (the
// 64 x Ci fp32 accumulator block takes 256 VGPRs per wave at Ci = 1024).  It owns 64 "outer" rows -- query rows in
// FWD / DQ, key rows in DV / DK -- and loops over blocks of 64 "inner" rows.  Per inner block:
//   1. score tile S[64 q][64 k] = Q K^T over Ci in chunks of 32, both operands staged global -> LDS as k-major tiles
//      (conflict-free ds_read_b32 fragments), double-buffered, one barrier per chunk; wave w computes the 32 x 32
//      quadrant (w >> 1, w & 1) over the whole contraction, so no cross-wave reduction is needed
//      (DK / DQ: a second tile dP = dY G^T the same way);
//   2. the tile is transformed in registers / LDS (FWD: online softmax; DV: P = exp(S - lse); DK, DQ:
//      dS = P (dP - D)) and left in LDS in the orientation the next contraction reads as its A operand;
//   3. acc[64][Ci] += T (or T^T) times the inner (FWD, DQ) / outer-loop (DV, DK) rows of g / dY / theta / phi, whose
//      B fragments (k = row, n = 32 consecutive columns) are 128-byte coalesced reads straight from global memory / L2.
#include "gemm_common.h"

namespace {

constexpr int AT = 64;            // rows per block (outer and inner)
constexpr int AKC = 32;           // contraction chunk of the score tiles
constexpr int ALD = 65;           // LDS row stride of every [*][64] tile (k-major staging tiles, score tiles)
constexpr int ANT = 256;
constexpr int A_MAXCT = 8;        // 32-column accumulator tiles per wave  => Ci <= 4 * 8 * 32 = 1024

enum AttnMode { ATT_FWD = 0, ATT_DV = 1, ATT_DK = 2, ATT_DQ = 3 };

struct AttnArgs {
    const float* q; const float* k; const float* v;      // theta, phi, g: rows of length ci, row strides ldq / ldk / ldv
    const float* dy;                                     // backward: gradient of y, row stride lddy
    const float* lse;                                    // backward: [frames * L] log-sum-exp of every score row
    const float* dsum;                                   // DK / DQ: [frames * L] D = rowsum(dy * y)
    float* out;                                          // FWD: y; DV: dg; DK: dphi; DQ: dtheta (row stride ldo)
    float* lse_out;                                      // FWD: receives the row log-sum-exp
    int L, ci;
    long long ldq, ldk, ldv, lddy, ldo;
    long long fq, fk, fv, fdy, fo;                       // frame strides (elements)
};

#define GLF_MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0)

// stage rows [r0, r0 + 64) x columns [k0, k0 + 32) of a row-major matrix into a k-major LDS tile dst[32][ALD];
// rows beyond L are clamped to L - 1 (their scores are masked / their outputs never stored)
struct Staged { float4 a, b; };
__device__ __forceinline__ Staged stage_load(const float* __restrict__ src, long long ld, int r0, int L, int k0, int tid) {
    const int row = tid >> 3, kq = tid & 7;
    const int ra = min(r0 + row, L - 1), rb = min(r0 + row + 32, L - 1);
    Staged s;
    s.a = *reinterpret_cast<const float4*>(src + (long long)ra * ld + k0 + 4 * kq);
    s.b = *reinterpret_cast<const float4*>(src + (long long)rb * ld + k0 + 4 * kq);
    return s;
}
__device__ __forceinline__ void stage_store(float* __restrict__ dst, const Staged& s, int tid) {
    const int row = tid >> 3, kq = tid & 7;
    float* d = dst + (4 * kq) * ALD + row;
    d[0] = s.a.x; d[ALD] = s.a.y; d[2 * ALD] = s.a.z; d[3 * ALD] = s.a.w;
    d[32] = s.b.x; d[ALD + 32] = s.b.y; d[2 * ALD + 32] = s.b.z; d[3 * ALD + 32] = s.b.w;
}

// S quadrant (rt, ct) of X[x0 .. x0+64) . Y[y0 .. y0+64)^T over `ci` columns.  `stg` = 4 tiles of [32][ALD] (two buffers
// of an X tile and a Y tile).  Every thread of the workgroup must call it (barriers inside).
__device__ __forceinline__ f32x16 score_tile(const float* __restrict__ X, long long ldx, int x0, const float* __restrict__ Y, long long ldy,
                                             int y0, int L, int ci, float* __restrict__ stg, int tid, int lane, int rt, int ct) {
    constexpr int TILE = AKC * ALD;
    f32x16 s = {0};
    Staged sx = stage_load(X, ldx, x0, L, 0, tid), sy = stage_load(Y, ldy, y0, L, 0, tid);
    __syncthreads();                                  // the previous user of the staging tiles is done
    stage_store(stg, sx, tid);
    stage_store(stg + TILE, sy, tid);
    __syncthreads();
    const int nchunk = ci / AKC;
    const int hl = lane >> 5, l31 = lane & 31;
    for (int c = 0; c < nchunk; ++c) {
        const float* xs = stg + (c & 1) * 2 * TILE;
        const float* ys = xs + TILE;
        const bool more = c + 1 < nchunk;
        if (more) {
            sx = stage_load(X, ldx, x0, L, (c + 1) * AKC, tid);
            sy = stage_load(Y, ldy, y0, L, (c + 1) * AKC, tid);
        }
#pragma unroll
        for (int ks = 0; ks < AKC / 2; ++ks) {
            const float a = xs[(2 * ks + hl) * ALD + 32 * rt + l31];
            const float b = ys[(2 * ks + hl) * ALD + 32 * ct + l31];
            s = GLF_MFMA_F32(a, b, s);
        }
        if (more) {
            float* nx = stg + ((c + 1) & 1) * 2 * TILE;
            stage_store(nx, sx, tid);
            stage_store(nx + TILE, sy, tid);
        }
        __syncthreads();
    }
    return s;
}

// acc[rt][jj] += sum_k T[k][32 rt + m] * Z[z0 + k][32 j + n] for this wave's column tiles j = wave + 4 jj:
// A fragments from the k-major LDS tile T ([64][ALD]), B fragments straight from global memory (rows clamped to L - 1:
// the matching T entries are exactly zero).
__device__ __forceinline__ void accumulate(f32x16 (&acc)[2][A_MAXCT], const float* __restrict__ T, const float* __restrict__ Z, long long ldz,
                                           int z0, int L, int nct, int wave, int lane) {
    const int hl = lane >> 5, l31 = lane & 31;
#pragma unroll
    for (int jj = 0; jj < A_MAXCT; ++jj) {
        const int j = wave + 4 * jj;
        if (j < nct) {
            const float* zc = Z + 32 * j + l31;
            // B fragments of the next eight k-steps are in flight while the current eight are multiplied; the scheduling
            // barriers keep hipcc from hoisting the loads of ALL column tiles to the top (it spilled 170 registers)
            float b[8], bn[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) b[u] = zc[(long long)min(z0 + 2 * u + hl, L - 1) * ldz];
#pragma unroll 1
            for (int kb = 0; kb < AT / 2; kb += 8) {
                if (kb + 8 < AT / 2) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) bn[u] = zc[(long long)min(z0 + 2 * (kb + 8 + u) + hl, L - 1) * ldz];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float a0 = T[(2 * (kb + u) + hl) * ALD + l31];
                    const float a1 = T[(2 * (kb + u) + hl) * ALD + 32 + l31];
                    acc[0][jj] = GLF_MFMA_F32(a0, b[u], acc[0][jj]);
                    acc[1][jj] = GLF_MFMA_F32(a1, b[u], acc[1][jj]);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) b[u] = bn[u];
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(ANT, 1) void attn_softmax_kernel(const AttnArgs args) {
    const int L = args.L, ci = args.ci, nct = ci / 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int rt = wave >> 1, ct = wave & 1;
    const int hl = lane >> 5, l31 = lane & 31;
    const int o0 = blockIdx.x * AT;                           // first outer row
    const long long fr = blockIdx.y;
    const float* __restrict__ Q = args.q + fr * args.fq;
    const float* __restrict__ K = args.k + fr * args.fk;
    const float* __restrict__ V = args.v + fr * args.fv;
    const float* __restrict__ DY = args.dy ? args.dy + fr * args.fdy : nullptr;
    const float* __restrict__ LSE = args.lse ? args.lse + fr * L : nullptr;
    const float* __restrict__ DS = args.dsum ? args.dsum + fr * L : nullptr;
    float* __restrict__ OUT = args.out + fr * args.fo;

    extern __shared__ __attribute__((aligned(16))) float smem_a[];
    constexpr int TILE = AKC * ALD;
    float* stg = smem_a;                                      // 4 staging tiles
    float* Ts = smem_a + 4 * TILE;                            // score tile as stored by the waves, [64][ALD]
    float* Pt = Ts + AT * ALD;                                // FWD: P transposed [key][q]
    float* rowm = Pt + AT * ALD;                              // FWD: running max, running sum, rescale factor per query row
    float* rowl = rowm + AT;
    float* rowa = rowl + AT;

    f32x16 acc[2][A_MAXCT];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < A_MAXCT; ++j) acc[i][j] = f32x16{0};

    if (MODE == ATT_FWD) {
        if (tid < AT) { rowm[tid] = -INFINITY; rowl[tid] = 0.f; rowa[tid] = 0.f; }
    }
    const int nblk = (L + AT - 1) / AT;
    for (int ib = 0; ib < nblk; ++ib) {
        const int i0 = ib * AT;                               // first inner row
        // q rows / key rows of the score tile: the queries are the outer block in FWD / DQ, the inner one in DV / DK
        const int q0 = (MODE == ATT_FWD || MODE == ATT_DQ) ? o0 : i0;
        const int k0 = (MODE == ATT_FWD || MODE == ATT_DQ) ? i0 : o0;
        f32x16 s = score_tile(Q, args.ldq, q0, K, args.ldk, k0, L, ci, stg, tid, lane, rt, ct);
        f32x16 dp = {0};
        if (MODE == ATT_DK || MODE == ATT_DQ) dp = score_tile(DY, args.lddy, q0, V, args.ldv, k0, L, ci, stg, tid, lane, rt, ct);
        // this lane's tile entries: key = k0 + 32 ct + l31, query rows q0 + 32 rt + (r & 3) + 8 (r >> 2) + 4 hl
        const int key = k0 + 32 * ct + l31;
        const bool key_ok = key < L;
        if (MODE == ATT_FWD) {
#pragma unroll
            for (int r = 0; r < 16; ++r) Ts[(32 * rt + (r & 3) + 8 * (r >> 2) + 4 * hl) * ALD + 32 * ct + l31] = s[r];
            __syncthreads();
            // online softmax: 4 lanes per query row, 16 keys each
            const int row = tid >> 2, part = tid & 3;
            float sv[16];
            float mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool ok = i0 + 16 * part + i < L;
                sv[i] = ok ? Ts[row * ALD + 16 * part + i] : -INFINITY;
                mx = fmaxf(mx, sv[i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
            const float m_old = rowm[row], m_new = fmaxf(m_old, mx);       // every inner block holds >= 1 valid key: m_new is finite
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = expf(sv[i] - m_new);                       // exp(-inf) = 0 for masked keys
                sum += p;
                Pt[(16 * part + i) * ALD + row] = p;
            }
            sum += __shfl_xor(sum, 1, 64);
            sum += __shfl_xor(sum, 2, 64);
            __syncthreads();                                               // every lane has read rowm before it is rewritten
            if (part == 0) {
                const float alpha = expf(m_old - m_new);                   // 0 on the first block (m_old = -inf)
                rowa[row] = alpha;
                rowm[row] = m_new;
                rowl[row] = rowl[row] * alpha + sum;
            }
            __syncthreads();
            // rescale the accumulators by alpha[row] -- skipped when no row of the block moved its maximum (the usual case
            // after the first few key blocks) -- then acc += P V
            const float my_alpha = rowa[lane];
            if (__ballot(my_alpha != 1.f) != 0ull) {
#pragma unroll
                for (int jj = 0; jj < A_MAXCT; ++jj) {
                    if (wave + 4 * jj < nct) {
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) acc[i][jj][r] *= rowa[32 * i + (r & 3) + 8 * (r >> 2) + 4 * hl];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            accumulate(acc, Pt, V, args.ldv, i0, L, nct, wave, lane);
        } else {
            // P = exp(S - lse[q]) (0 for rows / keys beyond L); DK, DQ: dS = P (dP - D[q])
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ql = 32 * rt + (r & 3) + 8 * (r >> 2) + 4 * hl;
                const int qg = q0 + ql;
                const bool ok = key_ok && qg < L;
                const int qc = min(qg, L - 1);
                float t = ok ? expf(s[r] - LSE[qc]) : 0.f;
                if (MODE == ATT_DK || MODE == ATT_DQ) t = ok ? t * (dp[r] - DS[qc]) : 0.f;
                if (MODE == ATT_DQ) Ts[(32 * ct + l31) * ALD + ql] = t;              // [key][q]: A[m = q][k = key]
                else Ts[ql * ALD + 32 * ct + l31] = t;                                 // [q][key]: A[m = key][k = q]
            }
            __syncthreads();
            if (MODE == ATT_DV) accumulate(acc, Ts, DY, args.lddy, i0, L, nct, wave, lane);        // dg_j += P^T dY_i
            else if (MODE == ATT_DK) accumulate(acc, Ts, Q, args.ldq, i0, L, nct, wave, lane);     // dphi_j += dS^T theta_i
            else accumulate(acc, Ts, K, args.ldk, i0, L, nct, wave, lane);                          // dtheta_i += dS phi_j
        }
        // the next score_tile() begins with a barrier before it overwrites the staging tiles; Ts / Pt are rewritten only
        // after that call's barriers, i.e. after every wave has finished this block's accumulate()
    }

    // epilogue: rows o0 + 32 i + (r & 3) + 8 (r >> 2) + 4 hl, columns 32 (wave + 4 jj) + l31
    if (MODE == ATT_FWD) __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rl = 32 * i + (r & 3) + 8 * (r >> 2) + 4 * hl;
            const int row = o0 + rl;
            if (row < L) {
                float scale = 1.f;
                if (MODE == ATT_FWD) scale = 1.f / rowl[rl];
#pragma unroll
                for (int jj = 0; jj < A_MAXCT; ++jj) {
                    const int j = wave + 4 * jj;
                    if (j < nct) OUT[(long long)row * args.ldo + 32 * j + l31] = acc[i][jj][r] * scale;
                }
            }
        }
    }
    if (MODE == ATT_FWD && tid < AT && o0 + tid < L) args.lse_out[fr * L + o0 + tid] = rowm[tid] + logf(rowl[tid]);
}

constexpr size_t SMEM_ATTN = (4 * AKC * ALD + 2 * AT * ALD + 3 * AT) * sizeof(float);

// D[row] = sum_c dy[row][c] * y[row][c]: one wavefront per row
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const float* __restrict__ dy, long long lddy, const float* __restrict__ y, long long ldy,
                                                          float* __restrict__ out, long long rows, int ci) {
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.f;
    for (int c = lane; c < ci; c += 64) s = fmaf(dy[row * lddy + c], y[row * ldy + c], s);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (lane == 0) out[row] = s;
}

int attn_check(const glf_attn_params* p, const void* q, const void* k, const void* v, const void* out) {
    GLF_REQUIRE(p && q && k && v && out, GLF_ERR_NULL, "attn_softmax: null argument");
    GLF_REQUIRE(p->frames > 0 && p->frames <= 65535 && p->L > 0, GLF_ERR_BAD_SHAPE, "attn_softmax: frames (%d) / L (%d) out of range", p->frames, p->L);
    GLF_REQUIRE(p->ci > 0 && p->ci % 32 == 0 && p->ci <= 32 * 4 * A_MAXCT, GLF_ERR_UNSUPPORTED,
                "attn_softmax: Ci must be a multiple of 32 and <= %d (got %d)", 32 * 4 * A_MAXCT, p->ci);
    GLF_REQUIRE(p->ldq % 4 == 0 && p->ldk % 4 == 0 && p->ldv % 4 == 0 && p->ldq >= p->ci && p->ldk >= p->ci && p->ldv >= p->ci,
                GLF_ERR_BAD_SHAPE, "attn_softmax: row strides must be multiples of 4 and >= Ci");
    GLF_REQUIRE(aligned16(q) && aligned16(k) && aligned16(v), GLF_ERR_BAD_SHAPE, "attn_softmax: theta / phi / g must be 16-byte aligned");
    return GLF_OK;
}

AttnArgs attn_args(const glf_attn_params* p, const float* q, const float* k, const float* v) {
    AttnArgs a{};
    a.q = q; a.k = k; a.v = v;
    a.L = p->L; a.ci = p->ci;
    a.ldq = p->ldq; a.ldk = p->ldk; a.ldv = p->ldv;
    a.fq = (long long)p->L * p->ldq; a.fk = (long long)p->L * p->ldk; a.fv = (long long)p->L * p->ldv;
    return a;
}

}  // namespace

namespace glf {
int init_attn_attrs() {
    hipError_t e;
#define SET_ATTR(fn)                                                                                        \
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_ATTN); \
    if (e != hipSuccess) return fail(GLF_ERR_LAUNCH, "hipFuncSetAttribute(" #fn "): %s", hipGetErrorString(e));
    SET_ATTR((attn_softmax_kernel<ATT_FWD>))
    SET_ATTR((attn_softmax_kernel<ATT_DV>))
    SET_ATTR((attn_softmax_kernel<ATT_DK>))
    SET_ATTR((attn_softmax_kernel<ATT_DQ>))
#undef SET_ATTR
    return GLF_OK;
}
}  // namespace glf

extern "C" int glf_attn_softmax_fwd(const float* theta, const float* phi, const float* g, float* y, float* lse,
                                    const glf_attn_params* p, glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    if (int rc = attn_check(p, theta, phi, g, y)) return rc;
    GLF_REQUIRE(lse != nullptr, GLF_ERR_NULL, "attn_softmax_fwd: lse must be given (it is what backward recomputes the scores against)");
    GLF_REQUIRE(p->ldy >= p->ci, GLF_ERR_BAD_SHAPE, "attn_softmax_fwd: ldy < Ci");
    AttnArgs a = attn_args(p, theta, phi, g);
    a.out = y; a.ldo = p->ldy; a.fo = (long long)p->L * p->ldy; a.lse_out = lse;
    dim3 grid((p->L + AT - 1) / AT, p->frames);
    hipLaunchKernelGGL((attn_softmax_kernel<ATT_FWD>), grid, dim3(ANT), SMEM_ATTN, glf::S(stream), a);
    return glf::check_launch("attn_softmax_fwd");
}

extern "C" int glf_attn_softmax_bwd(const float* theta, const float* phi, const float* g, const float* y, const float* dy, const float* lse,
                                    float* dtheta, float* dphi, float* dg, float* dsum_ws, const glf_attn_params* p, glf_stream_t stream) {
    if (int rc = glf::ensure_init()) return rc;
    if (int rc = attn_check(p, theta, phi, g, dtheta)) return rc;
    GLF_REQUIRE(y && dy && lse && dphi && dg && dsum_ws, GLF_ERR_NULL, "attn_softmax_bwd: null argument");
    GLF_REQUIRE(p->ldy >= p->ci && p->lddy >= p->ci && p->lddy % 4 == 0 && aligned16(dy), GLF_ERR_BAD_SHAPE, "attn_softmax_bwd: bad dy / y strides");
    GLF_REQUIRE(p->ldd >= p->ci, GLF_ERR_BAD_SHAPE, "attn_softmax_bwd: gradient row stride < Ci");
    const long long rows = (long long)p->frames * p->L;
    hipLaunchKernelGGL(attn_rowdot_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, glf::S(stream), dy, (long long)p->lddy, y,
                       (long long)p->ldy, dsum_ws, rows, p->ci);
    AttnArgs a = attn_args(p, theta, phi, g);
    a.dy = dy; a.lddy = p->lddy; a.fdy = (long long)p->L * p->lddy;
    a.lse = lse; a.dsum = dsum_ws;
    a.ldo = p->ldd; a.fo = (long long)p->L * p->ldd;
    dim3 grid((p->L + AT - 1) / AT, p->frames);
    a.out = dg;
    hipLaunchKernelGGL((attn_softmax_kernel<ATT_DV>), grid, dim3(ANT), SMEM_ATTN, glf::S(stream), a);
    a.out = dphi;
    hipLaunchKernelGGL((attn_softmax_kernel<ATT_DK>), grid, dim3(ANT), SMEM_ATTN, glf::S(stream), a);
    a.out = dtheta;
    hipLaunchKernelGGL((attn_softmax_kernel<ATT_DQ>), grid, dim3(ANT), SMEM_ATTN, glf::S(stream), a);
    return glf::check_launch("attn_softmax_bwd");
}
