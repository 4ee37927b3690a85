// Multi-tensor weight refresh: every image the contraction kernels derive from the parameters -- the tap-major re-layouts
// of k x k conv weights (forward / wgrad and dgrad orders), transposed copies of 1x1 / linear weights, the stacked
// theta | phi | g operand of a fusion block, each parameter's max magnitude and the packed pre-split fp16 images of all of
// those -- recomputed in FOUR launches per weight update (one per dependency level), whatever the number of parameters.
// The per-tensor entry points (glf_oihw_to_tap_major, glf_transpose2d, glf_amax, glf_split_f16_packed ...) cost one launch
// per image: ~1 000 launches per update for the 3-view model, which is what a training step paid after every optimizer step.
//
// A job table lives in device memory (caller-owned, built once per model); workgroup b of a pass finds its job by binary
// search over the jobs' first-workgroup indices.  Values are bit-identical to the per-tensor kernels (pure data movement,
// and split4h / pow2_scale of split_f16.h for the packed images).
#include "glf_common.h"
#include "split_f16.h"

namespace {

constexpr int WB = 256;                 // threads per workgroup
constexpr int ELEMS_PER_WG = 4096;      // elements of the destination one workgroup produces (streaming kinds)

__device__ __forceinline__ void job_copy(const glf_weight_job& j, long long wg) {
    const long long n = j.d0, base = wg * ELEMS_PER_WG;
    const long long hi = base + ELEMS_PER_WG < n ? base + ELEMS_PER_WG : n;
    if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(j.src) | reinterpret_cast<uintptr_t>(j.dst)) & 15u) == 0) {
        for (long long i = base + 4 * threadIdx.x; i < hi; i += 4 * WB)
            *reinterpret_cast<float4*>(j.dst + i) = *reinterpret_cast<const float4*>(j.src + i);
    } else {
        for (long long i = base + threadIdx.x; i < hi; i += WB) j.dst[i] = j.src[i];
    }
}

__device__ __forceinline__ void job_zero(const glf_weight_job& j, long long wg) {
    const long long n = j.d0, base = wg * ELEMS_PER_WG;
    const long long hi = base + ELEMS_PER_WG < n ? base + ELEMS_PER_WG : n;
    for (long long i = base + threadIdx.x; i < hi; i += WB) j.dst[i] = 0.f;
}

__device__ __forceinline__ void job_amax(const glf_weight_job& j, long long wg) {
    const long long n = j.d0, base = wg * ELEMS_PER_WG;
    const long long hi = base + ELEMS_PER_WG < n ? base + ELEMS_PER_WG : n;
    float m = 0.f;
    for (long long i = base + threadIdx.x; i < hi; i += WB) m = fmaxf(m, fabsf(j.src[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float sm[WB / 64];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(j.amax), __float_as_uint(m));     // non-negative floats order like their bits
    }
}

// dst[t][co][ci] = src[co][ci][t]
__device__ __forceinline__ void job_tap_major(const glf_weight_job& j, long long wg) {
    const long long cc = (long long)j.d0 * j.d1, taps = j.d2, n = cc * taps, base = wg * ELEMS_PER_WG;
    const long long hi = base + ELEMS_PER_WG < n ? base + ELEMS_PER_WG : n;
    for (long long o = base + threadIdx.x; o < hi; o += WB) {
        const long long t = o / cc, i = o - t * cc;
        j.dst[o] = j.src[i * taps + t];
    }
}

// dst[t][ci][co] = src[co][ci][t]
__device__ __forceinline__ void job_tap_major_t(const glf_weight_job& j, long long wg) {
    const int cout = j.d0, cin = j.d1, taps = j.d2;
    const long long n = (long long)cout * cin * taps, base = wg * ELEMS_PER_WG;
    const long long hi = base + ELEMS_PER_WG < n ? base + ELEMS_PER_WG : n;
    for (long long o = base + threadIdx.x; o < hi; o += WB) {
        const int co = (int)(o % cout);
        const long long r = o / cout;
        const int ci = (int)(r % cin), t = (int)(r / cin);
        j.dst[o] = j.src[((long long)co * cin + ci) * taps + t];
    }
}

// dst[c][r] = src[r][c], one 32 x 32 tile per workgroup through padded LDS
__device__ __forceinline__ void job_transpose(const glf_weight_job& j, long long wg) {
    __shared__ float tile[32][33];
    const int rows = j.d0, cols = j.d1;
    const int tiles_c = (cols + 31) / 32;
    const int r0 = (int)(wg / tiles_c) * 32, c0 = (int)(wg % tiles_c) * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int r = r0 + i, c = c0 + tx;
        tile[i][tx] = (r < rows && c < cols) ? j.src[(long long)r * cols + c] : 0.f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, r = r0 + tx;
        if (c < cols && r < rows) j.dst[(long long)c * rows + r] = tile[tx][i];
    }
}

// dst = packed pre-split image of src (d0 elements, a multiple of 4; both 16-byte aligned), scaled by *amax
__device__ __forceinline__ void job_pack(const glf_weight_job& j, long long wg) {
    float sc, inv;
    pow2_scale(j.amax, sc, inv);
    const long long n = j.d0, base = wg * ELEMS_PER_WG;
    const long long hi = base + ELEMS_PER_WG < n ? base + ELEMS_PER_WG : n;
    for (long long i = base + 4 * threadIdx.x; i < hi; i += 4 * WB) {
        const SplitH s = split4h(*reinterpret_cast<const float4*>(j.src + i), sc);
        const float2 h = __builtin_bit_cast(float2, s.h), l = __builtin_bit_cast(float2, s.l);
        *reinterpret_cast<float4*>(j.dst + i) = make_float4(h.x, h.y, l.x, l.y);
    }
}

// dst (bf16) = round-to-nearest-even(src), d0 elements (a multiple of 4; src 16-byte, dst 8-byte aligned): the operand images of
// the 16-bit-storage contractions (glf_s16_gemm_*), made from the fp32 master weights once per update
__device__ __forceinline__ void job_cvt_bf16(const glf_weight_job& j, long long wg) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    const long long n = j.d0, base = wg * ELEMS_PER_WG;
    const long long hi = base + ELEMS_PER_WG < n ? base + ELEMS_PER_WG : n;
    unsigned short* dst = reinterpret_cast<unsigned short*>(j.dst);
    for (long long i = base + 4 * threadIdx.x; i < hi; i += 4 * WB) {
        const float4 v = *reinterpret_cast<const float4*>(j.src + i);
        const f32x2_ a = {v.x, v.y}, b = {v.z, v.w};
        *reinterpret_cast<uint2*>(dst + i) = make_uint2(__builtin_bit_cast(unsigned, __builtin_convertvector(a, bf16x2_)),
                                                        __builtin_bit_cast(unsigned, __builtin_convertvector(b, bf16x2_)));
    }
}

__global__ __launch_bounds__(WB) void weights_refresh_kernel(const glf_weight_job* __restrict__ jobs, int first, int count) {
    // job of this workgroup: the last one whose first_wg <= blockIdx.x
    int lo = 0, hi = count - 1;
    const long long b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[first + mid].first_wg <= b) lo = mid; else hi = mid - 1;
    }
    const glf_weight_job j = jobs[first + lo];
    const long long wg = b - j.first_wg;
    switch (j.kind) {
        case GLF_WJ_COPY: job_copy(j, wg); break;
        case GLF_WJ_AMAX: job_amax(j, wg); break;
        case GLF_WJ_TAP_MAJOR: job_tap_major(j, wg); break;
        case GLF_WJ_TAP_MAJOR_T: job_tap_major_t(j, wg); break;
        case GLF_WJ_TRANSPOSE: job_transpose(j, wg); break;
        case GLF_WJ_PACK: job_pack(j, wg); break;
        case GLF_WJ_ZERO: job_zero(j, wg); break;
        case GLF_WJ_CVT_BF16: job_cvt_bf16(j, wg); break;
        default: break;
    }
}

long long job_workgroups(const glf_weight_job& j) {
    switch (j.kind) {
        case GLF_WJ_COPY: case GLF_WJ_AMAX: case GLF_WJ_PACK: case GLF_WJ_ZERO: case GLF_WJ_CVT_BF16: return ((long long)j.d0 + ELEMS_PER_WG - 1) / ELEMS_PER_WG;
        case GLF_WJ_TAP_MAJOR: case GLF_WJ_TAP_MAJOR_T: return ((long long)j.d0 * j.d1 * j.d2 + ELEMS_PER_WG - 1) / ELEMS_PER_WG;
        case GLF_WJ_TRANSPOSE: return (long long)((j.d0 + 31) / 32) * ((j.d1 + 31) / 32);
        default: return -1;
    }
}

}  // namespace

extern "C" int glf_weights_plan(glf_weight_job* jobs_host, int n_jobs, int* pass_first, int* pass_count, int64_t* pass_wgs) {
    GLF_REQUIRE(jobs_host && pass_first && pass_count && pass_wgs && n_jobs >= 0, GLF_ERR_NULL, "weights_plan: null argument");
    for (int p = 0; p < GLF_WJ_PASSES; ++p) { pass_first[p] = 0; pass_count[p] = 0; pass_wgs[p] = 0; }
    int prev = -1;
    for (int i = 0; i < n_jobs; ++i) {
        glf_weight_job& j = jobs_host[i];
        GLF_REQUIRE(j.pass >= 0 && j.pass < GLF_WJ_PASSES && j.pass >= prev, GLF_ERR_BAD_SHAPE,
                    "weights_plan: job %d: pass %d out of range or jobs not sorted by pass", i, j.pass);
        GLF_REQUIRE((j.src || j.kind == GLF_WJ_ZERO) && (j.dst || j.kind == GLF_WJ_AMAX), GLF_ERR_NULL, "weights_plan: job %d: null tensor", i);
        GLF_REQUIRE(j.d0 > 0 && (j.kind == GLF_WJ_COPY || j.kind == GLF_WJ_AMAX || j.kind == GLF_WJ_PACK || j.kind == GLF_WJ_ZERO || j.kind == GLF_WJ_CVT_BF16 || j.d1 > 0), GLF_ERR_BAD_SHAPE,
                    "weights_plan: job %d: bad extents", i);
        if (j.kind == GLF_WJ_AMAX || j.kind == GLF_WJ_PACK) GLF_REQUIRE(j.amax, GLF_ERR_NULL, "weights_plan: job %d needs an amax scalar", i);
        if (j.kind == GLF_WJ_PACK)
            GLF_REQUIRE((j.d0 & 3) == 0 && ((reinterpret_cast<uintptr_t>(j.src) | reinterpret_cast<uintptr_t>(j.dst)) & 15u) == 0, GLF_ERR_BAD_SHAPE,
                        "weights_plan: job %d: packed images need 16-byte aligned tensors of 4n elements", i);
        if (j.kind == GLF_WJ_CVT_BF16)
            GLF_REQUIRE((j.d0 & 3) == 0 && (reinterpret_cast<uintptr_t>(j.src) & 15u) == 0 && (reinterpret_cast<uintptr_t>(j.dst) & 7u) == 0, GLF_ERR_BAD_SHAPE,
                        "weights_plan: job %d: bf16 images need aligned tensors of 4n elements", i);
        if (j.kind == GLF_WJ_TAP_MAJOR || j.kind == GLF_WJ_TAP_MAJOR_T) GLF_REQUIRE(j.d2 > 0, GLF_ERR_BAD_SHAPE, "weights_plan: job %d: taps must be > 0", i);
        const long long w = job_workgroups(j);
        GLF_REQUIRE(w > 0, GLF_ERR_UNSUPPORTED, "weights_plan: job %d: unknown kind %d", i, j.kind);
        if (j.pass != prev) { pass_first[j.pass] = i; prev = j.pass; }
        j.first_wg = pass_wgs[j.pass];
        pass_wgs[j.pass] += w;
        pass_count[j.pass] += 1;
        GLF_REQUIRE(pass_wgs[j.pass] < 2147483647LL, GLF_ERR_BAD_SHAPE, "weights_plan: pass %d grid out of range", j.pass);
    }
    return GLF_OK;
}

extern "C" int glf_weights_refresh(const glf_weight_job* jobs_dev, const int* pass_first, const int* pass_count, const int64_t* pass_wgs,
                                   float* amax_arena, int64_t amax_floats, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(jobs_dev && pass_first && pass_count && pass_wgs, GLF_ERR_NULL, "weights_refresh: null argument");
    if (amax_arena && amax_floats > 0) {          // the maxima are taken with atomicMax: start every slot from 0
        hipError_t e = hipMemsetAsync(amax_arena, 0, (size_t)amax_floats * sizeof(float), glf::S(s));
        if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "weights_refresh: hipMemsetAsync: %s", hipGetErrorString(e));
    }
    for (int p = 0; p < GLF_WJ_PASSES; ++p) {
        if (pass_count[p] <= 0) continue;
        hipLaunchKernelGGL(weights_refresh_kernel, dim3((unsigned)pass_wgs[p]), dim3(WB), 0, glf::S(s), jobs_dev, pass_first[p], pass_count[p]);
        if (int rc = glf::check_launch("weights_refresh")) return rc;
    }
    return GLF_OK;
}
