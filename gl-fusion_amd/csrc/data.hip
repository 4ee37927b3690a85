// the GPU side of the reference's data path and evaluation harness.
//   * glf_prepare_frames: the per-sample transform chain of datasets/loader.py:460-498 (AddChannel -> Resized(144, 144,
//     mode='nearest') -> Center/RandSpatialCrop(112, 112) -> EnsureType) fused with the label handling of
//     loader.py:298-330, 358-414 (class id -> one of 5 part channels per view, images / 255), reading a raw
//     [H0][W0][T] volume and writing the model's layout directly: frames [T][1][112][112], masks [T][5][112][112].
//   * glf_overlap_counts_nchw: tp / fp / fn / tn of sigmoid(logit) > 0.5 against the mask PER CLASS CHANNEL
//     (main.py:537-543 slices the accumulated predictions per part), int64, one pass over [N][C][HW].
// HBM-bound byte / index work: one thread per output pixel, coalesced along x.
#include "glf_common.h"

namespace {

__device__ __forceinline__ float sigmoid_d(float x) { return 1.0f / (1.0f + expf(-x)); }

// nearest-neighbour source index exactly as ATen's upsample_nearest (legacy 'nearest'): floor(dst * (in / out)) in
// fp32, clamped to in - 1
__device__ __forceinline__ int nearest_src(int dst, float scale, int in) {
    const int s = (int)floorf((float)dst * scale);
    return s < in - 1 ? s : in - 1;
}

// img / lab: [H0][W0][T] (T fastest, the NIfTI volume order nibabel hands over); out_img [T][1][oh][ow]; out_mask [T][5][oh][ow]
__global__ __launch_bounds__(256) void prepare_frames_kernel(const float* __restrict__ img, const float* __restrict__ lab,
                                                             float* __restrict__ out_img, float* __restrict__ out_mask, int H0, int W0, int T,
                                                             int rs, int oh, int ow, int oy, int ox, int4 chan_lo, int chan_4, float img_div) {
    const long long total = (long long)T * oh * ow;
    const float sy = (float)H0 / (float)rs, sx = (float)W0 / (float)rs;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int x = (int)(i % ow);
        const int y = (int)((i / ow) % oh);
        const int t = (int)(i / ((long long)ow * oh));
        const int ys = nearest_src(y + oy, sy, H0), xs = nearest_src(x + ox, sx, W0);
        const long long src = ((long long)ys * W0 + xs) * T + t;
        if (out_img) out_img[i] = img[src] / img_div;              // a true division: bit-identical to `images / 255.0` (loader.py:327)
        if (out_mask) {
            const int cls = (int)lab[src];                      // 0 = background, 1..4 = parts of this view
            const int ch = cls == 1 ? chan_lo.x : cls == 2 ? chan_lo.y : cls == 3 ? chan_lo.z : cls == 4 ? chan_lo.w : -1;
            (void)chan_4;
            float* m = out_mask + ((long long)t * 5) * oh * ow + (long long)y * ow + x;
#pragma unroll
            for (int c = 0; c < 5; ++c) m[(long long)c * oh * ow] = (c == ch) ? 1.f : 0.f;
        }
    }
}

// counts[c][4] += (tp, fp, fn, tn) of channel c; x, t: [N][C][hw]
__global__ __launch_bounds__(256) void overlap_nchw_kernel(const float* __restrict__ x, const float* __restrict__ t,
                                                           unsigned long long* __restrict__ counts, int N, int C, long long hw) {
    const int c = blockIdx.y;
    unsigned long long tp = 0, fp = 0, fn = 0, tn = 0;
    const long long per = (long long)N * hw;
    for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < per; i += (long long)gridDim.x * blockDim.x) {
        const long long n = i / hw, p = i - n * hw;
        const long long o = (n * C + c) * hw + p;
        const bool pr = sigmoid_d(x[o]) > 0.5f;               // main.py:250, 385
        const bool gt = t[o] != 0.f;
        tp += pr && gt; fp += pr && !gt; fn += !pr && gt; tn += !pr && !gt;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        tp += __shfl_xor(tp, o, 64); fp += __shfl_xor(fp, o, 64); fn += __shfl_xor(fn, o, 64); tn += __shfl_xor(tn, o, 64);
    }
    if ((threadIdx.x & 63) == 0) {
        unsigned long long* k = counts + 4 * c;
        atomicAdd(k + 0, tp); atomicAdd(k + 1, fp); atomicAdd(k + 2, fn); atomicAdd(k + 3, tn);
    }
}

}  // namespace

extern "C" int glf_prepare_frames(const float* img, const float* lab, float* out_img, float* out_mask, int H0, int W0, int T,
                                  int resize, int out_h, int out_w, int off_y, int off_x, const int* class_to_channel, float img_div,
                                  glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE((img && out_img) || (lab && out_mask), GLF_ERR_NULL, "prepare_frames: nothing to do (image and label both missing)");
    GLF_REQUIRE((img != nullptr) == (out_img != nullptr) && (lab != nullptr) == (out_mask != nullptr), GLF_ERR_NULL,
                "prepare_frames: an input needs its output and vice versa");
    GLF_REQUIRE(H0 > 0 && W0 > 0 && T > 0 && resize > 0 && out_h > 0 && out_w > 0, GLF_ERR_BAD_SHAPE, "prepare_frames: sizes must be > 0");
    GLF_REQUIRE(img_div != 0.f, GLF_ERR_BAD_SHAPE, "prepare_frames: img_div must not be 0");
    GLF_REQUIRE(off_y >= 0 && off_x >= 0 && off_y + out_h <= resize && off_x + out_w <= resize, GLF_ERR_BAD_SHAPE,
                "prepare_frames: crop window [%d+%d, %d+%d] leaves the %d x %d resized image", off_y, out_h, off_x, out_w, resize, resize);
    int4 lo = make_int4(-1, -1, -1, -1);
    if (lab) {
        GLF_REQUIRE(class_to_channel != nullptr, GLF_ERR_NULL, "prepare_frames: class_to_channel (4 ints: channel of class 1..4, -1 = none) missing");
        for (int i = 0; i < 4; ++i)
            GLF_REQUIRE(class_to_channel[i] >= -1 && class_to_channel[i] < 5, GLF_ERR_BAD_SHAPE, "prepare_frames: channel index out of range");
        lo = make_int4(class_to_channel[0], class_to_channel[1], class_to_channel[2], class_to_channel[3]);
    }
    const long long total = (long long)T * out_h * out_w;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(prepare_frames_kernel, dim3((unsigned)blocks), dim3(256), 0, glf::S(s), img, lab, out_img, out_mask, H0, W0, T, resize,
                       out_h, out_w, off_y, off_x, lo, 0, img_div);
    return glf::check_launch("prepare_frames");
}

extern "C" int glf_overlap_counts_nchw(const float* logits, const float* target, int64_t* counts, int n, int c, int64_t hw, glf_stream_t s) {
    if (int rc = glf::ensure_init()) return rc;
    GLF_REQUIRE(logits && target && counts, GLF_ERR_NULL, "overlap_counts_nchw: null argument");
    GLF_REQUIRE(n > 0 && c > 0 && c <= 65535 && hw > 0, GLF_ERR_BAD_SHAPE, "overlap_counts_nchw: bad extents");
    hipError_t e = hipMemsetAsync(counts, 0, (size_t)c * 4 * sizeof(int64_t), glf::S(s));
    if (e != hipSuccess) return glf::fail(GLF_ERR_LAUNCH, "overlap_counts_nchw: memset: %s", hipGetErrorString(e));
    long long blocks = ((long long)n * hw + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(overlap_nchw_kernel, dim3((unsigned)blocks, c), dim3(256), 0, glf::S(s), logits, target,
                       reinterpret_cast<unsigned long long*>(counts), n, c, (long long)hw);
    return glf::check_launch("overlap_counts_nchw");
}
