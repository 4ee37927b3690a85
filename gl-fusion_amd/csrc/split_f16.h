// Split-fp16 operand arithmetic shared by the contraction kernels (gemm_f16s.hip) and every kernel that writes a packed
// pre-split image (glf_split_f16_packed, the multi-tensor weight refresh, BatchNorm apply): ONE definition, so an image made
// anywhere is bit-identical to what the contraction kernels compute in their own staging path.
#pragma once
#include <hip/hip_runtime.h>

namespace {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct SplitH { f16x4 h, l; };

// x*s = h + 2^-11 l: 2 pk_mul + 2 cvt_pk + 4 cvt_f32 + 2 pk_add + 2 pk_mul + 2 cvt_pk per four elements
__device__ __forceinline__ SplitH split4h(const float4 v, const float s) {
    const f32x2 s2 = {s, s}, k2 = {2048.f, 2048.f};
    const f32x2 v01 = {v.x, v.y}, v23 = {v.z, v.w};
    const f32x2 x01 = v01 * s2, x23 = v23 * s2;
    const f16x2 h01 = __builtin_convertvector(x01, f16x2), h23 = __builtin_convertvector(x23, f16x2);
    const f32x2 r01 = (x01 - __builtin_convertvector(h01, f32x2)) * k2;
    const f32x2 r23 = (x23 - __builtin_convertvector(h23, f32x2)) * k2;
    const f16x2 l01 = __builtin_convertvector(r01, f16x2), l23 = __builtin_convertvector(r23, f16x2);
    SplitH o;
    o.h = __builtin_shufflevector(h01, h23, 0, 1, 2, 3);
    o.l = __builtin_shufflevector(l01, l23, 0, 1, 2, 3);
    return o;
}

// a float4 of the packed pre-split image {h0 h1 h2 h3 | l0 l1 l2 l3} -> the two pieces (a bit-cast)
__device__ __forceinline__ SplitH unpack4h(const float4 v) {
    SplitH o;
    o.h = __builtin_bit_cast(f16x4, make_float2(v.x, v.y));
    o.l = __builtin_bit_cast(f16x4, make_float2(v.z, v.w));
    return o;
}

// power-of-two operand scale from its max magnitude: amax*s in [2^13, 2^14); inv = 1/s.  No pointer, zero,
// denormal-range or non-finite amax: s = 1.
__device__ __forceinline__ void pow2_scale(const float* amax, float& s, float& inv) {
    s = 1.f; inv = 1.f;
    if (amax) {
        const int e = (int)((__float_as_uint(*amax) >> 23) & 0xffu);
        if (e >= 20 && e <= 250) {
            s = __uint_as_float((unsigned)(267 - e) << 23);
            inv = __uint_as_float((unsigned)(e - 13) << 23);
        }
    }
}

// (shared by norm.hip and pointwise.hip; 256-thread workgroups)
// by-product of the applies for the f16x3 contraction kernels: max |value written|, one atomic per block
// (*amax_out must hold a non-negative float, normally 0, before the launch)
__device__ __forceinline__ void block_amax(float m, float* __restrict__ amax_out) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
    __shared__ float sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(sm[0], sm[1]), fmaxf(sm[2], sm[3]));
        if (m > 0.f) atomicMax(reinterpret_cast<unsigned*>(amax_out), __float_as_uint(m));
    }
}

}  // namespace
