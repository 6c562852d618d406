// sg3_split.h -- operand handling shared by the split-precision (fp16 x 3) MFMA kernels.
#pragma once
#include "sg3_common.h"

namespace sg3 {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef _Float16 v2h __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <typename T> struct bufld;
template <> struct bufld<float> {
    static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, (int)soff, 0));
    }
    // One element kept RAW until the caller converts it (a conversion at the request would wait for the data there).
    static __device__ __forceinline__ unsigned ldraw(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
        return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, (int)soff, 0);
    }
    static __device__ __forceinline__ float fromraw(unsigned v) { return __builtin_bit_cast(float, v); }
    // Two adjacent elements with one load, kept RAW (raw2) until the caller needs them (unpack2).
    // NOTE: the elements are copied to scalars before __builtin_bit_cast: hipcc 7.2 (clang 22.0.0git roc-7.2.0) compiles
    // __builtin_bit_cast(float, v.y) on an ext-vector element expression as a read of element 0.
    typedef unsigned raw2 __attribute__((ext_vector_type(2)));
    static __device__ __forceinline__ raw2 ld2(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
        return __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, (int)soff, 0);
    }
    static __device__ __forceinline__ void unpack2(raw2 v, float& a, float& b) {
        const unsigned lo = v.x, hi = v.y;
        a = __builtin_bit_cast(float, lo); b = __builtin_bit_cast(float, hi);
    }
};
template <> struct bufld<_Float16> {
    static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
        return (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(r, (int)off, (int)soff, 0));
    }
    static __device__ __forceinline__ unsigned ldraw(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
        return (unsigned)__builtin_amdgcn_raw_buffer_load_b16(r, (int)off, (int)soff, 0);
    }
    static __device__ __forceinline__ float fromraw(unsigned v) { return (float)__builtin_bit_cast(_Float16, (unsigned short)v); }
    typedef unsigned raw2;
    static __device__ __forceinline__ raw2 ld2(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff) {
        return __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, (int)soff, 0);
    }
    static __device__ __forceinline__ void unpack2(raw2 v, float& a, float& b) {
        const v2h h = __builtin_bit_cast(v2h, v);
        a = (float)h.x; b = (float)h.y;
    }
};

// x = hi + lo with hi = the top 11 significand bits of x (exactly representable in fp16 for normal-range values) and
// lo = fp16(x - hi); two values per call, packed for the MFMA operand registers
__device__ __forceinline__ v2h round2(float x0, float x1) {          // plain fp16 form: round to nearest even
    typedef float f2 __attribute__((ext_vector_type(2)));
    return __builtin_convertvector((f2){x0, x1}, v2h);
}
// a * b as ONE single-issue v_mul_f32, hidden from hipcc's SLP vectoriser: adjacent fp32 multiplies are otherwise paired into
// v_pk_mul_f32, the one vector class that does not execute beside the SIMD's other wave's matrix instructions
// (profiles/r04_mfma_valu_coissue.txt: a packed fp32 instruction costs its full ~10 cycles of matrix-pipe time there)
__device__ __forceinline__ float mul_single(float a, float b) {
    float r;
    asm("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ void split2(float x0, float x1, v2h& hi, v2h& lo) {
    const float h0 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x0) & 0xffffe000u);
    const float h1 = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x1) & 0xffffe000u);
    hi = __builtin_bit_cast(v2h, __builtin_amdgcn_cvt_pkrtz(h0, h1));
    lo = __builtin_bit_cast(v2h, __builtin_amdgcn_cvt_pkrtz(x0 - h0, x1 - h1));
}

} // namespace sg3
