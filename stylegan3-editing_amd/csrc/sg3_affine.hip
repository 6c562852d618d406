// sg3_affine.hip -- the style vectors of ALL synthesis layers in one launch, and the transform algebra of SynthesisInput in one.
//
// Reference (models/stylegan3/networks_stylegan3.py): every SynthesisLayer runs its own FullyConnectedLayer on its own latent,
//     styles = addmm(b * bias_gain, w_latent, (weight * weight_gain).t())            (:88-96, :349; ToRGB: * 1/sqrt(fan_in), :350-352)
// and SynthesisInput.forward builds per-sample frequencies / phases / amplitudes from four affine outputs with ~35 small torch
// ops (:204-230).  On the GPU each of those is a launch of ~5 us that a captured graph replays one after the other: 0.45 ms of a
// 21 ms FFHQ-1024 forward.  The arithmetic is a few MFLOP; here it is two launches.
//
// affine_batch_kernel: one wave per output row r (a style channel of some layer): the row's weights stay in registers (gain
// folded in by the caller, exactly `weight * weight_gain` rounded once as in the reference), each sample's latent is read through
// L2, the dot product is reduced over the wave, then  (dot + bias[r]) * scale[r]  in the reference's order.  Layer j's styles
// form a dense [N, C_j] block at out + N * rowStart[j] -- what the convolution's prep pass reads.
// input_transform_kernel: thread (n, c) of one workgroup row: t' = t / |t[:2]| (optional), M = R(t') T(t') U (3x3 products in
// the reference's association, k ascending), then phases / freqs / amplitudes of channel c.
#include "sg3_common.h"
#include <cmath>

namespace sg3 {

__global__ void __launch_bounds__(256)
affine_batch_kernel(sg3_affine_batch_params p) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);                    // wave-uniform
    if (r >= p.rows) return;
    int j = 0;
    while (j + 1 < p.layers && r >= p.rowStart[j + 1]) j++;
    const int r0 = p.rowStart[j], cj = p.rowStart[j + 1] - r0;
    const float* wrow = p.weight + (size_t)r * p.wDim;
    const float* lat = p.ws + (size_t)p.wsIndex[j] * p.wsStrideL;
    const float b = p.bias ? p.bias[r] : 0.f;
    const float sc = p.scale ? p.scale[r] : 1.f;
    float* out = p.out + (size_t)p.N * r0 + (r - r0);
    for (int n = 0; n < p.N; n++) {
        const float* x = lat + (size_t)n * p.wsStrideN;
        float acc = 0.f;
        for (int k = lane; k < p.wDim; k += 64) acc = __builtin_fmaf(x[k], wrow[k], acc);
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
        if (lane == 0) out[(size_t)n * cj] = (acc + b) * sc;
    }
}

// c = a @ b for 3x3 row-major matrices, k ascending with one rounding per step (the fp32 matrix product of the BLAS library)
__device__ __forceinline__ void mat3(const float (&a)[9], const float (&b)[9], float (&c)[9]) {
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) {
            float s = a[3 * i] * b[k];
            s = __builtin_fmaf(a[3 * i + 1], b[3 + k], s);
            s = __builtin_fmaf(a[3 * i + 2], b[6 + k], s);
            c[3 * i + k] = s;
        }
}

__global__ void __launch_bounds__(256)
input_transform_kernel(sg3_input_transform_params p) {
    const int n = blockIdx.y;
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= p.C) return;
    float t[4];
#pragma unroll
    for (int i = 0; i < 4; i++) t[i] = p.t[4 * n + i];
    if (p.normalise) {
        const float nrm = sqrtf(t[0] * t[0] + t[1] * t[1]);
#pragma unroll
        for (int i = 0; i < 4; i++) t[i] = t[i] / nrm;
    }
    const float mr[9] = {t[0], -t[1], 0.f, t[1], t[0], 0.f, 0.f, 0.f, 1.f};
    const float mt[9] = {1.f, 0.f, -t[2], 0.f, 1.f, -t[3], 0.f, 0.f, 1.f};
    float u[9], a[9], m[9];
#pragma unroll
    for (int i = 0; i < 9; i++) u[i] = p.user[(size_t)n * p.userStrideN + i];
    mat3(mr, mt, a);
    mat3(a, u, m);
    const float f0 = p.freqs[2 * c], f1 = p.freqs[2 * c + 1];
    // phases + (freqs @ M[:2, 2:]), freqs @ M[:2, :2]: K = 2 products, k ascending
    const float ph = p.phases[c] + __builtin_fmaf(f1, m[5], f0 * m[2]);
    const float g0 = __builtin_fmaf(f1, m[3], f0 * m[0]);
    const float g1 = __builtin_fmaf(f1, m[4], f0 * m[1]);
    const float nrm = sqrtf(g0 * g0 + g1 * g1);
    float amp = 1.f - (nrm - p.bandwidth) / (p.samplingRate / 2.f - p.bandwidth);
    amp = fminf(fmaxf(amp, 0.f), 1.f);
    const size_t o = (size_t)n * p.C + c;
    p.outFreqs[2 * o] = g0; p.outFreqs[2 * o + 1] = g1;
    p.outPhases[o] = ph;
    p.outAmps[o] = amp;
}

} // namespace sg3

extern "C" {

int sg3_affine_batch(const sg3_affine_batch_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->ws && p->weight && p->rowStart && p->wsIndex && p->out, "affine_batch: null tensor");
    SG3_REQUIRE(p->N > 0 && p->wDim > 0 && p->layers > 0 && p->rows > 0, "affine_batch: empty problem");
    hipLaunchKernelGGL(affine_batch_kernel, dim3((unsigned)ceil_div(p->rows, 4)), dim3(256), 0, (hipStream_t)stream, *p);
    SG3_LAUNCH_CHECK("affine_batch_kernel");
    return SG3_OK;
}

int sg3_input_transform(const sg3_input_transform_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->t && p->user && p->freqs && p->phases && p->outFreqs && p->outPhases && p->outAmps, "input_transform: null tensor");
    SG3_REQUIRE(p->N > 0 && p->C > 0 && p->N <= 65535, "input_transform: bad shape");
    SG3_REQUIRE(p->userStrideN == 0 || p->userStrideN == 9, "input_transform: the user transform is [3,3] (stride 0) or [N,3,3] (stride 9)");
    hipLaunchKernelGGL(input_transform_kernel, dim3((unsigned)ceil_div(p->C, 256), (unsigned)p->N), dim3(256), 0, (hipStream_t)stream, *p);
    SG3_LAUNCH_CHECK("input_transform_kernel");
    return SG3_OK;
}

} // extern "C"
