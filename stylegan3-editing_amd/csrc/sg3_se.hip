// sg3_se.hip -- the squeeze-and-excitation tail of an IR-SE residual unit in two launches.
//
// Reference (models/setgan/encoder/encoders/helpers.py:57-73 SEModule, :117-120 bottleneck_IR_SE.forward):
//     g   = sigmoid(fc2(relu(fc1(mean_hw(res)))))        two bias-free 1x1 convolutions on a [N,C,1,1] tensor
//     out = shortcut + res * g
// As torch ops that is a mean, two tiny GEMMs, relu, sigmoid and an addcmul per unit -- seven launches of ~5 us each, 24 units
// per IR-SE50 forward.  Here: se_mean_kernel (one wave per (n, c) plane) and se_apply_kernel, whose workgroups each recompute the
// C/16-wide hidden vector of their sample (at most 32 x 512 MACs) and the gates of their 8 channels before streaming
// shortcut + res * gate.  The shortcut may be a strided view (the stride-2 units subsample their input: MaxPool2d(1, 2)).
#include "sg3_common.h"
#include <cmath>

namespace sg3 {

__global__ void __launch_bounds__(256)
se_mean_kernel(const float* __restrict__ x, float* __restrict__ mean, int planes, int hw) {
    const int lane = threadIdx.x & 63;
    const int pl = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pl >= planes) return;
    const float* src = x + (size_t)pl * hw;
    float s = 0.f;
    if ((hw & 3) == 0) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        for (int i = lane; i < hw / 4; i += 64) { const float4 v = s4[i]; s += (v.x + v.y) + (v.z + v.w); }
    } else {
        for (int i = lane; i < hw; i += 64) s += src[i];
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) mean[pl] = s / (float)hw;
}

constexpr int SE_CG = 8;           // channels per workgroup of the apply kernel
constexpr int SE_MAXC = 2048, SE_MAXR = 128;

__global__ void __launch_bounds__(256)
se_apply_kernel(sg3_se_params p, int chunks) {
    __shared__ float sh[SE_MAXR];
    __shared__ float sg[SE_CG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bid = blockIdx.x;
    const int chunk = bid % chunks; bid /= chunks;
    const int groups = (p.C + SE_CG - 1) / SE_CG;
    const int cg = bid % groups, n = bid / groups;
    // One round trip to memory for the whole gate computation: the means and the fc1 rows are read together (nothing between them
    // waits on LDS), and every thread requests its fc2 entry before the barrier that publishes the hidden vector.
    const float* mean = p.mean + (size_t)n * p.C;
    const int gk = tid >> 5, gr = tid & 31;                                 // gate channel (SE_CG = 8 x 32 lanes) and hidden unit
    const int gc = cg * SE_CG + gk;
    const float f2 = (gc < p.C && gr < p.R) ? p.fc2[(size_t)gc * p.R + gr] : 0.f;
    for (int r = wave; r < p.R; r += 4) {                                   // hidden = relu(fc1 @ mean)
        const float* w = p.fc1 + (size_t)r * p.C;
        float s = 0.f;
        for (int i = lane; i < p.C; i += 64) s = __builtin_fmaf(w[i], mean[i], s);
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
        if (lane == 0) sh[r] = fmaxf(s, 0.f);
    }
    __syncthreads();
    {                                                                       // gate = sigmoid(fc2 @ hidden): 32 lanes per channel
        float s = gr < p.R ? f2 * sh[gr] : 0.f;
        if (gc < p.C)
            for (int r = gr + 32; r < p.R; r += 32) s = __builtin_fmaf(p.fc2[(size_t)gc * p.R + r], sh[r], s);
#pragma unroll
        for (int m = 16; m > 0; m >>= 1) s += __shfl_xor(s, m);
        if (gr == 0) sg[gk] = 1.f / (1.f + expf(-s));
    }
    __syncthreads();
    const int hw = p.H * p.W;
    const int per = ((hw + chunks - 1) / chunks + 3) & ~3;      // multiple of 4: float4 lanes
    const int i0 = chunk * per, i1 = min(hw, i0 + per);
    for (int k = 0; k < SE_CG; k++) {
        const int c = cg * SE_CG + k;
        if (c >= p.C) break;
        const float g = sg[k];
        const float* res = p.res + ((size_t)n * p.C + c) * hw;
        float* out = p.out + ((size_t)n * p.C + c) * hw;
        const float* sc = p.shortcut + (size_t)n * p.scStride[0] + (size_t)c * p.scStride[1];
        if (p.scStride[3] == 1 && p.scStride[2] == p.W && ((hw | i0 | i1) & 3) == 0 && (((uintptr_t)sc | (uintptr_t)res | (uintptr_t)out) & 15) == 0) {
            const float4* r4 = reinterpret_cast<const float4*>(res);
            const float4* s4 = reinterpret_cast<const float4*>(sc);
            float4* o4 = reinterpret_cast<float4*>(out);
            for (int i = i0 / 4 + tid; i < i1 / 4; i += 256) {
                const float4 a = r4[i], b = s4[i];
                o4[i] = make_float4(__builtin_fmaf(a.x, g, b.x), __builtin_fmaf(a.y, g, b.y), __builtin_fmaf(a.z, g, b.z), __builtin_fmaf(a.w, g, b.w));
            }
        } else {
            for (int i = i0 + tid; i < i1; i += 256) {
                const int y = i / p.W, x = i - y * p.W;
                out[i] = __builtin_fmaf(res[i], g, sc[(size_t)y * p.scStride[2] + (size_t)x * p.scStride[3]]);
            }
        }
    }
}

} // namespace sg3

extern "C" int sg3_se_residual(const sg3_se_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->res && p->shortcut && p->fc1 && p->fc2 && p->mean && p->out, "se_residual: null tensor");
    SG3_REQUIRE(p->N > 0 && p->C > 0 && p->H > 0 && p->W > 0 && p->R > 0, "se_residual: empty tensor");
    SG3_REQUIRE(p->C <= SE_MAXC && p->R <= SE_MAXR, "se_residual: at most 2048 channels and 128 hidden units");
    SG3_REQUIRE((long long)p->N * p->C * p->H * p->W < 0x7fffffffLL, "se_residual: tensor too large");
    hipStream_t st = (hipStream_t)stream;
    const int planes = p->N * p->C, hw = p->H * p->W;
    hipLaunchKernelGGL(se_mean_kernel, dim3((unsigned)ceil_div(planes, 4)), dim3(256), 0, st, p->res, p->mean, planes, hw);
    SG3_LAUNCH_CHECK("se_mean_kernel");
    // enough workgroups to fill the chip: 8 planes of <= 4096 pixels per workgroup
    const int chunks = std::max(1, std::min(64, ceil_div(hw, 4096)));
    const long long blocks = (long long)p->N * ceil_div(p->C, SE_CG) * chunks;
    hipLaunchKernelGGL(se_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, st, *p, chunks);
    SG3_LAUNCH_CHECK("se_apply_kernel");
    return SG3_OK;
}
