// sg3_modgrad.hip -- gradient of the per-sample effective weights of modulated_conv2d with respect to w and s.
//
// Reference (models/stylegan3/networks_stylegan3.py:39-56): the effective weights of sample n are
//     wn = w * rsqrt(mean_{i,t} w^2)   per output channel o            (a[o])
//     sn = s * rsqrt(mean_{n,i} s^2)   over the whole batch            (b)
//     m  = wn[o,i,t] * sn[n,i]
//     d  = rsqrt(sum_{i,t} m^2 + 1e-8) per (n, o)                      (demodulation)
//     w_eff = m * d * g[n,i]                                            (g = input_gain; without demodulation: w * s * g)
// and PTI's backward needs dL/dw, dL/ds given G = dL/dw_eff (the weight-gradient kernel's output).  Autograd differentiates
// that chain op by op: ~70 launches of ~5 us per layer and step, 1000 of the 1330 launches of a PTI step.  The same algebra
// in closed form, four launches per layer:
//     dE = G g;  t1 = sum_{i,t} dE m;  dM = dE d - d^3 t1 m                                   (kernel 1, block per (n, o), in place)
//     dSn[n,i] = sum_{o,t} dM wn                                                              (kernel 2, wave per (n, i))
//     dWn = sum_n dM sn;  r1 = sum_{i,t} dWn w;  dW = a dWn - a^3 w r1 / (I T)                (kernel 3, block per o)
//     r2 = sum_{n,i} dSn s;  dS = b dSn - b^3 s r2 / (N I)                                    (kernel 4, one block)
// All sums are fixed-order tree reductions (no atomics): gradients are reproducible run to run.
#include "sg3_common.h"
#include <cmath>

namespace sg3 {

__device__ __forceinline__ float block_sum(float v, float* red) {      // 256 threads; result to every thread
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

__device__ __forceinline__ float gain_of(const sg3_modgrad_params& p, int n, int i) {
    if (p.inputGainMode == 1) return p.inputGain[0];
    if (p.inputGainMode == 2) return p.inputGain[i];
    if (p.inputGainMode == 3) return p.inputGain[(size_t)n * p.I + i];
    return 1.f;
}

// b = rsqrt(mean s^2) over the whole batch, recomputed per block (N I <= a few thousand values)
__device__ __forceinline__ float style_norm(const sg3_modgrad_params& p, float* red) {
    float q = 0.f;
    for (int j = threadIdx.x; j < p.N * p.I; j += 256) { const float v = p.s[j]; q += v * v; }
    return rsqrtf(block_sum(q, red) / (float)(p.N * p.I));
}

// kernel 1: G[n,o,:,:] -> dM in place; also a[o] (from the n = 0 blocks)
__global__ void __launch_bounds__(256)
modgrad_dm_kernel(sg3_modgrad_params p) {
    __shared__ float red[4];
    const int o = blockIdx.x, n = blockIdx.y;
    const int IT = p.I * p.T;
    const float* w = p.w + (size_t)o * IT;
    float* G = p.G + ((size_t)n * p.O + o) * IT;
    if (!p.demodulate) {                                   // w_eff = w s g: dM = G g
        for (int j = threadIdx.x; j < IT; j += 256) G[j] *= gain_of(p, n, j / p.T);
        if (n == 0 && threadIdx.x == 0) p.a[o] = 1.f;
        return;
    }
    float q = 0.f;
    for (int j = threadIdx.x; j < IT; j += 256) { const float v = w[j]; q += v * v; }
    const float a = rsqrtf(block_sum(q, red) / (float)IT);
    const float b = style_norm(p, red);
    if (n == 0 && threadIdx.x == 0) p.a[o] = a;
    const float* s = p.s + (size_t)n * p.I;
    float m2 = 0.f, t1 = 0.f;
    for (int j = threadIdx.x; j < IT; j += 256) {
        const int i = j / p.T;
        const float m = (w[j] * a) * (s[i] * b);
        const float dE = G[j] * gain_of(p, n, i);
        m2 += m * m; t1 += dE * m;
    }
    m2 = block_sum(m2, red);
    t1 = block_sum(t1, red);
    const float d = rsqrtf(m2 + 1e-8f);
    const float c = d * d * d * t1;
    for (int j = threadIdx.x; j < IT; j += 256) {
        const int i = j / p.T;
        const float m = (w[j] * a) * (s[i] * b);
        G[j] = G[j] * gain_of(p, n, i) * d - c * m;
    }
}

// kernel 2: dSn[n,i] = sum_{o,t} dM[n,o,i,t] wn[o,i,t]; one wave per (n, i), lanes over o
__global__ void __launch_bounds__(256)
modgrad_ds_kernel(sg3_modgrad_params p) {
    const int lane = threadIdx.x & 63;
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (idx >= p.N * p.I) return;
    const int n = idx / p.I, i = idx - n * p.I;
    const int IT = p.I * p.T;
    float acc = 0.f;
    for (int o = lane; o < p.O; o += 64) {
        const float* g = p.G + ((size_t)n * p.O + o) * IT + (size_t)i * p.T;
        const float* w = p.w + (size_t)o * IT + (size_t)i * p.T;
        float v = 0.f;
        for (int t = 0; t < p.T; t++) v = __builtin_fmaf(g[t], w[t], v);
        acc = __builtin_fmaf(v, p.a[o], acc);
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) acc += __shfl_xor(acc, m);
    if (lane == 0) p.dSn[idx] = acc;
}

// kernel 3: dW[o,:,:]
__global__ void __launch_bounds__(256)
modgrad_dw_kernel(sg3_modgrad_params p) {
    __shared__ float red[4];
    const int o = blockIdx.x;
    const int IT = p.I * p.T;
    const float* w = p.w + (size_t)o * IT;
    const float b = p.demodulate ? style_norm(p, red) : 1.f;
    const float a = p.a[o];
    float r1 = 0.f;
    for (int j = threadIdx.x; j < IT; j += 256) {
        const int i = j / p.T;
        float v = 0.f;
        for (int n = 0; n < p.N; n++) v = __builtin_fmaf(p.G[((size_t)n * p.O + o) * IT + j], p.s[(size_t)n * p.I + i] * b, v);
        p.dW[(size_t)o * IT + j] = v;                      // dWn for now
        r1 = __builtin_fmaf(v, w[j], r1);
    }
    if (!p.demodulate) return;                             // dW = dWn
    r1 = block_sum(r1, red);
    const float c = a * a * a * r1 / (float)IT;
    for (int j = threadIdx.x; j < IT; j += 256) p.dW[(size_t)o * IT + j] = a * p.dW[(size_t)o * IT + j] - c * w[j];
}

// kernel 4: dS
__global__ void __launch_bounds__(256)
modgrad_dsfinal_kernel(sg3_modgrad_params p) {
    __shared__ float red[4];
    const int NI = p.N * p.I;
    if (!p.demodulate) {
        for (int j = threadIdx.x; j < NI; j += 256) p.dS[j] = p.dSn[j];
        return;
    }
    const float b = style_norm(p, red);
    float r2 = 0.f;
    for (int j = threadIdx.x; j < NI; j += 256) r2 = __builtin_fmaf(p.dSn[j], p.s[j], r2);
    r2 = block_sum(r2, red);
    const float c = b * b * b * r2 / (float)NI;
    for (int j = threadIdx.x; j < NI; j += 256) p.dS[j] = b * p.dSn[j] - c * p.s[j];
}

// wt[i][o][T-1-t] = w[o][i][t] * a[o], a = rsqrt(mean w[o]^2) or 1: one workgroup per output channel (sg3_modconv_transpose_weights)
__global__ void __launch_bounds__(256)
modconv_transpose_weights_kernel(const float* __restrict__ w, float* __restrict__ wt, int O, int I, int T, int normalise) {
    __shared__ float red[4];
    const int o = blockIdx.x, IT = I * T;
    const float* wo = w + (size_t)o * IT;
    float a = 1.f;
    if (normalise) {
        float q = 0.f;
        for (int j = threadIdx.x; j < IT; j += 256) { const float v = wo[j]; q += v * v; }
        a = rsqrtf(block_sum(q, red) / (float)IT);
    }
    for (int j = threadIdx.x; j < IT; j += 256) {
        const int i = j / T, t = j - i * T;
        wt[((size_t)i * O + o) * T + (T - 1 - t)] = wo[j] * a;
    }
}

} // namespace sg3

extern "C" int sg3_modconv_transpose_weights(const float* w, float* wt, int O, int I, int k, int normalise, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(w && wt && O > 0 && I > 0 && (k == 1 || k == 3), "modconv_transpose_weights: bad arguments");
    hipLaunchKernelGGL(modconv_transpose_weights_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, w, wt, O, I, k * k, normalise ? 1 : 0);
    SG3_LAUNCH_CHECK("modconv_transpose_weights_kernel");
    return SG3_OK;
}

extern "C" int sg3_modulation_backward(const sg3_modgrad_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->G && p->w && p->s && p->dW && p->dS && p->a && p->dSn, "modulation_backward: null tensor");
    SG3_REQUIRE(p->N > 0 && p->O > 0 && p->I > 0 && p->T > 0 && p->N <= 65535, "modulation_backward: bad shape");
    SG3_REQUIRE(p->inputGainMode >= 0 && p->inputGainMode <= 3 && (p->inputGainMode == 0 || p->inputGain), "modulation_backward: bad input gain");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(modgrad_dm_kernel, dim3((unsigned)p->O, (unsigned)p->N), dim3(256), 0, st, *p);
    SG3_LAUNCH_CHECK("modgrad_dm_kernel");
    hipLaunchKernelGGL(modgrad_ds_kernel, dim3((unsigned)ceil_div(p->N * p->I, 4)), dim3(256), 0, st, *p);
    SG3_LAUNCH_CHECK("modgrad_ds_kernel");
    hipLaunchKernelGGL(modgrad_dw_kernel, dim3((unsigned)p->O), dim3(256), 0, st, *p);
    SG3_LAUNCH_CHECK("modgrad_dw_kernel");
    hipLaunchKernelGGL(modgrad_dsfinal_kernel, dim3(1), dim3(256), 0, st, *p);
    SG3_LAUNCH_CHECK("modgrad_dsfinal_kernel");
    return SG3_OK;
}
