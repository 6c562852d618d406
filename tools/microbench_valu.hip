// Measures the sustained fp32 VALU rate of gfx950 for v_fma_f32 vs v_pk_fma_f32 at 1..8 waves per SIMD.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_valu.hip -o /tmp/mb && /tmp/mb
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(256) k(float* out, int iters, float a, float b) {
    v2f x[8];
#pragma unroll
    for (int i = 0; i < 8; i++) x[i] = (v2f){(float)threadIdx.x + i, (float)i};
    v2f va = {a, a}, vb = {b, b};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (PK) x[i] = __builtin_elementwise_fma(x[i], va, vb);
                else { x[i].x = __builtin_fmaf(x[i].x, a, b); asm volatile("" : "+v"(x[i].x)); x[i].y = __builtin_fmaf(x[i].y, a, b); asm volatile("" : "+v"(x[i].y)); }
            }
    }
    float s = 0; for (int i = 0; i < 8; i++) s += x[i].x + x[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    float* d; hipMalloc(&d, 256 * 256 * 8 * 4 * 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pk = 0; pk < 2; pk++)
        for (int bpc = 1; bpc <= 8; bpc *= 2) {
            int blocks = 256 * bpc, iters = 20000;
            for (int rep = 0; rep < 2; rep++) {
                hipEventRecord(e0);
                if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                else    hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.0001f, 0.5f);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double fma = (double)blocks * 256 * iters * 64 * 2;   // scalar fmas
            printf("%s waves/SIMD=%d  %.1f TFLOP/s (%.3f ms)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", bpc, 2 * fma / ms * 1e-9, ms);
        }
    return 0;
}
