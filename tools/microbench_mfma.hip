// Sustained rate of v_mfma_f32_32x32x16_f16 on gfx950 with operands in registers (no memory traffic), alone and with the
// LDS fragment reads of the convolution kernels mixed in (0.5 ds_read_b128 per MFMA), at 1 and 2 waves per SIMD, for a run
// long enough (~0.3 s) that the clocks settle where the power limit puts them.  This is the ceiling the split-precision
// convolution kernels are compared with in DESIGN.md.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_mfma.hip -o /tmp/mm && /tmp/mm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDS, int RANDOM>
__global__ void __launch_bounds__(256, 2) k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 sm[64 * 8 * 8];
    const int lane = threadIdx.x & 63;
    // RANDOM = 1: operands with random significands and signs (what real activations and weights look like to the
    // multiplier array: the power drawn, and with it the clock the chip can hold, depends on the bits that toggle)
    for (int i = threadIdx.x; i < 64 * 8 * 8; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
        sm[i] = RANDOM ? (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f)) : (_Float16)(0.001f * (i & 63));
    }
    __syncthreads();
    v8h a[3], b[2];
#pragma unroll
    for (int i = 0; i < 3; i++) a[i] = *reinterpret_cast<const v8h*>(sm + (lane + i) * 8);
#pragma unroll
    for (int i = 0; i < 2; i++) b[i] = *reinterpret_cast<const v8h*>(sm + (lane + 8 + i) * 8);
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    v8h a2[3];
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 6; u += 2) {                  // 36 MFMAs per trip, 18 fragment reads when LDS, one group ahead
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 3; i++) a2[i] = *reinterpret_cast<const v8h*>(sm + ((lane + i + u + it) & 63) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 6; m++)
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m % 3], b[m & 1], acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (LDS) {
#pragma unroll
                for (int i = 0; i < 3; i++) a[i] = *reinterpret_cast<const v8h*>(sm + ((lane + i + u + 1 + 2 * it) & 63) * 8);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int m = 0; m < 6; m++)
                acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(LDS ? a2[m % 3] : a[m % 3], b[m & 1], acc[m & 3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* d; hipMalloc(&d, 256 * 2 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rnd = 0; rnd < 2; rnd++)
    for (int lds = 0; lds < 2; lds++)
        for (int wps = 1; wps <= 2; wps++) {
            const int blocks = 256 * wps, iters = 120000 / wps;
            const double mfma = (double)blocks * 4 * iters * 36;
            printf("%s %s  waves/SIMD=%d  PFLOP/s per ~60 ms launch, back to back:", rnd ? "random operands" : "smooth operands", lds ? "MFMA + ds_read_b128" : "MFMA only          ", wps);
            for (int rep = 0; rep < 24; rep++) {
                float ms = 0;
                hipEventRecord(e0);
                if (rnd) { if (lds) hipLaunchKernelGGL((k<1, 1>), dim3(blocks), dim3(256), 0, 0, d, iters); else hipLaunchKernelGGL((k<0, 1>), dim3(blocks), dim3(256), 0, 0, d, iters); }
                else     { if (lds) hipLaunchKernelGGL((k<1, 0>), dim3(blocks), dim3(256), 0, 0, d, iters); else hipLaunchKernelGGL((k<0, 0>), dim3(blocks), dim3(256), 0, 0, d, iters); }
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
                if (rep % 3 == 2) printf(" %.2f", mfma * 32768.0 / ms * 1e-12);
            }
            printf("\n");
        }
    return 0;
}
