// v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 on RANDOM operands: the same 64x64 output tile per wave, the same K = 32
// step, the same LDS fragment traffic (8 ds_read_b128 per step), 2 waves per SIMD; PFLOP/s and the in-kernel clock
// (s_memtime / s_memrealtime) after ~1 s of back-to-back launches.  MI355X_MICROARCH.md (DVFS give-back, item 7) reports the
// 16x16x32 shape 1.12-1.15 x faster at equal cycles per FLOP where the chip lowers its clock under load; this checks it on the
// operand mix of the split-precision convolution before any kernel is restructured around it.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_mfma_shape.hip -o /tmp/ms && /tmp/ms
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int LDS>
__global__ void __launch_bounds__(256, 2) k(float* out, unsigned long long* stamps, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 sm[128 * 8 * 8];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 128 * 8 * 8; i += 256) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
        sm[i] = (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f));
    }
    __syncthreads();
    v8h a[4], b[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { a[i] = *reinterpret_cast<const v8h*>(sm + (lane + i) * 8); b[i] = *reinterpret_cast<const v8h*>(sm + (lane + 16 + i) * 8); }
    f32x16 acc32[4];
    f32x4 acc16[16];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc32[j][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 16; j++) acc16[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; it++) {
        v8h a2[4], b2[4];
        if (LDS) {
#pragma unroll
            for (int i = 0; i < 4; i++) { a2[i] = *reinterpret_cast<const v8h*>(sm + ((lane + i + 3 * it) & 127) * 8); b2[i] = *reinterpret_cast<const v8h*>(sm + ((lane + 40 + i + 5 * it) & 127) * 8); }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (SHAPE == 32) {
            // 2 x 2 blocks of 32x32, two K = 16 halves: a[rowblock * 2 + khalf], b[colblock * 2 + khalf]
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int m = 0; m < 4; m++)
                    acc32[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(m >> 1) * 2 + kh], b[(m & 1) * 2 + kh], acc32[m], 0, 0, 0);
        } else if (SHAPE == 1616) {
            // legacy K = 16 instruction on the same 16x16 blocks: two per block and K = 32 step (operands: 4 halfs per lane)
            typedef _Float16 v4h __attribute__((ext_vector_type(4)));
#pragma unroll
            for (int kh = 0; kh < 2; kh++)
#pragma unroll
                for (int m = 0; m < 16; m++) {
                    const v4h av = kh ? (v4h){a[m >> 2][4], a[m >> 2][5], a[m >> 2][6], a[m >> 2][7]} : (v4h){a[m >> 2][0], a[m >> 2][1], a[m >> 2][2], a[m >> 2][3]};
                    const v4h bv = kh ? (v4h){b[m & 3][4], b[m & 3][5], b[m & 3][6], b[m & 3][7]} : (v4h){b[m & 3][0], b[m & 3][1], b[m & 3][2], b[m & 3][3]};
                    acc16[m] = __builtin_amdgcn_mfma_f32_16x16x16f16(av, bv, acc16[m], 0, 0, 0);
                }
        } else {
            // 4 x 4 blocks of 16x16, one K = 32 step: a[rowblock], b[colblock]
#pragma unroll
            for (int m = 0; m < 16; m++)
                acc16[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m >> 2], b[m & 3], acc16[m], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (LDS) {
#pragma unroll
            for (int i = 0; i < 4; i++) { a[i] = a2[i]; b[i] = b2[i]; }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc32[j][r];
#pragma unroll
    for (int j = 0; j < 16; j++) s += acc16[j][0] + acc16[j][1] + acc16[j][2] + acc16[j][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) { stamps[2 * (blockIdx.x * 4 + (threadIdx.x >> 6))] = c1 - c0; stamps[2 * (blockIdx.x * 4 + (threadIdx.x >> 6)) + 1] = r1 - r0; }
}

template <int SHAPE, int LDS>
static void run(float* d, unsigned long long* st) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 512, iters = 60000;
    float ms = 0; double total = 0;
    while (total < 1000.0) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(blocks), dim3(256), 0, 0, d, st, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1); total += ms;
    }
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < blocks * 4; i++) if (h[2 * i + 1]) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flop = (double)blocks * 4 * iters * 2.0 * 64 * 64 * 32;
    printf("%-9s %-22s %6.3f PFLOP/s  %7.2f ms  clock %.3f GHz  %.1f cycles per K=32 step of a 64x64 tile (ideal 256)\n", SHAPE == 32 ? "32x32x16" : (SHAPE == 16 ? "16x16x32" : "16x16x16"),
           LDS ? "+ 8 ds_read_b128/step" : "operands in registers", flop / ms * 1e-12, ms, ghz[ghz.size() / 2], ms * 1e-3 * ghz[ghz.size() / 2] * 1e9 / iters / 2.0);
    fflush(stdout);
}

int main() {
    float* d; unsigned long long* st;
    (void)hipMalloc(&d, 512 * 256 * 4); (void)hipMalloc(&st, 512 * 4 * 16);
    for (int rep = 0; rep < 2; rep++) {
        run<32, 0>(d, st); run<16, 0>(d, st); run<1616, 0>(d, st); run<32, 1>(d, st); run<16, 1>(d, st);
    }
    return 0;
}
