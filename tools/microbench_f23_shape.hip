// The matrix phase of modconv_f23_kernel in two instruction shapes: v_mfma_f32_32x32x16_f16 as shipped, and
// v_mfma_f32_16x16x32_f16, which needs K = 32 per instruction = a 32-channel chunk.  Same work per "patch row" step in both:
//   2 x ds_read_b128 (the B fragment hi | lo of one patch row) feed 3 filter rows x {hi hi, hi lo, lo hi} products of a
//   32-channel x 32-pixel block over K = 16 (32x32x16: 9 instructions of 32 cycles) or of 2 x 16-channel blocks x 16 pixel pairs
//   over K = 32 (16x16x32: 18 instructions of 16 cycles, 32 channels x 16 pairs x K 32 = the same MACs),
// A fragments resident in registers (random fp16 bits, hi and lo halves as in the split), accumulators per tile row, 2 waves per
// SIMD, one 512-thread workgroup per CU, ~1 s per line; PFLOP/s and the in-kernel clock.  This is the CEILING of what the shape
// can give the kernel: staging, A streaming and the output transform are not in it.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_f23_shape.hip -o /tmp/fs && /tmp/fs
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, int ROWS>
__global__ void __launch_bounds__(512, 2) k(float* out, unsigned long long* stamps, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 sm[4096 * 8];           // 64 KB: a B image to read fragments from
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096 * 8; i += 512) {
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
        const float v = ((int)(h & 0xffff) - 32768) * (1.0f / 32768.0f);
        sm[i] = (_Float16)(((i >> 12) & 1) ? v * 4.8e-4f : v);               // alternate planes: hi values, lo values (2^-11 of them)
    }
    __syncthreads();
    constexpr int NA = SHAPE == 32 ? 6 : 12;
    v8h a[NA];
#pragma unroll
    for (int i = 0; i < NA; i++) a[i] = *reinterpret_cast<const v8h*>(sm + ((lane * 7 + i * 64 + ((i & 1) ? 512 : 0)) & 4095) * 8);
    f32x16 acc32[ROWS];
    f32x4 acc16[2 * ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc32[j][r] = 0.f;
#pragma unroll
    for (int j = 0; j < 2 * ROWS; j++) acc16[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    v8h bh = *reinterpret_cast<const v8h*>(sm + (wave * 64 + lane) * 8), bl = *reinterpret_cast<const v8h*>(sm + (512 + wave * 64 + lane) * 8);
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int q = 0; q < ROWS; q++) {                                       // one patch row per step; its fragment feeds three tile rows
            const int o = ((it * ROWS + q) * 64 + wave * 640 + lane) & 2047;
            const v8h nh = *reinterpret_cast<const v8h*>(sm + (o & ~512) * 8), nl = *reinterpret_cast<const v8h*>(sm + (o | 512) * 8);
            __builtin_amdgcn_sched_barrier(0);
            if (SHAPE == 32) {
#pragma unroll
                for (int ky = 0; ky < 3; ky++) {
                    const int r = (q + ROWS - ky) % ROWS;
                    acc32[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ky + 1], bh, acc32[r], 0, 0, 0);       // lo(A) hi(B)
                    acc32[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ky], bl, acc32[r], 0, 0, 0);           // hi(A) lo(B)
                    acc32[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[2 * ky], bh, acc32[r], 0, 0, 0);           // hi(A) hi(B)
                }
            } else {
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int mb = 0; mb < 2; mb++) {
                        const int r = 2 * ((q + ROWS - ky) % ROWS) + mb;
                        acc16[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[4 * ky + 2 * mb + 1], bh, acc16[r], 0, 0, 0);
                        acc16[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[4 * ky + 2 * mb], bl, acc16[r], 0, 0, 0);
                        acc16[r] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[4 * ky + 2 * mb], bh, acc16[r], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);
            bh = nh; bl = nl;
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < ROWS; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc32[j][r];
#pragma unroll
    for (int j = 0; j < 2 * ROWS; j++) s += acc16[j][0] + acc16[j][1] + acc16[j][2] + acc16[j][3];
    out[blockIdx.x * 512 + threadIdx.x] = s;
    if (lane == 0) { stamps[2 * (blockIdx.x * 8 + wave)] = c1 - c0; stamps[2 * (blockIdx.x * 8 + wave) + 1] = r1 - r0; }
}

template <int SHAPE, int ROWS>
static void run(float* d, unsigned long long* st) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256, iters = 6000;
    float ms = 0; double total = 0;
    while (total < 1000.0) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, ROWS>), dim3(blocks), dim3(512), 0, 0, d, st, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1); total += ms;
    }
    std::vector<unsigned long long> h(blocks * 16);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < blocks * 8; i++) if (h[2 * i + 1]) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double flop = (double)blocks * 8 * iters * ROWS * 9.0 * 2.0 * 32 * 32 * 16;
    const double clk = ghz[ghz.size() / 2];
    printf("%-9s %d tile rows  %6.3f PFLOP/s  %7.2f ms  clock %.3f GHz  %.1f cycles per patch-row step and SIMD (two waves: ideal 576)\n",
           SHAPE == 32 ? "32x32x16" : "16x16x32", ROWS, flop / ms * 1e-12, ms, clk, ms * 1e-3 * clk * 1e9 / iters / ROWS);
    fflush(stdout);
}

int main() {
    float* d; unsigned long long* st;
    (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 8 * 16);
    for (int rep = 0; rep < 2; rep++) {
        run<32, 7>(d, st); run<16, 7>(d, st); run<32, 3>(d, st); run<16, 3>(d, st);
    }
    return 0;
}
