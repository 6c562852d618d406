// Do matrix instructions and vector instructions of DIFFERENT waves on one SIMD overlap on gfx950?
// One 512-thread workgroup per CU = two waves per SIMD (waves w and w + 4 share SIMD w).  Each wave runs one of two roles:
//   M: n x v_mfma_f32_32x32x16_f16 on four independent accumulator blocks (random fp16 operands in registers)
//   V: 8 n x v_pk_fma_f32 on eight independent registers (random operands)      -- 8 n x 4 cycles = the n x 32 cycles of role M
// and the kernel is timed (in-kernel s_memtime of the slowest wave, median over workgroups) for the role assignments
//   MM both waves of a SIMD matrix role      VV both vector role      MV waves 0-3 matrix, waves 4-7 vector
//   M- / V-: the second wave idle            Mv / Vm : second wave with a quarter of the other role's work
// If the two pipes overlapped, MV would take what M- takes; if a matrix instruction holds the SIMD's vector issue for its whole
// duration, MV takes M- + V-.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_coissue.hip -o /tmp/mc && /tmp/mc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ __forceinline__ float rnd(unsigned i) {
    unsigned h = i * 2654435761u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
    return ((int)(h & 0xffffff) - 0x800000) * (1.0f / 8388608.0f);
}

// role per wave half: 0 idle, 1 matrix (nM instructions), 2 vector (nV instructions)
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* stamps, int role0, int n0, int role1, int n1) {
    const int wave = threadIdx.x >> 6, half = wave >> 2;
    const int role = half ? role1 : role0, n = half ? n1 : n0;
    v8h a, b;
#pragma unroll
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)rnd(threadIdx.x * 16 + i + blockIdx.x * 7919); b[i] = (_Float16)rnd(threadIdx.x * 16 + 8 + i + blockIdx.x * 104729); }
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[j][r] = 0.f;
    v2f x[8], y[8];
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = (v2f){rnd(threadIdx.x * 32 + i), rnd(threadIdx.x * 32 + 8 + i)}; y[i] = (v2f){0.f, 0.f}; }
    const v2f t = {rnd(blockIdx.x + 1) * 0.5f, rnd(blockIdx.x + 77) * 0.5f};
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < n; it += 4) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < n; it += 8) {
#pragma unroll
            for (int i = 0; i < 8; i++) y[i] = __builtin_elementwise_fma(x[i], t, y[i]);
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[j][r];
#pragma unroll
    for (int i = 0; i < 8; i++) s += y[i].x + y[i].y;
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wave] = c1 - c0;
}

static void run(const char* name, float* d, unsigned long long* st, int r0, int n0, int r1, int n1) {
    const int blocks = 256;
    for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k, dim3(blocks), dim3(512), 0, 0, d, st, r0, n0, r1, n1);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> first, second;
    for (int b = 0; b < blocks; b++) {
        unsigned long long m0 = 0, m1 = 0;
        for (int w = 0; w < 4; w++) { m0 = std::max(m0, h[b * 8 + w]); m1 = std::max(m1, h[b * 8 + 4 + w]); }
        first.push_back((double)m0); second.push_back((double)m1);
    }
    std::sort(first.begin(), first.end()); std::sort(second.begin(), second.end());
    printf("%-44s waves 0-3: %9.0f cycles   waves 4-7: %9.0f cycles\n", name, first[blocks / 2], second[blocks / 2]);
    fflush(stdout);
}

int main() {
    float* d; unsigned long long* st;
    (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 8 * 8);
    const int n = 4096;                       // matrix instructions per wave: 4096 x 32 = 131 k cycles
    printf("expected alone: matrix role %d cycles (32 per instruction), vector role %d cycles (4 per instruction)\n", n * 32, 8 * n * 4);
    run("M-  matrix | idle", d, st, 1, n, 0, 0);
    run("V-  vector | idle", d, st, 2, 8 * n, 0, 0);
    run("MM  matrix | matrix", d, st, 1, n, 1, n);
    run("VV  vector | vector", d, st, 2, 8 * n, 2, 8 * n);
    run("MV  matrix | vector (equal pipe time)", d, st, 1, n, 2, 8 * n);
    run("Mv  matrix | vector (a quarter)", d, st, 1, n, 2, 2 * n);
    run("Mv8 matrix | vector (an eighth)", d, st, 1, n, 2, n);
    run("Vm  vector | matrix (a quarter)", d, st, 2, 8 * n, 1, n / 4);
    return 0;
}
