// Do matrix instructions and vector instructions of DIFFERENT waves on one SIMD overlap on gfx950 -- and for WHICH vector
// instruction classes?  (Round 3 measured v_pk_fma_f32 only; MI355X_MICROARCH.md names packed-f32 VALU as the one class that does
// not hide beside MFMAs.  Round 4 adds the classes the F(2,3) staging block really issues.)
// One 512-thread workgroup per CU = two waves per SIMD (waves w and w + 4 share SIMD w).  Each wave runs one of these roles:
//   M : n x v_mfma_f32_32x32x16_f16 on four independent accumulator blocks (random fp16 operands in registers)
//   V<kind>: nv instructions of ONE class on eight independent destination registers (asm, so the class is what is issued):
//       pkfma  v_pk_fma_f32        fma  v_fma_f32       cvt  v_cvt_pkrtz_f16_f32      mix  v_fma_mixlo_f16
//       dpp    v_mov_b32_dpp row_shl:1                  rdl  v_readlane_b32 (to an SGPR)          dsw  ds_write_b128 (own 16 B slot)
//       mix6   the staging block's own mix per channel pair: 2 v_mul + 4 v_fma + 1 cvt_pkrtz + 2 fma_mix (9 instructions)
// and the kernel is timed (in-kernel s_memtime of the slowest wave of each half, median over workgroups) for
//   V-  : vector role alone              MV : waves 0-3 matrix, waves 4-7 vector          MVp: the same, vector waves at s_setprio 1
//   VM  : waves 0-3 vector, 4-7 matrix   (which wave is older matters for the arbitration)
// hidden = 1 - (T_MV(vector wave) - T_M) / T_V : 1 = the vector work vanished behind the partner's MFMAs, 0 = it was serialised.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_coissue.hip -o /tmp/mc && /tmp/mc
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned i) {
    unsigned h = i * 2654435761u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
    return ((int)(h & 0xffffff) - 0x800000) * (1.0f / 8388608.0f);
}

enum { K_PKFMA, K_FMA, K_CVT, K_MIX, K_DPP, K_RDL, K_DSW, K_MIX6, K_COUNT };
static const char* kNames[K_COUNT] = {"v_pk_fma_f32", "v_fma_f32", "v_cvt_pkrtz_f16_f32", "v_fma_mixlo_f16", "v_mov_b32_dpp", "v_readlane_b32", "ds_write_b128",
                                      "staging mix (2 mul 4 fma 1 cvt 2 mix)"};
static const int kPerIter[K_COUNT] = {8, 8, 8, 8, 8, 8, 8, 9};

// role per wave half: 0 idle, 1 matrix (n instructions), 2 vector (n loop trips of the kind's block); prio: s_setprio of the vector role
template <int KIND>
__global__ void __launch_bounds__(512) k(float* out, unsigned long long* stamps, int role0, int n0, int role1, int n1, int vprio) {
    __shared__ u32x4 slots[512];
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
    float xs[8], ys[8];
#pragma unroll
    for (int i = 0; i < 8; i++) {
        x[i] = (v2f){rnd(threadIdx.x * 32 + i), rnd(threadIdx.x * 32 + 8 + i)}; y[i] = (v2f){0.f, 0.f};
        xs[i] = rnd(threadIdx.x * 32 + 16 + i); ys[i] = 0.f;
    }
    const v2f t = {rnd(blockIdx.x + 1) * 0.5f, rnd(blockIdx.x + 77) * 0.5f};
    const float ts = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, rnd(blockIdx.x + 5) * 0.5f)));
    u32x4 wv = {threadIdx.x, 1u, 2u, 3u};
    unsigned sacc = 0;
    __syncthreads();
    if (role == 2 && vprio) __builtin_amdgcn_s_setprio(1);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    if (role == 1) {
        for (int it = 0; it < n; it += 4) {
#pragma unroll
            for (int j = 0; j < 4; j++) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
        }
    } else if (role == 2) {
        for (int it = 0; it < n; it++) {
            if constexpr (KIND == K_PKFMA) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(y[i]) : "v"(x[i]), "v"(t));
            } else if constexpr (KIND == K_FMA) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(ys[i]) : "v"(xs[i]), "s"(ts));
            } else if constexpr (KIND == K_CVT) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(ys[i]) : "v"(xs[i]), "v"(xs[(i + 1) & 7]));
            } else if constexpr (KIND == K_MIX) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(ys[i]) : "v"(xs[i]), "v"(xs[(i + 1) & 7]));
            } else if constexpr (KIND == K_DPP) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(ys[i]) : "v"(xs[i]));
            } else if constexpr (KIND == K_RDL) {
#pragma unroll
                for (int i = 0; i < 8; i++) { unsigned s; asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(s) : "v"(xs[i]), "n"(3)); sacc ^= s; }
            } else if constexpr (KIND == K_DSW) {
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("ds_write_b128 %0, %1" :: "v"((unsigned)(threadIdx.x * 16)), "v"(wv) : "memory");
            } else {
                // one channel pair's worth of the F(2,3) staging block in single-issue form
                float t2, t3;
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t2) : "s"(ts), "v"(xs[2]));
                asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t3) : "s"(ts), "v"(xs[3]));
                asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(ys[0]) : "v"(xs[0]), "s"(ts), "v"(t2));
                asm volatile("v_fma_f32 %0, %1, %2, -%3" : "=v"(ys[1]) : "v"(xs[1]), "s"(ts), "v"(t3));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(ys[2]) : "v"(xs[1]), "s"(ts), "v"(t2));
                asm volatile("v_fma_f32 %0, -%1, %2, %3" : "=v"(ys[3]) : "v"(xs[1]), "s"(ts), "v"(t2));
                asm volatile("v_cvt_pkrtz_f16_f32 %0, %1, %2" : "=v"(ys[4]) : "v"(ys[0]), "v"(ys[1]));
                asm volatile("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=&v"(ys[5]) : "v"(ys[4]), "v"(ys[0]));
                asm volatile("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(ys[5]) : "v"(ys[4]), "v"(ys[1]));
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime();
    float s = (float)sacc;
#pragma unroll
    for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 16; r++) s += acc[j][r];
#pragma unroll
    for (int i = 0; i < 8; i++) s += y[i].x + y[i].y + ys[i];
    if (KIND == K_DSW) { __syncthreads(); s += (float)slots[threadIdx.x ^ 1].x; }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) stamps[blockIdx.x * 8 + wave] = c1 - c0;
}

template <int KIND>
static void run(float* d, unsigned long long* st, int r0, int n0, int r1, int n1, int vprio, double& t0, double& t1) {
    const int blocks = 256;
    for (int i = 0; i < 100; i++) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(512), 0, 0, d, st, r0, n0, r1, n1, vprio);
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
    t0 = first[blocks / 2]; t1 = second[blocks / 2];
}

template <int KIND>
static void kind(float* d, unsigned long long* st, int nM) {
    // vector work sized to a QUARTER of the matrix role's pipe time at 4 cycles per instruction (the staging block's share in the
    // F(2,3) kernel is about that), so a fully hidden vector role ends long before the matrix role does
    const int trips = nM * 32 / 4 / 4 / kPerIter[KIND];
    const int nV = trips * kPerIter[KIND];
    double a0, a1, m0, m1;
    run<KIND>(d, st, 1, nM, 0, 0, 0, m0, m1);                    const double tM = m0;
    run<KIND>(d, st, 0, 0, 2, trips, 0, a0, a1);                 const double tV = a1;
    printf("%-40s %6d instr  alone %8.0f cyc (%5.2f / instr)", kNames[KIND], nV, tV, tV / nV);
    run<KIND>(d, st, 1, nM, 2, trips, 0, a0, a1);                // matrix older
    printf(" | M(old) V(young): M %8.0f V %8.0f  extra/instr %5.2f", a0, a1, (std::max(a0, a1) - tM) / nV);
    run<KIND>(d, st, 1, nM, 2, trips, 1, a0, a1);                // vector wave at priority 1
    printf(" | V prio 1: M %8.0f V %8.0f  extra/instr %5.2f", a0, a1, (std::max(a0, a1) - tM) / nV);
    run<KIND>(d, st, 2, trips, 1, nM, 0, a0, a1);                // vector older
    printf(" | V(old) M(young): V %8.0f M %8.0f  extra/instr %5.2f\n", a0, a1, (std::max(a0, a1) - tM) / nV);
    fflush(stdout);
}

int main() {
    float* d; unsigned long long* st;
    (void)hipMalloc(&d, 256 * 512 * 4); (void)hipMalloc(&st, 256 * 8 * 8);
    const int n = 4096;                       // matrix instructions per wave: 4096 x 32 = 131 k cycles
    double m0, m1;
    run<K_FMA>(d, st, 1, n, 0, 0, 0, m0, m1);
    printf("matrix role alone: %d x v_mfma_f32_32x32x16_f16 = %.0f cycles (%.2f per instruction)\n", n, m0, m0 / n);
    run<K_FMA>(d, st, 1, n, 1, n, 0, m0, m1);
    printf("matrix | matrix: %.0f / %.0f cycles\n", m0, m1);
    printf("'extra/instr' = (time until BOTH waves of the SIMD are done - matrix role alone) / vector instructions: 0 = hidden, 'alone' value = serialised\n");
    kind<K_PKFMA>(d, st, n); kind<K_FMA>(d, st, n); kind<K_CVT>(d, st, n); kind<K_MIX>(d, st, n);
    kind<K_DPP>(d, st, n); kind<K_RDL>(d, st, n); kind<K_DSW>(d, st, n); kind<K_MIX6>(d, st, n);
    return 0;
}
