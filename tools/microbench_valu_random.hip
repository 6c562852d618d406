// fp32 VALU ceiling of gfx950 for v_pk_fma_f32 on SMOOTH vs RANDOM operands, with the in-kernel clock
// (s_memtime / s_memrealtime) each run holds after ~1 s of back-to-back launches -- the VALU analogue of
// tools/microbench_mfma.hip (profiles/r01_mfma_ceiling.txt).  It answers VERDICT r1 "what's weak": a bare loop on smooth
// numbers sustained 147 TFLOP/s (~2.25 GHz) while flrelu_stream_kernel runs its packed FMAs at ~1.4 GHz.
//   mode 0  acc = fma(acc, 1.0001, 0.5)                      smooth operands, registers only (tools/microbench_valu.hip)
//   mode 1  acc[i] = fma(x[(i+r)&7], tap[r], acc[i])         random significands / signs in x and taps, registers only
//   mode 2  the arithmetic of one flrelu up-2 row per trip: 8 samples from LDS (random table, address moves with the trip),
//           H-up 2 x (mul + 5 fma), V-up 2 rows x 2 x (mul + 5 fma), lrelu + clamp (pk_mul, 4 max, 4 med3), V-down 2 x 12 fma,
//           one LDS row hand-off + 12-tap H-down per trip; taps in scalar registers; no global traffic in the loop
//   mode 3  mode 2 + the kernel's HBM stream: per row every lane loads 2 dwords (prefetched three rows ahead) and stores one
//           8-byte pair, each wave walking its own region of a 4 GiB buffer (8 B in + 8 B out per lane and row, like an up-2 layer)
//   mode 4  mode 3 with the V-down scatter (24 of the 76 packed ops of a row) moved to the matrix pipe as banded fp32 MFMA
//           blocks, v_mfma_f32_16x16x4_f32: 16 output rows x 16 columns x 4 input rows per instruction; a 12-tap stride-2 band
//           fills 12 of every 44 input rows of a 16-row block, so a row of 256 columns costs 11 instructions (32 matrix-pipe
//           cycles each).  Operands are taken from the registers as they lie (the real kernel would first have to move the
//           input rows across lane groups): an UPPER bound for the MFMA variant
//           (64 accumulator registers for the 16 rows in flight: 192 registers, 2 waves/SIMD; mode 6 = the same held to 128
//           registers for 4 waves/SIMD, which spills)
//   mode 5  the same with v_mfma_f32_4x4x1_16B_f32: 16 blocks of 4 output rows x 4 columns x 1 input row, whose B / D layout IS
//           the kernel's (lane = column): 18 instructions of 8 matrix-pipe cycles per row, no relayout needed
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/microbench_valu_random.hip -o /tmp/mv && /tmp/mv
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;          // explicit LDS address space: ds_* instructions, not flat_*
typedef __attribute__((address_space(3))) v2f lds_v2f;
typedef __attribute__((address_space(3))) v4f lds_v4f;

struct Taps { float t[12]; };

__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float a) { return (v2f){a, a}; }
__device__ __forceinline__ float rnd(unsigned i) {
    unsigned h = i * 2654435761u; h ^= h >> 13; h *= 2246822519u; h ^= h >> 16;
    return ((int)(h & 0xffffff) - 0x800000) * (1.0f / 8388608.0f);      // uniform in [-1, 1), 24 random significand bits
}

template <int MODE>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(MODE >= 5 ? 4 : 1, 8))) k(float* out, unsigned long long* stamps, int iters, Taps tp, const float* gin, float* gout, unsigned gmask) {
    __shared__ __attribute__((aligned(16))) float sm[4 * 1024 + 4 * 512];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    lds_f* tab = (lds_f*)sm + wave * 1024;          // this wave's random table
    lds_f* row = (lds_f*)sm + 4096 + wave * 512;    // this wave's hand-off row
    for (int i = lane; i < 1024; i += 64) tab[i] = rnd(i + 1024 * (blockIdx.x * 4 + wave));
    for (int i = lane; i < 512; i += 64) row[i] = 0.f;
    __syncthreads();
    v2f x[8], acc[12];
    constexpr bool M16 = MODE == 4 || MODE == 6;
    constexpr int NT = M16 ? 16 : 12, NM = M16 ? 11 : 18;       // accumulator tiles in flight, MFMAs per row
    v4f macc[NT];
    float band[4];
    if (MODE >= 4) {
#pragma unroll
        for (int i = 0; i < NT; i++) macc[i] = (v4f){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 4; i++) band[i] = rnd(threadIdx.x * 32 + i) * 0.6f;      // the band's coefficients, one A operand per lane
    }
#pragma unroll
    for (int i = 0; i < 8; i++) { x[i] = (v2f){rnd(threadIdx.x * 16 + 2 * i + blockIdx.x * 7919), rnd(threadIdx.x * 16 + 2 * i + 1 + blockIdx.x * 104729)}; acc[i] = MODE == 0 ? x[i] + splat((float)i) : splat(0.f); }
#pragma unroll
    for (int i = 8; i < 12; i++) acc[i] = splat(0.f);
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (MODE == 0) {
        const v2f va = splat(tp.t[0]), vb = splat(tp.t[1]);
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int r = 0; r < 12; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = fma2(acc[i], va, vb);
    } else if (MODE == 1) {
        for (int it = 0; it < iters; it++)
#pragma unroll
            for (int r = 0; r < 12; r++)
#pragma unroll
                for (int i = 0; i < 8; i++) acc[i] = fma2(x[(i + r) & 7], splat(tp.t[r]), acc[i]);
    } else if (MODE >= 2) {
        v2f w[6][2];
#pragma unroll
        for (int s = 0; s < 6; s++) { w[s][0] = x[s]; w[s][1] = x[(s + 2) & 7]; }
        // mode 3: this wave's stream position (floats); rows are 128 floats in, 128 floats out
        const unsigned wid = (blockIdx.x * 4 + wave);
        unsigned gpos = wid * (unsigned)(iters * 6 * 128);
        float pre[3][2];
        if (MODE >= 3) {
#pragma unroll
            for (int r = 0; r < 3; r++) { pre[r][0] = gin[((gpos + r * 128) & gmask) + lane]; pre[r][1] = gin[((gpos + r * 128) & gmask) + 64 + lane]; }
        }
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int S = 0; S < 6; S++) {             // six rows per trip, window slots resolved at compile time
                const int base = ((it * 6 + S) * 16 + 2 * lane) & 1015;
                if (MODE >= 3) {
                    // the staged input row goes through LDS like the kernel's (two dword writes), next row requested
                    tab[(base + 512) & 1023] = pre[S % 3][0]; tab[(base + 640) & 1023] = pre[S % 3][1];
                    const unsigned nxt = (gpos + 3 * 128) & gmask;
                    pre[S % 3][0] = gin[nxt + lane]; pre[S % 3][1] = gin[nxt + 64 + lane];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                float xs[8];
#pragma unroll
                for (int q = 0; q < 4; q++) { const v2f t = *reinterpret_cast<const volatile lds_v2f*>(tab + base + 2 * q); xs[2 * q] = t.x; xs[2 * q + 1] = t.y; }
#pragma unroll
                for (int g = 0; g < 2; g++) {
                    v2f a = splat(xs[g]) * (v2f){tp.t[0], tp.t[1]};
#pragma unroll
                    for (int t = 1; t < 6; t++) a = fma2(splat(xs[g + t]), (v2f){tp.t[2 * t], tp.t[2 * t + 1]}, a);
                    w[S][g] = a;
                }
#pragma unroll
                for (int j = 0; j < 2; j++) {
                    v2f u0 = w[(S + 1) % 6][0] * splat(tp.t[j]), u1 = w[(S + 1) % 6][1] * splat(tp.t[j]);
#pragma unroll
                    for (int t = 1; t < 6; t++) { u0 = fma2(w[(S + 1 + t) % 6][0], splat(tp.t[j + 2 * t]), u0); u1 = fma2(w[(S + 1 + t) % 6][1], splat(tp.t[j + 2 * t]), u1); }
                    const v2f s0 = u0 * splat(0.2f), s1 = u1 * splat(0.2f);
                    float a[4] = {__builtin_fmaxf(u0.x, s0.x), __builtin_fmaxf(u0.y, s0.y), __builtin_fmaxf(u1.x, s1.x), __builtin_fmaxf(u1.y, s1.y)};
#pragma unroll
                    for (int c = 0; c < 4; c++) a[c] = __builtin_amdgcn_fmed3f(a[c], -181.f, 181.f);
                    const v2f r0v = {a[0], a[1]}, r1v = {a[2], a[3]};
                    if (MODE >= 4) {
#pragma unroll
                        for (int m = j * (NM / 2 + NM % 2); m < (j == 0 ? NM / 2 + NM % 2 : NM); m++) {
                            const int tile = (S * NM + m) % NT;
                            if (M16) macc[tile] = __builtin_amdgcn_mfma_f32_16x16x4f32(band[m & 3], a[m & 3], macc[tile], 0, 0, 0);
                            else macc[tile] = __builtin_amdgcn_mfma_f32_4x4x1f32(band[m & 3], a[m & 3], macc[tile], 0, 0, 0);
                        }
                    } else
#pragma unroll
                    for (int r = 0; r < 6; r++) {
                        const int slot = (S + 5 - r) % 6;          // six output rows in flight, as in the kernel
                        acc[2 * slot] = fma2(r0v, splat(tp.t[2 * r + j]), acc[2 * slot]);
                        acc[2 * slot + 1] = fma2(r1v, splat(tp.t[2 * r + j]), acc[2 * slot + 1]);
                    }
                }
                // completed output row: hand-off through LDS, 12-tap H-down, result folded back into the oldest accumulator
                const int slot = S % 6;
                const int mt = (S * NM) % NT;                       // the accumulator tile that "completes" with this row (modes 4 / 5)
                if (MODE >= 4) *reinterpret_cast<volatile lds_v4f*>(row + 4 + 4 * lane) = macc[mt];
                else *reinterpret_cast<volatile lds_v4f*>(row + 4 + 4 * lane) = (v4f){acc[2 * slot].x, acc[2 * slot].y, acc[2 * slot + 1].x, acc[2 * slot + 1].y};
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                v2f pr[8];
#pragma unroll
                for (int q = 0; q < 4; q++) { const v4f t = *reinterpret_cast<const volatile lds_v4f*>(row + 4 * ((lane + q) & 63)); pr[2 * q] = (v2f){t.x, t.y}; pr[2 * q + 1] = (v2f){t.z, t.w}; }
                v2f y0 = pr[0] * (v2f){tp.t[0], tp.t[1]}, y1 = pr[1] * (v2f){tp.t[0], tp.t[1]};
#pragma unroll
                for (int q = 1; q < 6; q++) { y0 = fma2(pr[q], (v2f){tp.t[2 * q], tp.t[2 * q + 1]}, y0); y1 = fma2(pr[q + 1], (v2f){tp.t[2 * q], tp.t[2 * q + 1]}, y1); }
                if (MODE >= 3) { *reinterpret_cast<v2f*>(gout + (gpos & gmask) + 2 * lane) = (v2f){y0.x + y0.y, y1.x + y1.y}; gpos += 128; }
                if (MODE >= 4) macc[mt] = (v4f){(y0.x + y0.y) * 0.01f, (y1.x + y1.y) * 0.01f, 0.f, 0.f};
                else {
                    acc[2 * slot] = (v2f){(y0.x + y0.y) * 0.01f, (y1.x + y1.y) * 0.01f};      // keeps the accumulators bounded
                    acc[2 * slot + 1] = splat(0.f);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 12; i++) s += acc[i].x + acc[i].y;
    if (MODE >= 4) {
#pragma unroll
        for (int i = 0; i < NT; i++) s += macc[i].x + macc[i].y + macc[i].z + macc[i].w;
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) { stamps[2 * (blockIdx.x * 4 + wave)] = c1 - c0; stamps[2 * (blockIdx.x * 4 + wave) + 1] = r1 - r0; }
}

template <int MODE>
static void run(const char* name, int bpc, int iters, double pkPerIter, float* d, unsigned long long* st, Taps tp, const float* gin = nullptr, float* gout = nullptr, unsigned gmask = 0) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * bpc;
    float ms = 0; double total = 0;
    while (total < 1000.0) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, st, iters, tp, gin, gout, gmask);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1); total += ms;
    }
    std::vector<unsigned long long> h(blocks * 8);
    (void)hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> ghz;
    for (int i = 0; i < blocks * 4; i++) if (h[2 * i + 1]) ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    const double pk = (double)blocks * 4 * iters * pkPerIter;            // packed wave-instructions
    const double clk = ghz[ghz.size() / 2];
    printf("%-34s waves/SIMD=%d  %7.1f TFLOP/s  %6.3f ms  clock %.3f GHz  packed-op issue %.2f cycles/SIMD each\n", name, bpc, pk * 64 * 4 / ms * 1e-9, ms, clk,
           ms * 1e-3 * clk * 1e9 / (pk / 1024.0));
    fflush(stdout);
}

int main() {
    float* d; unsigned long long* st;
    (void)hipMalloc(&d, 256 * 8 * 256 * 4); (void)hipMalloc(&st, 256 * 8 * 4 * 16);
    // 4 GiB in + 4 GiB out: far beyond the 256 MiB Infinity Cache; random input significands
    const unsigned gfloats = 1u << 30, gmask = gfloats - 1;             // stream positions (multiples of 128 floats) wrap inside the buffers
    float *gin, *gout;
    (void)hipMalloc(&gin, (size_t)gfloats * 4 + 4096); (void)hipMalloc(&gout, (size_t)gfloats * 4 + 4096);
    {
        std::vector<float> h(1 << 24);
        unsigned sd = 99u;
        for (auto& v : h) { sd = sd * 1664525u + 1013904223u; v = ((int)(sd >> 8) - 0x800000) * (1.0f / 8388608.0f); }
        for (size_t off = 0; off < gfloats; off += h.size()) (void)hipMemcpy(gin + off, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
    Taps smooth; for (int i = 0; i < 12; i++) smooth.t[i] = 0; smooth.t[0] = 1.0001f; smooth.t[1] = 0.5f;
    Taps rt = {{0.7391f, -0.3127f, 0.4513f, -0.6589f, 0.2071f, -0.4259f, 0.1873f, -0.5311f, 0.3499f, -0.2203f, 0.6113f, -0.1871f}};
    for (int bpc = 2; bpc <= 8; bpc *= 2) {
        run<0>("pk_fma smooth, registers", bpc, 4000, 96, d, st, smooth);
        run<1>("pk_fma random, registers", bpc, 4000, 96, d, st, rt);
        // packed-FMA-class instructions per trip (6 rows): H-up 12, V-up 24, slope mul 2 x 2, V-down 24, H-down 12 -> 76 per row
        run<2>("flrelu up-2 row arithmetic + LDS", bpc, 700, 6 * 76, d, st, rt);
        run<3>("... + HBM stream 8 B in / 8 B out", bpc, 700, 6 * 76, d, st, rt, gin, gout, gmask);
        // the same 76 packed-op-equivalents of work per row, 24 of them on the matrix pipe: TFLOP/s and "cycles each" stay comparable
        run<4>("... V-down as 11 mfma_16x16x4_f32", bpc, 700, 6 * 76, d, st, rt, gin, gout, gmask);
        run<5>("... V-down as 18 mfma_4x4x1_f32", bpc, 700, 6 * 76, d, st, rt, gin, gout, gmask);
        run<6>("... 11 mfma_16x16x4, 128 regs (spills)", bpc, 700, 6 * 76, d, st, rt, gin, gout, gmask);
    }
    return 0;
}
