// sg3_head_gemm.hip -- the batched small-M GEMMs of the GradualStyleBlock heads of the ReStyle encoders (reference
// models/setgan/encoder/encoders/map2style.py:8-25: the 3x3 stride-2 convolutions on 8x8 ... 2x2 maps, then EqualLinear
// (models/stylegan2/model.py EqualLinear: x @ (W * scale).T + bias * lr_mul); restyle_psp_encoders.py:26-50 runs n_styles of them).
//
//     C[g][m][n] = sum_k act(A[g][m][k]) * W[g][k][n] + bias[g][n]          g = head, m = (image, pixel), act = LeakyReLU(slope)
//
// From the second level on every head has its own input and its own 9.4 MB of weights, and M = images x pixels is 8 ... 256 rows:
// the work is reading the weights ONCE (151 MB per level at 16 heads, 19 us at the HBM rate), not the arithmetic.  Layout for that:
//   * weights packed once per checkpoint (sg3_head_gemm_pack) into matrix-instruction B fragments, split fp16 hi | lo:
//     [g][n/32][k/16][hi|lo][lane][8 halfs], lane = (k block of 8) * 32 + column -- a wave reads one k step of one 32-column block
//     as two 1 KB runs;
//   * a workgroup = one 32-column block of one head (x one block of up to 128 rows): 16 x heads workgroups stream disjoint
//     contiguous weight ranges; its four waves take k steps wave, wave + 4, ... with NO barrier between them -- each wave builds the
//     A fragments it needs straight from global memory (lane = row, 8 consecutive k: two 16-byte loads; A is a few MB and stays
//     in L2), keeps DEPTH k steps of loads in flight (the bytes in flight, not the arithmetic, set the rate), and the four partial
//     sums meet once, in LDS, in a fixed order (results do not depend on timing);
//   * arithmetic as in the encoder's convolutions (sg3_conv2d.hip: x = hi + lo in fp16, lo*hi + hi*lo + hi*hi on
//     v_mfma_f32_32x32x16_f16, fp32 accumulation, fp32-equivalent), with the same range guard: an activation beyond the fp16
//     range raises *flag and the caller repeats its forward on the exact fp32 path.
// The previous level's LeakyReLU is applied to A on the way in (the unfold kernel does the same for the levels it feeds), so the
// EqualLinear needs no separate activation pass.
#include "sg3_common.h"
#include "sg3_split.h"

namespace sg3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct HeadGemmParams {
    const float* a; const v8h* wp; const float* bias; float* c; int* flag;
    int G, M, K, N;
    float slope;
};

__global__ void __launch_bounds__(256)
head_gemm_pack_kernel(const float* __restrict__ w, v8h* __restrict__ wp, int G, int K, int N, int* bad) {
    // one thread per (g, n block, k step, lane): 8 k values of one column -> a hi and a lo fragment entry
    const long long t = (long long)blockIdx.x * 256 + threadIdx.x;
    const int nk = K >> 4, nb = N >> 5;
    if (t >= (long long)G * nb * nk * 64) return;
    const int lane = (int)(t & 63);
    long long r = t >> 6;
    const int kk = (int)(r % nk); r /= nk;
    const int b = (int)(r % nb); const int g = (int)(r / nb);
    const int col = b * 32 + (lane & 31), k0 = kk * 16 + (lane >> 5) * 8;
    const float* src = w + ((size_t)g * K + k0) * N + col;
    v8h hi, lo;
    float peak = 0.f;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const float v0 = src[(size_t)e * N], v1 = src[(size_t)(e + 1) * N];
        peak = fmaxf(peak, fmaxf(fabsf(v0), fabsf(v1)));
        v2h h, l; split2(v0, v1, h, l);
        hi[e] = h.x; hi[e + 1] = h.y; lo[e] = l.x; lo[e + 1] = l.y;
    }
    v8h* dst = wp + ((size_t)(g * nb + b) * nk + kk) * 128 + lane;
    dst[0] = hi; dst[64] = lo;
    if (!(peak <= 65000.f)) atomicOr(bad, 1);
}

template <int MB, int DEPTH>
__global__ void __launch_bounds__(256)
head_gemm_kernel(HeadGemmParams p) {
    __shared__ float red[3][MB][16][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nb = blockIdx.x, g = blockIdx.y, m0 = blockIdx.z * (32 * MB);
    const int row = lane & 31, kb = lane >> 5;
    const int nk = p.K >> 4;
    // this workgroup's weights: nk k steps of 2 KB; a k step past the end reads zeros (buffer range), so the pipeline needs no tail
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.wp + (size_t)(g * (p.N >> 5) + nb) * nk * 128), (short)0, (int)((unsigned)nk * 2048u), 0x00020000);
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.a + (size_t)g * p.M * p.K), (short)0, (int)((unsigned)p.M * (unsigned)p.K * 4u), 0x00020000);
    unsigned aoff[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++) {
        const int r = m0 + mb * 32 + row;
        aoff[mb] = r < p.M ? ((unsigned)r * (unsigned)p.K + kb * 8) * 4u : 0x80000000u;      // rows past M: out of range = zeros
    }
    const unsigned woff = lane * 16u;

    f32x16 acc[MB];
#pragma unroll
    for (int mb = 0; mb < MB; mb++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[mb][r] = 0.f;

    u32x4 sb[DEPTH][2], sa[DEPTH][MB][2];
    float peak = 0.f;
    const float slope = p.slope;

    // software pipeline over this wave's k steps: wave, wave + 4, ...; DEPTH of them in flight
    auto load = [&](int d, int kk) {
        const unsigned wo = (unsigned)kk * 2048u + woff;
        sb[d][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)wo, 0, 0);
        sb[d][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(wo + 1024u), 0, 0);
        const bool in = kk < nk;
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
            const unsigned ao = in ? aoff[mb] + (unsigned)kk * 64u : 0x80000000u;
            sa[d][mb][0] = __builtin_amdgcn_raw_buffer_load_b128(ar, (int)ao, 0, 0);
            sa[d][mb][1] = __builtin_amdgcn_raw_buffer_load_b128(ar, (int)(ao + 16u), 0, 0);
        }
    };
    auto compute = [&](int d) {
        const v8h bh = __builtin_bit_cast(v8h, sb[d][0]), bl = __builtin_bit_cast(v8h, sb[d][1]);
#pragma unroll
        for (int mb = 0; mb < MB; mb++) {
            const u32x4 q0 = sa[d][mb][0], q1 = sa[d][mb][1];
            const unsigned raw[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
            unsigned hw[4], lw[4];
#pragma unroll
            for (int e = 0; e < 4; e++) {
                float v0 = __builtin_bit_cast(float, raw[2 * e]), v1 = __builtin_bit_cast(float, raw[2 * e + 1]);
                v0 = v0 < 0.f ? v0 * slope : v0; v1 = v1 < 0.f ? v1 * slope : v1;
                peak = fmaxf(peak, fmaxf(fabsf(v0), fabsf(v1)));
                v2h h, l; split2(v0, v1, h, l);
                hw[e] = __builtin_bit_cast(unsigned, h); lw[e] = __builtin_bit_cast(unsigned, l);
            }
            const v8h ah = __builtin_bit_cast(v8h, (u32x4){hw[0], hw[1], hw[2], hw[3]});
            const v8h al = __builtin_bit_cast(v8h, (u32x4){lw[0], lw[1], lw[2], lw[3]});
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[mb], 0, 0, 0);
            acc[mb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[mb], 0, 0, 0);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; d++) load(d, wave + 4 * d);
    for (int kk = wave; kk < nk; kk += 4 * DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            compute(d);                                   // a k step past the end multiplies zeros
            load(d, kk + 4 * (d + DEPTH));
        }
    }
    if (!(peak <= 65000.f)) atomicOr(p.flag, 1);

    if (wave) {
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int r = 0; r < 16; r++) red[wave - 1][mb][r][lane] = acc[mb][r];
    }
    __syncthreads();
    if (wave == 0) {
        const int col = nb * 32 + row;
        const float b = p.bias ? p.bias[(size_t)g * p.N + col] : 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; mb++)
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int m = m0 + mb * 32 + (r & 3) + 8 * (r >> 2) + 4 * kb;
                const float v = ((acc[mb][r] + red[0][mb][r][lane]) + red[1][mb][r][lane]) + red[2][mb][r][lane] + b;
                if (m < p.M) p.c[((size_t)g * p.M + m) * p.N + col] = v;
            }
    }
}

template <int MB, int DEPTH>
static int launch_head_gemm(const HeadGemmParams& p, hipStream_t st) {
    const dim3 grid((unsigned)(p.N >> 5), (unsigned)p.G, (unsigned)ceil_div(p.M, 32 * MB));
    hipLaunchKernelGGL((head_gemm_kernel<MB, DEPTH>), grid, dim3(256), 0, st, p);
    SG3_LAUNCH_CHECK("head_gemm_kernel");
    return SG3_OK;
}

} // namespace sg3

extern "C" long long sg3_head_gemm_packed_halfs(int G, int K, int N) {
    if (G <= 0 || K <= 0 || N <= 0 || (K & 15) || (N & 31)) return -1;
    return (long long)G * K * N * 2;
}

extern "C" int sg3_head_gemm_pack(const float* w, void* packed, int G, int K, int N, int* rangeFlag, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(w && packed && rangeFlag, "head_gemm_pack: null tensor");
    SG3_REQUIRE(G > 0 && K > 0 && N > 0 && !(K & 15) && !(N & 31), "head_gemm_pack: K must be a multiple of 16 and N of 32");
    const long long threads = (long long)G * (N >> 5) * (K >> 4) * 64;
    SG3_REQUIRE(threads < (1ll << 39), "head_gemm_pack: too large");
    hipLaunchKernelGGL(head_gemm_pack_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       w, (v8h*)packed, G, K, N, rangeFlag);
    SG3_LAUNCH_CHECK("head_gemm_pack_kernel");
    return SG3_OK;
}

extern "C" int sg3_head_gemm(const sg3_head_gemm_params* q, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(q && q->a && q->wPacked && q->c && q->rangeFlag, "head_gemm: null tensor");
    SG3_REQUIRE(q->G > 0 && q->M > 0 && q->K > 0 && q->N > 0, "head_gemm: empty operand");
    SG3_REQUIRE(!(q->K & 15) && !(q->N & 31), "head_gemm: K must be a multiple of 16 and N of 32");
    SG3_REQUIRE((long long)q->M * q->K * 4 < (1ll << 31), "head_gemm: one head's A matrix must stay below 2 GB");
    SG3_REQUIRE(q->G < 65536 && ceil_div(q->M, 32) < 65536, "head_gemm: grid too large");
    HeadGemmParams p;
    p.a = q->a; p.wp = (const v8h*)q->wPacked; p.bias = q->bias; p.c = q->c; p.flag = q->rangeFlag;
    p.G = q->G; p.M = q->M; p.K = q->K; p.N = q->N; p.slope = q->slope;
    hipStream_t st = (hipStream_t)stream;
    if (p.M <= 32) return launch_head_gemm<1, 6>(p, st);
    if (p.M <= 64) return launch_head_gemm<2, 4>(p, st);
    return launch_head_gemm<4, 3>(p, st);
}
