// sg3_conv2d.hip -- plain NCHW convolution as an fp32-exact MFMA implicit GEMM, for the ReStyle encoder.
//
// Replaces, for eval-mode inference, the Conv2d (+ BatchNorm2d + PReLU / LeakyReLU) module calls of the IR-SE50
// backbone and the GradualStyleBlock heads (reference models/setgan/encoder/encoders/helpers.py:98-120,
// restyle_psp_encoders.py:26-50, map2style.py:15-24).  Same operand staging scheme as the modulated convolution
// (sg3_modconv.hip): weights pre-packed [O][I/KC][tap][KC], the input patch of KC channels staged once per K chunk in
// LDS, channel-interleaved by 4 so that the im2col view of a tap is one ds_read_b128 per lane, next chunk fetched
// global->registers during the MFMA loop.  Differences: stride 1 or 2; a per-input-channel affine (the BatchNorm that
// precedes the first convolution of a bottleneck) applied while staging, only inside the image so zero padding stays
// zero; epilogue bias (folded BatchNorm / conv bias) and PReLU / leaky ReLU.
#include "sg3_common.h"

namespace sg3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct PlainConvParams {
    const float* x; const float* wp; const float* inScale; const float* inShift; const float* bias; const float* slope; float* out;
    int N, I, O, H, W, outH, outW, stride, pad, act;
    int nch, xTiles, yTiles, mTiles, totalBlocks;
};

template <int KS, int STRIDE, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(256)
conv2d_mfma_kernel(PlainConvParams p) {
    constexpr int TAPS = KS * KS, KC = (KS == 3) ? 8 : 16;
    constexpr int BM = WM * TM * 32;
    constexpr int ROWS = WN * TN;
    constexpr int PH = (ROWS - 1) * STRIDE + KS, PW = 31 * STRIDE + KS;
    constexpr int AS = TAPS * KC + 4;
    constexpr int PLANE = PH * PW * 4;
    constexpr int NPL = KC / 4;
    constexpr int A_V4 = BM * TAPS * KC / 4;
    constexpr int A_PER = (A_V4 + 255) / 256;
    constexpr int B_EL = KC * PH * PW;
    constexpr int B_PER = (B_EL + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    __shared__ __attribute__((aligned(16))) float smem[BM * AS + NPL * PLANE];
    float* sA = smem;
    float* sB = smem + BM * AS;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    int bid = blockIdx.x;
    {
        const int nb = p.totalBlocks, q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int mt = bid % p.mTiles; bid /= p.mTiles;
    const int xt = bid % p.xTiles; bid /= p.xTiles;
    const int yt = bid % p.yTiles; const int n = bid / p.yTiles;
    const int o0 = mt * BM, x0 = xt * 32, y0 = yt * ROWS;
    const float* xin = p.x + (size_t)n * p.I * p.H * p.W;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    f32x4 ra[A_PER];
    float rb[B_PER];

    auto fetch = [&](int ch) {
#pragma unroll
        for (int q = 0; q < A_PER; q++) {
            const int v = tid + 256 * q;
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (A_V4 % 256 == 0 || v < A_V4) {
                const int row = v / (TAPS * KC / 4), col = v % (TAPS * KC / 4);
                if (o0 + row < p.O)
                    val = *reinterpret_cast<const f32x4*>(p.wp + ((size_t)(o0 + row) * p.nch + ch) * (TAPS * KC) + col * 4);
            }
            ra[q] = val;
        }
#pragma unroll
        for (int q = 0; q < B_PER; q++) {
            const int e = tid + 256 * q;
            float val = 0.f;
            if (B_EL % 256 == 0 || e < B_EL) {
                const int px = e % PW, t = e / PW, py = t % PH, c = t / PH;
                const int ci = ch * KC + c, gy = y0 * STRIDE - p.pad + py, gx = x0 * STRIDE - p.pad + px;
                if (ci < p.I && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W) {
                    val = xin[((size_t)ci * p.H + gy) * p.W + gx];
                    if (p.inScale) val = val * p.inScale[ci] + (p.inShift ? p.inShift[ci] : 0.f);
                }
            }
            rb[q] = val;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int q = 0; q < A_PER; q++) {
            const int v = tid + 256 * q;
            if (A_V4 % 256 == 0 || v < A_V4) {
                const int row = v / (TAPS * KC / 4), col = v % (TAPS * KC / 4);
                *reinterpret_cast<f32x4*>(sA + row * AS + col * 4) = ra[q];
            }
        }
#pragma unroll
        for (int q = 0; q < B_PER; q++) {
            const int e = tid + 256 * q;
            if (B_EL % 256 == 0 || e < B_EL) {
                const int px = e % PW, t = e / PW, py = t % PH, c = t / PH;
                sB[(c >> 2) * PLANE + (py * PW + px) * 4 + (c & 3)] = rb[q];
            }
        }
    };

    fetch(0);
    for (int ch = 0; ch < p.nch; ch++) {
        __syncthreads();
        stage();
        __syncthreads();
        if (ch + 1 < p.nch) fetch(ch + 1);
#pragma unroll
        for (int tap = 0; tap < TAPS; tap++) {
            const int ky = tap / KS, kx = tap % KS;
#pragma unroll
            for (int c8 = 0; c8 < KC / 8; c8++) {
                f32x4 fa[TM], fb[TN];
#pragma unroll
                for (int a = 0; a < TM; a++)
                    fa[a] = *reinterpret_cast<const f32x4*>(sA + ((wm * TM + a) * 32 + li) * AS + tap * KC + c8 * 8 + 4 * lh);
#pragma unroll
                for (int b = 0; b < TN; b++)
                    fb[b] = *reinterpret_cast<const f32x4*>(sB + (2 * c8 + lh) * PLANE + (((wn * TN + b) * STRIDE + ky) * PW + li * STRIDE + kx) * 4);
#pragma unroll
                for (int q = 0; q < 4; q++)
#pragma unroll
                    for (int a = 0; a < TM; a++)
#pragma unroll
                        for (int b = 0; b < TN; b++)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[a][q], fb[b][q], acc[a][b], 0, 0, 0);
            }
        }
    }

    float* outp = p.out + (size_t)n * p.O * p.outH * p.outW;
    const int gx = x0 + li;
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = o0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (o >= p.O) continue;
            const float bv = p.bias ? p.bias[o] : 0.f;
            const float sl = p.act == 1 ? p.slope[o] : (p.act == 2 ? p.slope[0] : 1.f);
#pragma unroll
            for (int b = 0; b < TN; b++) {
                const int gy = y0 + wn * TN + b;
                if (gy < p.outH && gx < p.outW) {
                    float v = acc[a][b][r] + bv;
                    if (p.act) v = v < 0.f ? v * sl : v;
                    outp[((size_t)o * p.outH + gy) * p.outW + gx] = v;
                }
            }
        }
}

__global__ void __launch_bounds__(256)
conv2d_pack_kernel(const float* w, const float* outScale, float* wp, int O, int I, int taps, int kc, int nch) {
    const int o = blockIdx.x;
    const float sc = outScale ? outScale[o] : 1.f;
    float* dst = wp + (size_t)o * nch * taps * kc;
    for (int j = threadIdx.x; j < nch * taps * kc; j += 256) {
        const int c = j % kc, t = (j / kc) % taps, ch = j / (kc * taps);
        const int i = ch * kc + c;
        dst[j] = i < I ? w[((size_t)o * I + i) * taps + t] * sc : 0.f;
    }
}

template <int KS, int STRIDE, int WM, int WN, int TM, int TN>
static int launch_plain(const sg3_conv2d_params& q, hipStream_t st) {
    constexpr int BM = WM * TM * 32, ROWS = WN * TN, KC = (KS == 3) ? 8 : 16;
    PlainConvParams p;
    p.x = q.x; p.wp = q.wPacked; p.inScale = q.inScale; p.inShift = q.inShift; p.bias = q.bias; p.slope = q.slope; p.out = q.out;
    p.N = q.N; p.I = q.I; p.O = q.O; p.H = q.H; p.W = q.W; p.stride = q.stride; p.pad = q.pad; p.act = q.act;
    p.outH = (q.H + 2 * q.pad - KS) / STRIDE + 1; p.outW = (q.W + 2 * q.pad - KS) / STRIDE + 1;
    p.nch = ceil_div(q.I, KC);
    p.xTiles = ceil_div(p.outW, 32); p.yTiles = ceil_div(p.outH, ROWS); p.mTiles = ceil_div(q.O, BM);
    const long long total = (long long)p.xTiles * p.yTiles * p.mTiles * q.N;
    if (total > 0x7fffffffLL) { set_error("conv2d: grid too large"); return SG3_BAD_ARG; }
    p.totalBlocks = (int)total;
    hipLaunchKernelGGL((conv2d_mfma_kernel<KS, STRIDE, WM, WN, TM, TN>), dim3((unsigned)total), dim3(256), 0, st, p);
    SG3_LAUNCH_CHECK("conv2d_mfma_kernel");
    return SG3_OK;
}

template <int KS, int STRIDE>
static int dispatch_plain(const sg3_conv2d_params& q, hipStream_t st) {
    // small feature maps (16x16 and below) get the 4-row tile, large ones the 128-channel tile when O allows
    const int outH = (q.H + 2 * q.pad - KS) / STRIDE + 1;
    if (q.O >= 128 && outH >= 8) return launch_plain<KS, STRIDE, 2, 2, 2, 2>(q, st);
    if (q.O > 32) return launch_plain<KS, STRIDE, 2, 2, 1, 1>(q, st);          // 64 x (2 rows x 32)
    return launch_plain<KS, STRIDE, 1, 4, 1, 1>(q, st);                        // 32 x (4 rows x 32)
}

} // namespace sg3

extern "C" {

int sg3_conv2d_pack(const float* w, const float* outScale, float* wPacked, int O, int I, int k, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(w && wPacked && O > 0 && I > 0 && (k == 1 || k == 3), "conv2d_pack: bad arguments");
    const int kc = k == 3 ? 8 : 16;
    hipLaunchKernelGGL(conv2d_pack_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, w, outScale, wPacked, O, I, k * k, kc, ceil_div(I, kc));
    SG3_LAUNCH_CHECK("conv2d_pack_kernel");
    return SG3_OK;
}

int sg3_conv2d(const sg3_conv2d_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->x && p->wPacked && p->out, "conv2d: null tensor");
    SG3_REQUIRE(p->N > 0 && p->I > 0 && p->O > 0 && p->H > 0 && p->W > 0, "conv2d: empty tensor");
    SG3_REQUIRE(p->k == 1 || p->k == 3, "conv2d: kernel size must be 1 or 3");
    SG3_REQUIRE(p->stride == 1 || p->stride == 2, "conv2d: stride must be 1 or 2");
    SG3_REQUIRE(p->pad >= 0 && p->pad <= p->k / 2 + 1, "conv2d: bad padding");
    SG3_REQUIRE(p->act >= 0 && p->act <= 2, "conv2d: act must be 0, 1 or 2");
    SG3_REQUIRE(p->act == 0 || p->slope, "conv2d: slope missing");
    SG3_REQUIRE(p->H + 2 * p->pad >= p->k && p->W + 2 * p->pad >= p->k, "conv2d: empty output");
    hipStream_t st = (hipStream_t)stream;
    if (p->k == 3) return p->stride == 1 ? dispatch_plain<3, 1>(*p, st) : dispatch_plain<3, 2>(*p, st);
    return p->stride == 1 ? dispatch_plain<1, 1>(*p, st) : dispatch_plain<1, 2>(*p, st);
}

} // extern "C"
