// sg3_modconv.hip -- StyleGAN3 modulated convolution as ONE batch-wide implicit GEMM on the gfx950 matrix cores.
//
// Replaces the grouped convolution of modulated_conv2d (reference models/stylegan3/networks_stylegan3.py:24-63; the
// reference expands the weights N-fold to [N*O,I,k,k] and calls cuDNN with groups=N, :59-62).  Here
//
//     out[n,o,y,x] = dcoef[n,o] * sum_{i,ky,kx} wn[o,i,ky,kx] * ( x[n,i,y+ky-pad,x+kx-pad] * sIn[n,i] )
//
// so all samples share one weight matrix:  M = O (out channels), N = pixels, K = I*k*k.
//
//   * arithmetic: v_mfma_f32_32x32x2_f32 -- exact fp32 products and fp32 accumulation (bitwise an fmaf chain), which
//     is what the 1e-4 end-to-end parity target needs; 4 k-steps are fetched per ds_read_b128 for each operand;
//   * A (weights) is pre-packed by the prep kernel as [O][I/KC][taps][KC] so a workgroup's A tile is a straight
//     16-byte-per-lane copy into LDS rows of taps*KC+4 floats (the +4 makes the 32-row ds_read_b128 conflict free);
//   * B is never materialised: the input patch (tile + halo) of KC channels is staged once in LDS, scaled by sIn on
//     the way in, channel-interleaved by 4 so that the im2col view for tap (ky,kx) is one conflict-free
//     ds_read_b128 per lane at a shifted offset;
//   * the next K-chunk is fetched global->registers while the current one is in the MFMA loop (issue-early /
//     write-late), one barrier pair per chunk;
//   * tile shapes are picked per layer so odd StyleGAN3 channel counts (323, 203, 81, 51, 32, 3) keep the M
//     dimension busy: BM in {128, 96, 64, 32} with more pixel rows per workgroup as BM shrinks;
//   * block ids are renumbered so the M-tiles that share an input patch, and neighbouring patches, sit on one XCD.
//
// demodulation (dcoef) is applied in the epilogue; bias/activation belong to the following filtered_lrelu.
#include "sg3_common.h"

namespace sg3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KS> struct ConvK;
template <> struct ConvK<3> { static constexpr int TAPS = 9, KC = 8; };
template <> struct ConvK<1> { static constexpr int TAPS = 1, KC = 16; };

static inline int packed_kc(int k) { return k == 3 ? ConvK<3>::KC : ConvK<1>::KC; }

struct ConvParams {
    const void* x; const float* wp; const float* sIn; const float* dcoef; void* out;
    int N, I, O, H, W, outH, outW, pad;
    int nch;                         // K chunks = ceil(I / KC)
    int xTiles, yTiles, mTiles;
    int totalBlocks;
};

template <typename T, int KS, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(256)
modconv_mfma_kernel(ConvParams p) {
    constexpr int TAPS = ConvK<KS>::TAPS, KC = ConvK<KS>::KC;
    constexpr int BM = WM * TM * 32;
    constexpr int ROWS = WN * TN;
    constexpr int PH = ROWS + KS - 1, PW = 32 + KS - 1;
    constexpr int AS = TAPS * KC + 4;                  // LDS row stride of A (floats)
    constexpr int PLANE = PH * PW * 4;                 // floats per 4-channel plane of the patch
    constexpr int NPL = KC / 4;
    constexpr int A_V4 = BM * TAPS * KC / 4;           // float4 loads for the A tile
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

    // block -> (m tile, x tile, y tile, sample); consecutive logical ids share an XCD
    int bid = blockIdx.x;
    {
        const int nb = p.totalBlocks, q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    const int mt = bid % p.mTiles; bid /= p.mTiles;
    const int xt = bid % p.xTiles; bid /= p.xTiles;
    const int yt = bid % p.yTiles; const int n = bid / p.yTiles;
    const int o0 = mt * BM, x0 = xt * 32, y0 = yt * ROWS;

    const T* xin = (const T*)p.x + (size_t)n * p.I * p.H * p.W;
    const float* sIn = p.sIn + (size_t)n * p.I;

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
        // A: rows o0 .. o0+BM-1 of the packed weights, chunk ch (rows beyond O read as zero)
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
        // B: input patch of KC channels, scaled by the per-sample style (zero outside the image / beyond I)
#pragma unroll
        for (int q = 0; q < B_PER; q++) {
            const int e = tid + 256 * q;
            float val = 0.f;
            if (B_EL % 256 == 0 || e < B_EL) {
                const int px = e % PW, t = e / PW, py = t % PH, c = t / PH;
                const int ci = ch * KC + c, gy = y0 - p.pad + py, gx = x0 - p.pad + px;
                if (ci < p.I && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W)
                    val = io<T>::ld(xin + ((size_t)ci * p.H + gy) * p.W + gx) * sIn[ci];
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
        __syncthreads();                 // previous chunk's fragment reads are done
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
                    fb[b] = *reinterpret_cast<const f32x4*>(sB + (2 * c8 + lh) * PLANE + ((wn * TN + b + ky) * PW + li + kx) * 4);
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

    // epilogue: demodulate and store (C layout: column = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))
    T* outp = (T*)p.out + (size_t)n * p.O * p.outH * p.outW;
    const int gx = x0 + li;
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int o = o0 + (wm * TM + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (o >= p.O) continue;
            const float d = p.dcoef ? p.dcoef[(size_t)n * p.O + o] : 1.f;
#pragma unroll
            for (int b = 0; b < TN; b++) {
                const int gy = y0 + wn * TN + b;
                if (gy < p.outH && gx < p.outW)
                    io<T>::st(outp + ((size_t)o * p.outH + gy) * p.outW + gx, acc[a][b][r] * d);
            }
        }
}

// ---------------------------------------------------------------------------
// prep A: one workgroup per output channel: normalise the filter, pack it, and emit wsq[o][i] = sum_taps wn^2
__global__ void __launch_bounds__(256)
modconv_prep_w_kernel(sg3_modconv_prep_params p, int kc, int nch) {
    __shared__ float red[256];
    const int o = blockIdx.x, taps = p.k * p.k, len = p.I * taps;
    const float* w = p.w + (size_t)o * len;
    float scale = 1.f;
    if (p.demodulate) {
        float s = 0.f;
        for (int j = threadIdx.x; j < len; j += 256) { float v = w[j]; s += v * v; }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
        scale = rsqrtf(red[0] / (float)len);
    }
    float* dst = p.wPacked + (size_t)o * nch * taps * kc;
    for (int j = threadIdx.x; j < nch * taps * kc; j += 256) {
        const int c = j % kc, t = (j / kc) % taps, ch = j / (kc * taps);
        const int i = ch * kc + c;
        dst[j] = i < p.I ? w[i * taps + t] * scale : 0.f;
    }
    for (int i = threadIdx.x; i < p.I; i += 256) {
        float s = 0.f;
        for (int t = 0; t < taps; t++) { float v = w[i * taps + t] * scale; s += v * v; }
        p.wsq[(size_t)o * p.I + i] = s;
    }
}

// prep B: one workgroup per sample: normalise styles over the WHOLE batch, write sIn and dcoef
__global__ void __launch_bounds__(256)
modconv_prep_s_kernel(sg3_modconv_prep_params p) {
    __shared__ float red[256];
    extern __shared__ float s2[];                         // [I] squared normalised styles of this sample
    const int n = blockIdx.x;
    float scale = 1.f;
    if (p.demodulate) {
        float s = 0.f;
        for (int j = threadIdx.x; j < p.N * p.I; j += 256) { float v = p.s[j]; s += v * v; }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
        scale = rsqrtf(red[0] / (float)(p.N * p.I));
    }
    for (int i = threadIdx.x; i < p.I; i += 256) {
        const float sn = p.s[(size_t)n * p.I + i] * scale;
        float g = 1.f;
        if (p.inputGainMode == 1) g = p.inputGain[0];
        else if (p.inputGainMode == 2) g = p.inputGain[i];
        else if (p.inputGainMode == 3) g = p.inputGain[(size_t)n * p.I + i];
        p.sIn[(size_t)n * p.I + i] = sn * g;
        s2[i] = sn * sn;
    }
    __syncthreads();
    if (p.demodulate) {
        for (int o = threadIdx.x; o < p.O; o += 256) {
            const float* wq = p.wsq + (size_t)o * p.I;
            float s = 0.f;
            for (int i = 0; i < p.I; i++) s += wq[i] * s2[i];
            p.dcoef[(size_t)n * p.O + o] = rsqrtf(s + 1e-8f);
        }
    }
}

template <typename T, int KS, int WM, int WN, int TM, int TN>
static int launch_conv(const sg3_modconv_params& q, hipStream_t st) {
    constexpr int BM = WM * TM * 32, ROWS = WN * TN;
    ConvParams p;
    p.x = q.x; p.wp = q.wPacked; p.sIn = q.sIn; p.dcoef = q.dcoef; p.out = q.out;
    p.N = q.N; p.I = q.I; p.O = q.O; p.H = q.H; p.W = q.W; p.pad = q.pad;
    p.outH = q.H + 2 * q.pad - KS + 1; p.outW = q.W + 2 * q.pad - KS + 1;
    p.nch = ceil_div(q.I, ConvK<KS>::KC);
    p.xTiles = ceil_div(p.outW, 32); p.yTiles = ceil_div(p.outH, ROWS); p.mTiles = ceil_div(q.O, BM);
    const long long total = (long long)p.xTiles * p.yTiles * p.mTiles * q.N;
    if (total > 0x7fffffffLL) { set_error("modulated_conv2d: grid too large"); return SG3_BAD_ARG; }
    p.totalBlocks = (int)total;
    hipLaunchKernelGGL((modconv_mfma_kernel<T, KS, WM, WN, TM, TN>), dim3((unsigned)total), dim3(256), 0, st, p);
    SG3_LAUNCH_CHECK("modconv_mfma_kernel");
    return SG3_OK;
}

template <typename T, int KS>
static int dispatch_conv(const sg3_modconv_params& q, hipStream_t st) {
    // M tile from the channel count: the smallest BM in {32,64,96,128} that wastes the least of the last tile
    const int O = q.O;
    int best = 128; double bestEff = 0.0;
    const int cands[4] = {128, 96, 64, 32};
    for (int c = 0; c < 4; c++) {
        const int bm = cands[c];
        const double eff = (double)O / (double)(ceil_div(O, bm) * bm);
        if (eff > bestEff + 1e-9) { bestEff = eff; best = bm; }
    }
    switch (best) {
        case 128: return launch_conv<T, KS, 2, 2, 2, 2>(q, st);
        case 96:  return launch_conv<T, KS, 1, 4, 3, 1>(q, st);
        case 64:  return launch_conv<T, KS, 1, 4, 2, 2>(q, st);
        default:  return launch_conv<T, KS, 1, 4, 1, 4>(q, st);
    }
}

} // namespace sg3

extern "C" {

int64_t sg3_modconv_packed_floats(int O, int I, int k) {
    if (O <= 0 || I <= 0 || (k != 1 && k != 3)) return 0;
    const int kc = sg3::packed_kc(k);
    return (int64_t)O * sg3::ceil_div(I, kc) * (k * k) * kc;
}

int sg3_modulated_conv2d_prep(const sg3_modconv_prep_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->w && p->s && p->wPacked && p->wsq && p->sIn, "modulated_conv2d_prep: null tensor");
    SG3_REQUIRE(p->N > 0 && p->I > 0 && p->O > 0, "modulated_conv2d_prep: empty tensor");
    SG3_REQUIRE(p->k == 1 || p->k == 3, "modulated_conv2d_prep: kernel size must be 1 or 3");
    SG3_REQUIRE(!p->demodulate || p->dcoef, "modulated_conv2d_prep: dcoef missing");
    SG3_REQUIRE(p->inputGainMode >= 0 && p->inputGainMode <= 3, "modulated_conv2d_prep: bad inputGainMode");
    SG3_REQUIRE(p->inputGainMode == 0 || p->inputGain, "modulated_conv2d_prep: inputGain missing");
    SG3_REQUIRE((size_t)p->I * sizeof(float) <= 48 * 1024, "modulated_conv2d_prep: too many input channels");
    hipStream_t st = (hipStream_t)stream;
    const int kc = packed_kc(p->k), nch = ceil_div(p->I, kc);
    hipLaunchKernelGGL(modconv_prep_w_kernel, dim3(p->O), dim3(256), 0, st, *p, kc, nch);
    SG3_LAUNCH_CHECK("modconv_prep_w_kernel");
    hipLaunchKernelGGL(modconv_prep_s_kernel, dim3(p->N), dim3(256), (size_t)p->I * sizeof(float), st, *p);
    SG3_LAUNCH_CHECK("modconv_prep_s_kernel");
    return SG3_OK;
}

int sg3_modulated_conv2d(const sg3_modconv_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->x && p->wPacked && p->sIn && p->out, "modulated_conv2d: null tensor");
    SG3_REQUIRE(p->N > 0 && p->I > 0 && p->O > 0 && p->H > 0 && p->W > 0, "modulated_conv2d: empty tensor");
    SG3_REQUIRE(p->k == 1 || p->k == 3, "modulated_conv2d: kernel size must be 1 or 3");
    SG3_REQUIRE(p->pad >= 0 && p->pad <= p->k - 1, "modulated_conv2d: padding must be in [0, k-1]");
    SG3_REQUIRE(p->H + 2 * p->pad - p->k + 1 > 0 && p->W + 2 * p->pad - p->k + 1 > 0, "modulated_conv2d: empty output");
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == SG3_F32) return p->k == 3 ? dispatch_conv<float, 3>(*p, st) : dispatch_conv<float, 1>(*p, st);
    if (p->dtype == SG3_F16) return p->k == 3 ? dispatch_conv<_Float16, 3>(*p, st) : dispatch_conv<_Float16, 1>(*p, st);
    set_error("modulated_conv2d: unsupported dtype %d", p->dtype);
    return SG3_BAD_ARG;
}

} // extern "C"
