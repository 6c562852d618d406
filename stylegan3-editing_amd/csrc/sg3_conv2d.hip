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
#include "sg3_split.h"

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

// ---------------------------------------------------------------------------
// Split-precision form for the 3x3 convolutions (stride 1: the backbone's FLOPs; stride 2: the first unit of every stage, the
// ResNet34 blocks, with the patch columns de-interleaved in LDS so that the stride-2 im2col read stays conflict-free) (same arithmetic as
// modconv_f16x3_kernel in sg3_modconv.hip: x = hi + lo in fp16, Ah*Bh + Ah*Bl + Al*Bh on v_mfma_f32_32x32x16_f16, fp32
// accumulation, fp32-equivalent).  No bound on BatchNorm-ed activations is known ahead of time and none is needed for
// accuracy (fp16's exponent range covers 6e-5 .. 65504 at full split precision, smaller magnitudes lose only bits that
// are below 1e-11 absolute); what must not happen silently is overflow, so every workgroup tracks max |operand| while
// staging and raises `*flag` when it exceeds the fp16 range -- the caller then repeats the layer stack on the exact
// fp32 kernel (torch_utils/ops/plain_conv.py).  Weights are packed [O][I/16][tap][hi|lo][16] halfs by
// conv2d_pack_f16x3_kernel with the folded BatchNorm scale applied before the split.
template <int KS, int STRIDE, int WM, int WN, int TM, int TN>
__global__ void __launch_bounds__(256, (TM * TN <= 4) ? 2 : 1)
conv2d_f16x3_kernel(PlainConvParams p, int* flag) {
    constexpr int TAPS = KS * KS, KC = 16;
    constexpr int BM = WM * TM * 32;
    constexpr int ROWS = WN * TN;
    constexpr int PH = (ROWS - 1) * STRIDE + KS, PW = 31 * STRIDE + KS;
    constexpr int PWE = (PW + 1) / 2;                  // stride 2: even patch columns first (PWE of them), then the odd ones
    constexpr int NPIX = PH * PW;
    constexpr int AS = TAPS * 32 + 8;
    constexpr int AROW_V = TAPS * 32 / 8;
    constexpr int TPR = (BM == 32) ? 8 : (BM == 64 ? 4 : 2);
    constexpr int A_PER = (AROW_V + TPR - 1) / TPR;
    constexpr int BPLANE = NPIX * 8;
    constexpr int PX_PER = (NPIX + 255) / 256;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BM * TPR <= 256, "A staging map");

    extern __shared__ __attribute__((aligned(16))) _Float16 smh[];
    _Float16* sA = smh;
    _Float16* sB = smh + BM * AS;

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

    const unsigned HWb = (unsigned)(p.H * p.W) * 4u;
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(p.x + (size_t)n * p.I * p.H * p.W), (short)0, (int)((unsigned)p.I * HWb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wp, (short)0, (int)((unsigned)p.O * (unsigned)p.nch * (unsigned)(AROW_V * 16)), 0x00020000);
    const int arow = tid / TPR, acol = tid % TPR;
    const bool aOk = arow < BM;
    const unsigned aG = (aOk && o0 + arow < p.O) ? ((unsigned)(o0 + arow) * (unsigned)p.nch * (AROW_V * 16) + acol * 16) : 0x80000000u;
    _Float16* aL = sA + arow * AS + acol * 8;
    unsigned bG[PX_PER];
    int bL[PX_PER];
    float bM[PX_PER];                       // 1 inside the image, 0 in the zero padding: the input affine applies to real pixels only
#pragma unroll
    for (int q = 0; q < PX_PER; q++) {
        const int e = tid + 256 * q;
        const int px = e % PW, py = e / PW;
        const int gy = y0 * STRIDE - p.pad + py, gx = x0 * STRIDE - p.pad + px;
        const bool ok = e < NPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        bG[q] = ok ? (unsigned)(gy * p.W + gx) * 4u : 0x80000000u;
        const int pcol = STRIDE == 1 ? px : (px & 1) * PWE + (px >> 1);
        bL[q] = e < NPIX ? (py * PW + pcol) * 8 : -1;
        bM[q] = ok ? 1.f : 0.f;
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    u32x4 ra[A_PER];
    float rb[PX_PER][2][8];
    float peak = 0.f;
    // per-channel affine in front of the convolution (the folded BatchNorm): lane c of every wave requests scale / shift of
    // channel c of the chunk together with the pixels (range-checked: channels beyond I read 0 and silence the padded
    // channels); stage() broadcasts them with v_readlane.  Nothing in fetch() touches the loaded values, so their
    // s_waitcnt sits behind the MFMA loop of the previous chunk, not in front of it.
    const bool hasScale = p.inScale != nullptr, hasShift = hasScale && p.inShift != nullptr;
    const __amdgpu_buffer_rsrc_t scr = __builtin_amdgcn_make_buffer_rsrc((void*)p.inScale, (short)0, hasScale ? p.I * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t shr = __builtin_amdgcn_make_buffer_rsrc((void*)p.inShift, (short)0, hasShift ? p.I * 4 : 0, 0x00020000);
    float rsc = 0.f, rsh = 0.f;

    auto fetch = [&](int ch) {
        const int cl = ch * KC + (lane & 15);
        if (hasScale) rsc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(scr, cl * 4, 0, 0));
        else rsc = cl < p.I ? 1.f : 0.f;
        if (hasShift) rsh = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(shr, cl * 4, 0, 0));
        const unsigned aoff = aG + (unsigned)ch * (AROW_V * 16);
#pragma unroll
        for (int q = 0; q < A_PER; q++) {
            if (AROW_V % TPR == 0 || acol + q * TPR < AROW_V)
                ra[q] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(aoff + q * TPR * 16), 0, 0);
        }
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int ci = ch * KC + hf * 8 + c;                        // wave-uniform
                const unsigned coff = ci < p.I ? (unsigned)ci * HWb : 0u;
#pragma unroll
                for (int q = 0; q < PX_PER; q++) rb[q][hf][c] = bufld<float>::ld(xr, bG[q], coff);
            }
    };
    auto stage = [&]() {
        float sc[2][8], sh[2][8];
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                sc[hf][c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rsc), hf * 8 + c));
                sh[hf][c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rsh), hf * 8 + c));
            }
        if (aOk) {
#pragma unroll
            for (int q = 0; q < A_PER; q++)
                if (AROW_V % TPR == 0 || acol + q * TPR < AROW_V)
                    *reinterpret_cast<u32x4*>(aL + q * TPR * 8) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < PX_PER; q++) {
            if (bL[q] < 0) continue;
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                v2h h[4], l[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    // zero padding stays zero: the shift only applies inside the image (bM)
                    const float v0 = __builtin_fmaf(rb[q][hf][2 * c], sc[hf][2 * c], sh[hf][2 * c] * bM[q]);
                    const float v1 = __builtin_fmaf(rb[q][hf][2 * c + 1], sc[hf][2 * c + 1], sh[hf][2 * c + 1] * bM[q]);
                    peak = __builtin_fmaxf(peak, __builtin_fmaxf(__builtin_fabsf(v0), __builtin_fabsf(v1)));
                    split2(v0, v1, h[c], l[c]);
                }
                _Float16* dst = sB + (hf * 2) * BPLANE + bL[q];
                *reinterpret_cast<v8h*>(dst) = __builtin_shufflevector(__builtin_shufflevector(h[0], h[1], 0, 1, 2, 3), __builtin_shufflevector(h[2], h[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
                *reinterpret_cast<v8h*>(dst + BPLANE) = __builtin_shufflevector(__builtin_shufflevector(l[0], l[1], 0, 1, 2, 3), __builtin_shufflevector(l[2], l[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    struct Frags { v8h ah[TM], al[TM], bh[TN], bl[TN]; };
    auto load_frags = [&](Frags& f, int tap) {
        const int ky = tap / KS, kx = tap % KS;
#pragma unroll
        for (int a = 0; a < TM; a++) {
            const _Float16* src = sA + ((wm * TM + a) * 32 + li) * AS + tap * 32 + lh * 8;
            f.ah[a] = *reinterpret_cast<const v8h*>(src);
            f.al[a] = *reinterpret_cast<const v8h*>(src + 16);
        }
#pragma unroll
        for (int b = 0; b < TN; b++) {
            // stride 2: patch column 2 * li + kx sits at (kx & 1) * PWE + li + (kx >> 1)
            const int pcol = STRIDE == 1 ? li + kx : (kx & 1) * PWE + li + (kx >> 1);
            const _Float16* src = sB + (lh * 2) * BPLANE + (((wn * TN + b) * STRIDE + ky) * PW + pcol) * 8;
            f.bh[b] = *reinterpret_cast<const v8h*>(src);
            f.bl[b] = *reinterpret_cast<const v8h*>(src + BPLANE);
        }
    };
    auto mfma_tap = [&](const Frags& f) {
#pragma unroll
        for (int a = 0; a < TM; a++)
#pragma unroll
            for (int b = 0; b < TN; b++) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[a], f.bh[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[a], f.bl[b], acc[a][b], 0, 0, 0);
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[a], f.bh[b], acc[a][b], 0, 0, 0);
            }
    };

    fetch(0);
    for (int ch = 0; ch < p.nch; ch++) {
        __syncthreads();
        stage();
        __syncthreads();
        if (ch + 1 < p.nch) fetch(ch + 1);
        Frags f0, f1;
        load_frags(f0, 0);
#pragma unroll
        for (int tap = 0; tap < TAPS; tap += 2) {
            if (tap + 1 < TAPS) load_frags(f1, tap + 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_tap(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (tap + 1 < TAPS) {
                if (tap + 2 < TAPS) load_frags(f0, tap + 2);
                __builtin_amdgcn_sched_barrier(0);
                mfma_tap(f1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    // fp16 range check on the staged operands (NaN compares false and propagates to the output on its own)
    if (peak > 65000.f) atomicOr(flag, 1);

    float* outp = p.out + (size_t)n * p.O * p.outH * p.outW;
    const int gx = x0 + li;
    const unsigned planeB = (unsigned)(p.outH * p.outW) * 4u;
    // every offset a lane can form -- channels of the padded last M tile included -- must stay below 2^31, so that nothing
    // wraps around into the tensor: (padded O + one wave block) * plane < 2^31
    if ((unsigned long long)(p.mTiles * BM + 32) * planeB < 0x7fffffffULL) {
        // stores through a descriptor over this sample's output: the hardware range check drops channels beyond O and
        // (offsets from 2^31) columns beyond the row; no per-store predicate or 64-bit address arithmetic
        const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)outp, (short)0, (int)((unsigned)p.O * planeB), 0x00020000);
#pragma unroll
        for (int a = 0; a < TM; a++) {
            const int oL = o0 + (wm * TM + a) * 32 + 4 * lh;
            float bv[16], sl[16];
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const int o = min(oL + (r & 3) + 8 * (r >> 2), p.O - 1);                // channels beyond O are dropped by the store
                bv[r] = p.bias ? p.bias[o] : 0.f;
                sl[r] = p.act == 1 ? p.slope[o] : (p.act == 2 ? p.slope[0] : 1.f);
            }
            const unsigned laneBase = gx < p.outW ? (unsigned)oL * planeB + (unsigned)gx * 4u : 0x80000000u;
#pragma unroll
            for (int b = 0; b < TN; b++) {
                const int gy = y0 + wn * TN + b;
                if (gy >= p.outH) continue;                                              // wave-uniform
                const unsigned rowOff = laneBase + (unsigned)(gy * p.outW) * 4u;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    float v = acc[a][b][r] + bv[r];
                    if (p.act) v = v < 0.f ? v * sl[r] : v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orr, (int)(rowOff + (unsigned)((r & 3) + 8 * (r >> 2)) * planeB), 0, 0);
                }
            }
        }
        return;
    }
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
conv2d_pack_f16x3_kernel(const float* w, const float* outScale, float* wp, int O, int I, int nch, int taps) {
    const int o = blockIdx.x;
    const float sc = outScale ? outScale[o] : 1.f;
    _Float16* dst = reinterpret_cast<_Float16*>(wp) + (size_t)o * nch * taps * 32;
    for (int j = threadIdx.x; j < nch * taps * 16; j += 256) {
        const int c = j % 16, t = (j / 16) % taps, ch = j / (16 * taps);
        const int i = ch * 16 + c;
        const float v = i < I ? w[((size_t)o * I + i) * taps + t] * sc : 0.f;
        const _Float16 h = (_Float16)v;
        _Float16* d = dst + ((size_t)ch * taps + t) * 32 + c;
        d[0] = h;
        d[16] = (_Float16)(v - (float)h);
    }
}

// CUs of the current device, read once per device
static int device_cu_count() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

template <int KS, int STRIDE, int WM, int WN, int TM, int TN>
static int launch_plain_f16x3(const sg3_conv2d_params& q, hipStream_t st) {
    constexpr int BM = WM * TM * 32, ROWS = WN * TN;
    constexpr size_t ldsBytes = ((size_t)BM * (KS * KS * 32 + 8) + 4 * (size_t)((ROWS - 1) * STRIDE + KS) * (31 * STRIDE + KS) * 8) * sizeof(_Float16);
    PlainConvParams p;
    p.x = q.x; p.wp = q.wPacked; p.inScale = q.inScale; p.inShift = q.inShift; p.bias = q.bias; p.slope = q.slope; p.out = q.out;
    p.N = q.N; p.I = q.I; p.O = q.O; p.H = q.H; p.W = q.W; p.stride = STRIDE; p.pad = q.pad; p.act = q.act;
    p.outH = (q.H + 2 * q.pad - KS) / STRIDE + 1; p.outW = (q.W + 2 * q.pad - KS) / STRIDE + 1;
    p.nch = ceil_div(q.I, 16);
    p.xTiles = ceil_div(p.outW, 32); p.yTiles = ceil_div(p.outH, ROWS); p.mTiles = ceil_div(q.O, BM);
    const long long total = (long long)p.xTiles * p.yTiles * p.mTiles * q.N;
    if (total > 0x7fffffffLL) { set_error("conv2d: grid too large"); return SG3_BAD_ARG; }
    p.totalBlocks = (int)total;
    auto kern = conv2d_f16x3_kernel<KS, STRIDE, WM, WN, TM, TN>;
    if (ldsBytes > 64 * 1024)
        SG3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), ldsBytes, st, p, q.rangeFlag);
    SG3_LAUNCH_CHECK("conv2d_f16x3_kernel");
    return SG3_OK;
}

static int dispatch_plain_f16x3(const sg3_conv2d_params& q, hipStream_t st) {
    if (q.k == 1)                                                                    // the projection shortcuts: same kernel, one tap
        return q.stride == 2 ? launch_plain_f16x3<1, 2, 1, 4, 2, 1>(q, st) : launch_plain_f16x3<1, 1, 1, 4, 2, 1>(q, st);
    if (q.stride == 2) return launch_plain_f16x3<3, 2, 1, 4, 2, 1>(q, st);        // 64 x (4 rows x 32), 9 x 65 patch
    const int outH = q.H + 2 * q.pad - 2, outW = q.W + 2 * q.pad - 2;
    if (outH <= 16) return launch_plain_f16x3<3, 1, 1, 4, 2, 1>(q, st);           // 64 x (4 rows x 32): the 16x16 maps
    // The 8-row tile (230 registers: two workgroups per CU) needs a grid that offers every CU its two workgroups; the 32 x 32 maps
    // of a 16-frame batch (14 units of the IR-SE50 trunk, 28 launches) give 256 eight-row tiles = ONE four-wave workgroup per CU
    // (counters, round 4: 0.85 waves per SIMD, matrix pipes 0.32 busy at 2.45 GHz -- latency, not power).  Such grids take the
    // 4-row tile instead (153 registers, three workgroups per CU): twice the workgroups, each with half the rows.
    const long long tiles8 = (long long)q.N * ceil_div(q.O, 64) * ceil_div(outH, 8) * ceil_div(outW, 32);
    if (tiles8 < 2 * (long long)device_cu_count()) return launch_plain_f16x3<3, 1, 1, 4, 2, 1>(q, st);
    return launch_plain_f16x3<3, 1, 1, 4, 2, 2>(q, st);                           // 64 x (8 rows x 32)
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

int sg3_conv2d_pack(const float* w, const float* outScale, float* wPacked, int O, int I, int k, int precision, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(w && wPacked && O > 0 && I > 0 && (k == 1 || k == 3), "conv2d_pack: bad arguments");
    SG3_REQUIRE(precision == SG3_CONV_FP32 || precision == SG3_CONV_F16X3, "conv2d_pack: bad precision");
    if (precision == SG3_CONV_F16X3) {
        hipLaunchKernelGGL(conv2d_pack_f16x3_kernel, dim3(O), dim3(256), 0, (hipStream_t)stream, w, outScale, wPacked, O, I, ceil_div(I, 16), k * k);
        SG3_LAUNCH_CHECK("conv2d_pack_f16x3_kernel");
        return SG3_OK;
    }
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
    if (p->precision == SG3_CONV_F16X3) {
        SG3_REQUIRE(p->rangeFlag, "conv2d: the f16x3 form takes a range flag");
        SG3_REQUIRE((int64_t)p->I * p->H * p->W * 4 < (int64_t)1 << 31, "conv2d: f16x3 needs a sample below 2 GiB (32-bit offsets)");
        return dispatch_plain_f16x3(*p, st);
    }
    SG3_REQUIRE(p->precision == SG3_CONV_FP32, "conv2d: bad precision");
    if (p->k == 3) return p->stride == 1 ? dispatch_plain<3, 1>(*p, st) : dispatch_plain<3, 2>(*p, st);
    return p->stride == 1 ? dispatch_plain<1, 1>(*p, st) : dispatch_plain<1, 2>(*p, st);
}

} // extern "C"
