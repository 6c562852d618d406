// sg3_wgrad.hip -- per-sample weight gradient of the (modulated) convolution on the gfx950 matrix cores.
//
// PTI (reference inversion/scripts/run_pti_images.py:111-139) back-propagates through modulated_conv2d
// (models/stylegan3/networks_stylegan3.py:59-62, a grouped F.conv2d whose weight gradient cuDNN computes).  With the
// per-sample effective weights W_n[o,i,ky,kx] the forward is out[n] = conv(x[n], W_n), so
//
//     dW[n,o,i,ky,kx] = sum_{y,x} dy[n,o,y,x] * xp[n,i,y+ky,x+kx]          (xp = x zero-padded by `pad`)
//
// a GEMM per sample and tap with M = O, N = I and the PIXELS as the K dimension.  The chain rule from dW to the weight,
// style and gain parameters is a few small tensor ops on the Python side (torch_utils/ops/modulated_conv.py).
//
// Arithmetic: both operands split into fp16 halves (hi = top 11 significand bits, lo = fp16(x - hi)), three
// v_mfma_f32_32x32x16_f16 per product (Ah*Bh + Ah*Bl + Al*Bh), fp32 accumulation -- fp32-equivalent, as in the forward
// kernel.  Gradients are tiny, activations are not: each operand is multiplied by a caller-supplied power of two on the
// way in (device scalars: no host round trip) and the product of the inverses is applied in the epilogue.
//
// Work decomposition: a workgroup owns a 64 x 64 (o, i) tile of one sample, a band of dy rows and a group of 32-pixel
// column segments; its four waves each own one 32 x 32 (o, i) block and ALL taps (nine accumulators), so an A fragment is
// read once per K step for 27 MFMAs.  Tiles with fewer than four real 32 x 32 blocks (the thin 1024^2 layers: 32 -> 32 is ONE
// block, 51 -> 32 two) give the idle waves a share of the TAPS instead of a block of channel padding: two waves per block take
// 5 + 4 taps, four waves 3 + 2 + 2 + 2 (a wave-uniform range test around each tap's three MFMAs; staging is unchanged).  Per (row, segment): the dy segment and the new xp row segment are split and staged
// in LDS (the three xp rows a dy row touches live in a ring, so every xp row is staged once per band); the column shift
// kx of a tap is taken in registers from a 16-half window (two aligned ds_read_b128): kx = 2 is a register rename, kx = 1
// four v_alignbyte.  The K dimension is split over workgroups; partial sums go to [split][n][tap][o][i] (coalesced
// stores) and are reduced by the caller -- deterministic, no float atomics.
#include "sg3_common.h"
#include "sg3_split.h"

namespace sg3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

struct WgradParams {
    const void* x; const void* dy; float* partial; const float* scaleX; const float* scaleDy;
    int N, I, O, H, W, OH, OW, pad;
    int oTiles, iTiles, bandRows, nBands, segsPerGroup, nSegGroups, nSegs;
    int amax;                        // scaleX / scaleDy point at max |x| / max |dy| instead: the power-of-two scales are derived here
};

// 2^-e with e = ceil(log2(amax / 2^15)): brings the peak magnitude just below 2^15 (what the caller's torch ops computed:
// exp2(-ceil(log2(clamp_min(amax, 1e-30) / 32768))))
__device__ __forceinline__ float pow2_scale(float amax) {
    return exp2f(-ceilf(log2f(fmaxf(amax, 1e-30f) / 32768.0f)));
}

template <typename T, int KS>
__global__ void __launch_bounds__(256, 2)
wgrad_f16x3_kernel(WgradParams p) {
    constexpr int TAPS = KS * KS;
    constexpr int KT = 32;                              // dy pixels per staged segment (two K steps of 16)
    constexpr int XW = KT + 16;                         // xp pixels per staged row segment (shift window + alignment)
    constexpr int DP = KT + 8, XP = XW + 8;             // LDS row pitches in halfs (conflict-free b128 rows)
    constexpr int DPLANE = 64 * DP, XPLANE = 64 * XP;   // halfs per (part) plane
    constexpr int RING = KS;                            // xp rows alive per dy row

    __shared__ __attribute__((aligned(16))) _Float16 sD[2 * DPLANE];
    __shared__ __attribute__((aligned(16))) _Float16 sX[RING * 2 * XPLANE];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    int t = blockIdx.x;
    const int it = t % p.iTiles; t /= p.iTiles;
    const int ot = t % p.oTiles; const int n = t / p.oTiles;
    const int band = blockIdx.y, sg = blockIdx.z;
    const int o0 = ot * 64, i0 = it * 64;
    // real 32 x 32 blocks of this tile and the waves' shares: (block, tap range) -- all wave-uniform
    const int nOb = p.O - o0 > 32 ? 2 : 1, nIb = p.I - i0 > 32 ? 2 : 1;
    const int wpb = (KS == 3) ? 4 / (nOb * nIb) : 1;                       // waves per block: 1 | 2 | 4
    const int blk = wave / wpb, sub = wave % wpb;
    const int ob = blk % nOb, ib = blk / nOb;
    const bool blockReal = blk < nOb * nIb;                                // 1x1 kernels keep one block per wave: padding blocks idle
    int tLo = 0, tHi = TAPS;
    if (wpb == 2) { tLo = sub ? 5 : 0; tHi = sub ? 9 : 5; }
    else if (wpb == 4) { tLo = sub ? 1 + 2 * sub : 0; tHi = 3 + 2 * sub; }
    if (!blockReal) tHi = tLo = 0;
    const int y0 = band * p.bandRows, y1 = min(y0 + p.bandRows, p.OH);
    const int seg0 = sg * p.segsPerGroup, seg1 = min(seg0 + p.segsPerGroup, p.nSegs);

    const float scD = p.amax ? pow2_scale(p.scaleDy[0]) : p.scaleDy[0], scX = p.amax ? pow2_scale(p.scaleX[0]) : p.scaleX[0];
    const unsigned dyPlane = (unsigned)(p.OH * p.OW) * (unsigned)sizeof(T), xPlane = (unsigned)(p.H * p.W) * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.dy + (size_t)n * p.O * p.OH * p.OW), (short)0, (int)((unsigned)p.O * dyPlane), 0x00020000);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.x + (size_t)n * p.I * p.H * p.W), (short)0, (int)((unsigned)p.I * xPlane), 0x00020000);

    // staging maps: thread -> (row of the tile, group of consecutive pixels)
    const int srow = tid >> 2, sgrp = tid & 3;
    const bool dOk = o0 + srow < p.O, xOk = i0 + srow < p.I;
    const unsigned dRowOff = dOk ? (unsigned)(o0 + srow) * dyPlane : 0u;
    const unsigned xRowOff = xOk ? (unsigned)(i0 + srow) * xPlane : 0u;

    f32x16 acc[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; tp++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tp][r] = 0.f;

    // one row segment of dy -> sD; 8 pixels per thread
    auto stage_dy = [&](int y, int xs) {
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; e++) {
            const int px = xs + sgrp * 8 + e;
            const unsigned off = (dOk && px < p.OW) ? dRowOff + (unsigned)(y * p.OW + px) * (unsigned)sizeof(T) : 0x80000000u;
            v[e] = bufld<T>::ld(dr, off, 0) * scD;
        }
        v2h h[4], l[4];
#pragma unroll
        for (int c = 0; c < 4; c++) split2(v[2 * c], v[2 * c + 1], h[c], l[c]);
        _Float16* dst = sD + srow * DP + sgrp * 8;
        *reinterpret_cast<v8h*>(dst) = __builtin_shufflevector(__builtin_shufflevector(h[0], h[1], 0, 1, 2, 3), __builtin_shufflevector(h[2], h[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
        *reinterpret_cast<v8h*>(dst + DPLANE) = __builtin_shufflevector(__builtin_shufflevector(l[0], l[1], 0, 1, 2, 3), __builtin_shufflevector(l[2], l[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
    };
    // one row segment of the padded input (padded row yp, padded columns xs .. xs + XW - 1) -> ring slot; 12 pixels per thread
    auto stage_x = [&](int yp, int xs, int slot) {
        const int gy = yp - p.pad;
        const bool rowOk = xOk && (unsigned)gy < (unsigned)p.H;
        float v[12];
#pragma unroll
        for (int e = 0; e < 12; e++) {
            const int gx = xs + sgrp * 12 + e - p.pad;
            const unsigned off = (rowOk && (unsigned)gx < (unsigned)p.W) ? xRowOff + (unsigned)(gy * p.W + gx) * (unsigned)sizeof(T) : 0x80000000u;
            v[e] = bufld<T>::ld(xr, off, 0) * scX;
        }
        _Float16* dst = sX + slot * 2 * XPLANE + srow * XP + sgrp * 12;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            v2h h0, l0, h1, l1;
            split2(v[4 * c], v[4 * c + 1], h0, l0);
            split2(v[4 * c + 2], v[4 * c + 3], h1, l1);
            typedef _Float16 v4h __attribute__((ext_vector_type(4)));
            *reinterpret_cast<v4h*>(dst + 4 * c) = __builtin_shufflevector(h0, h1, 0, 1, 2, 3);
            *reinterpret_cast<v4h*>(dst + XPLANE + 4 * c) = __builtin_shufflevector(l0, l1, 0, 1, 2, 3);
        }
    };
    // (ky, all kx) of one K step: the 16-half window of this lane's input row, then the three column shifts
    auto taps_of_row = [&](int slot, int ky, int k0, v8h ah, v8h al) {
        if (ky * KS + KS <= tLo || ky * KS >= tHi) return;                 // none of this filter row's taps is this wave's
        const _Float16* src = sX + slot * 2 * XPLANE + (ib * 32 + li) * XP + k0 + 8 * lh;
        const u32x4 h0 = *reinterpret_cast<const u32x4*>(src), h1 = *reinterpret_cast<const u32x4*>(src + 8);
        const u32x4 l0 = *reinterpret_cast<const u32x4*>(src + XPLANE), l1 = *reinterpret_cast<const u32x4*>(src + XPLANE + 8);
        const unsigned wh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        const unsigned wl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
#pragma unroll
        for (int kx = 0; kx < KS; kx++) {
            u32x4 bh, bl;
            if (kx == 1) {          // one half to the left: funnel shift of neighbouring dwords
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    bh[j] = __builtin_amdgcn_alignbyte(wh[j + 1], wh[j], 2);
                    bl[j] = __builtin_amdgcn_alignbyte(wl[j + 1], wl[j], 2);
                }
            } else {                // 0 or 2 halfs: whole dwords
#pragma unroll
                for (int j = 0; j < 4; j++) { bh[j] = wh[j + kx / 2]; bl[j] = wl[j + kx / 2]; }
            }
            const v8h fbh = __builtin_bit_cast(v8h, bh), fbl = __builtin_bit_cast(v8h, bl);
            const int tp = ky * KS + kx;
            if (tp < tLo || tp >= tHi) continue;                           // wave-uniform
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, fbh, acc[tp], 0, 0, 0);
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fbl, acc[tp], 0, 0, 0);
            acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fbh, acc[tp], 0, 0, 0);
        }
    };

    for (int seg = seg0; seg < seg1; seg++) {
        const int xs = seg * KT;                        // first dy column of the segment = first padded-input column
        // ring warm-up: padded rows y0 .. y0 + KS - 2
        __syncthreads();
#pragma unroll
        for (int r = 0; r < KS - 1; r++) stage_x(y0 + r, xs, (y0 + r) % RING);
        for (int y = y0; y < y1; y++) {
            __syncthreads();                            // the previous row's fragment reads are done
            stage_dy(y, xs);
            stage_x(y + KS - 1, xs, (y + KS - 1) % RING);
            __syncthreads();
#pragma unroll
            for (int k0 = 0; k0 < KT; k0 += 16) {
                const _Float16* asrc = sD + (ob * 32 + li) * DP + k0 + 8 * lh;
                const v8h ah = *reinterpret_cast<const v8h*>(asrc), al = *reinterpret_cast<const v8h*>(asrc + DPLANE);
#pragma unroll
                for (int ky = 0; ky < KS; ky++) taps_of_row((y + ky) % RING, ky, k0, ah, al);
            }
        }
    }

    // partial[split][n][tap][o][i]; C layout: column (i) = lane & 31, row (o) = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    const float inv = 1.f / (scD * scX);
    const int split = band * p.nSegGroups + sg;
    float* outp = p.partial + ((size_t)split * p.N + n) * TAPS * p.O * p.I;
    const int gi = i0 + ib * 32 + li;
#pragma unroll
    for (int tp = 0; tp < TAPS; tp++) {
        if (tp < tLo || tp >= tHi) continue;                               // another wave's tap (or a padding block)
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int go = o0 + ob * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (go < p.O && gi < p.I) outp[((size_t)tp * p.O + go) * p.I + gi] = acc[tp][r] * inv;
        }
    }
}

template <typename T, int KS>
static int launch_wgrad(const sg3_wgrad_params& q, hipStream_t st) {
    WgradParams p;
    p.x = q.x; p.dy = q.dy; p.partial = q.partial; p.scaleX = q.scaleX; p.scaleDy = q.scaleDy; p.amax = q.scalesAreAmax;
    p.N = q.N; p.I = q.I; p.O = q.O; p.H = q.H; p.W = q.W; p.pad = q.pad;
    p.OH = q.H + 2 * q.pad - KS + 1; p.OW = q.W + 2 * q.pad - KS + 1;
    p.oTiles = ceil_div(q.O, 64); p.iTiles = ceil_div(q.I, 64);
    p.nSegs = ceil_div(p.OW, 32);
    p.nBands = q.nBands; p.nSegGroups = q.nSegGroups;
    p.bandRows = ceil_div(p.OH, q.nBands); p.segsPerGroup = ceil_div(p.nSegs, q.nSegGroups);
    dim3 g((unsigned)(p.oTiles * p.iTiles * q.N), (unsigned)q.nBands, (unsigned)q.nSegGroups), b(256);
    hipLaunchKernelGGL((wgrad_f16x3_kernel<T, KS>), g, b, 0, st, p);
    SG3_LAUNCH_CHECK("wgrad_f16x3_kernel");
    return SG3_OK;
}

} // namespace sg3

extern "C" {

int sg3_conv2d_wgrad_splits(int N, int I, int O, int H, int W, int k, int pad, int* nBands, int* nSegGroups) {
    using namespace sg3;
    SG3_REQUIRE(N > 0 && I > 0 && O > 0 && H > 0 && W > 0 && (k == 1 || k == 3) && nBands && nSegGroups, "conv2d_wgrad_splits: bad arguments");
    const int OH = H + 2 * pad - k + 1, OW = W + 2 * pad - k + 1;
    SG3_REQUIRE(OH > 0 && OW > 0, "conv2d_wgrad_splits: empty output");
    // enough workgroups to fill 256 CUs a few times over, with bands tall enough that the ring warm-up stays small
    const long long tiles = (long long)ceil_div(O, 64) * ceil_div(I, 64) * N;
    long long want = ceil_div64(2048, tiles);
    int bands = (int)(want < 1 ? 1 : want);
    const int maxBands = max(1, OH / 8);
    int groups = 1;
    if (bands > maxBands) {
        groups = (int)ceil_div64(bands, maxBands);
        bands = maxBands;
        const int nSegs = ceil_div(OW, 32);
        if (groups > nSegs) groups = nSegs;
    }
    // no empty bands / groups
    bands = ceil_div(OH, ceil_div(OH, bands));
    const int nSegs = ceil_div(OW, 32);
    groups = ceil_div(nSegs, ceil_div(nSegs, groups));
    *nBands = bands; *nSegGroups = groups;
    return SG3_OK;
}

int sg3_conv2d_wgrad(const sg3_wgrad_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->x && p->dy && p->partial && p->scaleX && p->scaleDy, "conv2d_wgrad: null tensor");
    SG3_REQUIRE(p->N > 0 && p->I > 0 && p->O > 0 && p->H > 0 && p->W > 0, "conv2d_wgrad: empty tensor");
    SG3_REQUIRE(p->k == 1 || p->k == 3, "conv2d_wgrad: kernel size must be 1 or 3");
    SG3_REQUIRE(p->pad >= 0 && p->pad <= p->k - 1, "conv2d_wgrad: padding must be in [0, k-1]");
    SG3_REQUIRE(p->dtype == SG3_F32 || p->dtype == SG3_F16, "conv2d_wgrad: unsupported dtype");
    SG3_REQUIRE(p->nBands > 0 && p->nSegGroups > 0, "conv2d_wgrad: bad split counts");
    SG3_REQUIRE((int64_t)p->I * p->H * p->W * 4 < (int64_t)1 << 31 && (int64_t)p->O * (p->H + 2) * (p->W + 2) * 4 < (int64_t)1 << 31,
                "conv2d_wgrad: a sample must stay below 2 GiB (32-bit offsets)");
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == SG3_F32) return p->k == 3 ? launch_wgrad<float, 3>(*p, st) : launch_wgrad<float, 1>(*p, st);
    return p->k == 3 ? launch_wgrad<_Float16, 3>(*p, st) : launch_wgrad<_Float16, 1>(*p, st);
}

} // extern "C"
