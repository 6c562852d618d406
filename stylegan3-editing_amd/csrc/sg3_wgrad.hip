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
// in LDS (the three xp rows a dy row touches live in a ring of four, so every xp row is staged once per band, and the samples of
// row y + 1 are requested before the MFMAs of row y and handed to LDS after them: one barrier per row); the column shift
// kx of a tap is taken in registers from a 16-half window (two aligned ds_read_b128): kx = 2 is a register rename, kx = 1
// four v_alignbyte.  The K dimension is split over workgroups; partial sums go to [split][n][tap][o][i] (coalesced
// stores) and are reduced by the caller -- deterministic, no float atomics.
#include "sg3_common.h"
#include "sg3_split.h"
#include <type_traits>
#include <cstdlib>

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
    static_assert(KT == 32, "the row loop issues exactly two K steps");
    constexpr int XW = KT + 16;                         // xp pixels per staged row segment (shift window + alignment)
    constexpr int DP = KT + 8, XP = XW + 8;             // LDS row pitches in halfs (conflict-free b128 rows)
    constexpr int DPLANE = 64 * DP, XPLANE = 64 * XP;   // halfs per (part) plane
    constexpr int RING = KS + 1;                        // xp rows alive per dy row + the one being staged for the next row

    extern __shared__ __attribute__((aligned(16))) _Float16 wgradLds[];
    _Float16* const sD = wgradLds;                      // [2 buffers][hi | lo][64 rows][DP]
    _Float16* const sX = wgradLds + 2 * 2 * DPLANE;     // [RING slots][hi | lo][64 rows][XP]

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

    // staging maps.  A request covers whole row pieces with consecutive lanes on consecutive pixels (2 rows x 32 pixels of dy, 4 rows x
    // 16 pixels of the input: 3 - 6 cache lines per instruction; a thread-owns-8-pixels map touches 16 - 32 lines per instruction and
    // made the L1 the limiter).  Rows of channel padding lie beyond num_records (= channels x plane) by themselves; an invalid column
    // starts from the out-of-range sentinel, to which the row term (< 2^31, see the size checks) is added.
    const int dRow0 = wave * 16 + (lane >> 5), dCol = lane & 31;           // dy: rows dRow0 + 2 j (j < 8), one column
    const int xRow0 = wave * 16 + (lane >> 4), xCol = lane & 15;           // x : rows xRow0 + 4 (j / 3), columns xCol + 16 (j % 3), j < 12
    const unsigned dRowOff = (unsigned)(o0 + dRow0) * dyPlane;
    const unsigned xRowOff = (unsigned)(i0 + xRow0) * xPlane;
    f32x16 acc[TAPS];
#pragma unroll
    for (int tp = 0; tp < TAPS; tp++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[tp][r] = 0.f;

    // Staging is split into its request and its hand-over so that the requests of row y + 1 fly during the MFMAs of row y
    auto load_dy = [&](int y, int xs, unsigned (&v)[8], bool rowOk) {
        const int px = xs + dCol;
        const unsigned base = (px < (rowOk ? p.OW : 0) ? (unsigned)(y * p.OW + px) * (unsigned)sizeof(T) : 0x80000000u) + dRowOff;
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = bufld<T>::ldraw(dr, base + (unsigned)(2 * j) * dyPlane, 0);
    };
    auto write_dy_pair = [&](const unsigned (&v)[8], int buf, int c) {     // rows dRow0 + 4 c and dRow0 + 4 c + 2
        _Float16* dst = sD + buf * 2 * DPLANE + dRow0 * DP + dCol;
        v2h h, l;
        split2(bufld<T>::fromraw(v[2 * c]) * scD, bufld<T>::fromraw(v[2 * c + 1]) * scD, h, l);
        dst[(4 * c) * DP] = h.x; dst[(4 * c + 2) * DP] = h.y;
        dst[DPLANE + (4 * c) * DP] = l.x; dst[DPLANE + (4 * c + 2) * DP] = l.y;
    };
    auto write_dy = [&](const unsigned (&v)[8], int buf) {
#pragma unroll
        for (int c = 0; c < 4; c++) write_dy_pair(v, buf, c);
    };
    // the padded input: padded row yp, padded columns xs .. xs + XW - 1
    auto load_x = [&](int yp, int xs, unsigned (&v)[12], bool wanted) {
        const int gy = yp - p.pad;
        const bool rowOk = wanted & ((unsigned)gy < (unsigned)p.H);         // uniform; a padding row has no valid column
        const unsigned wLimit = rowOk ? (unsigned)p.W : 0u;
        const unsigned rowTerm = xRowOff + (unsigned)((rowOk ? gy : 0) * p.W) * (unsigned)sizeof(T);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const int gx = xs + xCol + 16 * c - p.pad;
            const unsigned base = ((unsigned)gx < wLimit ? (unsigned)gx * (unsigned)sizeof(T) : 0x80000000u) + rowTerm;
#pragma unroll
            for (int r = 0; r < 4; r++) v[3 * r + c] = bufld<T>::ldraw(xr, base + (unsigned)(4 * r) * xPlane, 0);
        }
    };
    auto write_x_pair = [&](const unsigned (&v)[12], int slot, int c, int r) {    // column block c, rows xRow0 + 4 r and xRow0 + 4 r + 4
        _Float16* dst = sX + slot * 2 * XPLANE + xRow0 * XP + xCol;
        v2h h, l;
        split2(bufld<T>::fromraw(v[3 * r + c]) * scX, bufld<T>::fromraw(v[3 * (r + 1) + c]) * scX, h, l);
        dst[(4 * r) * XP + 16 * c] = h.x; dst[(4 * r + 4) * XP + 16 * c] = h.y;
        dst[XPLANE + (4 * r) * XP + 16 * c] = l.x; dst[XPLANE + (4 * r + 4) * XP + 16 * c] = l.y;
    };
    auto write_x = [&](const unsigned (&v)[12], int slot) {
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int r = 0; r < 4; r += 2) write_x_pair(v, slot, c, r);
    };
    // (ky, all kx) of one K step: the 16-half window of this lane's input row, then the three column shifts.  RANGE = this wave
    // owns only taps [tLo, tHi) (thin tiles; wave-uniform tests); without it the nine MFMAs of a filter row are straight-line code,
    // issued part by part over the three shifts so that neighbours in the matrix pipe never wait for each other's accumulator
    auto taps_of_row = [&](auto range, int slot, int ky, int k0, v8h ah, v8h al, auto&& between) {
        constexpr bool RANGE = decltype(range)::value;
        if (RANGE && (ky * KS + KS <= tLo || ky * KS >= tHi)) return;      // none of this filter row's taps is this wave's
        const _Float16* src = sX + slot * 2 * XPLANE + (ib * 32 + li) * XP + k0 + 8 * lh;
        const u32x4 h0 = *reinterpret_cast<const u32x4*>(src), h1 = *reinterpret_cast<const u32x4*>(src + 8);
        const u32x4 l0 = *reinterpret_cast<const u32x4*>(src + XPLANE), l1 = *reinterpret_cast<const u32x4*>(src + XPLANE + 8);
        const unsigned wh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        const unsigned wl[8] = {l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, l1.z, l1.w};
        v8h fbh[KS], fbl[KS];
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
            fbh[kx] = __builtin_bit_cast(v8h, bh); fbl[kx] = __builtin_bit_cast(v8h, bl);
        }
        if (RANGE) {
#pragma unroll
            for (int kx = 0; kx < KS; kx++) {
                const int tp = ky * KS + kx;
                if (tp < tLo || tp >= tHi) continue;                       // wave-uniform
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, fbh[kx], acc[tp], 0, 0, 0);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fbl[kx], acc[tp], 0, 0, 0);
                acc[tp] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fbh[kx], acc[tp], 0, 0, 0);
            }
        } else {
            // part by part over the shifts; `between` (a slice of the next row's hand-over, or nothing) follows each group of KS
            // MFMAs and is fenced there, so that its vector work runs in the shadow of the matrix pipe instead of after it
#pragma unroll
            for (int kx = 0; kx < KS; kx++) acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, fbh[kx], acc[ky * KS + kx], 0, 0, 0);
            between(ky * 3 + 0);
#pragma unroll
            for (int kx = 0; kx < KS; kx++) acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fbl[kx], acc[ky * KS + kx], 0, 0, 0);
            between(ky * 3 + 1);
#pragma unroll
            for (int kx = 0; kx < KS; kx++) acc[ky * KS + kx] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, fbh[kx], acc[ky * KS + kx], 0, 0, 0);
            between(ky * 3 + 2);
        }
    };
    auto mfma_step = [&](auto range, int y, int k0, auto&& between) {
        const _Float16* asrc = sD + (y & 1) * 2 * DPLANE + (ob * 32 + li) * DP + k0 + 8 * lh;
        const v8h ah = *reinterpret_cast<const v8h*>(asrc), al = *reinterpret_cast<const v8h*>(asrc + DPLANE);
#pragma unroll
        for (int ky = 0; ky < KS; ky++) taps_of_row(range, (y + ky) % RING, ky, k0, ah, al, between);
    };
    const bool wholeBlock = blockReal && tLo == 0 && tHi == TAPS;        // wave-uniform: one block, all taps (every wave of a full tile)

    auto run = [&](auto range) {
    unsigned vd[8], vx[12];
    for (int seg = seg0; seg < seg1; seg++) {
        const int xs = seg * KT;                        // first dy column of the segment = first padded-input column
        __syncthreads();                                // the previous segment's fragment reads are done
        // ring warm-up: padded rows y0 .. y0 + KS - 1 and the first dy row
#pragma unroll
        for (int r = 0; r < KS - 1; r++) { load_x(y0 + r, xs, vx, true); write_x(vx, (y0 + r) % RING); }
        load_dy(y0, xs, vd, true); load_x(y0 + KS - 1, xs, vx, true);
        write_dy(vd, y0 & 1); write_x(vx, (y0 + KS - 1) % RING);
        __syncthreads();
        for (int y = y0; y < y1; y++) {
            // One barrier per row: the next row's samples are requested before this row's MFMAs and handed to LDS between its two K
            // steps (their split runs in the shadow of the matrix pipe), into the other dy buffer and the ring slot that row y - 1
            // released (both last read before the previous barrier).  Branch-free: past the band's last row the requests are out of
            // range (zeros, no traffic) and land in buffers nobody reads
            const bool more = y + 1 < y1;
            load_dy(y + 1, xs, vd, more); load_x(y + KS, xs, vx, more);
            mfma_step(range, y, 0, [](int) {});
            if (decltype(range)::value || KS != 3) {
                write_dy(vd, (y + 1) & 1); write_x(vx, (y + KS) % RING);
                mfma_step(range, y, 16, [](int) {});
            } else {
                // ten slices of the hand-over (4 dy row pairs, 6 input pairs) behind the nine MFMA groups of the second K step
                mfma_step(range, y, 16, [&](int g) {
                    if (g < 4) write_dy_pair(vd, (y + 1) & 1, g);
                    else write_x_pair(vx, (y + KS) % RING, (g - 4) % 3, 2 * ((g - 4) / 3));
                    if (g == 8) write_x_pair(vx, (y + KS) % RING, 2, 2);
                    __builtin_amdgcn_sched_barrier(0);
                });
            }
            __syncthreads();
        }
    }
    };
    // two copies of the loop nest rather than a test inside it: a merge of the accumulators of two MFMA sites per row would double them
    if (wholeBlock) run(std::false_type{});
    else run(std::true_type{});

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
    // LDS image (halfs): two dy buffers + KS + 1 ring slots of the input, hi | lo planes of 64 rows each: 76 KB for 3x3, two workgroups per CU
    constexpr unsigned ldsBytes = (2u * 2u * 64u * (32 + 8) + (KS + 1) * 2u * 64u * (32 + 16 + 8)) * 2u;
    auto kern = wgrad_f16x3_kernel<T, KS>;
    SG3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    hipLaunchKernelGGL(kern, g, b, ldsBytes, st, p);
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
    // 3x3: ONE round of resident workgroups (two per CU on 256 CUs), never more: every further round costs a ring warm-up, a partial
    // image of 9 x 64 x 64 floats per workgroup and a ragged tail (measured on the T-1024 layers, batch 1: 512 -> 3.17 ms, 1024 ->
    // 3.37, 2048 -> 3.61, 256 -> 3.64; `SG3_WGRAD_WGS` overrides).  1x1 (ToRGB): short workgroups, a few rounds
    static const long long forced = [] { const char* e = getenv("SG3_WGRAD_WGS"); const long long v = e ? atoll(e) : 0; return v > 0 ? v : 0; }();
    const long long target = forced ? forced : (k == 3 ? 512 : 2048);
    const long long tiles = (long long)ceil_div(O, 64) * ceil_div(I, 64) * N;
    long long want = k == 3 ? target / tiles : ceil_div64(target, tiles);
    if (want < 1) want = 1;
    const int maxBands = max(1, OH / 8), nSegs = ceil_div(OW, 32);       // bands tall enough that the ring warm-up stays small
    int bands = 1, groups = 1;
    if (k == 3) {
        // the (bands, segment groups) pair with the most workgroups not beyond the target, after dropping empty bands / groups;
        // ties go to fewer groups (each group restarts the ring for every band)
        long long best = 0;
        for (int g = 1; g <= nSegs; g++) {
            const int gEff = ceil_div(nSegs, ceil_div(nSegs, g));
            const long long b0 = min((long long)maxBands, want / gEff);
            if (b0 < 1) break;
            const int bEff = ceil_div(OH, ceil_div(OH, (int)b0));
            if ((long long)bEff * gEff > best) { best = (long long)bEff * gEff; bands = bEff; groups = gEff; }
        }
    } else {
        bands = (int)want;
        if (bands > maxBands) {
            groups = (int)ceil_div64(bands, maxBands);
            bands = maxBands;
            if (groups > nSegs) groups = nSegs;
        }
        bands = ceil_div(OH, ceil_div(OH, bands));
        groups = ceil_div(nSegs, ceil_div(nSegs, groups));
    }
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
    // rows of channel padding (up to the next multiple of 64) are requested beyond num_records: their offsets must not wrap
    SG3_REQUIRE((int64_t)((p->I + 63) / 64 * 64) * p->H * p->W * 4 < (int64_t)1 << 31 && (int64_t)((p->O + 63) / 64 * 64) * (p->H + 2) * (p->W + 2) * 4 < (int64_t)1 << 31,
                "conv2d_wgrad: a sample padded to 64 channels must stay below 2 GiB (32-bit offsets)");
    hipStream_t st = (hipStream_t)stream;
    if (p->dtype == SG3_F32) return p->k == 3 ? launch_wgrad<float, 3>(*p, st) : launch_wgrad<float, 1>(*p, st);
    return p->k == 3 ? launch_wgrad<_Float16, 3>(*p, st) : launch_wgrad<_Float16, 1>(*p, st);
}

} // extern "C"
