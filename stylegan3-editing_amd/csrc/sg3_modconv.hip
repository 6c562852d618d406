// sg3_modconv.hip -- StyleGAN3 modulated convolution as ONE batch-wide implicit GEMM on the gfx950 matrix cores.
//
// Replaces the grouped convolution of modulated_conv2d (reference models/stylegan3/networks_stylegan3.py:24-63; the
// reference expands the weights N-fold to [N*O,I,k,k] and calls cuDNN with groups=N, :59-62).  Here
//
//     out[n,o,y,x] = dcoef[n,o] * sum_{i,ky,kx} wn[o,i,ky,kx] * ( x[n,i,y+ky-pad,x+kx-pad] * sIn[n,i] )
//
// so all samples share one weight matrix:  M = O (out channels), N = pixels, K = I*k*k.
//
// Kernels in this file (the host picks one per call, see sg3_modulated_conv2d):
//   * modconv_f16x3_kernel   3x3, split precision (fp16 hi/lo operands, three fp16 MFMAs per product, fp32-equivalent) or
//                            plain fp16 (the reference's mixed-precision layers): the default when |x| is bounded;
//   * modconv1_f16x3_kernel  1x1 (config R), same arithmetic, a plain GEMM over flat 256-pixel tiles;
//   * modconv_mfma_kernel    3x3 / 1x1, v_mfma_f32_32x32x2_f32: exact fp32 products (bitwise an fmaf chain), for inputs
//                            without a bound;
//   * modconv_1x1_small_kernel  ToRGB (O <= 4): HBM-bound, no matrix cores;
//   * modconv_prep_{w,s}_kernel  weight normalisation + packing, style normalisation, demodulation coefficients.
// Common scheme: A (weights) pre-packed per K chunk so a workgroup's A tile is a straight 16-byte-per-lane copy into
// padded LDS rows; B never materialised: the input patch (tile + halo) of one K chunk is staged once in LDS, scaled by
// sIn on the way in, channel-interleaved so that the im2col view of tap (ky,kx) is one conflict-free ds_read_b128 per
// lane at a shifted offset; the next chunk is fetched global->registers while the current one is in the MFMA loop;
// tile shapes per layer so that odd channel counts (323, 203, 81, 51, 32, 3) keep the M dimension busy; block ids
// renumbered so that the M tiles sharing an input patch, and neighbouring patches, sit on one XCD.
//
// demodulation (dcoef) is applied in the epilogue; bias/activation belong to the following filtered_lrelu.
#include "sg3_common.h"
#include "sg3_split.h"
#include "sg3_modconv_f23.h"
#include <algorithm>
#include <cstdlib>

#ifndef SG3_TAILPACK
#define SG3_TAILPACK 1          // 0 compiles the tap-packed tail chunk of the 3x3 kernel out (A/B builds)
#endif

namespace sg3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KS> struct ConvK;
template <> struct ConvK<3> { static constexpr int TAPS = 9, KC = 8; };
template <> struct ConvK<1> { static constexpr int TAPS = 1, KC = 16; };

static inline int packed_kc(int k) { return k == 3 ? ConvK<3>::KC : ConvK<1>::KC; }
// 16-channel chunks of the f16x3 packing; 1x1 kernels stage two chunks per step, so their count is padded to even
static inline int f16x3_chunks(int I, int k) { const int c = (I + 15) / 16; return k == 1 ? (c + 1) / 2 * 2 : c; }

struct ConvParams {
    const void* x; const float* wp; const float* sIn; const float* dcoef; void* out;
    int N, I, O, H, W, outH, outW, pad;
    int nch;                         // K chunks = ceil(I / KC)
    int xTiles, yTiles, mTiles;
    int totalBlocks;
    int outPitch;                    // elements between output rows (row-streaming 3x3 kernel; elsewhere = outW)
    int tailPack;                    // row-streaming 3x3 kernel: the last K chunk holds at most 4 channels (see the kernel)
    const float* epBias; float epClamp, epScale;     // ToRGB kernel only: out = clamp(conv + bias[o]) * scale (see sg3_modconv_params)
    int kSplits; float* partial;     // flat 3x3 kernel only: input channels split over kSplits workgroups per tile, raw sums to partial
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
// Split-precision variant (SG3_CONV_F16X3), 3x3 kernels.
//
// Every fp32 operand is split on the way into LDS into two fp16 halves, x = hi + lo with hi = fp16(x) and
// lo = fp16(x - hi), and each 16-deep K step runs three v_mfma_f32_32x32x16_f16:  Ah*Bh + Ah*Bl + Al*Bh.
// Products of two 11-bit significands are exact in the fp32 accumulator, so the only omission is lo*lo ~ 2^-22 of the
// product -- the same order as fp32's own rounding -- at 16/3 = 5.3x the rate of the fp32 MFMA.  The prep kernel has
// already split and packed the weights ([O][I/16][tap][hi|lo][16] halfs) and scaled sIn by a power of two per sample
// so |x*sIn| < 2^15 given the caller's bound on |x| (dcoef carries the inverse power).
// LDS image: A rows of 9 taps x (16 hi | 16 lo) halfs + 8 halfs of padding (conflict-free b128 rows); B as four
// planes [channel-half][hi|lo][py][px][8 halfs], so a lane's im2col fragment for tap (ky,kx) is one ds_read_b128.
// SPLIT = false is the plain fp16 form (SG3_CONV_F16): operands rounded to fp16 once, ONE MFMA per K step -- the
// arithmetic of the reference's fp16 layers (fp16 cuDNN convolution with fp32 accumulation).
// MFMA loop order: every wave owns ONE 32-channel M block and TN output rows; the column offset kx is outermost, the
// three A fragments of that column are held in registers while the patch rows stream through, and one B fragment
// serves up to three output rows (ky = 0..2): 18 + 6 (TN + 2) ds_read_b128 per 27 TN MFMAs (0.5 per MFMA at TN = 4;
// a tap-by-tap loop over 2x2 blocks needs 0.67).
template <typename T, int WM, int WN, int TN, bool SPLIT, bool PACK>
__global__ void __launch_bounds__(256, 2)              // two workgroups per CU
modconv_f16x3_kernel(ConvParams p) {
    constexpr int TM = 1;
    constexpr int KS = 3, TAPS = 9, KC = 16;
    constexpr int NPART = SPLIT ? 2 : 1;               // B planes per channel half: hi | lo
    constexpr int BM = WM * TM * 32;
    constexpr int ROWS = WN * TN;
    constexpr int PH = ROWS + KS - 1, PW = 32 + KS - 1;
    constexpr int NPIX = PH * PW;
    constexpr int AS = TAPS * 32 + 8;                  // halfs per A row in LDS (592 B)
    constexpr int AROW_V = TAPS * 32 / 8;              // 16-byte vectors per packed A row (36)
    constexpr int TPR = (BM == 32) ? 8 : (BM == 64 ? 4 : 2);     // threads that share one A row
    constexpr int A_PER = (AROW_V + TPR - 1) / TPR;    // vectors per thread (immediate offsets 16 * TPR apart)
    constexpr int BPLANE = NPIX * 8;                   // halfs per (channel-half, part) plane
    constexpr int PX_PER = (NPIX + 255) / 256;         // patch pixels per thread
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

    // ---- staging maps, fixed for the whole K loop (all addressing is descriptor + 32-bit offsets) ----
    const unsigned HWb = (unsigned)(p.H * p.W) * (unsigned)sizeof(T);            // bytes per channel plane
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.x + (size_t)n * p.I * p.H * p.W), (short)0, (int)((unsigned)p.I * HWb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wp, (short)0, (int)((unsigned)p.O * (unsigned)p.nch * (unsigned)(AROW_V * 16)), 0x00020000);
    // A: thread -> (row, first vector); its vectors are TPR * 16 bytes apart
    const int arow = tid / TPR, acol = tid % TPR;
    const bool aOk = arow < BM;
    const unsigned aG = (aOk && o0 + arow < p.O) ? ((unsigned)(o0 + arow) * (unsigned)p.nch * (AROW_V * 16) + acol * 16) : 0x80000000u;
    _Float16* aL = sA + arow * AS + acol * 8;
    // B: thread -> PX_PER patch pixels (out-of-image pixels get an out-of-range offset and read as zero)
    unsigned bG[PX_PER];
    int bL[PX_PER];
#pragma unroll
    for (int q = 0; q < PX_PER; q++) {
        const int e = tid + 256 * q;
        const int px = e % PW, py = e / PW;
        const int gy = y0 - p.pad + py, gx = x0 - p.pad + px;
        const bool ok = e < NPIX && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        bG[q] = ok ? (unsigned)(gy * p.W + gx) * (unsigned)sizeof(T) : 0x80000000u;
        bL[q] = e < NPIX ? e * 8 : -1;
    }
    // style scales of a chunk's 16 channels: lane c of every wave requests channel c together with the chunk's pixels (a
    // descriptor over this sample's row of sIn: channels beyond I read 0, which also silences the padded channels) and
    // stage() broadcasts them with v_readlane -- no scalar loads, waits or branches between the two barriers
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.sIn + (size_t)n * p.I), (short)0, p.I * 4, 0x00020000);
    float rsc;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;

    u32x4 ra[A_PER];
    float rb[PX_PER][2][8];

    auto fetch = [&](int ch) {
        rsc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, (ch * KC + (lane & 15)) * 4, 0, 0));
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
                // channel offset rides in the scalar offset (not range checked: padded channels alias channel 0 and
                // are multiplied by zero); the pixel offset is the range-checked vector offset
                const unsigned coff = ci < p.I ? (unsigned)ci * HWb : 0u;
                // NO arithmetic on the loaded values here: a use would put the s_waitcnt for these loads in front of
                // the MFMA loop and expose the memory latency every chunk; the style scale is applied in stage()
#pragma unroll
                for (int q = 0; q < PX_PER; q++) rb[q][hf][c] = bufld<T>::ld(xr, bG[q], coff);
            }
    };
    auto stage = [&](int ch) {
        float sc[2][8];
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int c = 0; c < 8; c++)
                sc[hf][c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rsc), hf * 8 + c));
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
                    // fp16 form: the two products would be SLP-packed into v_pk_mul_f32 (the split form's bit masks keep them apart)
                    const float v0 = SPLIT ? rb[q][hf][2 * c] * sc[hf][2 * c] : mul_single(rb[q][hf][2 * c], sc[hf][2 * c]);
                    const float v1 = SPLIT ? rb[q][hf][2 * c + 1] * sc[hf][2 * c + 1] : mul_single(rb[q][hf][2 * c + 1], sc[hf][2 * c + 1]);
                    if (SPLIT) split2(v0, v1, h[c], l[c]);
                    else h[c] = round2(v0, v1);
                }
                _Float16* dst = sB + (hf * NPART) * BPLANE + bL[q];
                *reinterpret_cast<v8h*>(dst) = __builtin_shufflevector(__builtin_shufflevector(h[0], h[1], 0, 1, 2, 3), __builtin_shufflevector(h[2], h[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
                if (SPLIT)
                    *reinterpret_cast<v8h*>(dst + BPLANE) = __builtin_shufflevector(__builtin_shufflevector(l[0], l[1], 0, 1, 2, 3), __builtin_shufflevector(l[2], l[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };

    // demodulation coefficients of this lane's 16 output channels: requested first, consumed by the epilogue (channels
    // beyond O read 0 through the range check)
    const int oL = o0 + wm * 32 + 4 * lh;                                                // this lane's first channel
    float d[16];
    {
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dcoef + (size_t)n * p.O), (short)0, p.O * 4, 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; r++)
            d[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, (oL + (r & 3) + 8 * (r >> 2)) * 4, 0, 0));
    }
    struct BFrag { v8h h, l; };
    auto load_b = [&](BFrag& f, int pr, int kx) {
        const _Float16* src = sB + (lh * NPART) * BPLANE + ((wn * TN + pr) * PW + li + kx) * 8;
        f.h = *reinterpret_cast<const v8h*>(src);
        if (SPLIT) f.l = *reinterpret_cast<const v8h*>(src + BPLANE);
    };
    auto mfma_row = [&](const BFrag& f, const v8h (&ah)[3], const v8h (&al)[3], int pr) {
        // the three products go round the (up to three) output rows this patch row feeds, so consecutive
        // MFMAs never wait on each other's accumulator
        if (SPLIT) {
#pragma unroll
            for (int ky = 0; ky < 3; ky++) { const int b = pr - ky; if (b >= 0 && b < TN) acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[ky], f.h, acc[0][b], 0, 0, 0); }
#pragma unroll
            for (int ky = 0; ky < 3; ky++) { const int b = pr - ky; if (b >= 0 && b < TN) acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ky], f.l, acc[0][b], 0, 0, 0); }
        }
#pragma unroll
        for (int ky = 0; ky < 3; ky++) { const int b = pr - ky; if (b >= 0 && b < TN) acc[0][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[ky], f.h, acc[0][b], 0, 0, 0); }
    };

    // A wave whose 32 channels lie beyond O (O = 323: the second M block of the sixth 64-channel tile) or whose rows lie beyond the
    // image stages and synchronises with its workgroup but issues no MFMAs: the matrix pipe and its power go to the CU's other waves
    const bool active = o0 + wm * 32 < p.O && y0 + wn * TN < p.outH;          // wave-uniform
    // PACK: the last chunk holds at most 4 channels and is peeled off the loop (below)
    const int nMain = PACK ? p.nch - 1 : p.nch;
    fetch(0);
    for (int ch = 0; ch < nMain; ch++) {
        __syncthreads();
        stage(ch);
        __syncthreads();
        if (ch + 1 < p.nch) fetch(ch + 1);
        if (!active) continue;
        // the wave that is in its MFMA loop gets issue priority over the other workgroup's wave on the same SIMD (which is
        // staging or waiting): L6 2331 -> 2236 us, L8 2230 -> 2116 us; no effect on the thin layers
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kx = 0; kx < 3; kx++) {
            v8h ah[3], al[3];
#pragma unroll
            for (int ky = 0; ky < 3; ky++) {
                const _Float16* src = sA + (wm * 32 + li) * AS + (ky * 3 + kx) * 32 + lh * 8;
                ah[ky] = *reinterpret_cast<const v8h*>(src);
                if (SPLIT) al[ky] = *reinterpret_cast<const v8h*>(src + 16);
            }
            BFrag b0, b1;
            load_b(b0, 0, kx);
#pragma unroll
            for (int pr = 0; pr < TN + 2; pr += 2) {
                if (pr + 1 < TN + 2) load_b(b1, pr + 1, kx);
                __builtin_amdgcn_sched_barrier(0);
                mfma_row(b0, ah, al, pr);
                __builtin_amdgcn_sched_barrier(0);
                if (pr + 1 < TN + 2) {
                    if (pr + 2 < TN + 2) load_b(b0, pr + 2, kx);
                    __builtin_amdgcn_sched_barrier(0);
                    mfma_row(b1, ah, al, pr + 1);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }
    if (PACK) {
        // Last chunk with at most 4 channels (I = 81, 51, 323 at FFHQ-1024: 1 or 3 of them): its 16 K slots would carry 4
        // channels of ONE tap and 12 zeros.  Instead the three column offsets of a filter row share a K step: slots
        // 0-3 / 4-7 / 8-11 = channels 0-3 of kx = 0 / 1 / 2 (slots 12-15: channels 4-7 of kx = 2, zero in the packed
        // weights).  Both operands come from the usual LDS images through two 8-byte reads per lane -- two lane-dependent
        // bases per operand plus immediate offsets -- and the chunk costs 9 TN MFMAs instead of 27 TN.  Peeled off the
        // loop: inside it, next to the prefetch registers, the extra code path made the whole kernel spill.
        __syncthreads();
        stage(p.nch - 1);
        __syncthreads();
        typedef _Float16 v4h __attribute__((ext_vector_type(4)));
        if (active) {
        auto two = [](const _Float16* a, const _Float16* b) {
            return __builtin_shufflevector(*reinterpret_cast<const v4h*>(a), *reinterpret_cast<const v4h*>(b), 0, 1, 2, 3, 4, 5, 6, 7);
        };
        const _Float16* a0 = sA + (wm * 32 + li) * AS + (lh ? 2 * 32 : 0);
        const _Float16* a1 = sA + (wm * 32 + li) * AS + (lh ? 2 * 32 + 4 : 32);
        const _Float16* q0 = sB + (wn * TN * PW + li) * 8 + (lh ? 2 * 8 : 0);      // channel-half 0, hi plane
        const _Float16* q1 = sB + (wn * TN * PW + li) * 8 + (lh ? 2 * 8 + 4 : 8);
        v8h ah[3], al[3];
#pragma unroll
        for (int ky = 0; ky < 3; ky++) {
            ah[ky] = two(a0 + ky * 96, a1 + ky * 96);
            if (SPLIT) al[ky] = two(a0 + ky * 96 + 16, a1 + ky * 96 + 16);
        }
#pragma unroll
        for (int pr = 0; pr < TN + 2; pr++) {
            BFrag f;
            f.h = two(q0 + pr * PW * 8, q1 + pr * PW * 8);
            if (SPLIT) f.l = two(q0 + pr * PW * 8 + BPLANE, q1 + pr * PW * 8 + BPLANE);
            __builtin_amdgcn_sched_barrier(0);
            mfma_row(f, ah, al, pr);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
    }

    T* outp = (T*)p.out + (size_t)n * p.O * p.outH * p.outPitch;
    const int gx = x0 + li;
    const unsigned planeB = (unsigned)(p.outH * p.outPitch) * (unsigned)sizeof(T);      // bytes per output channel plane
    // every offset a lane can form -- channels of the padded last M tile included -- must stay below 2^31, so that nothing
    // wraps around into the tensor: (padded O + one wave block) * plane < 2^31
    if ((unsigned long long)(p.mTiles * BM + 32) * planeB < 0x7fffffffULL) {
        // Stores through a descriptor over this sample's output: byte offset = channel * plane + row + column in 32 bits,
        // and the hardware range check drops what lies outside -- channels beyond O (offset >= O planes) and, by
        // starting at 2^31, the columns beyond the image.  No per-store predicate, no 64-bit address arithmetic: with
        // two to eight K chunks per tile (the thin 1024^2 layers) the predicated form cost as much as a K chunk.
        const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)outp, (short)0, (int)((unsigned)p.O * planeB), 0x00020000);
        const unsigned laneBase = gx < p.outW ? (unsigned)oL * planeB + (unsigned)gx * (unsigned)sizeof(T) : 0x80000000u;
#pragma unroll
        for (int b = 0; b < TN; b++) {
            const int gy = y0 + wn * TN + b;
            if (gy >= p.outH) continue;                                                  // wave-uniform
            const unsigned rowOff = laneBase + (unsigned)(gy * p.outPitch) * (unsigned)sizeof(T);
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const unsigned off = rowOff + (unsigned)((r & 3) + 8 * (r >> 2)) * planeB;
                const float v = acc[0][b][r] * d[r];
                if (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orr, (int)off, 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), orr, (int)off, 0, 0);
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
            const float d = p.dcoef[(size_t)n * p.O + o];
#pragma unroll
            for (int b = 0; b < TN; b++) {
                const int gy = y0 + wn * TN + b;
                if (gy < p.outH && gx < p.outW)
                    io<T>::st(outp + ((size_t)o * p.outH + gy) * p.outPitch + gx, acc[a][b][r] * d);
            }
        }
}

// ---------------------------------------------------------------------------
// 3x3 kernel for NARROW outputs (the 36^2 .. 84^2 layers: 38 .. 86 output columns).  modconv_f16x3_kernel tiles an output row
// into 32-column pieces, so a 38-column row fills 59 % of its two tiles.  Here the MFMA's 32 pixel columns are a run of 32
// consecutive pixels of the FLATTENED output plane (crossing row ends), as in the 1x1 kernel: a 38 x 38 plane is 45.1 runs
// (fill 98 %).  A workgroup (4 waves: 2 M blocks x 2 pixel groups) takes 2 TN consecutive runs; the staged patch is every input
// row those runs touch (+ 2), full width (+ 2): at most FLAT_NPIX pixels.  Every lane keeps the patch offset of its pixel for
// each of its TN runs; a tap adds (ky PW + kx).  No row streaming (a run's neighbours in y are not a run): tap by tap, the A
// fragments of a tap held over the TN runs.  Same packed weights, style / demodulation handling and arithmetic as the row kernel.
constexpr int FLAT_NPIX = 512;                 // patch pixels staged per chunk (two per thread)

template <typename T, int TN, bool SPLIT>
__global__ void __launch_bounds__(256, 2)
modconv_flat_kernel(ConvParams p) {
    constexpr int TAPS = 9, KC = 16;
    constexpr int NPART = SPLIT ? 2 : 1;
    constexpr int BM = 64, WN = 2;
    constexpr int RUN = WN * TN * 32;                  // flat pixels per workgroup
    constexpr int AS = TAPS * 32 + 8;
    constexpr int AROW_V = TAPS * 32 / 8;              // 36
    constexpr int TPR = 4;                             // threads that share one A row (64 rows)
    constexpr int A_PER = AROW_V / TPR;                // 9
    constexpr int BPLANE = FLAT_NPIX * 8;
    constexpr int PX_PER = FLAT_NPIX / 256;

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
    // Small grids (batch 1 on the 36^2 maps: 96 tiles for 256 CUs, each walking all 32 K chunks with nobody to hide its latency):
    // the K chunks of a tile are split over kSplits workgroups, innermost in the block order so that they run side by side; each
    // writes its raw sums to partial[ks] and modconv_split_reduce_kernel adds them in order and applies the demodulation.
    const int ks = bid % p.kSplits; bid /= p.kSplits;
    const int chBegin = (int)((long long)ks * p.nch / p.kSplits), chEnd = (int)((long long)(ks + 1) * p.nch / p.kSplits);
    const int mt = bid % p.mTiles; bid /= p.mTiles;
    const int xt = bid % p.xTiles; const int n = bid / p.xTiles;             // xTiles = runs of RUN flat pixels per plane
    const int o0 = mt * BM;
    const int P = p.outH * p.outW;
    const int f0 = xt * RUN;
    const int ya = f0 / p.outW;                                              // first output row touched
    const int PW = p.outW + 2;                                               // patch width: input columns -pad .. outW - pad + 1
    const int prows = ((f0 + RUN < P ? f0 + RUN : P) - 1) / p.outW - ya + 3;  // patch rows this workgroup's runs read

    const unsigned HWb = (unsigned)(p.H * p.W) * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.x + (size_t)n * p.I * p.H * p.W), (short)0, (int)((unsigned)p.I * HWb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wp, (short)0, (int)((unsigned)p.O * (unsigned)p.nch * (unsigned)(AROW_V * 16)), 0x00020000);
    const int arow = tid / TPR, acol = tid % TPR;
    const unsigned aG = (o0 + arow < p.O) ? ((unsigned)(o0 + arow) * (unsigned)p.nch * (AROW_V * 16) + acol * 16) : 0x80000000u;
    _Float16* aL = sA + arow * AS + acol * 8;
    // B: thread -> PX_PER patch pixels, patch pixel e = py * PW + px <-> input (ya - pad + py, -pad + px)
    unsigned bG[PX_PER]; int bL[PX_PER];
#pragma unroll
    for (int q = 0; q < PX_PER; q++) {
        const int e = tid + 256 * q;
        const int py = e / PW, px = e - py * PW;
        const int gy = ya - p.pad + py, gx = px - p.pad;
        const bool ok = py < prows && (unsigned)gy < (unsigned)p.H && (unsigned)gx < (unsigned)p.W;
        bG[q] = ok ? (unsigned)(gy * p.W + gx) * (unsigned)sizeof(T) : 0x80000000u;
        bL[q] = e * 8;
    }
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.sIn + (size_t)n * p.I), (short)0, p.I * 4, 0x00020000);
    float rsc;
    // this lane's pixel in each of its TN runs: patch offset (halfs) and flat index
    int pb[TN], pf[TN];
#pragma unroll
    for (int b = 0; b < TN; b++) {
        const int f = f0 + (wn * TN + b) * 32 + li;
        const int y = f / p.outW, x = f - y * p.outW;
        pf[b] = f < P ? f : -1;
        pb[b] = f < P ? ((y - ya) * PW + x) * 8 : 0;                          // runs beyond the plane read pixel 0 and are not stored
    }

    f32x16 acc[TN];
#pragma unroll
    for (int b = 0; b < TN; b++)
#pragma unroll
        for (int r = 0; r < 16; r++) acc[b][r] = 0.f;
    u32x4 ra[A_PER];
    float rb[PX_PER][2][8];

    auto fetch = [&](int ch) {
        rsc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, (ch * KC + (lane & 15)) * 4, 0, 0));
        const unsigned aoff = aG + (unsigned)ch * (AROW_V * 16);
#pragma unroll
        for (int q = 0; q < A_PER; q++) ra[q] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(aoff + q * TPR * 16), 0, 0);
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const int ci = ch * KC + hf * 8 + c;                        // wave-uniform
                const unsigned coff = ci < p.I ? (unsigned)ci * HWb : 0u;
#pragma unroll
                for (int q = 0; q < PX_PER; q++) rb[q][hf][c] = bufld<T>::ld(xr, bG[q], coff);
            }
    };
    auto stage = [&]() {
        float sc[2][8];
#pragma unroll
        for (int hf = 0; hf < 2; hf++)
#pragma unroll
            for (int c = 0; c < 8; c++)
                sc[hf][c] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rsc), hf * 8 + c));
#pragma unroll
        for (int q = 0; q < A_PER; q++) *reinterpret_cast<u32x4*>(aL + q * TPR * 8) = ra[q];
#pragma unroll
        for (int q = 0; q < PX_PER; q++)
#pragma unroll
            for (int hf = 0; hf < 2; hf++) {
                v2h h[4], l[4];
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    // fp16 form: the two products would be SLP-packed into v_pk_mul_f32 (the split form's bit masks keep them apart)
                    const float v0 = SPLIT ? rb[q][hf][2 * c] * sc[hf][2 * c] : mul_single(rb[q][hf][2 * c], sc[hf][2 * c]);
                    const float v1 = SPLIT ? rb[q][hf][2 * c + 1] * sc[hf][2 * c + 1] : mul_single(rb[q][hf][2 * c + 1], sc[hf][2 * c + 1]);
                    if (SPLIT) split2(v0, v1, h[c], l[c]);
                    else h[c] = round2(v0, v1);
                }
                _Float16* dst = sB + (hf * NPART) * BPLANE + bL[q];
                *reinterpret_cast<v8h*>(dst) = __builtin_shufflevector(__builtin_shufflevector(h[0], h[1], 0, 1, 2, 3), __builtin_shufflevector(h[2], h[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
                if (SPLIT)
                    *reinterpret_cast<v8h*>(dst + BPLANE) = __builtin_shufflevector(__builtin_shufflevector(l[0], l[1], 0, 1, 2, 3), __builtin_shufflevector(l[2], l[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
            }
    };

    const int oL = o0 + wm * 32 + 4 * lh;
    float d[16];
    {
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dcoef + (size_t)n * p.O), (short)0, p.O * 4, 0x00020000);
#pragma unroll
        for (int r = 0; r < 16; r++)
            d[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, (oL + (r & 3) + 8 * (r >> 2)) * 4, 0, 0));
    }
    const _Float16* aBase = sA + (wm * 32 + li) * AS + lh * 8;
    const _Float16* bBase = sB + (lh * NPART) * BPLANE;

    fetch(chBegin);
    for (int ch = chBegin; ch < chEnd; ch++) {
        __syncthreads();
        stage();
        __syncthreads();
        if (ch + 1 < chEnd) fetch(ch + 1);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int tap = 0; tap < TAPS; tap++) {
            const int toff = ((tap / 3) * PW + tap % 3) * 8;                 // wave-uniform
            const v8h ah = *reinterpret_cast<const v8h*>(aBase + tap * 32);
            v8h al;
            if (SPLIT) al = *reinterpret_cast<const v8h*>(aBase + tap * 32 + 16);
#pragma unroll
            for (int b = 0; b < TN; b++) {
                if (f0 + (wn * TN + b) * 32 >= P) continue;                  // wave-uniform: a run beyond the plane issues no MFMAs
                const _Float16* src = bBase + pb[b] + toff;
                const v8h bh = *reinterpret_cast<const v8h*>(src);
                if (SPLIT) {
                    const v8h bl = *reinterpret_cast<const v8h*>(src + BPLANE);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc[b], 0, 0, 0);
                    acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc[b], 0, 0, 0);
                }
                acc[b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[b], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_setprio(0);
    }

    if (p.kSplits > 1) {
        // raw partial sums, fp32, [ks][n][o][P]: the reduction applies the demodulation coefficient
        float* pp = p.partial + ((size_t)ks * p.N + n) * p.O * P;
        const unsigned planeF = (unsigned)P * 4u;
        const __amdgpu_buffer_rsrc_t prr = __builtin_amdgcn_make_buffer_rsrc((void*)pp, (short)0, (int)((unsigned)p.O * planeF), 0x00020000);
#pragma unroll
        for (int b = 0; b < TN; b++) {
            const unsigned base = pf[b] >= 0 ? (unsigned)oL * planeF + (unsigned)pf[b] * 4u : 0x80000000u;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const float v = acc[b][r];               // a scalar first: hipcc 7.2 compiles __builtin_bit_cast of a vector ELEMENT as a read of element 0
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), prr, (int)(base + (unsigned)((r & 3) + 8 * (r >> 2)) * planeF), 0, 0);
            }
        }
        return;
    }
    // the runs are contiguous in the dense output plane: 128-byte store segments
    T* outp = (T*)p.out + (size_t)n * p.O * P;
    const unsigned planeB = (unsigned)P * (unsigned)sizeof(T);
    const bool desc = (unsigned long long)(p.mTiles * BM + 32) * planeB < 0x7fffffffULL;
    const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)outp, (short)0, desc ? (int)((unsigned)p.O * planeB) : 0, 0x00020000);
#pragma unroll
    for (int b = 0; b < TN; b++) {
        const unsigned base = pf[b] >= 0 ? (unsigned)oL * planeB + (unsigned)pf[b] * (unsigned)sizeof(T) : 0x80000000u;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int orow = (r & 3) + 8 * (r >> 2);
            const float v = acc[b][r] * d[r];
            if (desc) {
                if (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orr, (int)(base + (unsigned)orow * planeB), 0, 0);
                else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), orr, (int)(base + (unsigned)orow * planeB), 0, 0);
            } else if (pf[b] >= 0 && oL + orow < p.O) {
                io<T>::st(outp + (size_t)(oL + orow) * P + pf[b], v);
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Split-precision variant for 1x1 kernels (config R): a plain GEMM  out[O, pixels] = Wn[O, I] * (sIn * x)[I, pixels].
// Each staged input element is used once per output-channel tile (not 9 times as in the 3x3 kernel), so the tile is
// larger along M (128 x 256 pixels, eight waves, wave tile 64 x 64) to keep the L2 -> LDS stream at ~44 FLOP/B, two
// 16-channel K steps are staged per barrier, and the LDS image is double-buffered: one barrier per 32 input channels.
// LDS per buffer: A rows of (hi16|lo16) x 2 halfs + 8 halfs of padding; B as planes [k-step][channel-half][hi|lo][pixel][8].
// NBUF = 1: one LDS image (two barriers per stage) so that TWO workgroups fit a CU -- for the thin layers, whose few K
// stages leave a lone workgroup waiting on HBM at the start and the end of every tile.
// M16: the matrix instruction is v_mfma_f32_16x16x32_f16 (one per 16 x 16 output block and 32-channel stage) instead of
// v_mfma_f32_32x32x16_f16 (two per 32 x 32 block): the same cycles per FLOP and the same LDS fragment bytes, but the chip holds
// a 15 % higher clock on the small shape (profiles/r02_mfma_shape.txt).  A fragment = 16 rows x 32 channels: lane l holds row
// l % 16, channels 8g .. 8g+7 with g = l / 16 = (k-step, channel-half) of the staged image; B likewise for 16 pixels; the
// accumulator block holds rows 4 (l / 16) + i, pixel l % 16.
template <typename T, int WM, int WN, int TM, int TN, bool SPLIT, int NBUF, bool M16>
__global__ void __launch_bounds__(512, NBUF == 1 ? 4 : 2)      // second argument: waves per SIMD (a workgroup is two per SIMD)
modconv1_f16x3_kernel(ConvParams p) {
    constexpr int KSUB = 2, KC = 16 * KSUB;
    constexpr int NPART = SPLIT ? 2 : 1;
    constexpr int BM = WM * TM * 32;
    constexpr int ROWS = WN * TN;
    constexpr int NPIX = ROWS * 32;
    constexpr int AS = KSUB * 32 + 8;                  // halfs per A row in LDS
    constexpr int AROW_V = KSUB * 4;                   // 16-byte vectors per packed A row and stage
    constexpr int A_VEC = BM * AROW_V;
    constexpr int A_PER = (A_VEC + 511) / 512;
    constexpr int BPLANE = NPIX * 8;                   // halfs per (k-step, channel-half, part) plane
    constexpr int BUF = BM * AS + KSUB * 2 * NPART * BPLANE;   // halfs per LDS buffer
    static_assert(WM * WN == 8, "8 waves per workgroup");
    static_assert(NPIX == 256, "one patch pixel per thread pair");

    extern __shared__ __attribute__((aligned(16))) _Float16 smh[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    // staging task of this thread: a PAIR of adjacent pixels x the 8 channels of one (k-step, channel-half); wave-uniform group
    const int ssub = wave >> 2, kh = (wave >> 1) & 1;

    int bid = blockIdx.x;
    {
        const int nb = p.totalBlocks, q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
    }
    // grids smaller than the chip: the K stages of a tile split over kSplits workgroups, each writing its share (demodulation
    // applied: the epilogue is linear) to its own fp32 image partial[ks]; modconv_split_reduce_kernel adds the images in order
    const int ks = bid % p.kSplits; bid /= p.kSplits;
    const int chBegin = (int)((long long)ks * p.nch / p.kSplits), chEnd = (int)((long long)(ks + 1) * p.nch / p.kSplits);
    const int mt = bid % p.mTiles; bid /= p.mTiles;
    const int xt = bid % p.xTiles; bid /= p.xTiles;
    const int yt = bid % p.yTiles; const int n = bid / p.yTiles;
    // a 1x1 kernel has no halo, so the 256 pixels of a tile are a FLAT run of the H*W plane: every channel contributes
    // one contiguous 1 KB run (8-byte loads, full cache lines) instead of 8 row segments of 128 bytes
    const int P = p.H * p.W;
    const int o0 = mt * BM, p0 = (yt * p.xTiles + xt) * NPIX;

    const unsigned HWb = (unsigned)(p.H * p.W) * (unsigned)sizeof(T);
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)((const T*)p.x + (size_t)n * p.I * p.H * p.W), (short)0, (int)((unsigned)p.I * HWb), 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)p.wp, (short)0, (int)((unsigned)p.O * (unsigned)p.nch * (unsigned)(AROW_V * 16)), 0x00020000);
    // A: vector v of the stage -> row v / AROW_V, 16-byte column v % AROW_V
    unsigned aG[A_PER]; int aL[A_PER];
#pragma unroll
    for (int q = 0; q < A_PER; q++) {
        const int v = tid + 512 * q, row = v / AROW_V, col = v % AROW_V;
        const bool ok = v < A_VEC && o0 + row < p.O;
        aG[q] = ok ? ((unsigned)(o0 + row) * (unsigned)p.nch * (AROW_V * 16) + col * 16) : 0x80000000u;
        aL[q] = v < A_VEC ? row * AS + col * 8 : -1;
    }
    // B: this thread's pixel pair (pairs beyond the plane read the next channel or zero: those columns are never stored)
    const int e2 = tid & 127;
    const unsigned bG = (p0 + 2 * e2 < P) ? (unsigned)(p0 + 2 * e2) * (unsigned)sizeof(T) : 0x80000000u;
    // style scales of this wave's 8 channels: requested by lanes 0..7 with the pixels, broadcast with v_readlane in stage()
    // (channels beyond I read 0 through the range check, which also silences the padded channels)
    const __amdgpu_buffer_rsrc_t sr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.sIn + (size_t)n * p.I), (short)0, p.I * 4, 0x00020000);
    float rsc;

    f32x16 acc[TM][TN];                                // M16: the same 16 registers per 32 x 32 block, viewed as 2 x 2 blocks of 4
#pragma unroll
    for (int a = 0; a < TM; a++)
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[a][b][r] = 0.f;
    const int l16 = lane & 15, lg = lane >> 4;         // M16: row / pixel inside a 16-block, k group (k-step = lg >> 1, channel-half = lg & 1)

    u32x4 ra[A_PER];
    typename bufld<T>::raw2 rb[8];                     // raw pixel pairs, one per channel (unpacked in stage)

    auto fetch = [&](int ch) {
        rsc = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sr, (ch * KC + ssub * 16 + kh * 8 + (tid & 7)) * 4, 0, 0));
#pragma unroll
        for (int q = 0; q < A_PER; q++)
            ra[q] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)(aG[q] + (unsigned)ch * (AROW_V * 16)), 0, 0);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int ci = ch * KC + ssub * 16 + kh * 8 + c;                    // wave-uniform
            const unsigned coff = ci < p.I ? (unsigned)ci * HWb : 0u;
            rb[c] = bufld<T>::ld2(xr, bG, coff);       // no use of the values here: their s_waitcnt belongs after the MFMAs
        }
    };
    auto stage = [&](_Float16* buf, int ch) {
        _Float16* sA = buf;
        _Float16* sB = buf + BM * AS;
        float rv[8][2];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const float sc = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rsc), c));
            float v0, v1;
            bufld<T>::unpack2(rb[c], v0, v1);
            rv[c][0] = SPLIT ? v0 * sc : mul_single(v0, sc); rv[c][1] = SPLIT ? v1 * sc : mul_single(v1, sc);      // fp16 form: no v_pk_mul_f32 (see mul_single)
        }
#pragma unroll
        for (int q = 0; q < A_PER; q++)
            if (aL[q] >= 0) *reinterpret_cast<u32x4*>(sA + aL[q]) = ra[q];
#pragma unroll
        for (int j = 0; j < 2; j++) {
            v2h h[4], l[4];
#pragma unroll
            for (int c = 0; c < 4; c++) {
                if (SPLIT) split2(rv[2 * c][j], rv[2 * c + 1][j], h[c], l[c]);
                else h[c] = round2(rv[2 * c][j], rv[2 * c + 1][j]);
            }
            _Float16* dst = sB + ((ssub * 2 + kh) * NPART) * BPLANE + (2 * e2 + j) * 8;
            *reinterpret_cast<v8h*>(dst) = __builtin_shufflevector(__builtin_shufflevector(h[0], h[1], 0, 1, 2, 3), __builtin_shufflevector(h[2], h[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
            if (SPLIT)
                *reinterpret_cast<v8h*>(dst + BPLANE) = __builtin_shufflevector(__builtin_shufflevector(l[0], l[1], 0, 1, 2, 3), __builtin_shufflevector(l[2], l[3], 0, 1, 2, 3), 0, 1, 2, 3, 4, 5, 6, 7);
        }
    };
    struct Frags { v8h ah[TM], al[TM], bh[TN], bl[TN]; };
    auto load_frags = [&](Frags& f, const _Float16* buf, int sub) {
        const _Float16* sA = buf;
        const _Float16* sB = buf + BM * AS;
#pragma unroll
        for (int a = 0; a < TM; a++) {
            const _Float16* src = sA + ((wm * TM + a) * 32 + li) * AS + sub * 32 + lh * 8;
            f.ah[a] = *reinterpret_cast<const v8h*>(src);
            if (SPLIT) f.al[a] = *reinterpret_cast<const v8h*>(src + 16);
        }
#pragma unroll
        for (int b = 0; b < TN; b++) {
            const _Float16* src = sB + ((sub * 2 + lh) * NPART) * BPLANE + ((wn * TN + b) * 32 + li) * 8;
            f.bh[b] = *reinterpret_cast<const v8h*>(src);
            if (SPLIT) f.bl[b] = *reinterpret_cast<const v8h*>(src + BPLANE);
        }
    };
    // 16-row blocks of this wave's M range that hold real output channels (wave-uniform): blocks of pure channel padding
    // (O = 645 in 256-row tiles: 112 of 768 rows; O = 406: 96 of 512) issue no MFMAs
    const int act16 = min(max((p.O - o0 - wm * TM * 32 + 15) / 16, 0), 2 * TM);
    auto mfma_step = [&](const Frags& f) {
#pragma unroll
        for (int a = 0; a < TM; a++) {
            if (2 * a >= act16) continue;
#pragma unroll
            for (int b = 0; b < TN; b++) {
                if (SPLIT) {
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.al[a], f.bh[b], acc[a][b], 0, 0, 0);
                    acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[a], f.bl[b], acc[a][b], 0, 0, 0);
                }
                acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ah[a], f.bh[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    // M16: one stage (32 channels) = one K step.  Fragments of one 16-row block of A at a time (its 2 TN x 2 B fragments are
    // loaded once per stage and kept): 3 products per 16 x 16 block, the same order (lo*hi, hi*lo, hi*hi) as the 32-wide form.
    auto stage_m16 = [&](const _Float16* buf) {
        const _Float16* sA = buf;
        const _Float16* sB = buf + BM * AS;
        v8h bh[TN][2], bl[TN][2];
#pragma unroll
        for (int b = 0; b < TN; b++)
#pragma unroll
            for (int pb = 0; pb < 2; pb++) {
                const _Float16* src = sB + (lg * NPART) * BPLANE + ((wn * TN + b) * 32 + pb * 16 + l16) * 8;
                bh[b][pb] = *reinterpret_cast<const v8h*>(src);
                if (SPLIT) bl[b][pb] = *reinterpret_cast<const v8h*>(src + BPLANE);
            }
#pragma unroll
        for (int a = 0; a < TM; a++)
#pragma unroll
            for (int rb_ = 0; rb_ < 2; rb_++) {
                if (2 * a + rb_ >= act16) continue;
                const _Float16* src = sA + ((wm * TM + a) * 32 + rb_ * 16 + l16) * AS + (lg >> 1) * 32 + (lg & 1) * 8;
                const v8h ah = *reinterpret_cast<const v8h*>(src);
                v8h al;
                if (SPLIT) al = *reinterpret_cast<const v8h*>(src + 16);
#pragma unroll
                for (int b = 0; b < TN; b++)
#pragma unroll
                    for (int pb = 0; pb < 2; pb++) {
                        f32x4 c = {acc[a][b][(rb_ * 2 + pb) * 4], acc[a][b][(rb_ * 2 + pb) * 4 + 1], acc[a][b][(rb_ * 2 + pb) * 4 + 2], acc[a][b][(rb_ * 2 + pb) * 4 + 3]};
                        if (SPLIT) {
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[b][pb], c, 0, 0, 0);
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[b][pb], c, 0, 0, 0);
                        }
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[b][pb], c, 0, 0, 0);
#pragma unroll
                        for (int i = 0; i < 4; i++) acc[a][b][(rb_ * 2 + pb) * 4 + i] = c[i];
                    }
            }
    };

    fetch(chBegin);
    stage(smh + (NBUF == 2 ? (chBegin & 1) * BUF : 0), chBegin);
    __syncthreads();
    for (int ch = chBegin; ch < chEnd; ch++) {
        const _Float16* cur = smh + (NBUF == 2 ? (ch & 1) * BUF : 0);
        const bool more = ch + 1 < chEnd;
        if (more) fetch(ch + 1);
        if (M16) {
            stage_m16(cur);
        } else if (TM * TN <= 4) {
            Frags f0, f1;
            load_frags(f0, cur, 0);
            load_frags(f1, cur, 1);
            __builtin_amdgcn_sched_barrier(0);
            mfma_step(f0);
            mfma_step(f1);
            __builtin_amdgcn_sched_barrier(0);
        } else {                                            // eight accumulator blocks: one fragment set at a time
            Frags f;
            load_frags(f, cur, 0);
            mfma_step(f);
            load_frags(f, cur, 1);
            mfma_step(f);
        }
        if (NBUF == 2) {
            if (more) stage(smh + ((ch + 1) & 1) * BUF, ch + 1);   // the other buffer: its readers passed the previous barrier
            __syncthreads();
        } else {
            __syncthreads();                                       // every wave is done reading the image
            if (more) { stage(smh, ch + 1); __syncthreads(); }
        }
    }

    // (a K-split call has T = float: its partial images are fp32)
    T* outp = (p.kSplits > 1 ? (T*)p.partial + (size_t)ks * p.N * p.O * P : (T*)p.out) + (size_t)n * p.O * P;
    if (M16) {
        // accumulator register (rb * 2 + pb) * 4 + i of block (a, b): channel (wm TM + a) 32 + rb 16 + 4 lg + i, pixel (wn TN + b) 32 + pb 16 + l16
        const unsigned planeB16 = (unsigned)P * (unsigned)sizeof(T);
        const bool desc = (unsigned long long)(p.mTiles * BM + 32) * planeB16 < 0x7fffffffULL;
        const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)outp, (short)0, desc ? (int)((unsigned)p.O * planeB16) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dcoef + (size_t)n * p.O), (short)0, p.O * 4, 0x00020000);
#pragma unroll
        for (int a = 0; a < TM; a++)
#pragma unroll
            for (int rb_ = 0; rb_ < 2; rb_++) {
                const int oL = o0 + (wm * TM + a) * 32 + rb_ * 16 + 4 * lg;
                float d[4];
#pragma unroll
                for (int i = 0; i < 4; i++) d[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, (oL + i) * 4, 0, 0));
#pragma unroll
                for (int b = 0; b < TN; b++)
#pragma unroll
                    for (int pb = 0; pb < 2; pb++) {
                        const int px = p0 + (wn * TN + b) * 32 + pb * 16 + l16;
                        if (desc) {
                            const unsigned base = px < P ? (unsigned)oL * planeB16 + (unsigned)px * (unsigned)sizeof(T) : 0x80000000u;
#pragma unroll
                            for (int i = 0; i < 4; i++) {
                                const float v = acc[a][b][(rb_ * 2 + pb) * 4 + i] * d[i];
                                if (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orr, (int)(base + (unsigned)i * planeB16), 0, 0);
                                else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), orr, (int)(base + (unsigned)i * planeB16), 0, 0);
                            }
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; i++)
                                if (oL + i < p.O && px < P) io<T>::st(outp + (size_t)(oL + i) * P + px, acc[a][b][(rb_ * 2 + pb) * 4 + i] * d[i]);
                        }
                    }
            }
        return;
    }
    const unsigned planeB = (unsigned)P * (unsigned)sizeof(T);
    // every offset a lane can form -- channels of the padded last M tile included -- must stay below 2^31, so that nothing
    // wraps around into the tensor: (padded O + one wave block) * plane < 2^31
    if ((unsigned long long)(p.mTiles * BM + 32) * planeB < 0x7fffffffULL) {
        // descriptor stores, as in the 3x3 kernel: the range check drops channels beyond O and (offsets from 2^31) pixels
        // beyond the plane
        const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc((void*)outp, (short)0, (int)((unsigned)p.O * planeB), 0x00020000);
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dcoef + (size_t)n * p.O), (short)0, p.O * 4, 0x00020000);
#pragma unroll
        for (int a = 0; a < TM; a++) {
            const int oL = o0 + (wm * TM + a) * 32 + 4 * lh;
            float d[16];
#pragma unroll
            for (int r = 0; r < 16; r++)
                d[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, (oL + (r & 3) + 8 * (r >> 2)) * 4, 0, 0));
#pragma unroll
            for (int b = 0; b < TN; b++) {
                const int px = p0 + (wn * TN + b) * 32 + li;
                const unsigned base = px < P ? (unsigned)oL * planeB + (unsigned)px * (unsigned)sizeof(T) : 0x80000000u;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const unsigned off = base + (unsigned)((r & 3) + 8 * (r >> 2)) * planeB;
                    const float v = acc[a][b][r] * d[r];
                    if (sizeof(T) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), orr, (int)off, 0, 0);
                    else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), orr, (int)off, 0, 0);
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
            const float d = p.dcoef[(size_t)n * p.O + o];
#pragma unroll
            for (int b = 0; b < TN; b++) {
                const int px = p0 + (wn * TN + b) * 32 + li;
                if (px < P) io<T>::st(outp + (size_t)o * P + px, acc[a][b][r] * d);
            }
        }
}

// ---------------------------------------------------------------------------
// 1x1 kernels with very few output channels (the ToRGB layer: 32 -> 3): HBM-bound, no matrix cores.
// out[n,o,p] = dcoef[n,o] * sum_i wn[o,i] * sIn[n,i] * x[n,i,p]; a thread owns 4 consecutive pixels (16-byte loads
// per channel plane), the modulated weights sit in LDS.
// clamp of the fused ToRGB output stage: a NaN stays a NaN, as through the reference's clamp (filtered_lrelu.cu:412-419 /
// bias_act.cu: `if (fabsf(v) > clamp) v = copysign(clamp, v)`); fminf / fmaxf alone return the finite operand
static __device__ __forceinline__ float ep_clamp(float v, float c) { return v != v ? v : fminf(fmaxf(v, -c), c); }

template <typename T, int OMAX>
__global__ void __launch_bounds__(256)
modconv_1x1_small_kernel(ConvParams p, int vec) {
    extern __shared__ float sw[];                       // [I][OMAX]
    const int n = blockIdx.y;
    const int HW = p.H * p.W;
    for (int j = threadIdx.x; j < p.I * OMAX; j += 256) {
        const int i = j / OMAX, o = j % OMAX;
        float v = 0.f;
        if (o < p.O) v = p.wp[((size_t)o * p.nch + i / 16) * 16 + (i % 16)] * p.sIn[(size_t)n * p.I + i] * (p.dcoef ? p.dcoef[(size_t)n * p.O + o] : 1.f);
        sw[j] = v;
    }
    __syncthreads();
    const T* xin = (const T*)p.x + (size_t)n * p.I * HW;
    T* outp = (T*)p.out + (size_t)n * p.O * HW;
    const int gs = gridDim.x * 256;
    if (vec) {
        for (int q = blockIdx.x * 256 + threadIdx.x; q * 4 < HW; q += gs) {
            float acc[OMAX][4];
#pragma unroll
            for (int o = 0; o < OMAX; o++) { acc[o][0] = acc[o][1] = acc[o][2] = acc[o][3] = 0.f; }
#pragma unroll 4
            for (int i = 0; i < p.I; i++) {
                float xv[4];
                if (sizeof(T) == 4) { const f32x4 t = *reinterpret_cast<const f32x4*>(xin + (size_t)i * HW + q * 4); xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; }
                else { for (int e = 0; e < 4; e++) xv[e] = io<T>::ld(xin + (size_t)i * HW + q * 4 + e); }
#pragma unroll
                for (int o = 0; o < OMAX; o++) { const float w = sw[i * OMAX + o];
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[o][e] = fmaf(w, xv[e], acc[o][e]); }
            }
            for (int o = 0; o < p.O && o < OMAX; o++) {
                if (p.epBias) {
                    const float bo = p.epBias[o];
#pragma unroll
                    for (int e = 0; e < 4; e++) acc[o][e] = ep_clamp(acc[o][e] + bo, p.epClamp) * p.epScale;
                }
                if (sizeof(T) == 4) *reinterpret_cast<f32x4*>(outp + (size_t)o * HW + q * 4) = (f32x4){acc[o][0], acc[o][1], acc[o][2], acc[o][3]};
                else for (int e = 0; e < 4; e++) io<T>::st(outp + (size_t)o * HW + q * 4 + e, acc[o][e]);
            }
        }
    } else {
        for (int q = blockIdx.x * 256 + threadIdx.x; q < HW; q += gs) {
            float acc[OMAX];
#pragma unroll
            for (int o = 0; o < OMAX; o++) acc[o] = 0.f;
            for (int i = 0; i < p.I; i++) {
                const float xv = io<T>::ld(xin + (size_t)i * HW + q);
#pragma unroll
                for (int o = 0; o < OMAX; o++) acc[o] = fmaf(sw[i * OMAX + o], xv, acc[o]);
            }
            for (int o = 0; o < p.O && o < OMAX; o++) {
                if (p.epBias) acc[o] = ep_clamp(acc[o] + p.epBias[o], p.epClamp) * p.epScale;
                io<T>::st(outp + (size_t)o * HW + q, acc[o]);
            }
        }
    }
}

template <typename T>
static int launch_conv_1x1_small(const sg3_modconv_params& q, hipStream_t st) {
    ConvParams p;
    p.x = q.x; p.wp = q.wPacked; p.sIn = q.sIn; p.dcoef = q.dcoef; p.out = q.out;
    p.N = q.N; p.I = q.I; p.O = q.O; p.H = q.H; p.W = q.W; p.pad = 0; p.outH = q.H; p.outW = q.W;
    p.nch = ceil_div(q.I, ConvK<1>::KC);
    p.xTiles = p.yTiles = p.mTiles = 1; p.totalBlocks = 0;
    p.epBias = q.epilogueBias;
    p.epClamp = q.epilogueClamp >= 0.f ? q.epilogueClamp : INFINITY;
    p.epScale = q.epilogueScale != 0.f ? q.epilogueScale : 1.f;
    const int HW = q.H * q.W;
    const int vec = (HW % 4 == 0) && (((size_t)q.x | (size_t)q.out) % 16 == 0) ? 1 : 0;
    int bx = ceil_div(vec ? HW / 4 : HW, 256);
    if (bx > 2048) bx = 2048;
    hipLaunchKernelGGL((modconv_1x1_small_kernel<T, 4>), dim3(bx, q.N), dim3(256), (size_t)q.I * 4 * sizeof(float), st, p, vec);
    SG3_LAUNCH_CHECK("modconv_1x1_small_kernel");
    return SG3_OK;
}

// ---------------------------------------------------------------------------
// prep A: one workgroup per output channel: normalise the filter, pack it, and emit wsq[o][i] = sum_taps wn^2
static __device__ __forceinline__ void prep_w_body(const sg3_modconv_prep_params& p, int kc, int nch, int o, float* red) {
    const int taps = p.k * p.k, len = p.I * taps;
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
    if (p.precision == SG3_CONV_F16X3_F23 || p.precision == SG3_CONV_F16_F23) {
        f23_pack_row(w, scale, o, p.O, p.I, nch, reinterpret_cast<_Float16*>(p.wPacked), p.precision == SG3_CONV_F16X3_F23);      // transform-domain layout (sg3_modconv_f23.h)
    } else if (p.precision != SG3_CONV_FP32) {
        // [o][chunk][tap][hi|lo][16] halfs
        _Float16* dsth = reinterpret_cast<_Float16*>(p.wPacked) + (size_t)o * nch * taps * 32;
        for (int j = threadIdx.x; j < nch * taps * 16; j += 256) {
            const int c = j % 16, t = (j / 16) % taps, ch = j / (16 * taps);
            const int i = ch * 16 + c;
            const float v = i < p.I ? w[i * taps + t] * scale : 0.f;
            const _Float16 h = (_Float16)v;
            _Float16* d = dsth + ((size_t)ch * taps + t) * 32 + c;
            d[0] = h;
            d[16] = (_Float16)(v - (float)h);
        }
    } else {
        float* dst = p.wPacked + (size_t)o * nch * taps * kc;
        for (int j = threadIdx.x; j < nch * taps * kc; j += 256) {
            const int c = j % kc, t = (j / kc) % taps, ch = j / (kc * taps);
            const int i = ch * kc + c;
            dst[j] = i < p.I ? w[i * taps + t] * scale : 0.f;
        }
    }
    for (int i = threadIdx.x; i < p.I; i += 256) {
        float s = 0.f;
        for (int t = 0; t < taps; t++) { float v = w[i * taps + t] * scale; s += v * v; }
        p.wsq[(size_t)o * p.I + i] = s;
    }
}
__global__ void __launch_bounds__(256)
modconv_prep_w_kernel(sg3_modconv_prep_params p, int kc, int nch) {
    __shared__ float red[256];
    prep_w_body(p, kc, nch, blockIdx.x, red);
}

// The prep work of several layers in one launch each (sg3_modulated_conv2d_prep_batch): a single layer's prep kernels have a
// few hundred small workgroups and are latency-bound (13 + 20 us); fifteen layers' worth run side by side in about the time of one.
#define SG3_PREP_BATCH_MAX 16
struct PrepBatch {
    sg3_modconv_prep_params p[SG3_PREP_BATCH_MAX];
    int first[SG3_PREP_BATCH_MAX + 1];     // first workgroup of each entry (prefix sums)
    int kc[SG3_PREP_BATCH_MAX], nch[SG3_PREP_BATCH_MAX], gy[SG3_PREP_BATCH_MAX];
    int count;
};
static __device__ __forceinline__ int batch_entry(const PrepBatch& b, int block) {
    int l = 0;
    while (l + 1 < b.count && block >= b.first[l + 1]) l++;
    return l;
}
__global__ void __launch_bounds__(256)
modconv_prep_w_batch_kernel(PrepBatch b) {
    __shared__ float red[256];
    const int l = batch_entry(b, (int)blockIdx.x);
    prep_w_body(b.p[l], b.kc[l], b.nch[l], (int)blockIdx.x - b.first[l], red);
}

// prep B: gridDim.y workgroups per sample: normalise styles over the WHOLE batch, write sIn (first workgroup) and this
// workgroup's share of dcoef (one output channel per wave at a time: coalesced rows of wsq, shuffle reduction)
static __device__ __forceinline__ void prep_s_body(const sg3_modconv_prep_params& p, int n, int by, int gy, float* red, float* s2) {
    float scale = 1.f;
    if (p.demodulate) {
        float s = 0.f;
        for (int j = threadIdx.x; j < p.N * p.I; j += 256) { float v = p.s[j]; s += v * v; }
        red[threadIdx.x] = s;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st]; __syncthreads(); }
        scale = rsqrtf(red[0] / (float)(p.N * p.I));
    }
    float smax = 0.f;
    for (int i = threadIdx.x; i < p.I; i += 256) {
        const float sn = p.s[(size_t)n * p.I + i] * scale;
        float g = 1.f;
        if (p.inputGainMode == 1) g = p.inputGain[0];
        else if (p.inputGainMode == 2) g = p.inputGain[i];
        else if (p.inputGainMode == 3) g = p.inputGain[(size_t)n * p.I + i];
        const float v = sn * g;
        s2[i] = sn * sn;
        s2[p.I + i] = v;
        smax = fmaxf(smax, fabsf(v));
    }
    // f16x3 / f16: per-sample power-of-two scale bringing max |x * sIn| just below 2^15
    float down = 1.f, up = 1.f;
    if (p.precision != SG3_CONV_FP32) {
        __syncthreads();
        red[threadIdx.x] = smax;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) { if (threadIdx.x < st) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + st]); __syncthreads(); }
        const float peak = red[0] * (p.xBoundDev ? p.xBoundDev[0] : p.xBound);
        int e = 0;
        if (peak > 0.f && peak < 3.0e38f) e = (int)ceilf(log2f(peak / 32768.f));    // either direction: tiny operands (gradients) are scaled up
        if (p.precision == SG3_CONV_F16X3_F23 || p.precision == SG3_CONV_F16_F23) e += 1;                               // the input transform forms sums of two samples
        down = ldexpf(1.f, -e); up = ldexpf(1.f, e);
    }
    __syncthreads();
    if (by == 0)
        for (int i = threadIdx.x; i < p.I; i += 256) p.sIn[(size_t)n * p.I + i] = s2[p.I + i] * down;
    if (p.demodulate) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        for (int o = by * 4 + wave; o < p.O; o += gy * 4) {
            const float* wq = p.wsq + (size_t)o * p.I;
            float s = 0.f;
            for (int i = lane; i < p.I; i += 64) s += wq[i] * s2[i];
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m);
            if (lane == 0) p.dcoef[(size_t)n * p.O + o] = rsqrtf(s + 1e-8f) * up;
        }
    } else if (p.precision != SG3_CONV_FP32) {
        for (int o = by * 256 + threadIdx.x; o < p.O; o += gy * 256) p.dcoef[(size_t)n * p.O + o] = up;
    }
}
__global__ void __launch_bounds__(256)
modconv_prep_s_kernel(sg3_modconv_prep_params p) {
    __shared__ float red[256];
    extern __shared__ float s2[];                         // [I] squared normalised styles | [I] scaled styles of this sample
    prep_s_body(p, blockIdx.x, blockIdx.y, gridDim.y, red, s2);
}
__global__ void __launch_bounds__(256)
modconv_prep_s_batch_kernel(PrepBatch b) {
    __shared__ float red[256];
    extern __shared__ float s2[];
    const int l = batch_entry(b, (int)blockIdx.x);
    const int local = (int)blockIdx.x - b.first[l], gy = b.gy[l];
    prep_s_body(b.p[l], local / gy, local % gy, gy, red, s2);
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

template <typename T, int WM, int WN, int TN, bool SPLIT, bool PACK>
static int launch_conv_f16x3(const sg3_modconv_params& q, hipStream_t st) {
    constexpr int BM = WM * 32, ROWS = WN * TN;
    constexpr int PH = ROWS + 2, PW = 34;
    constexpr size_t ldsBytes = ((size_t)BM * (9 * 32 + 8) + (SPLIT ? 4 : 2) * (size_t)PH * PW * 8) * sizeof(_Float16);
    ConvParams p;
    p.x = q.x; p.wp = q.wPacked; p.sIn = q.sIn; p.dcoef = q.dcoef; p.out = q.out;
    p.N = q.N; p.I = q.I; p.O = q.O; p.H = q.H; p.W = q.W; p.pad = q.pad;
    p.outH = q.H + 2 * q.pad - 2; p.outW = q.W + 2 * q.pad - 2;
    p.nch = ceil_div(q.I, 16);
    p.xTiles = ceil_div(p.outW, 32); p.yTiles = ceil_div(p.outH, ROWS); p.mTiles = ceil_div(q.O, BM);
    const long long total = (long long)p.xTiles * p.yTiles * p.mTiles * q.N;
    if (total > 0x7fffffffLL) { set_error("modulated_conv2d: grid too large"); return SG3_BAD_ARG; }
    p.totalBlocks = (int)total;
    p.outPitch = q.outRowStride > 0 ? q.outRowStride : p.outW;
    p.tailPack = PACK ? 1 : 0;
    auto kern = modconv_f16x3_kernel<T, WM, WN, TN, SPLIT, PACK>;
    if (ldsBytes > 64 * 1024)
        SG3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)total), dim3(256), ldsBytes, st, p);
    SG3_LAUNCH_CHECK("modconv_f16x3_kernel");
    return SG3_OK;
}

// SG3_CONV3_ROWS=1 in the environment keeps the narrow 3x3 layers on the row-tile kernel (A/B timing; read once)
static bool conv3_use_flat() {
    static const bool v = [] { const char* e = getenv("SG3_CONV3_ROWS"); return !(e && e[0] == '1'); }();
    return v;
}

// out[n][o][f] = d[n][o] * sum_ks partial[ks][n][o][f] (ks in order: reproducible; d = 1 when dcoef is null: the 1x1 kernel applies it
// before storing its partial image), the second half of a K-split convolution
template <typename T>
__global__ void __launch_bounds__(256)
modconv_split_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ dcoef, T* __restrict__ out, int kSplits, long long planes, int P) {
    const long long total = planes * P;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        float v = 0.f;
        for (int k = 0; k < kSplits; k++) v += partial[(size_t)k * total + idx];
        out[idx] = (T)(dcoef ? v * dcoef[idx / P] : v);
    }
}

// CUs of the current device, read once per device
static int conv_cu_count() {
    static int cus[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cus[dev] == 0) {
        int n = 0;
        cus[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return cus[dev];
}

// K splits of a flat-kernel call with `tiles` workgroups and `nch` K chunks: none when the grid gives every CU a workgroup; else up to
// four, at least four chunks each
static int flat_forced_splits() {
    static const int forced = [] { const char* e = getenv("SG3_FLAT_SPLITS"); return e ? atoi(e) : 0; }();      // A/B timing: 1 = never split, 2 .. 4 = always
    return forced;
}
static int flat_k_splits(long long tiles, int nch) {
    const int forced = flat_forced_splits();
    if (forced >= 1) return std::min(forced, std::min(4, nch));
    const int cus = conv_cu_count();
    if (tiles >= cus) return 1;
    const int s = (int)std::min<long long>(std::min<long long>(4, (2LL * cus) / tiles), nch / 4);
    return s >= 2 ? s : 1;
}

// does the patch of every RUN-pixel piece of an outH x outW plane fit FLAT_NPIX pixels?  rows touched <= (RUN - 2) / outW + 2
static bool flat_fits(int outW, int run) { return ((run - 2) / outW + 2 + 2) * (outW + 2) <= FLAT_NPIX; }

template <typename T, int TN, bool SPLIT>
static int launch_conv_flat(const sg3_modconv_params& q, hipStream_t st) {
    constexpr int RUN = 2 * TN * 32;
    constexpr size_t ldsBytes = ((size_t)64 * (9 * 32 + 8) + (SPLIT ? 4 : 2) * (size_t)FLAT_NPIX * 8) * sizeof(_Float16);
    static_assert(ldsBytes <= 80 * 1024, "two workgroups per CU");
    ConvParams p;
    p.x = q.x; p.wp = q.wPacked; p.sIn = q.sIn; p.dcoef = q.dcoef; p.out = q.out;
    p.N = q.N; p.I = q.I; p.O = q.O; p.H = q.H; p.W = q.W; p.pad = q.pad;
    p.outH = q.H + 2 * q.pad - 2; p.outW = q.W + 2 * q.pad - 2;
    p.nch = ceil_div(q.I, 16);
    p.xTiles = ceil_div(p.outH * p.outW, RUN); p.yTiles = 1; p.mTiles = ceil_div(q.O, 64);
    const long long total = (long long)p.xTiles * p.mTiles * q.N;
    if (total > 0x7fffffffLL) { set_error("modulated_conv2d: grid too large"); return SG3_BAD_ARG; }
    p.outPitch = p.outW; p.tailPack = 0;
    p.kSplits = 1; p.partial = nullptr;
    const long long planes = (long long)q.N * q.O, P = (long long)p.outH * p.outW;
    if (q.splitScratch && q.dcoef) {
        const int ksp = flat_k_splits(total, p.nch);
        if (ksp > 1 && (long long)ksp * planes * P <= q.splitScratchFloats && (long long)ksp * total <= 0x7fffffffLL) { p.kSplits = ksp; p.partial = q.splitScratch; }
    }
    p.totalBlocks = (int)(total * p.kSplits);
    auto kern = modconv_flat_kernel<T, TN, SPLIT>;
    if (ldsBytes > 64 * 1024)
        SG3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)p.totalBlocks), dim3(256), ldsBytes, st, p);
    SG3_LAUNCH_CHECK("modconv_flat_kernel");
    if (p.kSplits > 1) {
        const long long elems = planes * P;
        const unsigned blocks = (unsigned)std::min<long long>((elems + 255) / 256, 4096);
        hipLaunchKernelGGL((modconv_split_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, (const float*)p.partial, q.dcoef, (T*)q.out, p.kSplits, planes, (int)P);
        SG3_LAUNCH_CHECK("modconv_split_reduce_kernel");
    }
    return SG3_OK;
}

template <typename T, bool SPLIT>
static int dispatch_conv_f16x3(const sg3_modconv_params& q, hipStream_t st) {
    // Two row-streaming tiles, two workgroups per CU each.  The 64-channel tile stages the patch once for twice the
    // channels; the 32-channel tile pads the channel count less.  Measured at FFHQ-1024 (batch 8): O = 81 (96 vs 128
    // padded rows) 1.76 vs 2.10 ms, O = 203 (224 vs 256) 2.40 vs 2.54 ms, O = 323 (352 vs 384) 1.86 vs 1.77 ms: the
    // small tile wins when it saves at least ~10 % of the rows.
    const int O = q.O;
    const int t32 = ceil_div(O, 32) * 32, t64 = ceil_div(O, 64) * 64;
    // last K chunk with 1..4 channels (and at least one full chunk before it): the kernel packs its taps (PACK)
    const bool pack = SG3_TAILPACK && q.I > 16 && q.I % 16 >= 1 && q.I % 16 <= 4;
    // The plain fp16 form (one MFMA per K step) reads 0.75 LDS fragments per MFMA with four rows per wave -- LDS-bandwidth bound,
    // where the split form (0.5 per MFMA, three MFMAs per fragment pair) is not: its waves take taller stacks of rows (six: 0.61; five in the 32-channel tile: 0.67; eight rows spill).
    // SG3_CONV_F16_ROWS4=1 keeps four rows (A/B timing).
    static const bool tallF16 = [] { const char* e = getenv("SG3_CONV_F16_ROWS4"); return !(e && e[0] == '1'); }();
    const bool tall = !SPLIT && tallF16 && q.H + 2 * q.pad - 2 >= 128;
    if (O <= 32 || t32 * 10 <= t64 * 9) {                                                          //  32 x (16 rows x 32)
        if constexpr (!SPLIT) {
            if (tall) return pack ? launch_conv_f16x3<T, 1, 4, 5, SPLIT, true>(q, st) : launch_conv_f16x3<T, 1, 4, 5, SPLIT, false>(q, st);
        }
        return pack ? launch_conv_f16x3<T, 1, 4, 4, SPLIT, true>(q, st) : launch_conv_f16x3<T, 1, 4, 4, SPLIT, false>(q, st);
    }
    if constexpr (!SPLIT) {
        if (tall) return pack ? launch_conv_f16x3<T, 2, 2, 6, SPLIT, true>(q, st) : launch_conv_f16x3<T, 2, 2, 6, SPLIT, false>(q, st);
    }
    // Narrow outputs with 64-channel tiles: runs of the flattened plane instead of 32-column row pieces (modconv_flat_kernel) when
    // that takes fewer rounds x MFMA blocks per wave than the best row tile.  SG3_CONV3_ROWS=1 keeps the row kernel.
    if (!pack && (q.outRowStride == 0 || q.outRowStride == q.W + 2 * q.pad - 2) && conv3_use_flat()) {
        const int outH = q.H + 2 * q.pad - 2, outW = q.W + 2 * q.pad - 2;
        const long long perM = (long long)q.N * ceil_div(O, 64);
        // time ~ rounds of 512 resident workgroups x (blocks per wave + staging); large grids are not quantised
        auto cost = [](long long wgs, int tn) { return (wgs <= 2048 ? (double)ceil_div64(wgs, 512) : wgs / 512.0) * (tn + 0.5); };
        const long long perRow = perM * ceil_div(outW, 32);
        const double rowCost = std::min(cost(perRow * ceil_div(outH, 8), 4), cost(perRow * ceil_div(outH, 10), 5));
        int best = 0; double bestCost = rowCost * 0.9;                       // the row kernel reads less LDS per MFMA: flat must save 10 %
        for (int tn = 4; tn >= 2; tn--) {
            if (!flat_fits(outW, 64 * tn)) continue;
            const double c = cost(perM * ceil_div(outH * outW, 64 * tn), tn);
            if (c < bestCost) { bestCost = c; best = tn; }
        }
        if (best == 4) return launch_conv_flat<T, 4, SPLIT>(q, st);
        if (best == 3) return launch_conv_flat<T, 3, SPLIT>(q, st);
        if (best == 2) return launch_conv_flat<T, 2, SPLIT>(q, st);
    }
    if (!pack) {
        // Small grids (the 36^2 .. 52^2 layers): 512 workgroups are resident at once, so the time goes with the number of
        // ROUNDS times the rows a workgroup computes.  Ten-row tiles turn the 640 workgroups of a 38-row output (8 images
        // x 8 channel tiles x 2 x 5) into exactly 512: one round of 5 rows per wave instead of two rounds of 4.
        const int outH = q.H + 2 * q.pad - 2, outW = q.W + 2 * q.pad - 2;
        const long long per = (long long)q.N * ceil_div(O, 64) * ceil_div(outW, 32);
        const long long wg8 = per * ceil_div(outH, 8), wg10 = per * ceil_div(outH, 10);
        if (wg8 <= 2048 && ceil_div64(wg10, 512) * 5 < ceil_div64(wg8, 512) * 4)
            return launch_conv_f16x3<T, 2, 2, 5, SPLIT, false>(q, st);                             //  64 x (10 rows x 32)
    }
    return pack ? launch_conv_f16x3<T, 2, 2, 4, SPLIT, true>(q, st) : launch_conv_f16x3<T, 2, 2, 4, SPLIT, false>(q, st);   //  64 x (8 rows x 32)
}

// SG3_CONV1_MFMA32=1 in the environment selects the 32x32x16 form of the 1x1 kernels (A/B timing; read once)
static bool conv1_use_m16() {
    static const bool v = [] { const char* e = getenv("SG3_CONV1_MFMA32"); return !(e && e[0] == '1'); }();
    return v;
}

template <typename T, int WM, int WN, int TM, int TN, bool SPLIT, int NBUF = 2>
static int launch_conv1_f16x3(const sg3_modconv_params& q, hipStream_t st) {
    constexpr int BM = WM * TM * 32, ROWS = WN * TN;
    constexpr size_t ldsBytes = NBUF * ((size_t)BM * (2 * 32 + 8) + (SPLIT ? 8 : 4) * (size_t)ROWS * 32 * 8) * sizeof(_Float16);
    ConvParams p;
    p.x = q.x; p.wp = q.wPacked; p.sIn = q.sIn; p.dcoef = q.dcoef; p.out = q.out;
    p.N = q.N; p.I = q.I; p.O = q.O; p.H = q.H; p.W = q.W; p.pad = 0;
    p.outH = q.H; p.outW = q.W;
    p.nch = f16x3_chunks(q.I, 1) / 2;                  // stages of 32 channels
    p.xTiles = ceil_div(q.H * q.W, ROWS * 32); p.yTiles = 1; p.mTiles = ceil_div(q.O, BM);     // flat 256-pixel tiles
    const long long total = (long long)p.xTiles * p.yTiles * p.mTiles * q.N;
    if (total > 0x7fffffffLL) { set_error("modulated_conv2d: grid too large"); return SG3_BAD_ARG; }
    p.kSplits = 1; p.partial = nullptr;
    if (q.splitScratch && std::is_same<T, float>::value) {
        // this kernel's workgroups are eight waves with two resident per CU: measured neutral from ~100 tiles up (R-1024 batch 4: 15.1 vs
        // 15.0 ms), +1.7 % at 48, +5.6 % at 24 -- so it splits below a quarter of a tile per CU
        const int ksp = flat_k_splits(4 * total, p.nch);
        const long long elems = (long long)q.N * q.O * q.H * q.W;
        if (ksp > 1 && (long long)ksp * elems <= q.splitScratchFloats && (long long)ksp * total <= 0x7fffffffLL) { p.kSplits = ksp; p.partial = q.splitScratch; }
    }
    p.totalBlocks = (int)(total * p.kSplits);
    // the 16x16x32 form for the compute-bound tiles (R-1024, batch 8: 28.4 vs 32.2 ms over the 1024 .. 406-channel layers); the
    // thin HBM-bound layers keep 32x32x16: its stores are 128-byte row segments, the 16-wide blocks' 64-byte ones cost them 5 %
    const bool m16 = NBUF == 2 && conv1_use_m16();
    auto kern = m16 ? modconv1_f16x3_kernel<T, WM, WN, TM, TN, SPLIT, NBUF, true> : modconv1_f16x3_kernel<T, WM, WN, TM, TN, SPLIT, NBUF, false>;
    if (ldsBytes > 64 * 1024)
        SG3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
    hipLaunchKernelGGL(kern, dim3((unsigned)p.totalBlocks), dim3(512), ldsBytes, st, p);
    SG3_LAUNCH_CHECK("modconv1_f16x3_kernel");
    if (p.kSplits > 1) {
        const long long planes = (long long)q.N * q.O, P = (long long)q.H * q.W, elems = planes * P;
        const unsigned blocks = (unsigned)std::min<long long>((elems + 255) / 256, 4096);
        hipLaunchKernelGGL((modconv_split_reduce_kernel<T>), dim3(blocks), dim3(256), 0, st, (const float*)p.partial, (const float*)nullptr, (T*)q.out, p.kSplits, planes, (int)P);
        SG3_LAUNCH_CHECK("modconv_split_reduce_kernel");
    }
    return SG3_OK;
}

template <typename T, bool SPLIT>
static int dispatch_conv1_f16x3(const sg3_modconv_params& q, hipStream_t st) {
    // every staged input element is used once per output-channel tile, so the tile is as tall as the channel padding
    // allows: 256 rows (1024 -> 1024 @ 148^2 x 4: 0.88 ms against 1.05 ms with 128 rows and 1.25 ms with 64)
    // Few K stages (I <= 256: the 532^2 and 1044^2 layers of config R): HBM-bound, and ONE resident workgroup (100 KB of
    // double-buffered LDS) leaves the CU waiting on memory at the start and end of every tile.  These layers take the
    // 64-row tile with a single LDS image (42 KB, 110 registers): two workgroups per CU.  Measured at R-1024, batch 8:
    // L10 256->161 1232 -> 1099 us, L11 161->102 2600 -> 2358, L12 102->64 1373 -> 1103, L13 64->64 945 -> 808 (5.5 TB/s).
    // The 128-row tile needs 174 registers and spills under the two-workgroup bound.
    const bool thin = q.I <= 256;
    if (thin) return launch_conv1_f16x3<T, 1, 8, 2, 1, SPLIT, 1>(q, st);           //  64 x 256 pixels, two workgroups per CU
    if (q.O <= 64) return launch_conv1_f16x3<T, 1, 8, 2, 1, SPLIT>(q, st);         //  64 x 256 pixels
    const int t128 = ceil_div(q.O, 128) * 128, t256 = ceil_div(q.O, 256) * 256;
    if (t256 <= t128) return launch_conv1_f16x3<T, 2, 4, 4, 2, SPLIT>(q, st);     // 256 x 256 pixels
    return launch_conv1_f16x3<T, 2, 4, 2, 2, SPLIT>(q, st);                       // 128 x 256 pixels
}

} // namespace sg3

extern "C" {

int64_t sg3_modconv_packed_floats(int O, int I, int k, int precision) {
    if (O <= 0 || I <= 0 || (k != 1 && k != 3)) return 0;
    if (precision == SG3_CONV_F16X3_F23 || precision == SG3_CONV_F16_F23) return k == 3 ? sg3::f23_packed_floats(O, I, precision == SG3_CONV_F16X3_F23) : 0;
    if (precision == SG3_CONV_F16X3 || precision == SG3_CONV_F16) {
        return (int64_t)O * sg3::f16x3_chunks(I, k) * (k * k) * 16;   // 32 halfs = 16 floats per (chunk, tap)
    }
    const int kc = sg3::packed_kc(k);
    return (int64_t)O * sg3::ceil_div(I, kc) * (k * k) * kc;
}

static int prep_validate(const sg3_modconv_prep_params* p) {
    using namespace sg3;
    SG3_REQUIRE(p && p->w && p->s && p->wPacked && p->wsq && p->sIn, "modulated_conv2d_prep: null tensor");
    SG3_REQUIRE(p->N > 0 && p->I > 0 && p->O > 0, "modulated_conv2d_prep: empty tensor");
    SG3_REQUIRE(p->k == 1 || p->k == 3, "modulated_conv2d_prep: kernel size must be 1 or 3");
    SG3_REQUIRE(!p->demodulate || p->dcoef, "modulated_conv2d_prep: dcoef missing");
    SG3_REQUIRE(p->inputGainMode >= 0 && p->inputGainMode <= 3, "modulated_conv2d_prep: bad inputGainMode");
    SG3_REQUIRE(p->inputGainMode == 0 || p->inputGain, "modulated_conv2d_prep: inputGain missing");
    SG3_REQUIRE((size_t)p->I * 2 * sizeof(float) <= 48 * 1024, "modulated_conv2d_prep: too many input channels");
    SG3_REQUIRE(p->precision == SG3_CONV_FP32 || p->precision == SG3_CONV_F16X3 || p->precision == SG3_CONV_F16 || p->precision == SG3_CONV_F16X3_F23 ||
                p->precision == SG3_CONV_F16_F23, "modulated_conv2d_prep: bad precision");
    SG3_REQUIRE((p->precision != SG3_CONV_F16X3_F23 && p->precision != SG3_CONV_F16_F23) || p->k == 3, "modulated_conv2d_prep: the transform-domain forms are for 3x3 kernels");
    if (p->precision != SG3_CONV_FP32) {
        SG3_REQUIRE((p->xBound > 0.f || p->xBoundDev) && p->dcoef, "modulated_conv2d_prep: f16x3 needs xBound > 0 and a dcoef buffer");
    }
    return SG3_OK;
}

int64_t sg3_modconv_split_scratch_floats(const sg3_modconv_params* p) {
    using namespace sg3;
    // an upper bound for the one form that splits K over workgroups (3x3, split-precision / fp16 direct kernels on narrow maps with a
    // grid smaller than the chip): 4 partial images; 0 when the call cannot take it
    if (!p || (p->precision != SG3_CONV_F16X3 && p->precision != SG3_CONV_F16)) return 0;
    if (p->k == 1) {
        // the 1x1 GEMM kernel (config R): fp32 tensors, 256-pixel tiles x up to 256 rows; its stages are 32 channels
        if (p->dtype != SG3_F32 || p->pad != 0 || p->O <= 4) return 0;
        const long long P1 = (long long)p->H * p->W;
        const long long tiles1 = (long long)p->N * ceil_div(p->O, 256) * ceil_div((int)std::min<long long>(P1, 0x7fffffff), 256);
        if (flat_forced_splits() < 2 && (4 * tiles1 >= conv_cu_count() || ceil_div(p->I, 32) < 8)) return 0;
        return 4LL * p->N * p->O * P1;
    }
    if (p->k != 3 || !p->dcoef) return 0;
    const long long outH = p->H + 2 * p->pad - 2, outW = p->W + 2 * p->pad - 2;
    if (outH <= 0 || outW <= 0 || outW > 128) return 0;
    const long long tiles = (long long)p->N * ceil_div(p->O, 64) * ceil_div((int)(outH * outW), 128);      // the smallest flat tile
    if (flat_forced_splits() < 2 && (tiles >= conv_cu_count() || ceil_div(p->I, 16) < 8)) return 0;
    return 4LL * p->N * p->O * outH * outW;
}

int sg3_modconv_f23_supported(int dtype, int I, int O, int H, int W, int k, int pad, int outRowStride) {
    return sg3::f23_supported(dtype, I, O, H, W, k, pad, outRowStride) ? 1 : 0;
}

int sg3_modconv_f23_force_rows(int rows) { return sg3::f23_force_rows(rows); }

int sg3_modulated_conv2d_prep_batch(const sg3_modconv_prep_params* list, int count, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(list && count > 0, "modulated_conv2d_prep_batch: empty list");
    hipStream_t st = (hipStream_t)stream;
    for (int base = 0; base < count; base += SG3_PREP_BATCH_MAX) {
        PrepBatch bw, bs;
        const int m = std::min(SG3_PREP_BATCH_MAX, count - base);
        size_t lds = 0;
        bw.count = bs.count = m;
        bw.first[0] = bs.first[0] = 0;
        for (int j = 0; j < m; j++) {
            const sg3_modconv_prep_params* p = list + base + j;
            const int rc = prep_validate(p);
            if (rc != SG3_OK) return rc;
            const int kc = packed_kc(p->k);
            const int nch = p->precision != SG3_CONV_FP32 ? f16x3_chunks(p->I, p->k) : ceil_div(p->I, kc);
            const int gy = std::min(16, ceil_div(p->O, 32));
            bw.p[j] = bs.p[j] = *p;
            bw.kc[j] = bs.kc[j] = kc; bw.nch[j] = bs.nch[j] = nch; bw.gy[j] = bs.gy[j] = gy;
            bw.first[j + 1] = bw.first[j] + (p->reuseWeights ? 0 : p->O);        // an entry whose packed weights stand takes no workgroup
            bs.first[j + 1] = bs.first[j] + p->N * gy;
            lds = std::max(lds, (size_t)p->I * 2 * sizeof(float));
        }
        if (bw.first[m] > 0) {
            hipLaunchKernelGGL(modconv_prep_w_batch_kernel, dim3(bw.first[m]), dim3(256), 0, st, bw);
            SG3_LAUNCH_CHECK("modconv_prep_w_batch_kernel");
        }
        hipLaunchKernelGGL(modconv_prep_s_batch_kernel, dim3(bs.first[m]), dim3(256), lds, st, bs);
        SG3_LAUNCH_CHECK("modconv_prep_s_batch_kernel");
    }
    return SG3_OK;
}

int sg3_modulated_conv2d_prep(const sg3_modconv_prep_params* p, void* stream) {
    using namespace sg3;
    { const int rc = prep_validate(p); if (rc != SG3_OK) return rc; }
    hipStream_t st = (hipStream_t)stream;
    const int kc = packed_kc(p->k);
    const int nch = p->precision != SG3_CONV_FP32 ? f16x3_chunks(p->I, p->k) : ceil_div(p->I, kc);
    if (!p->reuseWeights) {
        hipLaunchKernelGGL(modconv_prep_w_kernel, dim3(p->O), dim3(256), 0, st, *p, kc, nch);
        SG3_LAUNCH_CHECK("modconv_prep_w_kernel");
    }
    hipLaunchKernelGGL(modconv_prep_s_kernel, dim3(p->N, min(16, ceil_div(p->O, 32))), dim3(256), (size_t)p->I * 2 * sizeof(float), st, *p);
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
    SG3_REQUIRE(p->dtype == SG3_F32 || p->dtype == SG3_F16, "modulated_conv2d: unsupported dtype");
    hipStream_t st = (hipStream_t)stream;
    {
        const int outW = p->W + 2 * p->pad - p->k + 1;
        const bool rowStream = p->k == 3 && (p->precision == SG3_CONV_F16X3 || p->precision == SG3_CONV_F16 || p->precision == SG3_CONV_F16X3_F23 ||
                                             p->precision == SG3_CONV_F16_F23);
        SG3_REQUIRE(p->outRowStride == 0 || p->outRowStride == outW || (rowStream && p->outRowStride > outW),
                    "modulated_conv2d: outRowStride must be 0 or outW (a larger pitch is supported by the 3x3 f16x3 / f16 kernels only)");
    }
    const bool torgb = p->precision == SG3_CONV_FP32 && p->k == 1 && p->pad == 0 && p->O <= 4 && (size_t)p->I * 4 * sizeof(float) <= 48 * 1024;
    SG3_REQUIRE(!p->epilogueBias || torgb, "modulated_conv2d: the bias / clamp / scale epilogue exists for the ToRGB kernel only (1x1, O <= 4, fp32 form)");
    if (p->precision == SG3_CONV_F16X3_F23 || p->precision == SG3_CONV_F16_F23) {
        SG3_REQUIRE(p->dcoef, "modulated_conv2d: the transform-domain forms need dcoef");
        SG3_REQUIRE(p->dtype == (p->precision == SG3_CONV_F16X3_F23 ? SG3_F32 : SG3_F16),
                    "modulated_conv2d: SG3_CONV_F16X3_F23 takes fp32 tensors, SG3_CONV_F16_F23 fp16 tensors");
        SG3_REQUIRE(f23_supported(p->dtype, p->I, p->O, p->H, p->W, p->k, p->pad, p->outRowStride),
                    "modulated_conv2d: the transform-domain forms take 3x3 kernels with even W, pad and row pitch (sg3_modconv_f23_supported)");
        return launch_conv_f23(*p, st);
    }
    if (p->precision == SG3_CONV_F16X3 || p->precision == SG3_CONV_F16) {
        SG3_REQUIRE(p->dcoef, "modulated_conv2d: f16x3 needs dcoef");
        SG3_REQUIRE(p->k == 3 || p->pad == 0, "modulated_conv2d: f16x3 1x1 kernels take no padding");
        SG3_REQUIRE(p->k == 3 || ((p->H * p->W) & 1) == 0, "modulated_conv2d: f16x3 1x1 kernels need an even number of pixels (pair loads)");
        SG3_REQUIRE((int64_t)p->I * p->H * p->W * 4 < (int64_t)1 << 31, "modulated_conv2d: f16x3 needs a sample below 2 GiB (32-bit offsets)");
        if (p->precision == SG3_CONV_F16) {
            if (p->k == 1) return p->dtype == SG3_F32 ? dispatch_conv1_f16x3<float, false>(*p, st) : dispatch_conv1_f16x3<_Float16, false>(*p, st);
            return p->dtype == SG3_F32 ? dispatch_conv_f16x3<float, false>(*p, st) : dispatch_conv_f16x3<_Float16, false>(*p, st);
        }
        if (p->k == 1) return p->dtype == SG3_F32 ? dispatch_conv1_f16x3<float, true>(*p, st) : dispatch_conv1_f16x3<_Float16, true>(*p, st);
        return p->dtype == SG3_F32 ? dispatch_conv_f16x3<float, true>(*p, st) : dispatch_conv_f16x3<_Float16, true>(*p, st);
    }
    SG3_REQUIRE(p->precision == SG3_CONV_FP32, "modulated_conv2d: bad precision");
    if (torgb)
        return p->dtype == SG3_F32 ? launch_conv_1x1_small<float>(*p, st) : launch_conv_1x1_small<_Float16>(*p, st);
    if (p->dtype == SG3_F32) return p->k == 3 ? dispatch_conv<float, 3>(*p, st) : dispatch_conv<float, 1>(*p, st);
    return p->k == 3 ? dispatch_conv<_Float16, 3>(*p, st) : dispatch_conv<_Float16, 1>(*p, st);
}

} // extern "C"
