// sg3_unfold.hip -- patches of a 3x3, stride-2, padding-1 convolution as GEMM rows, for the GradualStyleBlock heads of the ReStyle
// encoders (reference models/setgan/encoder/encoders/map2style.py:8-25: log2(spatial) x [Conv2d(3x3, stride 2, padding 1) +
// LeakyReLU] down to 1x1, then EqualLinear; restyle_psp_encoders.py:26-50 runs n_styles of them on one [N,512,16,16] map).
//
// The heads see 8x8 ... 1x1 maps: too few pixels for an implicit-GEMM tile, and from the second level on each head has its own
// input, so those levels are batched GEMMs over unfolded patches ([heads][N * pixels, 9 C] x [heads][9 C, C], weight-bandwidth
// bound: 151 MB per level).  This kernel writes the patch matrix in ONE launch per level from wherever the previous level left its
// result -- the channels-first strip the level-1 convolution writes, or the pixel-major [heads][N * pixels][C] rows of a GEMM --
// with the previous level's LeakyReLU applied on the way: the pad / nine strided slices / stack / permute / reshape chain of
// torch ops (about six launches and four passes over the data per level) is gone.
//
//     dst[g][(n * OH + oy) * OW + ox][tap * C + c] = act(src[g, n, c, 2 oy + ky - 1, 2 ox + kx - 1])   (0 outside the map)
//
// tap = ky * 3 + kx, columns ordered tap-major so that a row is nine runs of C consecutive channels (the caller packs the GEMM's
// weight rows in the same order).  src is addressed through five element strides (g, n, c, y, x).  Two forms:
//   channels contiguous (stride_c == 1): lanes along c, reads and writes both 256-byte runs;
//   otherwise (channels-first source): a workgroup takes (g, oy, tap, 64 channels) x 64 rows (n, ox), reads with lanes along the
//   rows (consecutive lanes = neighbouring pixels of one channel row), transposes through LDS, writes with lanes along c.
#include "sg3_common.h"

namespace sg3 {

struct UnfoldParams {
    const float* src; float* dst;
    long long sg, sn, sc, sy, sx;
    int G, N, C, IH, IW, OH, OW;
    float slope;
};

__device__ __forceinline__ float unfold_act(float v, float slope) { return v < 0.f ? v * slope : v; }

__global__ void __launch_bounds__(256)
unfold_rows_kernel(UnfoldParams p) {
    // one (row, tap) per blockIdx.x; threads along c
    int b = blockIdx.x;
    const int tap = b % 9; b /= 9;
    const int ox = b % p.OW; b /= p.OW;
    const int oy = b % p.OH; b /= p.OH;
    const int n = b % p.N; const int g = b / p.N;
    const int iy = 2 * oy + tap / 3 - 1, ix = 2 * ox + tap % 3 - 1;
    const bool in = (unsigned)iy < (unsigned)p.IH && (unsigned)ix < (unsigned)p.IW;
    const float* s = p.src + g * p.sg + n * p.sn + (in ? iy * p.sy + ix * p.sx : 0);
    float* d = p.dst + (((size_t)g * p.N + n) * p.OH * p.OW + (size_t)oy * p.OW + ox) * (size_t)(9 * p.C) + (size_t)tap * p.C;
    for (int c = threadIdx.x; c < p.C; c += 256) d[c] = in ? unfold_act(s[c * p.sc], p.slope) : 0.f;
}

__global__ void __launch_bounds__(256)
unfold_transpose_kernel(UnfoldParams p, int cChunks, int rChunks) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int b = blockIdx.x;
    const int rc = b % rChunks; b /= rChunks;
    const int cc = b % cChunks; b /= cChunks;
    const int tap = b % 9; b /= 9;
    const int oy = b % p.OH; const int g = b / p.OH;
    const int iy = 2 * oy + tap / 3 - 1, kx = tap % 3;
    const bool rowIn = (unsigned)iy < (unsigned)p.IH;                       // workgroup-uniform
    const int R = p.N * p.OW;
    // read: lane = row (n, ox) of this chunk, wave walks 16 channels
    const int r = rc * 64 + lane;
    const int n = r / p.OW, ox = r - n * p.OW;
    const int ix = 2 * ox + kx - 1;
    const bool in = rowIn && r < R && (unsigned)ix < (unsigned)p.IW;
    const float* s = p.src + g * p.sg + (in ? n * p.sn + iy * p.sy + ix * p.sx : 0);
#pragma unroll 4
    for (int i = 0; i < 16; i++) {
        const int cl = wave * 16 + i, c = cc * 64 + cl;
        tile[lane][cl] = (in && c < p.C) ? unfold_act(s[c * p.sc], p.slope) : 0.f;
    }
    __syncthreads();
    // write: lane = channel, wave walks 16 rows
    const int c = cc * 64 + lane;
#pragma unroll 4
    for (int j = 0; j < 16; j++) {
        const int rl = wave * 16 + j, rr = rc * 64 + rl;
        if (rr < R && c < p.C) {
            const int nn = rr / p.OW, oxx = rr - nn * p.OW;
            p.dst[(((size_t)g * p.N + nn) * p.OH * p.OW + (size_t)oy * p.OW + oxx) * (size_t)(9 * p.C) + (size_t)tap * p.C + c] = tile[rl][lane];
        }
    }
}

} // namespace sg3

extern "C" int sg3_unfold3x3s2(const sg3_unfold_params* q, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(q && q->src && q->dst, "unfold3x3s2: null tensor");
    SG3_REQUIRE(q->G > 0 && q->N > 0 && q->C > 0 && q->IH > 0 && q->IW > 0, "unfold3x3s2: empty tensor");
    UnfoldParams p;
    p.src = q->src; p.dst = q->dst;
    p.sg = q->srcStride[0]; p.sn = q->srcStride[1]; p.sc = q->srcStride[2]; p.sy = q->srcStride[3]; p.sx = q->srcStride[4];
    p.G = q->G; p.N = q->N; p.C = q->C; p.IH = q->IH; p.IW = q->IW; p.OH = (q->IH + 1) / 2; p.OW = (q->IW + 1) / 2;
    p.slope = q->slope;
    const long long rows = (long long)p.G * p.N * p.OH * p.OW;
    SG3_REQUIRE(rows * 9 < (1ll << 31), "unfold3x3s2: too many patch rows for one launch");
    hipStream_t st = (hipStream_t)stream;
    if (p.sc == 1) {
        hipLaunchKernelGGL(unfold_rows_kernel, dim3((unsigned)(rows * 9)), dim3(256), 0, st, p);
        SG3_LAUNCH_CHECK("unfold_rows_kernel");
    } else {
        const int cChunks = ceil_div(p.C, 64), rChunks = ceil_div(p.N * p.OW, 64);
        const long long blocks = (long long)p.G * p.OH * 9 * cChunks * rChunks;
        SG3_REQUIRE(blocks < (1ll << 31), "unfold3x3s2: too many tiles for one launch");
        hipLaunchKernelGGL(unfold_transpose_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, cChunks, rChunks);
        SG3_LAUNCH_CHECK("unfold_transpose_kernel");
    }
    return SG3_OK;
}
