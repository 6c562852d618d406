// sg3_modconv_f23.h -- interface between sg3_modconv.hip (prep kernels, dispatch) and sg3_modconv_f23.hip (the transform-domain
// 3x3 kernel, SG3_CONV_F16X3_F23).
#pragma once
#include "sg3_common.h"

namespace sg3 {

int64_t f23_packed_floats(int O, int I, bool split);      // split: hi | lo fragments (SG3_CONV_F16X3_F23); else one fragment per filter row (SG3_CONV_F16_F23)
// shapes the kernel takes: 3x3, fp32 or fp16 tensors, even W / pad / row pitch (aligned column pairs), 32-bit offsets
bool f23_supported(int dtype, int I, int O, int H, int W, int k, int pad, int outRowStride);
int launch_conv_f23(const sg3_modconv_params& q, hipStream_t st);
int f23_force_rows(int rows);          // 4 | 5 | 7, 0 = cost model; returns the previous setting (sg3_modconv_f23_force_rows)

// Packing of one output channel's filters (one workgroup per output channel, called from the prep kernels):
//   [M tile = o / 64][chunk][M block = (o / 32) % 2][xi][ky][hi|lo][lane][8 halfs],  lane = 32 (c / 8) + o % 32,  element = c % 8
// for input channel i = 16 chunk + c: the A operand of v_mfma_f32_32x32x16_f16 in the order a wave loads it (six consecutive
// 1 KB fragments per (chunk, M block, xi)).  U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2 of the filter
// row g = wn[o, i, ky, :].  The workgroup of the last channel also zeroes the rows of the padded channels O .. 64 mTiles - 1.
// `split` = false (SG3_CONV_F16_F23): U rounded to fp16 once (round to nearest even), no lo fragments -- three 1 KB fragments per
// (chunk, M block, xi): [M tile][chunk][M block][xi][ky][lane][8 halfs].
static __device__ __forceinline__ void f23_pack_row(const float* w, float scale, int o, int O, int I, int nch, _Float16* wp, bool split) {
    const int rows = (o == O - 1) ? ((O + 63) / 64) * 64 - O + 1 : 1;        // this channel, then the padded ones
    for (int rr = 0; rr < rows; rr++) {
        const int oo = o + rr;
        const int mt = oo >> 6, mb = (oo >> 5) & 1, row = oo & 31;
        for (int j = threadIdx.x; j < nch * 3 * 16; j += blockDim.x) {
            const int c = j & 15, ky = (j >> 4) % 3, ch = j / 48;
            const int i = ch * 16 + c;
            float g0 = 0.f, g1 = 0.f, g2 = 0.f;
            if (rr == 0 && i < I) { const float* g = w + (size_t)i * 9 + ky * 3; g0 = g[0] * scale; g1 = g[1] * scale; g2 = g[2] * scale; }
            const float U[4] = {g0, 0.5f * (g0 + g1 + g2), 0.5f * (g0 - g1 + g2), g2};
            const int ln = (c >> 3) * 32 + row, e = c & 7;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const _Float16 h = (_Float16)U[t];
                if (split) {
                    _Float16* dst = wp + ((((((size_t)mt * nch + ch) * 2 + mb) * 4 + t) * 3 + ky) * 2) * 512 + ln * 8 + e;
                    dst[0] = h;
                    dst[512] = (_Float16)(U[t] - (float)h);
                } else {
                    wp[(((((size_t)mt * nch + ch) * 2 + mb) * 4 + t) * 3 + ky) * 512 + ln * 8 + e] = h;
                }
            }
        }
    }
}

} // namespace sg3
