// sg3_upfirdn2d.hip -- generic pad / zero-insert upsample / FIR / decimate.
//
// Replaces upfirdn2d_plugin.upfirdn2d (reference torch_utils/ops/upfirdn2d.cpp:16-98,
// kernels torch_utils/ops/upfirdn2d.cu:29-200).  On the StyleGAN3 synthesis path this
// op is only the generic fallback behind filtered_lrelu (the fused kernel in
// sg3_filtered_lrelu.hip covers every SG3 layer), so it is one polyphase
// gather kernel: a thread owns one output pixel and visits only the filter taps that
// land on real (non zero-stuffed) input samples.  Taps are staged in LDS once per
// workgroup, already flipped and scaled, so the inner loop is LDS-broadcast + FMA.
#include "sg3_common.h"

namespace sg3 {

constexpr int UF_MAX_TAPS = 64 * 64;   // LDS-staged filter limit (16 KiB)

template <typename T>
__global__ void __launch_bounds__(256)
upfirdn2d_gather_kernel(sg3_upfirdn2d_params p) {
    typedef typename io<T>::acc_t S;
    __shared__ float s_f[UF_MAX_TAPS];
    const int ntaps = p.fH * p.fW;
    // stage taps as the correlation kernel g[ky][kx] that multiplies padded
    // sample (oy*down + ky, ox*down + kx): flipped unless p.flip.
    for (int i = threadIdx.x; i < ntaps; i += blockDim.x) {
        int ky = i / p.fW, kx = i - ky * p.fW;
        int sy = p.flip ? ky : p.fH - 1 - ky;
        int sx = p.flip ? kx : p.fW - 1 - kx;
        s_f[i] = p.f[sy * p.fStride[0] + sx * p.fStride[1]] * p.gain;
    }
    __syncthreads();

    const T* __restrict__ px = (const T*)p.x;
    T* __restrict__ py = (T*)p.y;
    const int64_t total = (int64_t)p.N * p.C * p.yH * p.yW;
    const int64_t gstride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gstride) {
        int ox = (int)(idx % p.yW); int64_t t = idx / p.yW;
        int oy = (int)(t % p.yH); t /= p.yH;
        int c = (int)(t % p.C); int n = (int)(t / p.C);
        // padded-upsampled coordinate of tap 0
        const int bx = ox * p.downx - p.padx0;
        const int by = oy * p.downy - p.pady0;
        int ix0 = ceil_div_s(bx, p.upx);              if (ix0 < 0) ix0 = 0;
        int ix1 = floor_div(bx + p.fW - 1, p.upx);    if (ix1 > p.xW - 1) ix1 = p.xW - 1;
        int iy0 = ceil_div_s(by, p.upy);              if (iy0 < 0) iy0 = 0;
        int iy1 = floor_div(by + p.fH - 1, p.upy);    if (iy1 > p.xH - 1) iy1 = p.xH - 1;
        const T* plane = px + (int64_t)n * p.xStride[0] + (int64_t)c * p.xStride[1];
        S acc = 0;
        for (int iy = iy0; iy <= iy1; iy++) {
            const int ky = iy * p.upy - by;
            const T* row = plane + (int64_t)iy * p.xStride[2];
            const float* frow = s_f + ky * p.fW;
            for (int ix = ix0; ix <= ix1; ix++) {
                const int kx = ix * p.upx - bx;
                acc += io<T>::ld(row + (int64_t)ix * p.xStride[3]) * (S)frow[kx];
            }
        }
        io<T>::st(py + (int64_t)n * p.yStride[0] + (int64_t)c * p.yStride[1] + (int64_t)oy * p.yStride[2] + (int64_t)ox * p.yStride[3], acc);
    }
}

template <typename T>
static int launch_upfirdn2d(const sg3_upfirdn2d_params& p, hipStream_t st) {
    const int64_t total = (int64_t)p.N * p.C * p.yH * p.yW;
    int64_t blocks = ceil_div64(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((upfirdn2d_gather_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    SG3_LAUNCH_CHECK("upfirdn2d_gather_kernel");
    return SG3_OK;
}

} // namespace sg3

extern "C" {

int sg3_upfirdn2d_shape(int xH, int xW, int fH, int fW, int upx, int upy, int downx, int downy,
                        int padx0, int padx1, int pady0, int pady1, int* yH, int* yW) {
    using namespace sg3;
    SG3_REQUIRE(xH > 0 && xW > 0 && fH > 0 && fW > 0, "upfirdn2d: empty tensor");
    SG3_REQUIRE(upx >= 1 && upy >= 1 && downx >= 1 && downy >= 1, "upfirdn2d: up and down must be at least 1");
    // reference torch_utils/ops/upfirdn2d.cpp:36-39
    int64_t ow = ((int64_t)xW * upx + padx0 + padx1 - fW + downx) / downx;
    int64_t oh = ((int64_t)xH * upy + pady0 + pady1 - fH + downy) / downy;
    SG3_REQUIRE(ow >= 1 && oh >= 1, "upfirdn2d: output must be at least 1x1");
    SG3_REQUIRE(ow <= INT32_MAX && oh <= INT32_MAX, "upfirdn2d: output is too large");
    if (yH) *yH = (int)oh;
    if (yW) *yW = (int)ow;
    return SG3_OK;
}

int sg3_upfirdn2d(const sg3_upfirdn2d_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->x && p->y && p->f, "upfirdn2d: null tensor");
    SG3_REQUIRE(p->N > 0 && p->C > 0 && p->xH > 0 && p->xW > 0, "upfirdn2d: x is empty");
    SG3_REQUIRE(p->yH > 0 && p->yW > 0, "upfirdn2d: output must be at least 1x1");
    SG3_REQUIRE(p->fH > 0 && p->fW > 0, "upfirdn2d: f is empty");
    SG3_REQUIRE(p->upx >= 1 && p->upy >= 1 && p->downx >= 1 && p->downy >= 1, "upfirdn2d: up and down must be at least 1");
    if ((int64_t)p->fH * p->fW > UF_MAX_TAPS) {
        set_error("upfirdn2d: filter %dx%d exceeds the %d-tap LDS staging limit", p->fH, p->fW, UF_MAX_TAPS);
        return SG3_NO_KERNEL;
    }
    hipStream_t st = (hipStream_t)stream;
    switch (p->dtype) {
        case SG3_F32: return launch_upfirdn2d<float>(*p, st);
        case SG3_F16: return launch_upfirdn2d<_Float16>(*p, st);
        case SG3_F64: return launch_upfirdn2d<double>(*p, st);
    }
    set_error("upfirdn2d: unsupported dtype %d", p->dtype);
    return SG3_BAD_ARG;
}

} // extern "C"
