// sg3_bias_act.hip -- fused bias + activation + gain + clamp with 1st/2nd
// derivative modes, and the in-place lrelu/sign kernel used by the generic
// filtered_lrelu composition.
//
// Semantics follow the reference kernels (torch_utils/ops/bias_act.cu:23-147
// and torch_utils/ops/filtered_lrelu.cu:1105-1211); the implementation is a
// grid-stride wave64 elementwise pass (HBM-bound, negligible on the synthesis
// path: it only runs on the [N,512] mapping activations).
#include "sg3_common.h"
#include <cmath>

namespace sg3 {

static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
}
const char* get_error() { return g_err; }

// ---- activation table: value, d/dx and d2/dx2 expressed the way the reference
// does (derivatives in terms of the saved output y/gain, or x for swish) ----
template <typename S> struct ActConst {
    static constexpr double selu_scale = 1.0507009873554804934193349852946;
    static constexpr double selu_alpha = 1.6732632423543772848170429916717;
};

template <typename S, int ACT>
__device__ inline S act_eval(int grad, S x, S xref, S yy, S alpha) {
    const S one = (S)1, two = (S)2, big = (S)80, half_big = (S)40;
    const S ss = (S)ActConst<S>::selu_scale, sa = (S)ActConst<S>::selu_alpha;
    if (ACT == 1) {                     // linear
        return (grad <= 1) ? x : (S)0;
    } else if (ACT == 2) {              // relu
        if (grad == 0) return x > 0 ? x : (S)0;
        if (grad == 1) return yy > 0 ? x : (S)0;
        return (S)0;
    } else if (ACT == 3) {              // lrelu
        if (grad == 0) return x > 0 ? x : x * alpha;
        if (grad == 1) return yy > 0 ? x : x * alpha;
        return (S)0;
    } else if (ACT == 4) {              // tanh
        if (grad == 0) {
            S c = exp(x), d = one / c;
            return x < -big ? -one : (x > big ? one : (c - d) / (c + d));
        }
        S t = x * (one - yy * yy);
        return grad == 1 ? t : t * (-two * yy);
    } else if (ACT == 5) {              // sigmoid
        if (grad == 0) return x < -big ? (S)0 : one / (exp(-x) + one);
        S t = x * yy * (one - yy);
        return grad == 1 ? t : t * (one - two * yy);
    } else if (ACT == 6) {              // elu
        if (grad == 0) return x >= 0 ? x : exp(x) - one;
        if (grad == 1) return yy >= 0 ? x : x * (yy + one);
        return yy >= 0 ? (S)0 : x * (yy + one);
    } else if (ACT == 7) {              // selu
        if (grad == 0) return x >= 0 ? ss * x : (ss * sa) * (exp(x) - one);
        if (grad == 1) return yy >= 0 ? x * ss : x * (yy + ss * sa);
        return yy >= 0 ? (S)0 : x * (yy + ss * sa);
    } else if (ACT == 8) {              // softplus
        if (grad == 0) return x > big ? x : log(exp(x) + one);
        if (grad == 1) return x * (one - exp(-yy));
        S c = exp(-yy); return x * c * (one - c);
    } else {                            // swish (ACT == 9)
        if (grad == 0) return x < -big ? (S)0 : x / (exp(-x) + one);
        S c = exp(xref), d = c + one;
        if (grad == 1) return xref > half_big ? x : x * c * (xref + d) / (d * d);
        return xref > half_big ? (S)0 : x * c * (xref * (two - d) + two * d) / (d * d * d);
    }
}

template <typename T, int ACT>
__global__ void __launch_bounds__(256)
bias_act_kernel(sg3_bias_act_params p) {
    typedef typename io<T>::acc_t S;
    const T* __restrict__ px = (const T*)p.x;
    const T* __restrict__ pb = (const T*)p.b;
    const T* __restrict__ pxr = (const T*)p.xref;
    const T* __restrict__ pyr = (const T*)p.yref;
    const T* __restrict__ pdy = (const T*)p.dy;
    T* __restrict__ py = (T*)p.y;
    const int G = p.grad;
    const S alpha = (S)p.alpha, gain = (S)p.gain, clampv = (S)p.clamp;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < p.sizeX; i += stride) {
        S x = io<T>::ld(px + i);
        S b = pb ? io<T>::ld(pb + (i / p.stepB) % p.sizeB) : (S)0;
        S xref = pxr ? io<T>::ld(pxr + i) : (S)0;
        S yref = pyr ? io<T>::ld(pyr + i) : (S)0;
        S dy = pdy ? io<T>::ld(pdy + i) : (S)1;
        S yy = gain != 0 ? yref / gain : (S)0;
        if (G == 0) x += b; else xref += b;
        S y = act_eval<S, ACT>(G, x, xref, yy, alpha);
        if (ACT == 9 && G != 0)     // swish keeps x, not y: rebuild y for the clamp mask
            yref = xref < (S)-80 ? (S)0 : xref / (exp(-xref) + (S)1) * gain;
        y *= gain * dy;
        if (clampv >= 0) {
            if (G == 0) y = (y > -clampv && y < clampv) ? y : (y >= 0 ? clampv : -clampv);
            else        y = (yref > -clampv && yref < clampv) ? y : (S)0;
        }
        io<T>::st(py + i, y);
    }
}

template <typename T>
static int launch_bias_act(const sg3_bias_act_params& p, hipStream_t st) {
    int64_t blocks = ceil_div64(p.sizeX, 256 * 4);
    if (blocks > 256 * 8) blocks = 256 * 8;       // 8 blocks per CU, grid-stride the rest
    if (blocks < 1) blocks = 1;
    dim3 g((unsigned)blocks), b(256);
#define SG3_BA_CASE(A) case A: hipLaunchKernelGGL((bias_act_kernel<T, A>), g, b, 0, st, p); break;
    switch (p.act) {
        SG3_BA_CASE(1) SG3_BA_CASE(2) SG3_BA_CASE(3) SG3_BA_CASE(4) SG3_BA_CASE(5)
        SG3_BA_CASE(6) SG3_BA_CASE(7) SG3_BA_CASE(8) SG3_BA_CASE(9)
        default: set_error("bias_act: unknown activation index %d", p.act); return SG3_BAD_ARG;
    }
#undef SG3_BA_CASE
    SG3_LAUNCH_CHECK("bias_act_kernel");
    return SG3_OK;
}

// ---------------------------------------------------------------------------
// in-place gain * lrelu * clamp with 2-bit sign write / read.
// One thread per element; a wave covers 64 consecutive x, so the 16-element
// (one uint32 of sign bits) groups are the four quarter-waves and are reduced
// with DPP-style xor shuffles inside each 16-lane group.
// MODE: 0 = plain, 1 = write signs, 2 = read signs.
template <typename T, int MODE>
__global__ void __launch_bounds__(256)
flrelu_act_kernel(sg3_filtered_lrelu_act_params p) {
    typedef typename io<T>::acc_t S;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int ymax = (MODE == 1) ? p.sH : p.H;
    const int qmax = p.N * p.C;
    T* base = (T*)p.x;
    const S gain = (S)p.gain, slope = (S)p.slope, clampv = (S)p.clamp;
    for (int q = blockIdx.z; q < qmax; q += gridDim.z)
    for (int y = blockIdx.y; y < ymax; y += gridDim.y) {
        const int n = q / p.C, c = q - n * p.C;
        if (MODE == 1) {
            uint32_t s = 0;
            if (x < p.W && y < p.H) {
                T* pv = base + (int64_t)n * p.xStride[0] + (int64_t)c * p.xStride[1] + (int64_t)y * p.xStride[2] + (int64_t)x * p.xStride[3];
                S v = io<T>::ld(pv) * gain;
                if (v < 0) { v *= slope; s = 1; }
                if (fabs(v) > clampv) { v = v < 0 ? -clampv : clampv; s = 2; }
                io<T>::st(pv, v);
            }
            s <<= ((threadIdx.x & 15) << 1);
            s |= __shfl_xor(s, 1); s |= __shfl_xor(s, 2); s |= __shfl_xor(s, 4); s |= __shfl_xor(s, 8);
            if (!(threadIdx.x & 15) && x < p.sW) {
                uint64_t is = (uint64_t)x + (uint64_t)p.sW * ((uint64_t)y + (uint64_t)p.sH * q);
                ((uint32_t*)p.s)[is >> 4] = s;
            }
        } else if (x < p.W) {
            T* pv = base + (int64_t)n * p.xStride[0] + (int64_t)c * p.xStride[1] + (int64_t)y * p.xStride[2] + (int64_t)x * p.xStride[3];
            S v = io<T>::ld(pv) * gain;
            if (MODE == 2) {
                uint32_t sx = (uint32_t)(x + p.sx), sy = (uint32_t)(y + p.sy);
                if (sx < (uint32_t)p.sW && sy < (uint32_t)p.sH) {
                    uint64_t is = (sx >> 2) + (uint64_t)(p.sW >> 2) * (sy + (uint64_t)p.sH * q);
                    uint32_t s = p.s[is] >> ((sx & 3) << 1);
                    if (s & 1) v *= slope;
                    if (s & 2) v = 0;
                }
            } else {
                if (v < 0) v *= slope;
                if (fabs(v) > clampv) v = v < 0 ? -clampv : clampv;
            }
            io<T>::st(pv, v);
        }
    }
}

template <typename T>
static int launch_act(const sg3_filtered_lrelu_act_params& p, hipStream_t st) {
    const int bx = 256;
    uint32_t gx = p.writeSigns ? (uint32_t)p.sW : (uint32_t)p.W;
    uint32_t gy = p.writeSigns ? (uint32_t)p.sH : (uint32_t)p.H;
    uint32_t gz = (uint32_t)(p.N * p.C);
    gx = (gx - 1) / bx + 1;
    if (gy > 65535u) gy = 65535u;
    if (gz > 65535u) gz = 65535u;
    dim3 g(gx, gy, gz), b(bx);
    if (p.writeSigns)      hipLaunchKernelGGL((flrelu_act_kernel<T, 1>), g, b, 0, st, p);
    else if (p.readSigns)  hipLaunchKernelGGL((flrelu_act_kernel<T, 2>), g, b, 0, st, p);
    else                   hipLaunchKernelGGL((flrelu_act_kernel<T, 0>), g, b, 0, st, p);
    SG3_LAUNCH_CHECK("flrelu_act_kernel");
    return SG3_OK;
}

} // namespace sg3

extern "C" {

int sg3_abi_version(void) { return SG3_ABI_VERSION; }
const char* sg3_last_error(void) { return sg3::get_error(); }

int sg3_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

int sg3_bias_act(const sg3_bias_act_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->x && p->y, "bias_act: null tensor");
    SG3_REQUIRE(p->sizeX >= 0, "bias_act: negative size");
    SG3_REQUIRE(p->grad >= 0 && p->grad <= 2, "bias_act: grad must be 0, 1 or 2");
    SG3_REQUIRE(!p->b || (p->sizeB > 0 && p->stepB > 0), "bias_act: bad bias geometry");
    if (p->sizeX == 0) return SG3_OK;
    hipStream_t st = (hipStream_t)stream;
    switch (p->dtype) {
        case SG3_F32: return launch_bias_act<float>(*p, st);
        case SG3_F16: return launch_bias_act<_Float16>(*p, st);
        case SG3_F64: return launch_bias_act<double>(*p, st);
    }
    set_error("bias_act: unsupported dtype %d", p->dtype);
    return SG3_BAD_ARG;
}

int sg3_filtered_lrelu_act(const sg3_filtered_lrelu_act_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->x, "filtered_lrelu_act: null tensor");
    SG3_REQUIRE(p->N > 0 && p->C > 0 && p->H > 0 && p->W > 0, "filtered_lrelu_act: x is empty");
    SG3_REQUIRE(!(p->writeSigns && p->readSigns), "filtered_lrelu_act: cannot read and write signs");
    if (p->writeSigns || p->readSigns) {
        SG3_REQUIRE(p->s, "filtered_lrelu_act: sign tensor missing");
        SG3_REQUIRE(p->sH > 0 && p->sW > 0 && (p->sW & 15) == 0, "filtered_lrelu_act: sign width must be a positive multiple of 16");
    }
    hipStream_t st = (hipStream_t)stream;
    switch (p->dtype) {
        case SG3_F32: return launch_act<float>(*p, st);
        case SG3_F16: return launch_act<_Float16>(*p, st);
        case SG3_F64: return launch_act<double>(*p, st);
    }
    set_error("filtered_lrelu_act: unsupported dtype %d", p->dtype);
    return SG3_BAD_ARG;
}

} // extern "C"
