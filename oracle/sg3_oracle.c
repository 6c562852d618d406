/*
 * sg3_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C restatement of the reference's `ref` operator semantics for the
 * StyleGAN3 synthesis hot path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product path
 * (stylegan3-editing_amd/) never does and fails loudly without its HIP library.
 *
 * Parity status: PINNED.  Every function here is checked in tests/ against
 * golden vectors generated in the build container by importing the reference's
 * own pure-PyTorch `_ref` implementations (tests/golden/make_golden.py).
 *
 * Each function cites the reference lines it restates (paths relative to the
 * reference tree).  All arithmetic is done in the element type (float or
 * double), accumulating in that type like the reference's conv-based path.
 *
 * Build: oracle/Makefile  ->  oracle/libsg3_oracle.so   (gcc -O2 -fopenmp)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define ORACLE_API __attribute__((visibility("default")))

static inline int64_t floor_div64(int64_t a, int64_t b) {
    int64_t q = a / b, r = a % b;
    return (r != 0 && ((r < 0) != (b < 0))) ? q - 1 : q;
}

ORACLE_API int sg3o_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

ORACLE_API void sg3o_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ------------------------------------------------------------------------
 * upfirdn2d  (torch_utils/ops/upfirdn2d.py:168-212, _upfirdn2d_ref)
 *   zero-insert upsample (:188-190) -> pad / crop (:193-194) -> filter scaled
 *   by gain and flipped unless flip_filter (:197-200) -> correlation with the
 *   result (:203-208) -> keep every down-th sample (:211).
 * f is [fh, fw] (a separable filter is applied by the caller as two calls,
 * each with gain^(1/2), exactly like the reference's f.ndim==1 branch).
 * x: [nc, xh, xw] contiguous planes; y: [nc, yh, yw].
 * ---------------------------------------------------------------------- */
#define DEFINE_UPFIRDN2D(NAME, T)                                                          \
ORACLE_API void NAME(const T* x, const float* f, T* y, int nc, int xh, int xw,             \
                     int fh, int fw, int upx, int upy, int downx, int downy,               \
                     int padx0, int padx1, int pady0, int pady1, int flip, double gain) {  \
    const int yw = (xw * upx + padx0 + padx1 - fw) / downx + 1;                            \
    const int yh = (xh * upy + pady0 + pady1 - fh) / downy + 1;                            \
    T* g = (T*)malloc(sizeof(T) * (size_t)fh * fw);                                        \
    for (int ky = 0; ky < fh; ky++) for (int kx = 0; kx < fw; kx++) {                      \
        int sy = flip ? ky : fh - 1 - ky, sx = flip ? kx : fw - 1 - kx;                    \
        g[ky * fw + kx] = (T)((T)f[sy * fw + sx] * (T)gain);                               \
    }                                                                                      \
    _Pragma("omp parallel for collapse(2) schedule(static)")                               \
    for (int p = 0; p < nc; p++) for (int oy = 0; oy < yh; oy++) {                         \
        const T* xp = x + (size_t)p * xh * xw;                                             \
        T* yp = y + ((size_t)p * yh + oy) * yw;                                            \
        const int by = oy * downy - pady0;                                                 \
        for (int ox = 0; ox < yw; ox++) {                                                  \
            const int bx = ox * downx - padx0;                                             \
            T acc = 0;                                                                     \
            for (int ky = 0; ky < fh; ky++) {                                              \
                int uy = by + ky;                                                          \
                if (uy < 0 || uy % upy != 0) continue;                                     \
                int iy = uy / upy; if (iy >= xh) continue;                                 \
                for (int kx = 0; kx < fw; kx++) {                                          \
                    int ux = bx + kx;                                                      \
                    if (ux < 0 || ux % upx != 0) continue;                                 \
                    int ix = ux / upx; if (ix >= xw) continue;                             \
                    acc += xp[(size_t)iy * xw + ix] * g[ky * fw + kx];                     \
                }                                                                          \
            }                                                                              \
            yp[ox] = acc;                                                                  \
        }                                                                                  \
    }                                                                                      \
    free(g);                                                                               \
}
DEFINE_UPFIRDN2D(sg3o_upfirdn2d_f32, float)
DEFINE_UPFIRDN2D(sg3o_upfirdn2d_f64, double)

/* ------------------------------------------------------------------------
 * bias_act  (torch_utils/ops/bias_act.py:92-121, _bias_act_ref; activation
 * table :22-32).  x is [outer, nb, inner]; b has nb entries or is NULL.
 * act: 1 linear, 2 relu, 3 lrelu, 4 tanh, 5 sigmoid, 6 elu, 7 selu,
 * 8 softplus, 9 swish.  clamp < 0 = none.
 * ---------------------------------------------------------------------- */
#define DEFINE_BIAS_ACT(NAME, T, EXP, LOG1P, TANH)                                         \
ORACLE_API void NAME(const T* x, const T* b, T* y, int64_t outer, int64_t nb,              \
                     int64_t inner, int act, double alpha, double gain, double clamp) {    \
    const T selu_s = (T)1.0507009873554804934193349852946;                                 \
    const T selu_a = (T)1.6732632423543772848170429916717;                                 \
    const int64_t n = outer * nb * inner;                                                  \
    _Pragma("omp parallel for schedule(static)")                                           \
    for (int64_t i = 0; i < n; i++) {                                                      \
        T v = x[i];                                                                        \
        if (b) v += b[(i / inner) % nb];                                                   \
        switch (act) {                                                                     \
            case 1: break;                                                                 \
            case 2: v = v > 0 ? v : 0; break;                                              \
            case 3: v = v > 0 ? v : v * (T)alpha; break;                                   \
            case 4: v = TANH(v); break;                                                    \
            case 5: v = (T)1 / ((T)1 + EXP(-v)); break;                                    \
            case 6: v = v > 0 ? v : EXP(v) - (T)1; break;                                  \
            case 7: v = selu_s * (v > 0 ? v : selu_a * (EXP(v) - (T)1)); break;            \
            case 8: v = v > (T)20 ? v : LOG1P(EXP(v)); break;                              \
            case 9: v = v / ((T)1 + EXP(-v)); break;                                       \
        }                                                                                  \
        if (gain != 1.0) v *= (T)gain;                                                     \
        if (clamp >= 0) { if (v > (T)clamp) v = (T)clamp; if (v < -(T)clamp) v = -(T)clamp; } \
        y[i] = v;                                                                          \
    }                                                                                      \
}
DEFINE_BIAS_ACT(sg3o_bias_act_f32, float, expf, log1pf, tanhf)
DEFINE_BIAS_ACT(sg3o_bias_act_f64, double, exp, log1p, tanh)

/* ------------------------------------------------------------------------
 * filtered_lrelu  (torch_utils/ops/filtered_lrelu.py:122-154,
 * _filtered_lrelu_ref): bias (:146) -> upfirdn2d(up, padding, gain=up^2)
 * (:147) -> lrelu * gain, clamp (:148) -> upfirdn2d(down) (:149).
 * fu/fd: sep != 0 -> 1-D taps applied along x then y, each with gain^(1/2)
 * (torch_utils/ops/upfirdn2d.py:197,206-208); otherwise full [n,n] filters.
 * x: [nc, xh, xw], b: [c] or NULL (plane p uses b[p % c]); y: [nc, yh, yw].
 * Materialises the whole upsampled buffer like the reference does.
 * ---------------------------------------------------------------------- */
#define DEFINE_FLRELU(NAME, T, UPFIRDN, SQRT)                                              \
ORACLE_API int NAME(const T* x, const float* fu, const float* fd, const T* b, T* y,        \
                    int nc, int c, int xh, int xw, int up, int down,                       \
                    int fu_n, int fu_sep, int fd_n, int fd_sep,                            \
                    int px0, int px1, int py0, int py1,                                    \
                    double gain, double slope, double clamp, int flip) {                   \
    const int cw = xw * up + px0 + px1 - (fu_n - 1);                                       \
    const int ch = xh * up + py0 + py1 - (fu_n - 1);                                       \
    if (cw < fd_n || ch < fd_n) return -2;                                                 \
    const size_t nin = (size_t)nc * xh * xw;                                               \
    T* xb = (T*)malloc(sizeof(T) * nin);                                                   \
    for (size_t i = 0; i < nin; i++) {                                                     \
        size_t p = i / ((size_t)xh * xw);                                                  \
        xb[i] = x[i] + (b ? b[p % c] : (T)0);                                              \
    }                                                                                      \
    T* u = (T*)malloc(sizeof(T) * (size_t)nc * ch * cw);                                   \
    const double g2 = (double)up * up;                                                     \
    if (fu_sep) {                                                                          \
        T* t = (T*)malloc(sizeof(T) * (size_t)nc * xh * cw);                               \
        UPFIRDN(xb, fu, t, nc, xh, xw, 1, fu_n, up, 1, 1, 1, px0, px1, 0, 0, flip, SQRT(g2)); \
        UPFIRDN(t, fu, u, nc, xh, cw, fu_n, 1, 1, up, 1, 1, 0, 0, py0, py1, flip, SQRT(g2));  \
        free(t);                                                                           \
    } else {                                                                               \
        UPFIRDN(xb, fu, u, nc, xh, xw, fu_n, fu_n, up, up, 1, 1, px0, px1, py0, py1, flip, g2); \
    }                                                                                      \
    free(xb);                                                                              \
    const size_t nu = (size_t)nc * ch * cw;                                                \
    _Pragma("omp parallel for schedule(static)")                                           \
    for (size_t i = 0; i < nu; i++) {                                                      \
        T v = u[i];                                                                        \
        v = v > 0 ? v : v * (T)slope;                                                      \
        if (gain != 1.0) v *= (T)gain;                                                     \
        if (clamp >= 0) { if (v > (T)clamp) v = (T)clamp; if (v < -(T)clamp) v = -(T)clamp; } \
        u[i] = v;                                                                          \
    }                                                                                      \
    if (fd_sep) {                                                                          \
        const int tw = (cw - fd_n) / down + 1;                                             \
        T* t = (T*)malloc(sizeof(T) * (size_t)nc * ch * tw);                               \
        UPFIRDN(u, fd, t, nc, ch, cw, 1, fd_n, 1, 1, down, 1, 0, 0, 0, 0, flip, 1.0);      \
        UPFIRDN(t, fd, y, nc, ch, tw, fd_n, 1, 1, 1, 1, down, 0, 0, 0, 0, flip, 1.0);      \
        free(t);                                                                           \
    } else {                                                                               \
        UPFIRDN(u, fd, y, nc, ch, cw, fd_n, fd_n, 1, 1, down, down, 0, 0, 0, 0, flip, 1.0); \
    }                                                                                      \
    free(u);                                                                               \
    return 0;                                                                              \
}
DEFINE_FLRELU(sg3o_filtered_lrelu_f32, float, sg3o_upfirdn2d_f32, sqrt)
DEFINE_FLRELU(sg3o_filtered_lrelu_f64, double, sg3o_upfirdn2d_f64, sqrt)

/* ------------------------------------------------------------------------
 * modulated_conv2d  (models/stylegan3/networks_stylegan3.py:24-63)
 *   pre-normalise w and s when demodulating (:40-42), modulate the weights
 *   per sample (:45-46), demodulate (:50-51), input gain (:54-56), grouped
 *   correlation with zero padding `pad` (:59-62).
 * x [n, ci, h, w], w [co, ci, k, k], s [n, ci], input_gain scalar.
 * y [n, co, h + 2*pad - k + 1, w + 2*pad - k + 1].
 * ---------------------------------------------------------------------- */
#define DEFINE_MODCONV(NAME, T, SQRT)                                                      \
ORACLE_API void NAME(const T* x, const T* w, const T* s, T* y, int n, int ci, int co,      \
                     int h, int wd, int k, int pad, int demodulate, double input_gain) {   \
    const int oh = h + 2 * pad - k + 1, ow = wd + 2 * pad - k + 1;                         \
    const int kk = k * k;                                                                  \
    T* wn = (T*)malloc(sizeof(T) * (size_t)co * ci * kk);                                  \
    T* sn = (T*)malloc(sizeof(T) * (size_t)n * ci);                                        \
    memcpy(wn, w, sizeof(T) * (size_t)co * ci * kk);                                       \
    memcpy(sn, s, sizeof(T) * (size_t)n * ci);                                             \
    if (demodulate) {                                                                      \
        for (int o = 0; o < co; o++) {                                                     \
            T m = 0; T* wo = wn + (size_t)o * ci * kk;                                     \
            for (int j = 0; j < ci * kk; j++) m += wo[j] * wo[j];                          \
            T r = (T)1 / SQRT(m / (T)(ci * kk));                                           \
            for (int j = 0; j < ci * kk; j++) wo[j] *= r;                                  \
        }                                                                                  \
        T m = 0; for (int j = 0; j < n * ci; j++) m += sn[j] * sn[j];                      \
        T r = (T)1 / SQRT(m / (T)(n * ci));                                                \
        for (int j = 0; j < n * ci; j++) sn[j] *= r;                                       \
    }                                                                                      \
    T* wm = (T*)malloc(sizeof(T) * (size_t)n * co * ci * kk);                              \
    for (int b = 0; b < n; b++) for (int o = 0; o < co; o++) {                             \
        T* d = wm + ((size_t)b * co + o) * ci * kk;                                        \
        const T* wo = wn + (size_t)o * ci * kk;                                            \
        T acc = 0;                                                                         \
        for (int i = 0; i < ci; i++) for (int t = 0; t < kk; t++) {                        \
            T v = wo[i * kk + t] * sn[b * ci + i]; d[i * kk + t] = v; acc += v * v;        \
        }                                                                                  \
        T dc = demodulate ? (T)1 / SQRT(acc + (T)1e-8) : (T)1;                             \
        for (int j = 0; j < ci * kk; j++) d[j] *= dc * (T)input_gain;                      \
    }                                                                                      \
    _Pragma("omp parallel for collapse(2) schedule(static)")                               \
    for (int b = 0; b < n; b++) for (int o = 0; o < co; o++) {                             \
        const T* d = wm + ((size_t)b * co + o) * ci * kk;                                  \
        T* yp = y + ((size_t)b * co + o) * oh * ow;                                        \
        for (int j = 0; j < oh * ow; j++) yp[j] = 0;                                       \
        for (int i = 0; i < ci; i++) {                                                     \
            const T* xp = x + ((size_t)b * ci + i) * h * wd;                               \
            for (int ky = 0; ky < k; ky++) for (int kx = 0; kx < k; kx++) {                \
                const T wv = d[i * kk + ky * k + kx];                                      \
                for (int oy = 0; oy < oh; oy++) {                                          \
                    int iy = oy + ky - pad; if (iy < 0 || iy >= h) continue;               \
                    int x0 = pad - kx; if (x0 < 0) x0 = 0;                                 \
                    int x1 = wd + pad - kx; if (x1 > ow) x1 = ow;                          \
                    const T* xr = xp + (size_t)iy * wd + (kx - pad);                       \
                    T* yr = yp + (size_t)oy * ow;                                          \
                    for (int ox = x0; ox < x1; ox++) yr[ox] += wv * xr[ox];                \
                }                                                                          \
            }                                                                              \
        }                                                                                  \
    }                                                                                      \
    free(wm); free(sn); free(wn);                                                          \
}
DEFINE_MODCONV(sg3o_modulated_conv2d_f32, float, sqrtf)
DEFINE_MODCONV(sg3o_modulated_conv2d_f64, double, sqrt)

/* ------------------------------------------------------------------------
 * plain NCHW correlation (torch.nn.functional.conv2d, the op behind the
 * encoder's convolutions: models/setgan/encoder/encoders/helpers.py:98-120),
 * stride s, zero padding p, optional bias.
 * ---------------------------------------------------------------------- */
ORACLE_API void sg3o_conv2d_f32(const float* x, const float* w, const float* bias, float* y,
                                int n, int ci, int co, int h, int wd, int k, int stride, int pad) {
    const int oh = (h + 2 * pad - k) / stride + 1, ow = (wd + 2 * pad - k) / stride + 1;
    #pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < n; b++) for (int o = 0; o < co; o++) {
        float* yp = y + ((size_t)b * co + o) * oh * ow;
        for (int j = 0; j < oh * ow; j++) yp[j] = bias ? bias[o] : 0.f;
        for (int i = 0; i < ci; i++) {
            const float* xp = x + ((size_t)b * ci + i) * h * wd;
            const float* wp = w + ((size_t)o * ci + i) * k * k;
            for (int ky = 0; ky < k; ky++) for (int kx = 0; kx < k; kx++) {
                const float wv = wp[ky * k + kx];
                for (int oy = 0; oy < oh; oy++) {
                    int iy = oy * stride + ky - pad; if (iy < 0 || iy >= h) continue;
                    for (int ox = 0; ox < ow; ox++) {
                        int ix = ox * stride + kx - pad; if (ix < 0 || ix >= wd) continue;
                        yp[(size_t)oy * ow + ox] += wv * xp[(size_t)iy * wd + ix];
                    }
                }
            }
        }
    }
}
