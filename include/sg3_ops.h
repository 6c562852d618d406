/*
 * sg3_ops.h -- C ABI of libsg3hip.so, the MI355X (gfx950) replacement for the
 * native plugins behind krylea/stylegan3-editing's `torch_utils.ops` package.
 *
 * Boundary rules (all entry points):
 *   - plain C: raw device pointers, sizes, strides; no torch / C++ types.
 *   - the CALLER owns and allocates every buffer (inputs, outputs, sign
 *     tensors) and computes their sizes with the sg3_*_shape helpers below
 *     (host-only, no GPU needed);
 *   - every launch is asynchronous on the `hipStream_t` passed as `void*
 *     stream` (NULL = the null stream); no call synchronises, allocates or
 *     keeps state between calls, so concurrent streams are safe (the
 *     reference keeps its taps in one global __constant__ buffer and is not:
 *     reference torch_utils/ops/filtered_lrelu.py:216-217);
 *   - return value: SG3_OK (0) on success, SG3_NO_KERNEL (-1) when no
 *     specialised kernel exists for the (up, down, taps) tuple -- the same
 *     meaning as the reference's rc (reference torch_utils/ops/
 *     filtered_lrelu.cpp:52-56) -- and < -1 for an invalid argument or a HIP
 *     launch error; sg3_last_error() returns a thread-local message.
 *
 * Each function cites the reference interface it replaces (paths relative to
 * the reference tree).
 */
#ifndef SG3_OPS_H
#define SG3_OPS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SG3_ABI_VERSION 1

/* exported from libsg3hip.so (the library is built with -fvisibility=hidden) */
#if defined(__GNUC__)
#define SG3_API __attribute__((visibility("default")))
#else
#define SG3_API
#endif

enum {
    SG3_OK          =  0,
    SG3_NO_KERNEL   = -1,  /* no specialised kernel: caller uses the generic composition */
    SG3_BAD_ARG     = -2,
    SG3_HIP_ERROR   = -3,
};

/* element types of activation tensors */
enum {
    SG3_F32 = 0,
    SG3_F16 = 1,
    SG3_F64 = 2,   /* bias_act / upfirdn2d / filtered_lrelu_act only */
};

SG3_API int         sg3_abi_version(void);
SG3_API const char* sg3_last_error(void);
/* number of HIP devices visible to the library (0 on a CPU-only host; never fails) */
SG3_API int         sg3_device_count(void);

/* ------------------------------------------------------------------------
 * filtered_lrelu  -- replaces filtered_lrelu_plugin.filtered_lrelu
 *   (torch_utils/ops/filtered_lrelu.cpp:16-209; parameter block
 *    torch_utils/ops/filtered_lrelu.h:14-53).
 *
 *   y = down_fd( clamp( lrelu( up_fu(x + b) * up^2 * gain ) ) )
 *
 * Shapes are NCHW; strides are in ELEMENTS (not bytes), order n,c,h,w.
 * fu/fd are float32 device arrays.  fuH == 0 (fdH == 0) means a separable
 * filter of fuW (fdW) taps applied along both axes; otherwise a full
 * fuH x fuW filter with row stride fuW.
 * signs: uint8 [N, C, sH, sWbytes], 2 bits per element of the upsampled
 * buffer (bit0 = negative, value 2 = clamped), 4 elements per byte along x
 * (torch_utils/ops/filtered_lrelu.cpp:87-94).  NULL when unused.
 * ---------------------------------------------------------------------- */
typedef struct sg3_filtered_lrelu_params {
    const void*    x;          /* [N,C,xH,xW] */
    void*          y;          /* [N,C,yH,yW] */
    const void*    b;          /* [C] bias, same dtype as x, or NULL (no bias) */
    uint8_t*       s;          /* sign tensor or NULL */
    const float*   fu;         /* upsampling taps */
    const float*   fd;         /* downsampling taps */
    int32_t        dtype;      /* SG3_F32 | SG3_F16 */
    int32_t        N, C;
    int32_t        xH, xW;
    int32_t        yH, yW;
    int64_t        xStride[4]; /* elements, n,c,h,w */
    int64_t        yStride[4];
    int64_t        bStride;
    int32_t        up, down;
    int32_t        fuW, fuH;   /* fuH == 0: separable */
    int32_t        fdW, fdH;
    int32_t        px0, py0;   /* left / top padding w.r.t. the upsampled image */
    int32_t        sH;         /* sign tensor height (rows) */
    int32_t        sWbytes;    /* sign tensor row pitch in bytes */
    int32_t        sx, sy;     /* offset between upsampled buffer and sign tensor */
    int32_t        swLimit;    /* active width of the sign tensor in bytes */
    float          gain;
    float          slope;
    float          clamp;      /* +inf = no clamp */
    int32_t        flip;       /* 1 = correlation (flip_filter=True) */
    int32_t        writeSigns;
    int32_t        readSigns;
    float*         ySumPartial; /* optional, readSigns calls only: [N*C, sg3_filtered_lrelu_sum_slots(...)] per-workgroup sums of
                                 * the outputs of each plane.  The caller adds them up: the bias gradient db = dx.sum([0,2,3])
                                 * of the adjoint pass (torch_utils/ops/filtered_lrelu.py:266-267) without re-reading dx */
    int32_t        fdMirror;    /* caller's promise about a 2-D down filter: fd[r][c] == fd[r][fdW-1-c] bit for bit (true of the
                                 * radial filters design_lowpass_filter builds, networks_stylegan3.py:370-391).  The kernel then
                                 * adds the two samples that share a tap before multiplying: 42 instead of 72 packed operations per
                                 * upsampled row and lane.  0 = no assumption */
    float*         yAbsMaxPartial; /* optional, readSigns calls only, same shape as ySumPartial: per-workgroup max |output|.  Their
                                 * maximum is max |dx| of the adjoint pass: the operand bound the split-precision gradient kernels
                                 * of the convolution in front need (sg3_conv2d_wgrad's amax mode), without a pass over dx */
} sg3_filtered_lrelu_params;

SG3_API int sg3_filtered_lrelu(const sg3_filtered_lrelu_params* p, void* stream);

/* number of partial sums per (n,c) plane that sg3_filtered_lrelu writes to ySumPartial for this output shape */
SG3_API int sg3_filtered_lrelu_sum_slots(int N, int C, int yH, int yW, int down);

/* db[c] = sum over (n, slot) of sumPartial[n][c][slot] (float32 [C]) and, when given, absMax[0] = max over absMaxPartial: the
 * per-workgroup values of one readSigns call folded in one small launch (fixed summation order) */
SG3_API int sg3_filtered_lrelu_finish_partials(const float* sumPartial, const float* absMaxPartial, int N, int C, int slots,
                                               float* db, float* absMax, void* stream);

/* Host-only query: how many (n,c) planes one wave of the streaming kernel works on for this call -- 2 for the plain forward on
 * narrow dense planes (output at most 54 columns wide, even C: the 36^2 .. 52^2 layers); 3 = mixed: rows cut into 120-column
 * strips with one plane per wave plus a remainder strip of at most 54 columns with two (the 148^2 .. 532^2 layers); 1 otherwise;
 * 0 when the call does not take the streaming kernel.  For tests and profiling; no reference counterpart. */
SG3_API int sg3_filtered_lrelu_planes_per_wave(const sg3_filtered_lrelu_params* p);

/* 1 when sg3_filtered_lrelu has a fused kernel for this tuple (host-only
 * query; mirrors the reference's choose_filtered_lrelu_kernel test call,
 * torch_utils/ops/filtered_lrelu.cpp:46-56). */
SG3_API int sg3_filtered_lrelu_has_kernel(int up, int down, int fuW, int fuH, int fdW, int fdH);

/* Output / sign-tensor geometry exactly as the reference host code derives it
 * (torch_utils/ops/filtered_lrelu.cpp:59-98).  Returns SG3_BAD_ARG when the
 * reference would raise.  Any out pointer may be NULL. */
SG3_API int sg3_filtered_lrelu_shape(int xH, int xW, int up, int down,
                             int fuW, int fuH, int fdW, int fdH,
                             int px0, int px1, int py0, int py1,
                             int* yH, int* yW,
                             int* sH, int* sWbytes, int* swLimit);

/* ------------------------------------------------------------------------
 * filtered_lrelu_act  -- replaces filtered_lrelu_plugin.filtered_lrelu_act_
 *   (torch_utils/ops/filtered_lrelu.cpp:213-290; filtered_lrelu.h:55-68).
 * In-place gain * lrelu * clamp on x with optional sign write / read.
 * sW is the sign tensor width in ELEMENTS (multiple of 16).
 * ---------------------------------------------------------------------- */
typedef struct sg3_filtered_lrelu_act_params {
    void*          x;          /* [N,C,H,W], modified in place */
    uint8_t*       s;          /* [N,C,sH,sW/4] or NULL */
    int32_t        dtype;      /* SG3_F32 | SG3_F16 | SG3_F64 */
    int32_t        N, C, H, W;
    int64_t        xStride[4]; /* elements */
    int32_t        sH, sW;
    int32_t        sx, sy;
    float          gain, slope, clamp;
    int32_t        writeSigns;
    int32_t        readSigns;
} sg3_filtered_lrelu_act_params;

SG3_API int sg3_filtered_lrelu_act(const sg3_filtered_lrelu_act_params* p, void* stream);

/* ------------------------------------------------------------------------
 * upfirdn2d  -- replaces upfirdn2d_plugin.upfirdn2d
 *   (torch_utils/ops/upfirdn2d.cpp:16-98; upfirdn2d.h:14-40).
 * f is a rank-2 float32 filter [fH, fW] with element strides fStride[2]
 * (h, w).  Separable filters are two calls, as in the reference Python
 * (torch_utils/ops/upfirdn2d.py:244-246).
 * ---------------------------------------------------------------------- */
typedef struct sg3_upfirdn2d_params {
    const void*    x;          /* [N,C,xH,xW] */
    const float*   f;          /* [fH,fW] */
    void*          y;          /* [N,C,yH,yW] */
    int32_t        dtype;      /* SG3_F32 | SG3_F16 | SG3_F64 */
    int32_t        N, C, xH, xW, yH, yW;
    int64_t        xStride[4];
    int64_t        yStride[4];
    int32_t        fH, fW;
    int64_t        fStride[2];
    int32_t        upx, upy, downx, downy;
    int32_t        padx0, pady0;
    int32_t        flip;
    float          gain;
} sg3_upfirdn2d_params;

SG3_API int sg3_upfirdn2d(const sg3_upfirdn2d_params* p, void* stream);

SG3_API int sg3_upfirdn2d_shape(int xH, int xW, int fH, int fW,
                        int upx, int upy, int downx, int downy,
                        int padx0, int padx1, int pady0, int pady1,
                        int* yH, int* yW);

/* ------------------------------------------------------------------------
 * bias_act  -- replaces bias_act_plugin.bias_act
 *   (torch_utils/ops/bias_act.cpp:32-92; bias_act.h:12-31; kernel semantics
 *    torch_utils/ops/bias_act.cu:23-147).
 * x is dense with `sizeX` elements; b (or NULL) has sizeB elements and
 * element i of x uses b[(i / stepB) % sizeB].  grad = 0 forward, 1 first
 * derivative, 2 second derivative.  act = 1..9 in the reference's order
 * (linear, relu, lrelu, tanh, sigmoid, elu, selu, softplus, swish).
 * clamp < 0 = none.
 * ---------------------------------------------------------------------- */
typedef struct sg3_bias_act_params {
    const void*    x;
    const void*    b;
    const void*    xref;
    const void*    yref;
    const void*    dy;
    void*          y;
    int32_t        dtype;      /* SG3_F32 | SG3_F16 | SG3_F64 */
    int32_t        grad;
    int32_t        act;
    float          alpha, gain, clamp;
    int64_t        sizeX;
    int32_t        sizeB;
    int32_t        stepB;
} sg3_bias_act_params;

SG3_API int sg3_bias_act(const sg3_bias_act_params* p, void* stream);

/* ------------------------------------------------------------------------
 * fourier_features -- the input features of SynthesisInput.forward
 *   (models/stylegan3/networks_stylegan3.py:236-241), channels-first:
 *
 *   out[n,c,y,x] = sin((grid[y,x,0]*freqs[n,c,0] + grid[y,x,1]*freqs[n,c,1]
 *                       + phases[n,c]) * 2*pi) * amps[n,c]
 *
 * `grid` is the sampling grid the caller built (F.affine_grid, :233-234),
 * `freqs` / `phases` / `amps` the transformed frequencies, phases and
 * amplitudes (:222-230).  Operation order as in the reference's torch ops,
 * so the result is bit-identical to them.  float32 only.
 * ---------------------------------------------------------------------- */
typedef struct sg3_fourier_params {
    const float*   grid;       /* [H,W,2] */
    const float*   freqs;      /* [N,C,2] */
    const float*   phases;     /* [N,C] */
    const float*   amps;       /* [N,C] */
    float*         out;        /* [N,C,H,W] */
    int32_t        N, C, H, W;
} sg3_fourier_params;

SG3_API int sg3_fourier_features(const sg3_fourier_params* p, void* stream);

/* ------------------------------------------------------------------------
 * input_transform -- the per-sample frequencies / phases / amplitudes of
 *   SynthesisInput.forward (models/stylegan3/networks_stylegan3.py:204-230): t' = t / |t[:2]| (when `normalise`; :206),
 *   M = R(t') @ T(t') @ user (:213-221), phases + freqs @ M[:2,2:], freqs @ M[:2,:2] (:224-225), amplitude fade-out (:228).
 *   One launch instead of ~35 small torch ops; products are formed k-ascending with one rounding per step like the
 *   reference's fp32 matmuls (differences, if any, are single roundings).  float32.
 * ---------------------------------------------------------------------- */
typedef struct sg3_input_transform_params {
    const float*   t;          /* [N,4] affine output (r_c, r_s, t_x, t_y) */
    const float*   user;       /* [3,3] (userStrideN = 0) or [N,3,3] (userStrideN = 9): SynthesisInput.transform */
    const float*   freqs;      /* [C,2] */
    const float*   phases;     /* [C] */
    float*         outFreqs;   /* [N,C,2] */
    float*         outPhases;  /* [N,C] */
    float*         outAmps;    /* [N,C] */
    int32_t        N, C;
    int32_t        normalise;  /* 1: divide t by |t[:2]| first (t straight from the affine layer) */
    int32_t        userStrideN;
    float          bandwidth, samplingRate;
} sg3_input_transform_params;

SG3_API int sg3_input_transform(const sg3_input_transform_params* p, void* stream);

/* ------------------------------------------------------------------------
 * affine_batch -- the affine layers (FullyConnectedLayer, linear activation) of every synthesis layer in one launch:
 *   out_j[n, i] = (ws[n, wsIndex[j], :] . weight[rowStart[j] + i, :] + bias[rowStart[j] + i]) * scale[rowStart[j] + i]
 *   (networks_stylegan3.py:88-96 per layer, called at :205 and :349; the ToRGB gain of :350-352 is `scale`).
 *   `weight` holds all layers' matrices one after the other with weight_gain already applied (`weight * weight_gain`, one
 *   rounding, as :89), `bias` likewise (`bias * bias_gain`).  Layer j's result is the dense block [N, C_j] at
 *   out + N * rowStart[j].  float32; the dot product is accumulated in another order than the BLAS call's (1e-7 relative).
 * ---------------------------------------------------------------------- */
typedef struct sg3_affine_batch_params {
    const float*   ws;         /* latents: element (n, l, k) at ws[n * wsStrideN + l * wsStrideL + k] */
    int64_t        wsStrideN, wsStrideL;
    const float*   weight;     /* [rows, wDim] */
    const float*   bias;       /* [rows] or NULL */
    const float*   scale;      /* [rows] or NULL (= 1) */
    const int32_t* rowStart;   /* [layers + 1] DEVICE array: first row of each layer, rowStart[layers] = rows */
    const int32_t* wsIndex;    /* [layers] DEVICE array: which latent each layer reads */
    float*         out;        /* [N * rows] */
    int32_t        N, wDim, layers, rows;
} sg3_affine_batch_params;

SG3_API int sg3_affine_batch(const sg3_affine_batch_params* p, void* stream);

/* ------------------------------------------------------------------------
 * modulated_conv2d -- replaces the grouped F.conv2d behind
 *   models/stylegan3/networks_stylegan3.py:24-63 (modulated_conv2d), reached
 *   through torch_utils/ops/conv2d_gradfix.py:36-39.
 *
 *   out[n,o,:,:] = dcoef[n,o] * sum_{i,ky,kx} wn[o,i,ky,kx] *
 *                  (x[n,i,:,:] * sIn[n,i]) (zero padded by `pad`)
 *
 * i.e. the per-sample modulation is applied to the INPUT channels and the
 * demodulation to the OUTPUT channels, so the whole batch shares one weight
 * matrix and runs as a single implicit GEMM (M = O, N = batch*pixels,
 * K = I*k*k) on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
 * Two calls:
 *   sg3_modulated_conv2d_prep  -> wPacked, sIn, dcoef   (tiny)
 *   sg3_modulated_conv2d       -> out
 * x / out are NCHW contiguous, dtype f32 or f16; accumulation is fp32.
 * outH = H + 2*pad - k + 1.  k is 1 or 3.
 * ---------------------------------------------------------------------- */

/* arithmetic of the implicit GEMM */
enum {
    SG3_CONV_FP32  = 0,  /* v_mfma_f32_32x32x2_f32: exact fp32 products */
    SG3_CONV_F16X3 = 1,  /* operands split x = hi + lo into two fp16 halves; hi*hi + hi*lo + lo*hi on
                          * v_mfma_f32_32x32x16_f16 with fp32 accumulation: every retained product is exact, the
                          * dropped lo*lo term is 2^-22 relative (fp32-equivalent), 5.3x the fp32 MFMA rate.
                          * needs a bound on |x| (xBound) to keep the halves in fp16 range */
    SG3_CONV_F16   = 2,  /* operands rounded to fp16 once, one v_mfma_f32_32x32x16_f16 per K step, fp32 accumulation:
                          * the arithmetic of the reference's fp16 layers (networks_stylegan3.py:59-62 with fp16 x and
                          * w.to(x.dtype)); same packing, xBound and power-of-two rescale as SG3_CONV_F16X3 */
    SG3_CONV_F16X3_F23 = 3, /* SG3_CONV_F16X3 in a transform domain along x (Winograd F(2,3): per pair of output columns four
                          * transform points of a 3-tap vertical filter instead of 2 x 9 taps): two thirds of the matrix
                          * instructions, the same split arithmetic on the transformed operands (inputs transformed in fp32 before
                          * the split, weights at pack time).  3x3 kernels on fp32 tensors with even W, pad and output row pitch
                          * only (sg3_modconv_f23_supported); its own packed layout, so prep and convolution must agree */
    SG3_CONV_F16_F23 = 4, /* SG3_CONV_F16 in the same transform domain, for fp16 tensors (the reference's `use_fp16` layers,
                          * networks_stylegan3.py:355-366): transformed inputs and weights rounded to fp16 once, one product per K step,
                          * fp32 accumulation, fp16 output.  Same shape rules; packed layout without lo fragments (half the size) */
};

/* 1 when sg3_modulated_conv2d takes this call with precision SG3_CONV_F16X3_F23 (dtype SG3_F32) / SG3_CONV_F16_F23 (dtype SG3_F16)
 * (host-only query, no launch) */
SG3_API int sg3_modconv_f23_supported(int dtype, int I, int O, int H, int W, int k, int pad, int outRowStride);

/* Diagnostic: force the rows-per-wave of the SG3_CONV_F16X3_F23 kernel (4 | 5 | 7; 0 = the launcher's own cost model, the default)
 * for every later launch of the process; returns the previous setting.  The tile height fixes the order of a tile's sums, so it is a
 * process setting made by an explicit call (tests walk all three heights; tools/ A/B runs) -- not an environment variable read per
 * launch, which could differ between a graph capture and a later eager call.  The environment variable SG3_F23_TN seeds it once. */
SG3_API int sg3_modconv_f23_force_rows(int rows);

/* number of floats (4-byte units) of the packed weight buffer for an [O,I,k,k] weight
 * (fp32: [O][ceil(I/KC)][k*k][KC] floats; f16x3 / f16: [O][chunks][k*k][hi|lo][16] halfs, chunks = ceil(I/16), rounded up to even for k = 1; zero padded). */
SG3_API int64_t sg3_modconv_packed_floats(int O, int I, int k, int precision);

/* Demodulation coefficients and input scales
 * (models/stylegan3/networks_stylegan3.py:39-56):
 *   wn    = w * rsqrt(mean(w^2 over I,k,k))          (per O)    -> wPacked
 *   sn    = s * rsqrt(mean(s^2 over N,I))
 *   dcoef = rsqrt( sum_{i,k} (wn[o,i,k] * sn[n,i])^2 + 1e-8 )   -> dcoef [N,O]
 *   sIn   = sn * inputGain                                      -> sIn [N,I]
 * With demodulate == 0: wn = w, sn = s, dcoef is not written.
 * inputGain is a DEVICE pointer (it derives from the magnitude_ema buffer,
 * :344) read according to inputGainMode: 0 none, 1 one scalar, 2 [I], 3 [N,I].
 * `wsq` is scratch [O,I] float32 (per-(o,i) sum over taps of wn^2).
 * ---------------------------------------------------------------------- */
typedef struct sg3_modconv_prep_params {
    const float*   w;          /* [O,I,k,k] */
    const float*   s;          /* [N,I] */
    float*         wPacked;    /* sg3_modconv_packed_floats(O,I,k) floats */
    float*         wsq;        /* [O,I] scratch */
    float*         sIn;        /* [N,I] */
    float*         dcoef;      /* [N,O] (NULL allowed when demodulate == 0) */
    const float*   inputGain;  /* device pointer or NULL */
    int32_t        inputGainMode;
    int32_t        N, I, O, k;
    int32_t        demodulate;
    int32_t        precision;  /* SG3_CONV_FP32 | SG3_CONV_F16X3 | SG3_CONV_F16 */
    float          xBound;     /* f16x3 only: max |x| the conv will see (> 0).  sIn is scaled by a power of two per
                                * sample so that |x * sIn| stays below 2^15, and dcoef (required, also without
                                * demodulation) carries the inverse */
    const float*   xBoundDev;  /* optional DEVICE pointer to that bound (overrides xBound when not NULL): lets a caller
                                * derive it on the device, e.g. max |dy| of a gradient, without a host round trip */
    int32_t        reuseWeights; /* 1: wPacked and wsq already hold the result of an earlier call with the same w, k, precision
                                  * and demodulate (inference with unchanged weights): only the style pass (sIn, dcoef) runs */
} sg3_modconv_prep_params;

SG3_API int sg3_modulated_conv2d_prep(const sg3_modconv_prep_params* p, void* stream);

/* The same for `count` independent (w, s) pairs -- the layers of one Generator.synthesis call, whose styles are all known
 * before the first convolution (networks_stylegan3.py:478-485 computes them layer by layer) -- in two launches instead of
 * 2 * count.  No reference counterpart: the reference folds this arithmetic into torch ops inside modulated_conv2d (:39-56). */
SG3_API int sg3_modulated_conv2d_prep_batch(const sg3_modconv_prep_params* list, int count, void* stream);

typedef struct sg3_modconv_params {
    const void*    x;          /* [N,I,H,W] */
    const float*   wPacked;    /* from sg3_modulated_conv2d_prep */
    const float*   sIn;        /* [N,I] */
    const float*   dcoef;      /* [N,O] or NULL (no demodulation) */
    void*          out;        /* [N,O,outH,outW] */
    int32_t        dtype;      /* SG3_F32 | SG3_F16 (x and out) */
    int32_t        N, I, O, H, W;
    int32_t        k;          /* 1 or 3 */
    int32_t        pad;
    int32_t        precision;  /* must match the prep call that produced wPacked */
    /* Optional epilogue of the ToRGB form only (k = 1, pad = 0, O <= 4, SG3_CONV_FP32): the bias + clamp that
     * SynthesisLayer.forward applies next through filtered_lrelu with up = down = 1, gain = 1, slope = 1
     * (networks_stylegan3.py:352-356), optionally followed by a scale such as SynthesisNetwork's output scale (:488-489):
     * out = clamp(conv + bias[o], +-epilogueClamp) * epilogueScale -- the same values without further passes over the
     * image.  epilogueBias NULL = no epilogue; epilogueClamp < 0 = no clamp; epilogueScale 0 = 1. */
    const float*   epilogueBias;
    float          epilogueClamp;
    float          epilogueScale;
    /* Row pitch of `out` in elements; 0 = dense (outW).  The caller may allocate [N,O,outH,pitch] with pitch a multiple of 128
     * bytes and hand the [..., :outW] view on: a 1046-float row is 4184 bytes, so every 128-byte store segment of a dense output
     * straddles two cache lines (3x3 split-precision / fp16 kernels only; other forms require 0 or outW). */
    int32_t        outRowStride;
    /* Optional scratch for small grids (batch 1 on the 36^2 .. 84^2 maps: fewer tiles than CUs): the 3x3 direct kernel and the 1x1
     * GEMM kernel (fp32 tensors) then split the input channels of a tile over up to four workgroups and a second launch adds their
     * partial sums in order.  NULL / too
     * small = no split (same result up to fp32 summation order).  sg3_modconv_split_scratch_floats gives a sufficient size. */
    float*         splitScratch;
    int64_t        splitScratchFloats;
} sg3_modconv_params;

SG3_API int sg3_modulated_conv2d(const sg3_modconv_params* p, void* stream);

/* floats of splitScratch that let this call split its K loop (0 = the call would not split) */
SG3_API int64_t sg3_modconv_split_scratch_floats(const sg3_modconv_params* p);

/* ------------------------------------------------------------------------
 * conv2d_wgrad -- per-sample weight gradient of the (modulated) convolution, the part of the backward of
 *   models/stylegan3/networks_stylegan3.py:59-62 (grouped F.conv2d) that PTI needs
 *   (inversion/scripts/run_pti_images.py:130-139):
 *
 *   dW[n,o,i,ky,kx] = sum_{y,x} dy[n,o,y,x] * x[n,i,y+ky-pad,x+kx-pad]
 *
 * The pixel (K) dimension is split over nBands x nSegGroups workgroup sets (sg3_conv2d_wgrad_splits proposes the
 * counts); each writes its partial sum to partial[band*nSegGroups+group][n][tap][o][i] and the caller adds them up
 * (deterministic; no float atomics).  Operands are multiplied by *scaleX / *scaleDy (device scalars, powers of two chosen
 * by the caller so that the scaled magnitudes peak near 2^15: fp16 hi/lo split inside) and the result by 1/(scaleX*scaleDy).
 * ---------------------------------------------------------------------- */
typedef struct sg3_wgrad_params {
    const void*    x;          /* [N,I,H,W] */
    const void*    dy;         /* [N,O,OH,OW], OH = H + 2*pad - k + 1 */
    float*         partial;    /* [nBands*nSegGroups, N, k*k, O, I] */
    const float*   scaleX;     /* device scalar */
    const float*   scaleDy;    /* device scalar */
    int32_t        dtype;      /* SG3_F32 | SG3_F16 (x and dy) */
    int32_t        N, I, O, H, W;
    int32_t        k;          /* 1 or 3 */
    int32_t        pad;
    int32_t        nBands, nSegGroups;
    int32_t        scalesAreAmax; /* 1: *scaleX / *scaleDy hold max |x| / max |dy| and the kernel derives the power-of-two scales
                                   * itself (2^-ceil(log2(amax / 2^15))): saves the caller eight tiny launches per layer */
} sg3_wgrad_params;

SG3_API int sg3_conv2d_wgrad_splits(int N, int I, int O, int H, int W, int k, int pad, int* nBands, int* nSegGroups);
SG3_API int sg3_conv2d_wgrad(const sg3_wgrad_params* p, void* stream);

/* ------------------------------------------------------------------------
 * conv2d -- the plain convolutions of the ReStyle encoder (IR-SE50 backbone and
 *   GradualStyleBlock heads): replaces the torch.nn.Conv2d / BatchNorm2d / PReLU /
 *   LeakyReLU module calls of models/setgan/encoder/encoders/helpers.py:98-120,
 *   restyle_psp_encoders.py:26-28 and map2style.py:15-19 for eval-mode inference.
 *
 *   out[n,o,y,x] = act( bias[o] + sum_{i,ky,kx} w[o,i,ky,kx] *
 *                       pre(x[n,i, y*stride+ky-pad, x*stride+kx-pad]) )
 *   pre(v) = v * inScale[i] + inShift[i] inside the image, 0 in the padding
 *            (an eval-mode BatchNorm in FRONT of the convolution, helpers.py:108);
 *   a BatchNorm BEHIND the convolution is folded into w / bias by the caller.
 *   act: 0 none, 1 PReLU with per-channel slope[o], 2 leaky ReLU with slope[0].
 * Exact fp32 matrix-core arithmetic (v_mfma_f32_32x32x2_f32), NCHW contiguous
 * fp32 tensors, k in {1,3}, stride in {1,2}.
 * wPacked comes from sg3_conv2d_pack (layout [O][ceil(I/KC)][k*k][KC]).
 * ---------------------------------------------------------------------- */
typedef struct sg3_conv2d_params {
    const float*   x;          /* [N,I,H,W] */
    const float*   wPacked;    /* sg3_modconv_packed_floats(O,I,k,precision) floats, written by sg3_conv2d_pack */
    const float*   inScale;    /* [I] or NULL */
    const float*   inShift;    /* [I] or NULL (NULL = 0) */
    const float*   bias;       /* [O] or NULL */
    const float*   slope;      /* act 1: [O]; act 2: [1]; else ignored */
    float*         out;        /* [N,O,outH,outW] */
    int32_t        N, I, O, H, W;
    int32_t        k, stride, pad;
    int32_t        act;
    int32_t        precision;  /* SG3_CONV_FP32 (exact) | SG3_CONV_F16X3 (1x1 / 3x3, stride 1 or 2: fp16 x 3 split, fp32-equivalent) */
    int32_t*       rangeFlag;  /* f16x3: device int, OR-ed with 1 when an operand left the fp16 range (the result of that
                                * call is then invalid and the caller repeats it with SG3_CONV_FP32); never cleared here */
} sg3_conv2d_params;

SG3_API int sg3_conv2d(const sg3_conv2d_params* p, void* stream);

/* w [O,I,k,k] (* outScale[o] when given: a folded BatchNorm) -> packed layout */
SG3_API int sg3_conv2d_pack(const float* w, const float* outScale, float* wPacked, int O, int I, int k, int precision, void* stream);

/* ------------------------------------------------------------------------
 * modulation_backward -- dL/dw and dL/ds of modulated_conv2d's per-sample effective weights, given G = dL/dw_eff
 *   (the output of sg3_conv2d_wgrad).  The reference gets them from autograd through its weight algebra
 *   (models/stylegan3/networks_stylegan3.py:39-56: pre-normalisation of w and s, modulation, demodulation, input gain);
 *   this is the same chain in closed form, four launches instead of ~70 (see csrc/sg3_modgrad.hip).  float32, dense tensors.
 *   G is overwritten (it holds dL/dm afterwards).  The input gain is a constant here (a buffer in the reference).
 * ---------------------------------------------------------------------- */
typedef struct sg3_modgrad_params {
    float*         G;          /* [N,O,I,T] dL/dw_eff; scratch on return */
    const float*   w;          /* [O,I,T] raw weights */
    const float*   s;          /* [N,I] raw styles */
    const float*   inputGain;  /* as in sg3_modconv_prep_params, or NULL */
    int32_t        inputGainMode;
    float*         dW;         /* [O,I,T] */
    float*         dS;         /* [N,I] */
    float*         a;          /* [O] scratch */
    float*         dSn;        /* [N,I] scratch */
    int32_t        N, O, I, T; /* T = k * k taps */
    int32_t        demodulate;
} sg3_modgrad_params;

SG3_API int sg3_modulation_backward(const sg3_modgrad_params* p, void* stream);

/* The weights of the DATA-GRADIENT convolution of modulated_conv2d in one launch: the reference's backward runs the grouped
 * convolution's adjoint (autograd of networks_stylegan3.py:59-62), i.e. a convolution with the pre-normalised (:41-42, when
 * `normalise`), transposed and flipped weights:  wt[i][o][k-1-ky][k-1-kx] = w[o][i][ky][kx] * rsqrt(mean_{i,ky,kx} w[o]^2).
 * w [O,I,k,k] and wt [I,O,k,k] dense fp32.  Replaces six elementwise / reduction launches per layer and PTI step. */
SG3_API int sg3_modconv_transpose_weights(const float* w, float* wt, int O, int I, int k, int normalise, void* stream);

/* ------------------------------------------------------------------------
 * se_residual -- the tail of an IR-SE residual unit (models/setgan/encoder/encoders/helpers.py:57-73 SEModule and
 *   :117-120 bottleneck_IR_SE.forward):  out = shortcut + res * sigmoid(fc2 @ relu(fc1 @ mean_hw(res))).
 *   Two launches (plane means; gates + apply) instead of the seven of the torch op chain.  float32, res / out dense NCHW;
 *   the shortcut is addressed through element strides (the stride-2 units read a subsampled view of their input).
 * ---------------------------------------------------------------------- */
typedef struct sg3_se_params {
    const float*   res;        /* [N,C,H,W] dense: the residual branch (after its last BatchNorm) */
    const float*   shortcut;   /* [N,C,H,W] through scStride */
    int64_t        scStride[4];/* elements, n,c,h,w */
    const float*   fc1;        /* [R,C] */
    const float*   fc2;        /* [C,R] */
    float*         mean;       /* [N,C] scratch (holds mean_hw(res) afterwards) */
    float*         out;        /* [N,C,H,W] dense; may alias res */
    int32_t        N, C, H, W, R;
} sg3_se_params;

SG3_API int sg3_se_residual(const sg3_se_params* p, void* stream);

/* ----------------------------------------------------------------------
 * unfold3x3s2 -- the patch matrix of a 3x3, stride-2, padding-1 convolution, for the batched GEMMs of the GradualStyleBlock heads
 *   (models/setgan/encoder/encoders/map2style.py:8-25: Conv2d(3x3, stride 2, padding 1) + LeakyReLU per level;
 *   restyle_psp_encoders.py:26-50).  One launch per level instead of the pad / slice / stack / permute chain:
 *       dst[g][(n*OH + oy)*OW + ox][tap*C + c] = act(src[g,n,c, 2*oy + ky - 1, 2*ox + kx - 1]),  0 outside the map,
 *   tap = ky*3 + kx, OH = (IH+1)/2, OW = (IW+1)/2, act = LeakyReLU(slope) of the previous level (slope 1 = none).
 *   src is addressed through element strides, so it may be the channels-first strip a convolution wrote or the
 *   [g][n*pixels][c] rows of the previous level's GEMM.  float32.
 * ---------------------------------------------------------------------- */
typedef struct sg3_unfold_params {
    const float*   src;
    int64_t        srcStride[5];   /* elements: g, n, c, y, x */
    float*         dst;            /* [G][N*OH*OW][9*C] dense */
    int32_t        G, N, C, IH, IW;
    float          slope;
} sg3_unfold_params;

SG3_API int sg3_unfold3x3s2(const sg3_unfold_params* p, void* stream);

/* ----------------------------------------------------------------------
 * Batched small-M GEMMs of the GradualStyleBlock heads (reference models/setgan/encoder/encoders/map2style.py:8-25: the 3x3
 * stride-2 convolutions of levels 2.. as GEMMs over unfolded patches, and the closing EqualLinear, models/stylegan2/model.py
 * EqualLinear.forward; restyle_psp_encoders.py:26-50 runs n_styles heads):
 *     c[g][m][n] = sum_k lrelu(a[g][m][k], slope) * w[g][k][n] + bias[g][n]
 *   float32 in and out, split-precision fp16 x 3 arithmetic on the matrix cores (fp32-equivalent), weight-bandwidth bound.
 *   sg3_head_gemm_pack writes w ([G][K][N] float32, K % 16 == 0, N % 32 == 0) as matrix-instruction fragments into `packed`
 *   (sg3_head_gemm_packed_halfs(G,K,N) fp16 elements, -1 for unsupported shapes); *rangeFlag is OR-ed with 1 when a weight
 *   (pack) or an activation (gemm) lies outside the fp16 range: the caller then uses its fp32 path.  bias may be NULL.
 * ---------------------------------------------------------------------- */
typedef struct sg3_head_gemm_params {
    const float*   a;              /* [G][M][K] dense */
    const void*    wPacked;        /* from sg3_head_gemm_pack */
    const float*   bias;           /* [G][N] or NULL */
    float*         c;              /* [G][M][N] dense */
    int32_t*       rangeFlag;
    int32_t        G, M, K, N;
    float          slope;          /* LeakyReLU applied to a on the way in; 1 = none */
} sg3_head_gemm_params;

SG3_API long long sg3_head_gemm_packed_halfs(int32_t G, int32_t K, int32_t N);
SG3_API int sg3_head_gemm_pack(const float* w, void* packed, int32_t G, int32_t K, int32_t N, int32_t* rangeFlag, void* stream);
SG3_API int sg3_head_gemm(const sg3_head_gemm_params* p, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SG3_OPS_H */
