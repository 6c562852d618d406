// sg3_filtered_lrelu.hip -- fused bias -> upsample FIR -> gain*lrelu -> clamp -> downsample FIR for gfx950.
//
// Replaces filtered_lrelu_plugin.filtered_lrelu (reference torch_utils/ops/filtered_lrelu.cpp:16-209 and the
// tile kernel torch_utils/ops/filtered_lrelu.cu:139-1099).  The reference kernel is a CUDA thread-block tile
// pipeline (input tile -> 4 shared-memory passes, 512-1024 threads, warp-32 sign packing).  This one is built
// for CDNA4 from scratch around two observations:
//
//   * the op is on the fp32 VALU / HBM ridge (~21 FLOP/B), so the design goal is "nothing but FMAs
//     in the instruction stream": no halo recompute in the vertical direction, no LDS operand per FMA, no
//     workgroup barriers;
//   * both vertical FIR passes and the non-linearity are column-local, so they can live entirely in registers.
//
// Kernel `flrelu_stream_kernel` -- ONE WAVE (64 lanes) owns a strip of up to 120 output columns of one (n,c)
// plane and streams down a chunk of output rows:
//
//     global row i  --(+bias, zero outside the image)-->  LDS row  --H-up (polyphase, taps in SGPRs)-->
//     4 upsampled columns per lane, kept as a 6-row sliding window in VGPRs
//       --V-up (polyphase) --> U new upsampled rows in registers --> *gain, lrelu, clamp
//       --V-down: every upsampled row is scattered into 6 rotating accumulators (FD/D output rows in flight)
//     completed accumulator row --> LDS row --H-down (even/odd polyphase, ds_read_b64 pairs)--> global store
//
//   Per input row and lane that is 1 LDS row write + 8 LDS dword reads + 4 LDS writes + 24 LDS dword reads
//   against ~100 (U=2) / ~190 (U=4) VALU FMAs; the only data that ever touches LDS is one input row and one
//   output row, and because a wave only talks to itself there is no s_barrier anywhere (LDS executes a wave's
//   DS instructions in order; the compiler is held back with wavefront-scope fences).  The window / accumulator
//   rotation is resolved at compile time by unrolling 6 input rows per loop iteration.
//   Global reads are issued 3 rows ahead into registers; reads and writes are 256-B contiguous per instruction.
//   Grid = planes x row-chunks x strips, renumbered so that blocks sharing halo columns/rows sit on one XCD (L2).
//   Lane fill (plain forward): planes at most 54 columns wide run TWO per wave (template G = 2: lanes 0-31 / 32-63, own LDS
//   segments, own bias and gain); rows of 120 n + (1..54) columns are cut into n full strips, one plane per wave, plus the
//   remainder strip with two planes per wave -- in one launch of the G = 2 kernel (StreamParams.wideBlocks) where that costs
//   no occupancy, in two launches otherwise (`launch_stream`).
//
// Supported by the streaming kernel (`stream_supported` / `stream_params_ok` below are the authority): fp32 / fp16 I/O with
// unit innermost stride, slope in [0,1], and
//   forward, plain or sign-writing (training):  up 2 (12 taps) | up 4 (24 taps), separable;  down 2 with a separable 12-tap
//       filter (config T, and the critically sampled layers of config R) or a full 12x12 filter (radial layers of config R;
//       RADIAL 1|2, or 5|6 when the caller promises mirror-symmetric rows);
//   adjoint (sign-reading):  up 2 with a separable 12-tap filter or a full 12x12 filter (RADIAL 3|4: the radial layers' down
//       filter ends up on the up side), down 2 (12 taps) | down 4 (24 taps), separable.
// `flrelu_pointwise_kernel` covers up = down = 1 with 1x1 filters (the ToRGB layer).  Anything else returns SG3_NO_KERNEL and
// the caller composes upfirdn2d + filtered_lrelu_act + upfirdn2d, exactly like the reference's rc = -1 path.
//
// Algorithmic traffic: C*(xH*xW + yH*yW)*sizeof(T) bytes per image -- the roofline figure bench.py uses.
#include "sg3_common.h"
#include <cmath>
#include <cstdlib>

// rows of input the streaming kernel keeps in flight per wave (each costs NL registers per lane): three rows are ~3 x 1.5 us
// of work, several HBM latencies
#ifndef SG3_PREFETCH_ROWS
#define SG3_PREFETCH_ROWS 3
#endif
// 1: the plain forward keeps the reference's behaviour on non-finite input (a NaN activation stays a NaN through the clamp); 0
// compiles the guard out (A/B timing builds)
#ifndef SG3_NAN_GUARD
#define SG3_NAN_GUARD 1
#endif

namespace sg3 {

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_f;          // explicit LDS address space: keeps ds_* instructions
typedef __attribute__((address_space(3))) v2f lds_v2f;
typedef __attribute__((address_space(3))) v4f lds_v4f;

struct StreamParams {
    const void* x; void* y; const void* b; const float* fu; const float* fd;
    int C, xH, xW, yH, yW;
    int planes;                    // N * C
    long long xsN, xsC, xsH;      // element strides (innermost stride is 1)
    long long ysN, ysC, ysH;
    long long bStride;
    int px0, py0;
    int TW;                        // output columns per strip (<= 122)
    int ox0Base;                   // first output column of strip 0 (one-plane blocks) / of the packed strip (two-plane blocks)
    int wideBlocks;                // G = 2 kernels: the first wideBlocks (logical) blocks work on ONE plane each, strips of TW columns
                                   // from column 0; the others on two planes, columns ox0Base .. yW - 1
    int CH;                        // output rows per chunk
    int nStrips, nChunks;
    int totalBlocks;
    float gain, slope, clamp;
    int flip;
    float* ysum;                   // optional [N*C][nChunks*nStrips]: sum of this block's outputs (adjoint passes only: their bias gradient)
    float* ymax;                   // optional, same shape: max |output| of this block (adjoint passes only: the operand bound of the
                                   // convolution gradients that read the result next)
    unsigned char* s;              // sign tensor [N*C][sH][sWb] (2 bits per upsampled sample, 4 per byte), or null
    int sH, sWb, sx, sy;           // rows, bytes per row, offset of the upsampled buffer inside the sign tensor
#ifdef SG3_STAMPS
    unsigned long long* stamps;    // diagnostic build (tools/flrelu_clock.hip): per block {shader cycles, 100 MHz ticks} of the wave's life
#endif
};
#ifdef SG3_STAMPS
static unsigned long long* g_stamps = nullptr;
#endif

__device__ __forceinline__ int to_sgpr_i(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float to_sgpr(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v)));
}

// Compiler-level ordering of this wave's LDS traffic (no instruction is emitted: a wave's DS ops execute in order).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ v2f fma2(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float a) { return (v2f){a, a}; }

// acc + a * tap, where the tap is ONE half of a scalar-register pair (two taps per SGPR pair, no duplicates).
// The low half is expressible in C (the compiler emits op_sel_hi:[1,0,1]); the high half needs op_sel on the SGPR
// source, which hipcc never selects (it copies the tap to a fresh pair instead and runs out of SGPRs), hence the asm.
template <int HALF>
__device__ __forceinline__ v2f fma_tap(v2f a, v2f tapPair, v2f acc) {
    if (HALF == 0) return fma2(a, __builtin_shufflevector(tapPair, tapPair, 0, 0), acc);
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "s"(tapPair), "v"(acc));
    return r;
}
template <int HALF>
__device__ __forceinline__ v2f mul_tap(v2f a, v2f tapPair) {
    if (HALF == 0) return a * __builtin_shufflevector(tapPair, tapPair, 0, 0);
    v2f r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(r) : "v"(a), "s"(tapPair));
    return r;
}

// Radial (2-D) down filter taps: correlation taps g[k][x] = fd[flip ? k : 11-k][flip ? x : 11-x].  A filter row (12 taps)
// is fetched through the scalar cache into three SGPR quads; without flip the row sits reversed in memory and the
// packed FMA swaps the halves of its scalar pair (op_sel) instead of spending instructions on it.  The loads are
// explicit asm: left to the compiler they are hoisted (constant memory) and spill the scalar file.
struct TapRow { v4f a, b, c; };
__device__ __forceinline__ void radial_row_issue(TapRow& t, const float* fd, int row) {
    const float* src = fd + row * 12;
    asm volatile("s_load_dwordx4 %0, %3, 0x0\n\ts_load_dwordx4 %1, %3, 0x10\n\ts_load_dwordx4 %2, %3, 0x20"
                 : "=&s"(t.a), "=&s"(t.b), "=&s"(t.c) : "s"(src));
}
__device__ __forceinline__ void radial_row_wait(TapRow& t) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t.a), "+s"(t.b), "+s"(t.c));
}
// first half of a filter row (taps 0..5), all a mirror-symmetric row needs
struct TapHalf { v4f a; v2f b; };
__device__ __forceinline__ void radial_half_issue(TapHalf& t, const float* fd, int row) {
    const float* src = fd + row * 12;
    asm volatile("s_load_dwordx4 %0, %2, 0x0\n\ts_load_dwordx2 %1, %2, 0x10" : "=&s"(t.a), "=&s"(t.b) : "s"(src));
}
__device__ __forceinline__ void radial_half_wait(TapHalf (&t)[6]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(t[0].a), "+s"(t[0].b), "+s"(t[1].a), "+s"(t[1].b), "+s"(t[2].a), "+s"(t[2].b),
                                          "+s"(t[3].a), "+s"(t[3].b), "+s"(t[4].a), "+s"(t[4].b), "+s"(t[5].a), "+s"(t[5].b));
}
// pair (g[k][2q], g[k][2q+1]) of the row (memory pair q with flip; memory pair 5-q, halves swapped, without)
template <bool FLIP>
__device__ __forceinline__ v2f radial_pair(const TapRow& t, int q) {
    const int j = FLIP ? q : 5 - q;
    const v4f v = j < 2 ? t.a : (j < 4 ? t.b : t.c);
    return (j & 1) ? (v2f){v.z, v.w} : (v2f){v.x, v.y};
}
template <bool FLIP>
__device__ __forceinline__ v2f radial_fma(v2f a, v2f tq, v2f acc) {
    v2f r;
    if (FLIP) asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(tq), "v"(acc));
    else      asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(tq), "v"(acc));
    return r;
}
// a + (b.y, b.x)
__device__ __forceinline__ v2f add_swapped(v2f a, v2f b) {
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
template <bool FLIP>
__device__ __forceinline__ v2f radial_mul(v2f a, v2f tq) {
    v2f r;
    if (FLIP) asm("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "s"(tq));
    else      asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "s"(tq));
    return r;
}

// 2-D UP filter (the adjoint of the radial layers: their 12x12 down filter becomes the up filter).  acc += x * (tap pair),
// x = one half of a register pair of adjacent input samples broadcast to both result lanes, the tap pair = one SGPR pair
// of a filter row, halves exchanged when SWAP (filter row stored in the opposite order).
template <int XHALF, bool SWAP>
__device__ __forceinline__ v2f fma_x_tap(v2f xpair, v2f tapPair, v2f acc) {
    v2f r;
    if (XHALF == 0 && !SWAP) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(xpair), "s"(tapPair), "v"(acc));
    if (XHALF == 1 && !SWAP) asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(xpair), "s"(tapPair), "v"(acc));
    if (XHALF == 0 && SWAP)  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(xpair), "s"(tapPair), "v"(acc));
    if (XHALF == 1 && SWAP)  asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(xpair), "s"(tapPair), "v"(acc));
    return r;
}
// taps (G[k][2s+1], G[k][2s]) of the correlation-form filter row held in `t` (memory row k with flip, 11-k without):
// memory pair s with the halves exchanged (flip), memory pair 5-s as it stands (no flip)
template <bool FLIP>
__device__ __forceinline__ v2f up2d_pair(const TapRow& t, int s) {
    const int j = FLIP ? s : 5 - s;
    const v4f v = j < 2 ? t.a : (j < 4 ? t.b : t.c);
    return (j & 1) ? (v2f){v.z, v.w} : (v2f){v.x, v.y};
}

template <int U, int D> struct StreamCfg {
    static constexpr int FU = 6 * U, FD = 6 * D;
    static constexpr int CPL = 4;                       // upsampled columns per lane
    static constexpr int GROUPS = CPL / U;              // column groups per lane that share one input window
    static constexpr int IWS = GROUPS * 64 + 6;         // input samples a wave needs per row (+ slack)
    static constexpr int NL = (IWS + 63) / 64;          // global loads per lane per row
    static constexpr int SIN = NL * 64;                 // LDS floats for the input row
    static constexpr int SOUT = 4 + 256 + 32;           // LDS floats for the output row (+ read-ahead of the last lanes)
    static constexpr int MAXTW = (256 - FD) / D;        // output columns a wave's 256 upsampled columns can complete
    // packed mode (two narrow planes per wave, 32 lanes each): input-row / output-row LDS segment of one plane, and the widest
    // output 32 lanes complete
    static constexpr int PSEG = GROUPS * 32 + 8;        // 72 (up 2) | 40 (up 4) input samples; 2 PSEG <= SIN
    static constexpr int POSEG = 4 * 32 + 16;           // 144 floats; 2 POSEG <= SOUT
    // output column 2l + 1 reads the exchanged row up to slot 4 + 4l + 13, and the 32 lanes write slots 4 - delta .. 131 - delta
    // (delta <= 3): l <= 26
    static constexpr int PACKTW = 2 * 27;               // 54 output columns (down 2)
    static_assert(2 * PSEG <= SIN && 2 * POSEG <= SOUT, "packed LDS rows fit the unpacked ones");
};

// element <-> fp32 through raw buffer instructions: the hardware range check (offset >= num_records reads 0 /
// drops the store) replaces every per-lane bounds branch
template <typename T> struct bufio;
template <> struct bufio<float> {
    static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, int byteOff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byteOff, 0, 0));
    }
    static __device__ __forceinline__ void st1(__amdgpu_buffer_rsrc_t r, int byteOff, float v) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, byteOff, 0, 0);
    }
    static __device__ __forceinline__ void st2(__amdgpu_buffer_rsrc_t r, int byteOff, float a, float b) {
        typedef unsigned u2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64((u2){__builtin_bit_cast(unsigned, a), __builtin_bit_cast(unsigned, b)}, r, byteOff, 0, 0);
    }
};
template <> struct bufio<_Float16> {
    static __device__ __forceinline__ float ld(__amdgpu_buffer_rsrc_t r, int byteOff) {
        return (float)__builtin_bit_cast(_Float16, __builtin_amdgcn_raw_buffer_load_b16(r, byteOff, 0, 0));
    }
    static __device__ __forceinline__ void st1(__amdgpu_buffer_rsrc_t r, int byteOff, float v) {
        __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(unsigned short, (_Float16)v), r, byteOff, 0, 0);
    }
    static __device__ __forceinline__ void st2(__amdgpu_buffer_rsrc_t r, int byteOff, float a, float b) {
        const unsigned lo = __builtin_bit_cast(unsigned short, (_Float16)a), hi = __builtin_bit_cast(unsigned short, (_Float16)b);
        __builtin_amdgcn_raw_buffer_store_b32(lo | (hi << 16), r, byteOff, 0, 0);
    }
};

template <typename T, int U, int D>
struct WaveState {
    typedef StreamCfg<U, D> Cfg;
    v2f w[6][2];                  // sliding window of H-upsampled rows: [slot][column pair]
    v2f xw[6][4];                 // 2-D up filter: sliding window of RAW input rows, 8 samples per lane as adjacent pairs
    v2f acc[6][2];                // output rows in flight: [slot][column pair]
    float pre[SG3_PREFETCH_ROWS][Cfg::NL];   // prefetched input samples (+bias) for the next rows
    float bcol[Cfg::NL];          // bias where this lane's input column exists, else 0
    int coff[Cfg::NL];            // byte offset of this lane's input columns inside a row
    v2f tuP[Cfg::FU / 2];         // up taps (x U) in REVERSED pairs (tu[2m+1], tu[2m]): the H-up column-pair operands
                                  // as they stand, and the source of the V-up tap splats (odd tap = low half)
    v2f tdP[Cfg::FD / 2];         // down taps (td[2m], td[2m+1]): V-down splats and H-down even/odd pairs
    unsigned sg[SG3_PREFETCH_ROWS][U];       // sign-read mode: prefetched sign bytes (this lane's byte | next byte << 8) per upsampled row
    float osum;                   // running sum of the outputs this lane stored (only kept when p.ysum is given)
    float omax;                   // running max |output| (only kept when p.ymax is given)
    int soff;                     // sign modes: byte offset, inside a sign row, of the byte holding this lane's first column
    int sq;                       // sign modes: position (0..3) of that column inside its byte (wave-uniform)
    int inBase, outBase;          // packed mode: this lane's window start in the input LDS row / its 4 samples' place in the output LDS row
    int ooff[2];                  // packed mode: byte offsets of this lane's two output columns from the first plane's row (out of range: none)
};

// SIGNS: 0 = plain forward; 1 = forward that also writes the sign tensor (training); 2 = adjoint pass: the stored signs
// replace the nonlinearity (gradient of lrelu + clamp).
// G: planes per wave.  2 = packed mode for narrow planes (yW <= PACKTW, one strip): lanes 0-31 work on plane 2k, lanes 32-63 on plane
// 2k + 1, with their own segments of the two LDS rows -- the 36^2 .. 52^2 layers use 21 .. 29 lanes of a wave otherwise, and the
// kernel's time there is its instruction count.  Plain forward only; tensors dense over (n, c) so that plane 2k + 1 sits one
// channel stride after plane 2k (also across the end of an image).
template <typename T, int U, int D, int VPH, int RADIAL, int SIGNS, int G = 1>
struct Stream {
    typedef StreamCfg<U, D> Cfg;
    typedef WaveState<T, U, D> State;

    // issue the loads of input row `i` into st.pre[slot]; rows outside the image get a zero-length descriptor
    static __device__ __forceinline__ void prefetch(State& st, int slot, const StreamParams& p, const T* __restrict__ plane,
                                                    const unsigned char* __restrict__ splane, int i) {
        const bool rowOk = (unsigned)i < (unsigned)p.xH;
        const T* row = plane + (long long)i * p.xsH;
        // packed: the descriptor spans the same row of both planes (the lanes' offsets carry the plane and the column range check)
        const int rowBytes = (G > 1 ? (int)p.xsC + p.xW : p.xW) * (int)sizeof(T);
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)row, (short)0, rowOk ? rowBytes : 0, 0x00020000);
        const float rowFlag = rowOk ? 1.f : 0.f;
#pragma unroll
        for (int q = 0; q < Cfg::NL; q++)
            st.pre[slot][q] = __builtin_fmaf(st.bcol[q], rowFlag, bufio<T>::ld(rs, st.coff[q]));
        if (SIGNS == 2) {
            // sign bytes of the U upsampled rows this input row will complete (rows / bytes outside the tensor read 0)
#pragma unroll
            for (int j = 0; j < U; j++) {
                const int sr = U * (i - 5) - (U - 1) + j + p.py0 + p.sy;
                const bool ok = (unsigned)sr < (unsigned)p.sH;
                const unsigned char* srow = splane + (ok ? (long long)sr * p.sWb : 0);
                const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc((void*)srow, (short)0, ok ? p.sWb : 0, 0x00020000);
                const unsigned b0 = __builtin_amdgcn_raw_buffer_load_b8(ss, st.soff, 0, 0);
                const unsigned b1 = __builtin_amdgcn_raw_buffer_load_b8(ss, st.soff + 1, 0, 0);
                st.sg[slot][j] = b0 | (b1 << 8);
            }
        }
    }

    // one input row: H-up into window slot S, then U upsampled rows through lrelu into the down accumulators
    template <int S>
    static __device__ __forceinline__ void step(State& st, const StreamParams& p, const T* __restrict__ plane, T* __restrict__ oplane,
                                                unsigned char* __restrict__ splane, lds_f* sIn, lds_f* sOut, int i, int delta, int lane,
                                                int oy0, int oy1, int ox0, int oxN, bool pairStore, float& liveGain, float& liveGain1, bool wide) {
        constexpr bool RDOWN = RADIAL == 1 || RADIAL == 2 || RADIAL >= 5;   // full 12x12 DOWN filter (config R forward)
        constexpr bool FOLD = RADIAL >= 5;                         // ... whose rows read the same in both directions
        constexpr bool UP2D = RADIAL == 3 || RADIAL == 4;          // full 12x12 UP filter (adjoint of those layers)
        // ---- input row -> LDS -> this lane's H-upsampled samples ----
        wave_lds_sync();                 // the previous row's sIn reads precede this row's writes
        constexpr int PS = S % SG3_PREFETCH_ROWS;        // prefetch slot of this row
#pragma unroll
        for (int q = 0; q < Cfg::NL; q++) sIn[lane + 64 * q] = st.pre[PS][q];
        // Non-finite input.  v_med3 returns the MINIMUM of its operands when one is a NaN, so the clamp below turns a NaN activation
        // into -clamp, while the reference's `if (fabsf(v) > clamp) v = copysign(clamp, v)` (filtered_lrelu.cu:412-419) lets it
        // through; a NaN-preserving clamp costs an instruction per upsampled sample (8 %), and a wave-uniform branch to one cost 5 %
        // (measured: it fences the scheduler).  Instead every sample a lane stages is classified (one v_cmp_class per load: NaN or
        // infinity, which meets taps of both signs in the filter) and from the row a wave has met one on, its output gain -- one
        // scalar register, overwritten in place -- is NaN: every output it still writes in this strip is NaN.  That is a superset of the
        // reference's NaN footprint (a failure stays loud and stays where it happened), at no cost inside the nonlinearity.  The
        // last load of a row holds the strip's 6 halo samples, which the neighbouring strip classifies as its own: left out here.
        if (SIGNS == 0 && SG3_NAN_GUARD && G > 1) {
            // packed: every load is the wave's own (no neighbouring strip); staging lane e = lane + 64 q belongs to the first plane
            // while e < PSEG -- each plane has its own gain register
#pragma unroll
            for (int q = 0; q < Cfg::NL; q++) {
                unsigned long long m;
                asm("v_cmp_class_f32 %0, %1, %2" : "=s"(m) : "v"(st.pre[PS][q]), "s"(0x207));
                const int n0 = Cfg::PSEG - 64 * q;                                   // lanes of this load that stage the first plane
                const unsigned long long first = n0 >= 64 ? ~0ull : (n0 <= 0 ? 0ull : ((1ull << (n0 & 63)) - 1ull));
                // one-plane blocks: both registers follow the one plane, the last load (halo) left out as in the one-plane kernel
                const unsigned long long mw = q < Cfg::NL - 1 ? m : 0ull;
                liveGain = (wide ? mw : (m & first)) != 0ull ? __builtin_bit_cast(float, 0x7fc00000u) : liveGain;
                liveGain1 = (wide ? mw : (m & ~first)) != 0ull ? __builtin_bit_cast(float, 0x7fc00000u) : liveGain1;
            }
        } else if (SIGNS == 0 && SG3_NAN_GUARD) {
#pragma unroll
            for (int q = 0; q < Cfg::NL - 1; q++) {
                unsigned long long m;            // asm: the file is built with -fno-honor-nans, which folds the NaN classes away
                asm("v_cmp_class_f32 %0, %1, %2" : "=s"(m) : "v"(st.pre[PS][q]), "s"(0x207));
                // the output gain itself carries the state (one scalar register that is live anyway: the radial kernels have none to spare)
                liveGain = m != 0ull ? __builtin_bit_cast(float, 0x7fc00000u) : liveGain;
            }
        }
        unsigned sgNow[U];               // this row's sign bytes: the prefetch below reuses their slot
#pragma unroll
        for (int j = 0; j < U; j++) sgNow[j] = SIGNS == 2 ? st.sg[PS][j] : 0u;
        prefetch(st, PS, p, plane, splane, i + SG3_PREFETCH_ROWS);
        wave_lds_sync();
        if (U == 2) {
            float xs[8];
            // volatile: keeps four ds_read_b64 (2 LDS cycles each); merged into ds_read2_b64 they run at a quarter of that
            const volatile lds_v2f* src = reinterpret_cast<const volatile lds_v2f*>(sIn + (G > 1 ? st.inBase : 2 * lane));
#pragma unroll
            for (int q = 0; q < 4; q++) { v2f t = src[q]; xs[2 * q] = t.x; xs[2 * q + 1] = t.y; if (UP2D) st.xw[S][q] = t; }
#pragma unroll
            for (int g = 0; g < 2 && !UP2D; g++) {
                v2f a = splat(xs[g]) * st.tuP[0];
#pragma unroll
                for (int t = 1; t < 6; t++) a = fma2(splat(xs[g + t]), st.tuP[t], a);
                st.w[S][g] = a;
            }
        } else {  // U == 4: the lane's four columns are the four phases of one input window
            float xs[6];
#pragma unroll
            for (int t = 0; t < 6; t++) xs[t] = sIn[(G > 1 ? st.inBase : lane) + t];
            v2f a0 = splat(xs[0]) * st.tuP[1], a1 = splat(xs[0]) * st.tuP[0];
#pragma unroll
            for (int t = 1; t < 6; t++) {
                a0 = fma2(splat(xs[t]), st.tuP[2 * t + 1], a0);    // columns 0,1: phases 3,2
                a1 = fma2(splat(xs[t]), st.tuP[2 * t], a1);        // columns 2,3: phases 1,0
            }
            st.w[S][0] = a0; st.w[S][1] = a1;
        }
        // ---- U new upsampled rows ----
        const float slope = p.slope, clampv = p.clamp / p.gain, gain = p.gain;
        // sign-write mode applies the gain before the nonlinearity; the 2-D up filter's taps carry no up^2 factor
        const float gainOut = (SIGNS == 1) ? 1.f : (UP2D ? p.gain * (float)(U * U) : ((SIGNS == 0 && SG3_NAN_GUARD) ? ((G > 1 && lane >= 32) ? liveGain1 : liveGain) : p.gain));
#pragma unroll
        for (int j = 0; j < U; j++) {
            const int kv = U - 1 - j;                          // vertical up phase of this row
            v2f u0, u1;
            // mirror-symmetric 12x12 down filter: the six half rows this upsampled row needs (36 taps) are requested now and
            // arrive under the vertical up filter and the nonlinearity
            TapHalf th[6];
            if (FOLD) {
                const int kp0 = (VPH + S * U + j) % D;
#pragma unroll
                for (int r = 0; r < 6; r++) radial_half_issue(th[r], p.fd, (RADIAL == 6) ? kp0 + r * D : 11 - (kp0 + r * D));
            }
            if (UP2D) {
                // 36 taps per upsampled sample: filter rows kv, kv + 2, .. against the six raw rows of the window; the rows
                // stream through the scalar cache like the radial down filter's, the next one loading under this one's FMAs
                constexpr bool FLIPU = RADIAL == 4;
                u0 = splat(0.f); u1 = splat(0.f);
                TapRow rowA, rowB;
                radial_row_issue(rowA, p.fu, FLIPU ? kv : 11 - kv);
#pragma unroll
                for (int t = 0; t < 6; t++) {
                    const int slot = (S + 1 + t) % 6;          // t = 0: oldest row (i - 5)
                    const int k = kv + 2 * t;
                    TapRow& cur = (t & 1) ? rowB : rowA;
                    TapRow& nxt = (t & 1) ? rowA : rowB;
                    radial_row_wait(cur);
                    if (t < 5) radial_row_issue(nxt, p.fu, FLIPU ? k + 2 : 11 - (k + 2));
#pragma unroll
                    for (int sx = 0; sx < 6; sx++) {
                        const v2f tq = up2d_pair<FLIPU>(cur, sx);
                        // column groups 0 / 1 read samples sx / sx + 1 of the row
                        u0 = (sx & 1) ? fma_x_tap<1, FLIPU>(st.xw[slot][sx / 2], tq, u0) : fma_x_tap<0, FLIPU>(st.xw[slot][sx / 2], tq, u0);
                        u1 = ((sx + 1) & 1) ? fma_x_tap<1, FLIPU>(st.xw[slot][(sx + 1) / 2], tq, u1) : fma_x_tap<0, FLIPU>(st.xw[slot][(sx + 1) / 2], tq, u1);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 6 && !UP2D; t++) {
                const int slot = (S + 1 + t) % 6;              // t = 0: oldest row (i - 5)
                const int k = kv + U * t;
                // tap k lives in the reversed pair k/2: odd k = low half, even k = high half
                if (t == 0) {
                    u0 = (k & 1) ? mul_tap<0>(st.w[slot][0], st.tuP[k / 2]) : mul_tap<1>(st.w[slot][0], st.tuP[k / 2]);
                    u1 = (k & 1) ? mul_tap<0>(st.w[slot][1], st.tuP[k / 2]) : mul_tap<1>(st.w[slot][1], st.tuP[k / 2]);
                } else if (k & 1) {
                    u0 = fma_tap<0>(st.w[slot][0], st.tuP[k / 2], u0);
                    u1 = fma_tap<0>(st.w[slot][1], st.tuP[k / 2], u1);
                } else {
                    u0 = fma_tap<1>(st.w[slot][0], st.tuP[k / 2], u0);
                    u1 = fma_tap<1>(st.w[slot][1], st.tuP[k / 2], u1);
                }
            }
            const int uy = U * (i - 5) - (U - 1) + j + p.py0;  // row of the upsampled buffer
            float a[4] = {u0.x, u0.y, u1.x, u1.y};
            if (SIGNS == 2) {
                // adjoint pass: the stored sign codes select the derivative of lrelu + clamp (1, slope or 0); the gain is
                // linear and applied once per output sample
                const unsigned bits = sgNow[j] >> (2 * st.sq);
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const unsigned code = bits >> (2 * c);
                    a[c] *= (code & 2u) ? 0.f : ((code & 1u) ? slope : 1.f);
                }
            } else if (SIGNS == 1) {
                // training forward: the reference's exact order (gain, lrelu, clamp; filtered_lrelu.cu sign write) so that
                // the sign codes agree with it: 1 = negative, 2 = clamped
                unsigned code = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const float v = a[c] * gain;
                    const bool neg = v < 0.f;
                    const float lv = neg ? v * slope : v;
                    const bool big = __builtin_fabsf(lv) > p.clamp;
                    a[c] = big ? __builtin_copysignf(p.clamp, lv) : lv;
                    code |= (big ? 2u : (neg ? 1u : 0u)) << (2 * c);
                }
                const int sr = uy + p.sy;
                if (uy >= oy0 * D && uy <= (oy1 - 1) * D + Cfg::FD - 1 && (unsigned)sr < (unsigned)p.sH) {    // wave-uniform
                    // a byte holds 4 consecutive columns; a lane's 4 columns start st.sq columns into a byte, so the byte is
                    // completed with the first st.sq columns of the next lane
                    const unsigned nxt = (unsigned)__shfl_down((int)code, 1);
                    const unsigned byte = st.sq == 0 ? code : ((code >> (2 * (4 - st.sq))) | (nxt << (2 * st.sq))) & 0xffu;
                    const int bofs = (st.sq == 0 || lane < 63) ? st.soff + (st.sq ? 1 : 0) : -1;
                    const __amdgpu_buffer_rsrc_t ss = __builtin_amdgcn_make_buffer_rsrc((void*)(splane + (long long)sr * p.sWb), (short)0, p.sWb, 0x00020000);
                    __builtin_amdgcn_raw_buffer_store_b8((unsigned char)byte, ss, bofs, 0, 0);
                }
            } else {
                // leaky ReLU (slope <= 1 so lrelu(v) = max(v, slope*v)) and clamp.  The activation gain g > 0 commutes with
                // both: clamp_c(lrelu(g*u)) = g * clamp_{c/g}(lrelu(u)), so g is applied once per OUTPUT sample after the
                // down filter instead of once per upsampled sample (clampv = clamp / gain here)
                const v2f s0 = u0 * splat(slope), s1 = u1 * splat(slope);
                a[0] = __builtin_fmaxf(u0.x, s0.x); a[1] = __builtin_fmaxf(u0.y, s0.y); a[2] = __builtin_fmaxf(u1.x, s1.x); a[3] = __builtin_fmaxf(u1.y, s1.y);
#pragma unroll
                for (int c = 0; c < 4; c++) a[c] = __builtin_amdgcn_fmed3f(a[c], -clampv, clampv);
            }
            const v2f r0 = {a[0], a[1]}, r1 = {a[2], a[3]};
            const int kp = (VPH + S * U + j) % D;              // down phase of this row (a trip starts at phase VPH)
            if (RDOWN) {
                // ---- full 2-D down filter (config R): the activated row is exchanged through LDS once (each lane reads
                // the 16 samples under its two output columns) and scattered into the six output rows it belongs to, one
                // even/odd polyphase pass per output row with that row's 12 taps.  The 144 taps do not fit the scalar
                // registers: they stream through the scalar cache (s_load) as SGPR pairs, off the LDS and VALU paths.
                lds_f* dstr = sOut + (4 - delta) + (G > 1 ? st.outBase : 4 * lane);
                dstr[0] = r0.x; dstr[1] = r0.y; dstr[2] = r1.x; dstr[3] = r1.y;
                wave_lds_sync();
                const lds_v4f* srcr = reinterpret_cast<const lds_v4f*>(sOut + 4 + (G > 1 ? st.outBase : 4 * lane));
                v2f pr[8];
#pragma unroll
                for (int q = 0; q < 4; q++) { const v4f t = srcr[q]; pr[2 * q] = (v2f){t.x, t.y}; pr[2 * q + 1] = (v2f){t.z, t.w}; }
                wave_lds_sync();
                const int headR = ((VPH + S * U + j) / D) % 6;
                constexpr bool FLIPPED = RADIAL == 2 || RADIAL == 6;
                // mirror-symmetric rows: the samples j and 11 - j under an output column share a tap, so they are added
                // first (three packed adds per column, the partner pair entering with its halves exchanged) and each of the
                // six output rows then costs three packed FMAs per column instead of six
                v2f fo[2][3];
#pragma unroll
                for (int q = 0; q < 3 && FOLD; q++) { fo[0][q] = add_swapped(pr[q], pr[5 - q]); fo[1][q] = add_swapped(pr[q + 1], pr[6 - q]); }
                if (FOLD) radial_half_wait(th);
#pragma unroll
                for (int r = 0; r < 6 && FOLD; r++) {
                    const int slot = (headR + 5 - r) % 6;          // r = 5: oldest output row (completes first)
                    const int k = kp + r * D;                      // filter row of this upsampled row in that output row
#pragma unroll
                    for (int q = 0; q < 3; q++) {
                        // memory order: taps (2q, 2q + 1) of the row
                        const v2f tq = q == 0 ? (v2f){th[r].a.x, th[r].a.y} : (q == 1 ? (v2f){th[r].a.z, th[r].a.w} : th[r].b);
                        if (k == 0 && q == 0) {
                            st.acc[slot][0] = radial_mul<true>(fo[0][0], tq);
                            st.acc[slot][1] = radial_mul<true>(fo[1][0], tq);
                        } else {
                            st.acc[slot][0] = radial_fma<true>(fo[0][q], tq, st.acc[slot][0]);
                            st.acc[slot][1] = radial_fma<true>(fo[1][q], tq, st.acc[slot][1]);
                        }
                    }
                }
                TapRow rowA, rowB;                                 // filter rows in flight: the next one loads under this one's FMAs
                if (!FOLD) radial_row_issue(rowA, p.fd, FLIPPED ? kp : 11 - kp);
#pragma unroll
                for (int r = 0; r < 6 && !FOLD; r++) {
                    const int slot = (headR + 5 - r) % 6;          // r = 5: oldest output row (completes first)
                    const int k = kp + r * D;                      // filter row of this upsampled row in that output row
                    TapRow& cur = (r & 1) ? rowB : rowA;
                    TapRow& nxt = (r & 1) ? rowA : rowB;
                    radial_row_wait(cur);
                    if (r < 5) radial_row_issue(nxt, p.fd, FLIPPED ? k + D : 11 - (k + D));
#pragma unroll
                    for (int q = 0; q < 6 && !FOLD; q++) {
                        const v2f tq = radial_pair<FLIPPED>(cur, q);
                        if (k == 0 && q == 0) {
                            st.acc[slot][0] = radial_mul<FLIPPED>(pr[0], tq);
                            st.acc[slot][1] = radial_mul<FLIPPED>(pr[1], tq);
                        } else {
                            st.acc[slot][0] = radial_fma<FLIPPED>(pr[q], tq, st.acc[slot][0]);
                            st.acc[slot][1] = radial_fma<FLIPPED>(pr[q + 1], tq, st.acc[slot][1]);
                        }
                    }
                }
                if (kp == D - 1) {
                    const int oy = (uy - (Cfg::FD - 1)) / D;       // exact
                    if (oy >= oy0 && oy < oy1) {                   // wave-uniform
                        const v2f y0 = st.acc[headR][0], y1 = st.acc[headR][1];
                        const float f0 = (y0.x + y0.y) * gainOut, f1 = (y1.x + y1.y) * gainOut;
                        if (SIGNS == 2) {
                            const float m0 = 2 * lane < oxN ? f0 : 0.f, m1 = 2 * lane + 1 < oxN ? f1 : 0.f;
                            st.osum += m0 + m1;
                            st.omax = __builtin_fmaxf(st.omax, __builtin_fmaxf(__builtin_fabsf(m0), __builtin_fabsf(m1)));
                        }
                        T* orow = oplane + (long long)oy * p.ysH + ox0;
                        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)orow, (short)0, (G > 1 ? (int)p.ysC + oxN : oxN) * (int)sizeof(T), 0x00020000);
                        if (pairStore) {
                            bufio<T>::st2(rs, G > 1 ? st.ooff[0] : 2 * lane * (int)sizeof(T), f0, f1);
                        } else {
                            bufio<T>::st1(rs, G > 1 ? st.ooff[0] : 2 * lane * (int)sizeof(T), f0);
                            bufio<T>::st1(rs, G > 1 ? st.ooff[1] : (2 * lane + 1) * (int)sizeof(T), f1);
                        }
                    }
                }
                continue;
            }
            // scatter into the FD/D output rows this upsampled row belongs to
            const int headNow = ((VPH + S * U + j) / D) % 6;   // output rows completed so far in this trip shift the head
#pragma unroll
            for (int r = 0; r < 6; r++) {
                const int slot = (headNow + 5 - r) % 6;        // r = 5: oldest output row (completes first)
                const int k = kp + r * D;
                if (k == 0) {                                  // first contribution to a new output row: no zeroing needed
                    st.acc[slot][0] = mul_tap<0>(r0, st.tdP[0]);
                    st.acc[slot][1] = mul_tap<0>(r1, st.tdP[0]);
                } else if (k & 1) {
                    st.acc[slot][0] = fma_tap<1>(r0, st.tdP[k / 2], st.acc[slot][0]);
                    st.acc[slot][1] = fma_tap<1>(r1, st.tdP[k / 2], st.acc[slot][1]);
                } else {
                    st.acc[slot][0] = fma_tap<0>(r0, st.tdP[k / 2], st.acc[slot][0]);
                    st.acc[slot][1] = fma_tap<0>(r1, st.tdP[k / 2], st.acc[slot][1]);
                }
            }
            if (kp == D - 1) {
                // output row complete: H-down through LDS and store
                const int oy = (uy - (Cfg::FD - 1)) / D;       // exact
                if (oy >= oy0 && oy < oy1) {                   // wave-uniform
                    const v2f o0 = st.acc[headNow][0], o1 = st.acc[headNow][1];
                    lds_f* dst = sOut + (4 - delta) + (G > 1 ? st.outBase : 4 * lane);
                    dst[0] = o0.x; dst[1] = o0.y; dst[2] = o1.x; dst[3] = o1.y;
                    wave_lds_sync();
                    T* orow = oplane + (long long)oy * p.ysH + ox0;
                    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)orow, (short)0, (G > 1 ? (int)p.ysC + oxN : oxN) * (int)sizeof(T), 0x00020000);
                    const lds_v4f* src = reinterpret_cast<const lds_v4f*>(sOut + 4 + (G > 1 ? st.outBase : 4 * lane));
                    if (D == 2) {
                        // lane l -> output columns 2l, 2l+1: upsampled samples 4l .. 4l+13 (+4 pad), four aligned b128 reads
                        v2f pr[8];
#pragma unroll
                        for (int q = 0; q < 4; q++) { const v4f t = src[q]; pr[2 * q] = (v2f){t.x, t.y}; pr[2 * q + 1] = (v2f){t.z, t.w}; }
                        v2f y0 = pr[0] * st.tdP[0], y1 = pr[1] * st.tdP[0];
#pragma unroll
                        for (int q = 1; q < Cfg::FD / 2; q++) { y0 = fma2(pr[q], st.tdP[q], y0); y1 = fma2(pr[q + 1], st.tdP[q], y1); }
                        const float f0 = (y0.x + y0.y) * gainOut, f1 = (y1.x + y1.y) * gainOut;
                        if (SIGNS == 2) {
                            const float m0 = 2 * lane < oxN ? f0 : 0.f, m1 = 2 * lane + 1 < oxN ? f1 : 0.f;
                            st.osum += m0 + m1;
                            st.omax = __builtin_fmaxf(st.omax, __builtin_fmaxf(__builtin_fabsf(m0), __builtin_fabsf(m1)));
                        }
#ifdef SG3_FUSION_BOUND
                        // diagnostic builds of tools/flrelu_fusion_bound.hip ONLY (never in the product library): what a fusion of this
                        // layer with the ToRGB convolution behind it could cost at least.  1: the activation plane is not stored at all
                        // (the outputs are only kept alive); 2: instead, the wave's share of the three RGB partial sums -- three packed
                        // multiplies with wave-uniform weights and three 8-byte LDS writes per lane and row; the exchange between the
                        // waves of a pixel (LDS reads, adds, barrier) and the store of the partial image are NOT included.
                        if (SG3_FUSION_BOUND >= 2) {
                            lds_v2f* rgbx = reinterpret_cast<lds_v2f*>(sOut + Cfg::SOUT);
#pragma unroll
                            for (int k = 0; k < 3; k++) rgbx[k * 64 + lane] = (v2f){f0, f1} * splat(p.slope * (float)(k + 1));
                        } else {
                            asm volatile("" :: "v"(f0), "v"(f1));
                        }
                        if (false) {
#else
                        if (pairStore) {
#endif
                            bufio<T>::st2(rs, G > 1 ? st.ooff[0] : 2 * lane * (int)sizeof(T), f0, f1);
                        } else {
                            bufio<T>::st1(rs, G > 1 ? st.ooff[0] : 2 * lane * (int)sizeof(T), f0);
                            bufio<T>::st1(rs, G > 1 ? st.ooff[1] : (2 * lane + 1) * (int)sizeof(T), f1);
                        }
                    } else {
                        // D == 4: lane l -> output column l: upsampled samples 4l .. 4l+23, six aligned b128 reads
                        v2f pr[12];
#pragma unroll
                        for (int q = 0; q < 6; q++) { const v4f t = src[q]; pr[2 * q] = (v2f){t.x, t.y}; pr[2 * q + 1] = (v2f){t.z, t.w}; }
                        v2f y0 = pr[0] * st.tdP[0];
#pragma unroll
                        for (int q = 1; q < Cfg::FD / 2; q++) y0 = fma2(pr[q], st.tdP[q], y0);
                        const float f0 = (y0.x + y0.y) * gainOut;
                        if (SIGNS == 2) {
                            const float m0 = lane < oxN ? f0 : 0.f;
                            st.osum += m0;
                            st.omax = __builtin_fmaxf(st.omax, __builtin_fabsf(m0));
                        }
                        bufio<T>::st1(rs, lane * (int)sizeof(T), f0);
                    }
                    wave_lds_sync();
                }
            }
        }
    }

    static __device__ __forceinline__ void run(const StreamParams& p) {
        static_assert((D == 2 || D == 4) && (6 * U) % D == 0, "streaming kernel: down is 2 or 4");
        constexpr bool RDOWN = RADIAL == 1 || RADIAL == 2 || RADIAL >= 5;
        constexpr bool UP2D = RADIAL == 3 || RADIAL == 4;
        static_assert(RADIAL >= 0 && RADIAL <= 6, "RADIAL: 0 separable, 1/2 12x12 down, 3/4 12x12 up, 5/6 12x12 down with mirror-symmetric rows");
        static_assert(!RDOWN || (D == 2 && SIGNS != 2), "radial down filter: forward passes (plain or sign-writing), down 2");
        static_assert(!UP2D || (U == 2 && SIGNS == 2), "2-D up filter: the adjoint pass, up 2");
        static_assert(G == 1 || (G == 2 && SIGNS == 0 && D == 2), "packed mode: plain forward, down 2, two planes");
#ifdef SG3_FUSION_BOUND
        __shared__ __attribute__((aligned(16))) float lds[Cfg::SIN + Cfg::SOUT + 3 * 128];      // + the diagnostic RGB rows
#else
        __shared__ __attribute__((aligned(16))) float lds[Cfg::SIN + Cfg::SOUT];
#endif
        lds_f* sIn = (lds_f*)lds;
        lds_f* sOut = (lds_f*)lds + Cfg::SIN;                               // row exchanged for the horizontal down pass
        const int lane = threadIdx.x;
#ifdef SG3_STAMPS
        const unsigned long long stampC = __builtin_amdgcn_s_memtime(), stampR = __builtin_amdgcn_s_memrealtime();
#endif

        // XCD-aware renumbering: consecutive logical blocks (adjacent strips / chunks of one plane) share an XCD's L2
        int bid = blockIdx.x;
        {
            const int nb = p.totalBlocks, q = nb >> 3, r = nb & 7, xcd = bid & 7, k = bid >> 3;
            bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
        }
        // a G = 2 kernel runs both forms: its first wideBlocks blocks take the full-width strips of the rows, one plane each, the
        // others the remainder strip (or the whole row, when it is narrow) of two planes each.  Wave-uniform.
        const bool wide = G > 1 && bid < p.wideBlocks;
        const bool two = G > 1 && !wide;
        const int pbid = two ? bid - p.wideBlocks : bid;
        const int strip = two ? 0 : pbid % p.nStrips;
        const int chunk = two ? pbid % p.nChunks : (pbid / p.nStrips) % p.nChunks;
        const int plane_id = two ? (pbid / p.nChunks) * 2 : pbid / (p.nStrips * p.nChunks);      // two: the first of this wave's planes
        const int n = plane_id / p.C, c = plane_id - n * p.C;

        const int ox0 = (G > 1 && wide) ? strip * p.TW : p.ox0Base + strip * p.TW;
        const int oxN = two ? p.yW - ox0 : min(p.TW, p.yW - ox0);
        const int oy0 = chunk * p.CH;
        const int oy1 = min(oy0 + p.CH, p.yH);

        const T* plane = (const T*)p.x + (long long)n * p.xsN + (long long)c * p.xsC;
        T* oplane = (T*)p.y + (long long)n * p.ysN + (long long)c * p.ysC;
        const float bias = p.b ? io<T>::ld((const T*)p.b + (long long)c * p.bStride) : 0.f;
        unsigned char* splane = SIGNS ? p.s + (long long)plane_id * p.sH * p.sWb : nullptr;
        // a pair store must not straddle the end of the row (and fp16 pairs must be dword aligned)
        const bool pairStore = ((oxN & 1) == 0) &&
            (sizeof(T) == 4 || (((((unsigned long long)oplane >> 1) + (unsigned long long)ox0) & 1) == 0 && (p.ysH & 1) == 0 && (G == 1 || (p.ysC & 1) == 0)));
        // packed: the second plane is the next one of the dense (n, c) order -- channel 0 of the next image after the last channel;
        // an odd number of planes leaves the last wave without one
        const bool second = two && plane_id + 1 < p.planes;
        const float bias1 = (second && p.b) ? io<T>::ld((const T*)p.b + (long long)(c + 1 == p.C ? 0 : c + 1) * p.bStride) : 0.f;

        State st;
        // taps -> scalar register pairs; effective correlation taps g[k] = f[flip ? k : taps-1-k]
        const float gU = (float)U;
#pragma unroll
        for (int m = 0; m < Cfg::FU / 2; m++) {
            if (UP2D) { st.tuP[m] = splat(0.f); continue; }          // 2-D up filter: rows stream from the scalar cache
            const float f1 = p.fu[p.flip ? 2 * m + 1 : Cfg::FU - 2 - 2 * m], f0 = p.fu[p.flip ? 2 * m : Cfg::FU - 1 - 2 * m];
            st.tuP[m] = (v2f){to_sgpr(f1 * gU), to_sgpr(f0 * gU)};
        }
        if (RDOWN) {
#pragma unroll
            for (int m = 0; m < Cfg::FD / 2; m++) st.tdP[m] = splat(0.f);      // unused: the 12x12 taps stream from the scalar cache
        } else {
#pragma unroll
            for (int m = 0; m < Cfg::FD / 2; m++) {
                const float f0 = p.fd[p.flip ? 2 * m : Cfg::FD - 1 - 2 * m], f1 = p.fd[p.flip ? 2 * m + 1 : Cfg::FD - 2 - 2 * m];
                st.tdP[m] = (v2f){to_sgpr(f0), to_sgpr(f1)};
            }
        }

        // horizontal geometry: first upsampled column of the strip, shifted left by delta so that every lane's
        // U-column groups start at an upsampled position == 1 (mod U) and share one input window
        const int delta = ((D * ox0 - p.px0 - 1) % U + U) % U;
        const int uxs = D * ox0 - delta;
        const int ibase = floor_div(uxs - p.px0 - 1, U) + 1;
#pragma unroll
        for (int q = 0; q < Cfg::NL; q++) {
            if (G > 1) {
                // staging slot e of the LDS row: plane e / PSEG, sample e % PSEG of that plane's window
                // (one-plane blocks: the whole row is that plane's)
                const int e = lane + 64 * q, g = (two && e >= Cfg::PSEG) ? 1 : 0;
                const int ix = ibase + e - g * Cfg::PSEG;
                const bool ok = (wide || e < 2 * Cfg::PSEG) && (unsigned)ix < (unsigned)p.xW && (g == 0 || second);
                st.coff[q] = ok ? (g * (int)p.xsC + ix) * (int)sizeof(T) : (int)0x80000000;
                st.bcol[q] = ok ? (g ? bias1 : bias) : 0.f;
                continue;
            }
            const int ix = ibase + lane + 64 * q;
            st.coff[q] = ix * (int)sizeof(T);                  // negative / beyond the row -> out of range -> reads 0
            st.bcol[q] = ((unsigned)ix < (unsigned)p.xW) ? bias : 0.f;
        }
        if (G > 1) {
            const int g = two ? lane >> 5 : 0, ll = two ? lane & 31 : lane;
            st.inBase = g * Cfg::PSEG + Cfg::GROUPS * ll;
            st.outBase = g * Cfg::POSEG + 4 * ll;
            const bool mine = g == 0 || second;
            st.ooff[0] = (mine && 2 * ll < oxN) ? (g * (int)p.ysC + 2 * ll) * (int)sizeof(T) : (int)0x80000000;
            st.ooff[1] = (mine && 2 * ll + 1 < oxN) ? (g * (int)p.ysC + 2 * ll + 1) * (int)sizeof(T) : (int)0x80000000;
        }

        st.osum = 0.f;
        st.omax = 0.f;
        float liveGain = p.gain;                  // the output gain; NaN from the row on in which this wave staged a non-finite sample
        float liveGain1 = p.gain;                 // packed mode: the second plane's
        if (SIGNS) {
            const int c0 = uxs + p.sx;                           // sign-tensor column of lane 0's first upsampled column
            st.sq = to_sgpr_i(((c0 % 4) + 4) % 4);
            st.soff = floor_div(c0, 4) + lane;
        } else {
            st.sq = 0; st.soff = 0;
        }

        // vertical geometry
        const int uyA = oy0 * D;
        const int uyB = (oy1 - 1) * D + Cfg::FD - 1;
        int iFirst = ceil_div_s(uyA - p.py0, U);                // first input row of the window that yields uyA
        if ((U * (iFirst - 5)) % D != 0) iFirst -= 1;           // trips start at down phase VPH: U * (iFirst - 5) = 0 (mod D)
        const int iLast = ceil_div_s(uyB - p.py0, U) + 5;       // step that yields uyB
        const int nBlocks = (iLast - iFirst + 1 + 5) / 6;

#pragma unroll
        for (int s = 0; s < 6; s++) {
            st.w[s][0] = splat(0.f); st.w[s][1] = splat(0.f);
#pragma unroll
            for (int q = 0; q < 4; q++) st.xw[s][q] = splat(0.f);
            st.acc[s][0] = splat(0.f); st.acc[s][1] = splat(0.f);
            if (s < SG3_PREFETCH_ROWS) prefetch(st, s, p, plane, splane, iFirst + s);
        }

        int i = iFirst;
        for (int blk = 0; blk < nBlocks; blk++, i += 6) {
            step<0>(st, p, plane, oplane, splane, sIn, sOut, i + 0, delta, lane, oy0, oy1, ox0, oxN, pairStore, liveGain, liveGain1, wide);
            step<1>(st, p, plane, oplane, splane, sIn, sOut, i + 1, delta, lane, oy0, oy1, ox0, oxN, pairStore, liveGain, liveGain1, wide);
            step<2>(st, p, plane, oplane, splane, sIn, sOut, i + 2, delta, lane, oy0, oy1, ox0, oxN, pairStore, liveGain, liveGain1, wide);
            step<3>(st, p, plane, oplane, splane, sIn, sOut, i + 3, delta, lane, oy0, oy1, ox0, oxN, pairStore, liveGain, liveGain1, wide);
            step<4>(st, p, plane, oplane, splane, sIn, sOut, i + 4, delta, lane, oy0, oy1, ox0, oxN, pairStore, liveGain, liveGain1, wide);
            step<5>(st, p, plane, oplane, splane, sIn, sOut, i + 5, delta, lane, oy0, oy1, ox0, oxN, pairStore, liveGain, liveGain1, wide);
            if ((6 * U / D) % 6 != 0) {
                // a trip completes 6U/D output rows; when that is 3 (U = 2, D = 4) the ring of output rows has turned by
                // half: swap the halves so that the compile-time slot numbering holds for the next trip
#pragma unroll
                for (int r = 0; r < 3; r++)
#pragma unroll
                    for (int h = 0; h < 2; h++) { const v2f t = st.acc[r][h]; st.acc[r][h] = st.acc[r + 3][h]; st.acc[r + 3][h] = t; }
            }
        }
#ifdef SG3_STAMPS
        if (p.stamps && lane == 0) {
            p.stamps[2 * (long long)blockIdx.x] = __builtin_amdgcn_s_memtime() - stampC;
            p.stamps[2 * (long long)blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - stampR;
        }
#endif
        if (SIGNS == 2 && p.ysum) {                         // bias gradient of the adjoint pass
            float v = st.osum;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m);
            if (lane == 0) p.ysum[(long long)plane_id * (p.nChunks * p.nStrips) + chunk * p.nStrips + strip] = v;
        }
        if (SIGNS == 2 && p.ymax) {                         // max |dx| of the adjoint pass
            float v = st.omax;
#pragma unroll
            for (int m = 32; m > 0; m >>= 1) v = __builtin_fmaxf(v, __shfl_xor(v, m));
            if (lane == 0) p.ymax[(long long)plane_id * (p.nChunks * p.nStrips) + chunk * p.nStrips + strip] = v;
        }
    }
};

template <typename T, int U, int D, int VPH, int RADIAL, int SIGNS, int G = 1>
__global__ void __launch_bounds__(64)
flrelu_stream_kernel(StreamParams p) {
    Stream<T, U, D, VPH, RADIAL, SIGNS, G>::run(p);
}

// ---------------------------------------------------------------------------
// up = down = 1, 1x1 filters: y = clamp(lrelu((x + b) * fu * gain)) * fd   (ToRGB layer: clamp(x + b))
struct PointParams {
    const void* x; void* y; const void* b;
    const float* fu; const float* fd;
    int N, C, H, W;
    long long xs[4], ys[4], bStride;
    int px0, py0, yH, yW;
    float gain, slope, clamp;
};

template <typename T>
__global__ void __launch_bounds__(256)
flrelu_pointwise_kernel(PointParams p) {
    const float fu = p.fu[0], fd = p.fd[0];
    const long long total = (long long)p.N * p.C * p.yH * p.yW;
    const long long gs = (long long)gridDim.x * blockDim.x;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gs) {
        const int ox = (int)(idx % p.yW); long long t = idx / p.yW;
        const int oy = (int)(t % p.yH); t /= p.yH;
        const int c = (int)(t % p.C); const int n = (int)(t / p.C);
        const int ix = ox - p.px0, iy = oy - p.py0;
        float v = 0.f;
        if ((unsigned)ix < (unsigned)p.W && (unsigned)iy < (unsigned)p.H)
            v = io<T>::ld((const T*)p.x + n * p.xs[0] + c * p.xs[1] + iy * p.xs[2] + ix * p.xs[3]) + (p.b ? io<T>::ld((const T*)p.b + c * p.bStride) : 0.f);
        v *= fu * p.gain;
        v = v < 0.f ? v * p.slope : v;
        v = fminf(fmaxf(v, -p.clamp), p.clamp);
        io<T>::st((T*)p.y + n * p.ys[0] + c * p.ys[1] + oy * p.ys[2] + ox * p.ys[3], v * fd);
    }
}

// ---------------------------------------------------------------------------
static bool stream_supported(int up, int down, int fuW, int fuH, int fdW, int fdH) {
    if (fuH == 12) {
        // adjoint of the radial (config R) layers: 12x12 up filter, up 2, separable down 2 (12 taps) or 4 (24 taps)
        return up == 2 && fuW == 12 && fdH == 0 && ((down == 2 && fdW == 12) || (down == 4 && fdW == 24));
    }
    if (fuH != 0) return false;                               // otherwise a separable up filter
    if (fdH != 0 && fdH != 12) return false;                  // separable, or full 12x12 (radial) down filter
    if (down == 2 && fdW == 12) return (up == 2 && fuW == 12) || (up == 4 && fuW == 24);
    // adjoint of the up-4 layers: up 2 (12 taps), down 4 (24 taps), separable
    return down == 4 && fdW == 24 && fdH == 0 && up == 2 && fuW == 12;
}

// the streaming kernel evaluates lrelu as max(v, slope * v), valid for 0 <= slope <= 1 (every StyleGAN3 layer)
static bool stream_params_ok(const sg3_filtered_lrelu_params& q) {
    if (q.fdH != 0 && ((uintptr_t)q.fd & 15) != 0) return false;  // 2-D taps are fetched as 16-byte scalar quads
    if (q.fuH != 0 && (((uintptr_t)q.fu & 15) != 0 || !q.readSigns)) return false;       // 2-D up filter: adjoint passes only
    const bool signs = q.writeSigns || q.readSigns;
    if (signs && (!q.s || q.sH <= 0 || q.sWbytes <= 0)) return false;
    if (q.readSigns && q.fdH != 0) return false;                                          // adjoint passes: separable down filter
    if (q.readSigns && q.up != 2) return false;                                            // adjoint passes upsample by 2
    if (q.down == 4 && !q.readSigns) return false;                                         // down 4 exists for the adjoint only
    return q.slope >= 0.f && q.slope <= 1.f && q.xStride[3] == 1 && q.yStride[3] == 1 && !(q.clamp < 0.f);
}

static bool pointwise_supported(int up, int down, int fuW, int fuH, int fdW, int fdH) {
    return up == 1 && down == 1 && fuW == 1 && fdW == 1 && fuH <= 1 && fdH <= 1;
}

// Work decomposition of the streaming kernel: strips of equal width, as wide as a wave's 256 upsampled columns can complete
// (120 for down 2, 58 for down 4); row chunks: enough waves to fill 256 CUs x 16 waves a few times over, but chunks tall
// enough that the 11-row warm-up stays small.
static void stream_grid(int N, int C, int yH, int yW, int down, int& nStrips, int& TW, int& nChunks, int& CH) {
    const int maxTW = (256 - 6 * down) / down;
    nStrips = ceil_div(yW, maxTW);
    TW = ceil_div(yW, nStrips);
    const long long planes = (long long)N * C;
    const long long wantWaves = 256LL * 16 * 4;
    int n = (int)ceil_div64(wantWaves, planes * nStrips);
    const int maxChunks = max(1, yH / 48);
    if (n > maxChunks) n = maxChunks;
    if (n < 1) n = 1;
    CH = ceil_div(yH, n);
    nChunks = ceil_div(yH, CH);
}

// Two planes per wave (packed mode of the streaming kernel) for the plain forward on strips at most `width` columns wide: dense
// (n, c) planes (the second plane of a wave sits one channel stride after the first), offsets inside a 2 GB descriptor.
// SG3_FLRELU_NOPACK=1 in the environment keeps one plane per wave (A/B timing; read once).
static bool stream_packs_planes(const sg3_filtered_lrelu_params& q, int width) {
    static const bool allowed = [] { const char* e = getenv("SG3_FLRELU_NOPACK"); return !(e && e[0] == '1'); }();
    const long long esz = q.dtype == SG3_F32 ? 4 : 2;
    const long long xsC = q.xStride[1], ysC = q.yStride[1];
    return allowed && !q.readSigns && !q.writeSigns && q.down == 2 && width <= StreamCfg<2, 2>::PACKTW &&
           q.xStride[0] == (long long)q.C * xsC && q.yStride[0] == (long long)q.C * ysC && xsC > 0 && ysC > 0 &&
           (xsC + q.xW) * esz < 0x7fffffffLL && (ysC + q.yW) * esz < 0x7fffffffLL;
}

// width of the full strips when a row is cut into full strips + a packed remainder, and their number (0: equal strips instead --
// no remainder, a remainder too wide to pack, or more waves than equal strips take)
constexpr int STREAM_FULL_TW = 120;
static int stream_full_strips(int yW, int nStripsEqual) {
    const int nFull = (yW - 1) / STREAM_FULL_TW, rem = yW - nFull * STREAM_FULL_TW;
    return (nFull >= 1 && nFull + 1 == nStripsEqual && rem <= StreamCfg<2, 2>::PACKTW) ? nFull : 0;
}

// The per-workgroup partial sums / maxima an adjoint launch left ([N][C][slots] each) -> the bias gradient db[c] = sum over (n, slot)
// and max |dx| = max over everything, in one small launch (as torch ops: a strided reduction and a max, ~19 us per layer and PTI step).
// One block: thread per channel, (n, slot) walked in order -- a fixed summation order, so db is reproducible.
__global__ void __launch_bounds__(256)
flrelu_finish_partials_kernel(const float* __restrict__ psum, const float* __restrict__ pmax, int N, int C, int slots,
                              float* __restrict__ db, float* __restrict__ amax) {
    __shared__ float red[4];
    float mx = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int n = 0; n < N; n++) {
            const size_t base = ((size_t)n * C + c) * slots;
            for (int k = 0; k < slots; k++) {
                s += psum[base + k];
                if (pmax) mx = __builtin_fmaxf(mx, pmax[base + k]);
            }
        }
        db[c] = s;
    }
    if (!amax) return;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) mx = __builtin_fmaxf(mx, __shfl_xor(mx, m));
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) amax[0] = __builtin_fmaxf(__builtin_fmaxf(red[0], red[1]), __builtin_fmaxf(red[2], red[3]));
}

template <typename T>
static int launch_stream(const sg3_filtered_lrelu_params& q, hipStream_t st) {
    StreamParams p;
    p.x = q.x; p.y = q.y; p.b = q.b; p.fu = q.fu; p.fd = q.fd;
    p.C = q.C; p.planes = q.N * q.C; p.xH = q.xH; p.xW = q.xW; p.yH = q.yH; p.yW = q.yW;
    p.xsN = q.xStride[0]; p.xsC = q.xStride[1]; p.xsH = q.xStride[2];
    p.ysN = q.yStride[0]; p.ysC = q.yStride[1]; p.ysH = q.yStride[2];
    p.bStride = q.bStride;
    p.px0 = q.px0; p.py0 = q.py0;
    p.gain = q.gain; p.slope = q.slope; p.clamp = q.clamp; p.flip = q.flip;

    p.s = q.s; p.sH = q.sH; p.sWb = q.sWbytes; p.sx = q.sx; p.sy = q.sy;
    p.ysum = q.ySumPartial;
    p.ymax = q.yAbsMaxPartial;
#ifdef SG3_STAMPS
    p.stamps = g_stamps;
#endif
    stream_grid(q.N, q.C, q.yH, q.yW, q.down, p.nStrips, p.TW, p.nChunks, p.CH);
    const long long planes = (long long)q.N * q.C;
    const long long total = planes * p.nStrips * p.nChunks;
    if (total > 0x7fffffffLL) { set_error("filtered_lrelu: grid too large"); return SG3_BAD_ARG; }
    p.totalBlocks = (int)total;

    p.ox0Base = 0;
    const bool packed = p.nStrips == 1 && stream_packs_planes(q, q.yW);    // the 36^2 .. 52^2 layers
    if (packed) p.totalBlocks = (int)((planes + 1) / 2 * p.nChunks);
    // Rows a little wider than a whole number of full strips (148 = 120 + 28, 276 = 2 x 120 + 36, 532 = 4 x 120 + 52 columns: the
    // 148^2 .. 532^2 layers): equal strips would leave 40 .. 57 of a wave's 64 lanes at work.  Instead the full-width strips go
    // first, one plane per wave, and the remainder strip follows in a second launch with two planes per wave:
    // nFull + 1/2 waves per plane and row chunk instead of nFull + 1.
    const int remFull = stream_full_strips(q.yW, p.nStrips);
    const bool mixed = !packed && remFull > 0 && stream_packs_planes(q, q.yW - remFull * STREAM_FULL_TW);
    // ONE launch of the two-plane kernel, whose first blocks take the full strips, where it holds as many waves per SIMD as the
    // one-plane kernel (separable: 120 / 124 registers at up 2, 142 / 144 at up 4; 12x12 down filter at up 4: 153 .. 160 against
    // 148 .. 156, three waves either way).  12x12 down filter at up 2: two launches (135 registers against 128: three waves, not four)
    const bool oneLaunch = mixed && (q.fdH == 0 || q.up == 4);
    p.wideBlocks = 0;
    if (mixed) {
        p.nStrips = remFull; p.TW = STREAM_FULL_TW;
        p.totalBlocks = (int)(planes * remFull * p.nChunks);
        if (oneLaunch) {
            p.wideBlocks = p.totalBlocks;
            p.ox0Base = remFull * STREAM_FULL_TW;
            const long long all = (long long)p.totalBlocks + (planes + 1) / 2 * p.nChunks;
            if (all > 0x7fffffffLL) { set_error("filtered_lrelu: grid too large"); return SG3_BAD_ARG; }
            p.totalBlocks = (int)all;
        }
    }

    const int vph = (((q.py0 - (q.up - 1)) % q.down) + q.down) % q.down;
    dim3 g((unsigned)p.totalBlocks), b(64);
#define SG3_STREAM_LAUNCH(U, D, V, R, S) hipLaunchKernelGGL((flrelu_stream_kernel<T, U, D, V, R, S>), g, b, 0, st, p)
#define SG3_STREAM_LAUNCH_V(U, R, S) do { if (vph == 0) SG3_STREAM_LAUNCH(U, 2, 0, R, S); else SG3_STREAM_LAUNCH(U, 2, 1, R, S); } while (0)
#define SG3_PACKED_LAUNCH(U, V, R) hipLaunchKernelGGL((flrelu_stream_kernel<T, U, 2, V, R, 0, 2>), g, b, 0, st, p)
#define SG3_PACKED_LAUNCH_V(U, R) do { if (vph == 0) SG3_PACKED_LAUNCH(U, 0, R); else SG3_PACKED_LAUNCH(U, 1, R); } while (0)
#define SG3_PACKED_UP(U) do { switch (variant) { \
        case 0: SG3_PACKED_LAUNCH_V(U, 0); break; case 1: SG3_PACKED_LAUNCH_V(U, 1); break; \
        case 2: SG3_PACKED_LAUNCH_V(U, 2); break; case 5: SG3_PACKED_LAUNCH_V(U, 5); break; \
        default: SG3_PACKED_LAUNCH_V(U, 6); break; } } while (0)
    // separable | 12x12 | 12x12 flipped | 12x12 with mirror-symmetric rows | the same, flipped
    const int variant = q.fdH == 0 ? 0 : (q.fdMirror ? (q.flip ? 6 : 5) : (q.flip ? 2 : 1));
#define SG3_FORWARD_UP(U, S) do { switch (variant) { \
        case 0: SG3_STREAM_LAUNCH_V(U, 0, S); break; case 1: SG3_STREAM_LAUNCH_V(U, 1, S); break; \
        case 2: SG3_STREAM_LAUNCH_V(U, 2, S); break; case 5: SG3_STREAM_LAUNCH_V(U, 5, S); break; \
        default: SG3_STREAM_LAUNCH_V(U, 6, S); break; } } while (0)
#define SG3_FORWARD_LAUNCH(S) do { if (q.up == 2) SG3_FORWARD_UP(2, S); else SG3_FORWARD_UP(4, S); } while (0)
    if (q.readSigns) {
#define SG3_ADJOINT_LAUNCH(R) do { \
        if (q.down == 2) SG3_STREAM_LAUNCH_V(2, R, 2); \
        else switch (vph) { \
            case 0: SG3_STREAM_LAUNCH(2, 4, 0, R, 2); break; \
            case 1: SG3_STREAM_LAUNCH(2, 4, 1, R, 2); break; \
            case 2: SG3_STREAM_LAUNCH(2, 4, 2, R, 2); break; \
            default: SG3_STREAM_LAUNCH(2, 4, 3, R, 2); break; \
        } } while (0)
        if (q.fuH == 0) SG3_ADJOINT_LAUNCH(0);
        else if (q.flip) SG3_ADJOINT_LAUNCH(4);
        else SG3_ADJOINT_LAUNCH(3);
#undef SG3_ADJOINT_LAUNCH
    } else if (q.writeSigns) {
        SG3_FORWARD_LAUNCH(1);
    } else if (packed || oneLaunch) {
        if (q.up == 2) SG3_PACKED_UP(2); else SG3_PACKED_UP(4);
    } else {
        SG3_FORWARD_LAUNCH(0);
        if (mixed) {
            SG3_LAUNCH_CHECK("flrelu_stream_kernel");
            p.ox0Base = remFull * STREAM_FULL_TW; p.nStrips = 1; p.TW = q.yW - p.ox0Base;
            p.totalBlocks = (int)((planes + 1) / 2 * p.nChunks);
            g = dim3((unsigned)p.totalBlocks);
            if (q.up == 2) SG3_PACKED_UP(2); else SG3_PACKED_UP(4);
        }
    }
#undef SG3_PACKED_UP
#undef SG3_PACKED_LAUNCH_V
#undef SG3_PACKED_LAUNCH
#undef SG3_FORWARD_LAUNCH
#undef SG3_FORWARD_UP
#undef SG3_STREAM_LAUNCH_V
#undef SG3_STREAM_LAUNCH
    SG3_LAUNCH_CHECK("flrelu_stream_kernel");
    return SG3_OK;
}

template <typename T>
static int launch_pointwise(const sg3_filtered_lrelu_params& q, hipStream_t st) {
    PointParams p;
    p.x = q.x; p.y = q.y; p.b = q.b; p.fu = q.fu; p.fd = q.fd;
    p.N = q.N; p.C = q.C; p.H = q.xH; p.W = q.xW; p.yH = q.yH; p.yW = q.yW;
    for (int i = 0; i < 4; i++) { p.xs[i] = q.xStride[i]; p.ys[i] = q.yStride[i]; }
    p.bStride = q.bStride; p.px0 = q.px0; p.py0 = q.py0;
    p.gain = q.gain; p.slope = q.slope; p.clamp = q.clamp;
    const long long total = (long long)q.N * q.C * q.yH * q.yW;
    long long blocks = ceil_div64(total, 256);
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL((flrelu_pointwise_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, p);
    SG3_LAUNCH_CHECK("flrelu_pointwise_kernel");
    return SG3_OK;
}

} // namespace sg3

extern "C" {

int sg3_filtered_lrelu_has_kernel(int up, int down, int fuW, int fuH, int fdW, int fdH) {
    return (sg3::stream_supported(up, down, fuW, fuH, fdW, fdH) || sg3::pointwise_supported(up, down, fuW, fuH, fdW, fdH)) ? 1 : 0;
}

int sg3_filtered_lrelu_sum_slots(int N, int C, int yH, int yW, int down) {
    if (N <= 0 || C <= 0 || yH <= 0 || yW <= 0 || (down != 2 && down != 4)) return 0;
    int nStrips, TW, nChunks, CH;
    sg3::stream_grid(N, C, yH, yW, down, nStrips, TW, nChunks, CH);
    return nStrips * nChunks;
}

int sg3_filtered_lrelu_finish_partials(const float* sumPartial, const float* absMaxPartial, int N, int C, int slots, float* db, float* absMax, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(sumPartial && db && N > 0 && C > 0 && slots > 0, "filtered_lrelu_finish_partials: bad arguments");
    SG3_REQUIRE(!absMax == !absMaxPartial, "filtered_lrelu_finish_partials: absMaxPartial and absMax come together");
    hipLaunchKernelGGL(flrelu_finish_partials_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, sumPartial, absMaxPartial, N, C, slots, db, absMax);
    SG3_LAUNCH_CHECK("flrelu_finish_partials_kernel");
    return SG3_OK;
}

int sg3_filtered_lrelu_shape(int xH, int xW, int up, int down, int fuW, int fuH, int fdW, int fdH,
                             int px0, int px1, int py0, int py1, int* yH, int* yW, int* sH, int* sWbytes, int* swLimit) {
    using namespace sg3;
    SG3_REQUIRE(xH > 0 && xW > 0, "x is empty");
    SG3_REQUIRE(up >= 1 && down >= 1, "up and down must be at least 1");
    SG3_REQUIRE(fuW >= 1 && fdW >= 1, "fu and fd must not be empty");
    // separable filters have the same extent along both axes (reference :63-66 uses size(0) for the height)
    const int64_t fut_w = fuW - 1, fut_h = (fuH ? fuH : fuW) - 1;
    const int64_t fdt_w = fdW - 1, fdt_h = (fdH ? fdH : fdW) - 1;
    const int64_t cw = (int64_t)xW * up + (px0 + px1) - fut_w;
    const int64_t ch = (int64_t)xH * up + (py0 + py1) - fut_h;
    SG3_REQUIRE(cw > fdt_w && ch > fdt_h, "upsampled buffer must be at least the size of downsampling filter");
    SG3_REQUIRE(cw <= INT32_MAX && ch <= INT32_MAX, "upsampled buffer is too large");
    const int64_t ow = (cw - fdt_w + (down - 1)) / down;
    const int64_t oh = (ch - fdt_h + (down - 1)) / down;
    SG3_REQUIRE(ow > 0 && oh > 0, "output must be at least 1x1");
    const int64_t sw_active = ow * down - (down - 1) + fdt_w;
    const int64_t sh = oh * down - (down - 1) + fdt_h;
    const int64_t sw = (sw_active + 15) & ~(int64_t)15;
    SG3_REQUIRE(sh <= INT32_MAX && (sw >> 2) <= INT32_MAX, "signs is too large");
    if (yH) *yH = (int)oh;
    if (yW) *yW = (int)ow;
    if (sH) *sH = (int)sh;
    if (sWbytes) *sWbytes = (int)(sw >> 2);
    if (swLimit) *swLimit = (int)((sw_active + 3) >> 2);
    return SG3_OK;
}

int sg3_filtered_lrelu_planes_per_wave(const sg3_filtered_lrelu_params* p) {
    using namespace sg3;
    if (!p || (p->dtype != SG3_F32 && p->dtype != SG3_F16)) return 0;
    const bool signs = p->writeSigns || p->readSigns;
    if (!signs && pointwise_supported(p->up, p->down, p->fuW, p->fuH, p->fdW, p->fdH)) return 0;
    if (!(stream_supported(p->up, p->down, p->fuW, p->fuH, p->fdW, p->fdH) && stream_params_ok(*p))) return 0;
    if (p->N <= 0 || p->C <= 0 || p->yH <= 0 || p->yW <= 0) return 0;
    int nStrips, TW, nChunks, CH;
    stream_grid(p->N, p->C, p->yH, p->yW, p->down, nStrips, TW, nChunks, CH);
    if (nStrips == 1) return stream_packs_planes(*p, p->yW) ? 2 : 1;
    const int nFull = stream_full_strips(p->yW, nStrips);
    return (nFull > 0 && stream_packs_planes(*p, p->yW - nFull * STREAM_FULL_TW)) ? 3 : 1;
}

int sg3_filtered_lrelu(const sg3_filtered_lrelu_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->x && p->y && p->fu && p->fd, "filtered_lrelu: null tensor");
    SG3_REQUIRE(p->N > 0 && p->C > 0 && p->xH > 0 && p->xW > 0, "filtered_lrelu: x is empty");
    SG3_REQUIRE(p->yH > 0 && p->yW > 0, "filtered_lrelu: output must be at least 1x1");
    SG3_REQUIRE(p->up >= 1 && p->down >= 1, "filtered_lrelu: up and down must be at least 1");
    SG3_REQUIRE(p->dtype == SG3_F32 || p->dtype == SG3_F16, "filtered_lrelu: x must be float16 or float32");
    SG3_REQUIRE(!(p->writeSigns && p->readSigns), "filtered_lrelu: cannot read and write signs in one call");
    hipStream_t st = (hipStream_t)stream;
    const bool signs = p->writeSigns || p->readSigns;
    if (!signs && pointwise_supported(p->up, p->down, p->fuW, p->fuH, p->fdW, p->fdH))
        return p->dtype == SG3_F32 ? launch_pointwise<float>(*p, st) : launch_pointwise<_Float16>(*p, st);
    if (stream_supported(p->up, p->down, p->fuW, p->fuH, p->fdW, p->fdH) && stream_params_ok(*p))
        return p->dtype == SG3_F32 ? launch_stream<float>(*p, st) : launch_stream<_Float16>(*p, st);
    set_error("filtered_lrelu: no fused kernel for up=%d down=%d fu=%dx%d fd=%dx%d signs=%d", p->up, p->down,
              p->fuW, p->fuH, p->fdW, p->fdH, (int)signs);
    return SG3_NO_KERNEL;
}

} // extern "C"
