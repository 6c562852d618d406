// sg3_modconv_f23.hip -- the 3x3 modulated convolution in a TRANSFORM DOMAIN along x (Winograd F(2,3)), split precision:
// 12 instead of 18 channel contractions per pair of output pixels, i.e. two thirds of the matrix instructions of
// modconv_f16x3_kernel at the same fp32-equivalent arithmetic.  Same operation as sg3_modconv.hip
// (reference models/stylegan3/networks_stylegan3.py:24-63, the grouped F.conv2d at :59-62):
//
//     out[n,o,y,x] = dcoef[n,o] * sum_{i,ky,kx} wn[o,i,ky,kx] * ( x[n,i,y+ky-pad,x+kx-pad] * sIn[n,i] )
//
// For an output pixel PAIR (columns 2p, 2p+1) with the four input columns d0..d3 = 2p-pad .. 2p-pad+3 of input row y+ky-pad:
//     V0 = d0 - d2    V1 = d1 + d2    V2 = d2 - d1    V3 = d1 - d3              (input transform, fp32, before the split)
//     U0 = g0         U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2            (weight transform of a filter row g, at pack time)
//     M_xi[o, y, p] = sum_{ky, i} U_xi[o,i,ky] * V_xi[i, y+ky, p]                (the matrix-core work: K = 3 I per xi)
//     out(2p) = M0 + M1 + M2      out(2p+1) = M1 - M2 - M3                       (output transform, once per tile)
// Each V and U value is split into fp16 hi + lo and every 16-channel K step runs three v_mfma_f32_32x32x16_f16
// (Uh Vh + Uh Vl + Ul Vh), exactly as the direct kernel does with x and w (T = float, SG3_CONV_F16X3_F23).  For fp16 tensors -- the
// reference's `use_fp16` layers, networks_stylegan3.py:355-366 -- the same kernel runs with V and U rounded to fp16 once and ONE
// product per K step (T = _Float16, SG3_CONV_F16_F23).
//
// Work split (one 512-thread workgroup per CU, 8 waves): wave = (xi, M block): it owns ONE transform point for 32 output
// channels and the whole pixel tile, so a wave touches a quarter of the weight image and keeps ONE accumulator set
// (TN x 16 registers); the four xi-waves of an M block meet once per tile, in the output transform, through LDS.
// Pixel tile: 2 TN rows x 16 pairs (32 output columns): the MFMA's 32 columns are (row group rg = 0 | 1) x 16 pairs, lanes of
// row group 1 sitting TN rows below those of group 0 -- a uniform address shift per patch row, so one B fragment (patch row q)
// serves the three output rows q, q-1, q-2 of BOTH groups, as in the row-streaming direct kernel: 2 ds_read_b128 per 9 MFMAs.
// A fragments (U) never pass through LDS: the pack kernel writes them in MFMA lane order per (M tile, chunk, M block, xi), and
// a wave loads its six 1 KB fragments of the next chunk straight into registers (a 6 KB contiguous run, L2 resident).
// B image (V): [xi][hi|lo][channel half][patch row][pair][8 channels] halfs, double buffered; staged by all waves: a thread
// owns (patch row, pair, channel half), loads 8 channels x 4 columns (two 8-byte loads per channel), transforms with SINGLE-ISSUE
// fp32 instructions (packed fp32 is the one vector class that stalls the SIMD's matrix pipe, profiles/r04_mfma_valu_coissue.txt),
// splits and writes eight 16-byte vectors; the style scales come through the scalar cache.  One barrier per 16-channel chunk.  Waves 4-7 (the second wave of every SIMD) run the
// chunk body in the order stage-then-MFMA, waves 0-3 MFMA-then-stage, so that one wave's staging arithmetic sits beside the
// other's matrix instructions instead of both waves alternating in lockstep.
#include "sg3_common.h"
#include "sg3_split.h"
#include "sg3_modconv_f23.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace sg3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct F23Params {
    const void* x; const void* wp; const float* sIn; const float* dcoef; void* out;
    int N, I, O, H, W, outH, outW, pad;
    int nch, xTiles, yTiles, mTiles, totalBlocks, outPitch;
#ifdef SG3_F23_STAMPS
    unsigned long long* stamps;      // diagnostic build (tools/f23_stamps.hip): [workgroup][wave][8] cycle sums, never in the product library
#endif
};

#ifdef SG3_F23_STAMPS
unsigned long long* g_f23_stamps = nullptr;
#define F23_STAMP(t) do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define F23_STAMP(t) do { } while (0)
#endif

// ---- accumulator-register helpers: every register is named literally (template integers spliced into the text) ----
// layout: a[0:23] / a[24:47] the A fragments of even / odd chunks (fragment f = 2 ky + (hi | lo) in a[AB + 4f : AB + 4f+3], AB = 0 | 24),
// a[48 + 16 b : 63 + 16 b] the accumulator block of tile row b.
template <int AB, int B, int F> __device__ __forceinline__ void f23_mfma(const v8h& bfrag) {
    asm volatile("v_mfma_f32_32x32x16_f16 a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" :: "v"(bfrag), "n"(48 + 16 * B), "n"(63 + 16 * B), "n"(AB + 4 * F), "n"(AB + 4 * F + 3));
}
template <int R> __device__ __forceinline__ void f23_acc_zero() { asm volatile("v_accvgpr_write_b32 a[%c0], 0" :: "n"(R)); }
template <int R> __device__ __forceinline__ float f23_acc_read() { float v; asm volatile("v_accvgpr_read_b32 %0, a[%c1]" : "=v"(v) : "n"(R)); return v; }
template <int R, int N> struct F23Seq {
    static __device__ __forceinline__ void zero() { f23_acc_zero<R>(); F23Seq<R + 1, N - 1>::zero(); }
    static __device__ __forceinline__ void read(float* dst) { *dst = f23_acc_read<R>(); F23Seq<R + 1, N - 1>::read(dst + 1); }
};
template <int R> struct F23Seq<R, 0> {
    static __device__ __forceinline__ void zero() {}
    static __device__ __forceinline__ void read(float*) {}
};
// the products of one B fragment (patch row Q) with the filter rows ky = 0..2 -> tile rows Q - ky: lo(A) hi(B), hi(A) lo(B), hi(A) hi(B)
template <int AB, int TN, int Q> __device__ __forceinline__ void f23_mfma_row(const v8h& bh, const v8h& bl) {
    if constexpr (Q >= 0 && Q < TN)         f23_mfma<AB, (Q >= 0 && Q < TN) ? Q : 0, 1>(bh);
    if constexpr (Q - 1 >= 0 && Q - 1 < TN) f23_mfma<AB, (Q - 1 >= 0 && Q - 1 < TN) ? Q - 1 : 0, 3>(bh);
    if constexpr (Q - 2 >= 0 && Q - 2 < TN) f23_mfma<AB, (Q - 2 >= 0 && Q - 2 < TN) ? Q - 2 : 0, 5>(bh);
    if constexpr (Q >= 0 && Q < TN)         f23_mfma<AB, (Q >= 0 && Q < TN) ? Q : 0, 0>(bl);
    if constexpr (Q - 1 >= 0 && Q - 1 < TN) f23_mfma<AB, (Q - 1 >= 0 && Q - 1 < TN) ? Q - 1 : 0, 2>(bl);
    if constexpr (Q - 2 >= 0 && Q - 2 < TN) f23_mfma<AB, (Q - 2 >= 0 && Q - 2 < TN) ? Q - 2 : 0, 4>(bl);
    if constexpr (Q >= 0 && Q < TN)         f23_mfma<AB, (Q >= 0 && Q < TN) ? Q : 0, 0>(bh);
    if constexpr (Q - 1 >= 0 && Q - 1 < TN) f23_mfma<AB, (Q - 1 >= 0 && Q - 1 < TN) ? Q - 1 : 0, 2>(bh);
    if constexpr (Q - 2 >= 0 && Q - 2 < TN) f23_mfma<AB, (Q - 2 >= 0 && Q - 2 < TN) ? Q - 2 : 0, 4>(bh);
}

// fp16 operand form (SG3_CONV_F16_F23): operands rounded once, ONE product per filter row; fragment ky in a[AB + 4 ky : AB + 4 ky + 3]
template <int AB, int TN, int Q> __device__ __forceinline__ void f23_mfma_row_f16(const v8h& b) {
    if constexpr (Q >= 0 && Q < TN)         f23_mfma<AB, (Q >= 0 && Q < TN) ? Q : 0, 0>(b);
    if constexpr (Q - 1 >= 0 && Q - 1 < TN) f23_mfma<AB, (Q - 1 >= 0 && Q - 1 < TN) ? Q - 1 : 0, 1>(b);
    if constexpr (Q - 2 >= 0 && Q - 2 < TN) f23_mfma<AB, (Q - 2 >= 0 && Q - 2 < TN) ? Q - 2 : 0, 2>(b);
}

// wave priority: 0 none | 1 matrix phase at priority 1 | 2 waves 4-7 at priority 1 throughout (the second-dispatched wave of every SIMD
// loses the issue arbitration to the older one: its staging block took 3-4x as long; measured -6 % on L5..L9) | 3 staging block at 1
// | 4 waves 0-3 at priority 1 throughout (as slow as none: L6 2035 vs 1878 us with mode 2)
// Round 4 (single-issue staging): modes 0 / 1 / 2 and three per-phase schemes measure within the run-to-run spread of each other
// (profiles/r04_f23_prio.txt); 2 stays.
#ifndef F23_PRIO_MODE
#define F23_PRIO_MODE 2
#endif
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));

// T = float: split precision (SG3_CONV_F16X3_F23, fp32 tensors); T = _Float16: the fp16 operand form (SG3_CONV_F16_F23, fp16 tensors:
// the reference's `use_fp16` layers, networks_stylegan3.py:355-366) -- operands rounded to fp16 once, one product per K step, no lo
// planes / fragments, fp16 stores.  Same work split, staging map, request / wait structure and output transform.
template <int TN, typename T>
__global__ void __launch_bounds__(512, 2)
modconv_f23_kernel(F23Params p) {
    constexpr bool SPLIT = sizeof(T) == 4;
    constexpr int EB = (int)sizeof(T);
    constexpr int NF = SPLIT ? 6 : 3;                  // A fragments per (chunk, M block, xi): 3 ky x (hi | lo)  |  3 ky
    constexpr int PR = 2 * TN + 2;                     // patch rows of a tile
    constexpr int PLANE = PR * 256;                    // bytes per (xi, part, channel half) plane: rows of 16 pairs x 16 B
    constexpr int BUF = (SPLIT ? 16 : 8) * PLANE;      // 4 xi x (hi | lo) x 2 channel halves  |  4 xi x 2 channel halves
    constexpr int FRAG = 1024;                         // bytes per A fragment (64 lanes x 16 B)
    constexpr int CHUNKB = 2 * 4 * NF * FRAG;          // packed weights per (M tile, chunk): 2 M blocks x 4 xi x fragments
    static_assert(PR <= 16, "staging map: 8 waves x 2 (row, channel half) pairs");

    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];           // 2 x BUF (>= 64 KB: the exchange area of the output transform)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int xi = wave & 3, mb = wave >> 2;
    const int li = lane & 31, lh = lane >> 5;
    const int rg = li >> 4, pq = li & 15;

    // PERSISTENT workgroups (one per CU: the B images take most of the LDS, so a tile's prologue and epilogue have no co-resident
    // workgroup to hide behind, and neither has the dispatch of a fresh workgroup): a workgroup walks through its share of the tiles.
    // Blocks b, b + 8, ... share an XCD; each XCD takes a contiguous range of the logical tile order (M tile fastest, so the M tiles
    // of one pixel tile -- same input patch -- and neighbouring patches run side by side on one L2), 'slots' tiles at a time.
    int tFirst, tEnd, tStep;
    {
        const int G = (int)gridDim.x, b = (int)blockIdx.x, nb = p.totalBlocks;
        if ((G & 7) == 0) {
            const int q = nb >> 3, r = nb & 7, xcd = b & 7, slots = G >> 3;
            const int start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
            tFirst = start + (b >> 3); tEnd = start + q + (xcd < r ? 1 : 0); tStep = slots;
        } else { tFirst = b; tEnd = nb; tStep = G; }
#ifdef SG3_F23_ONE_TILE
        {   // experiment: one tile per workgroup, grid = tiles (XCD-aware order as before)
            const int q = nb >> 3, r = nb & 7, xcd = b & 7, k = b >> 3;
            tFirst = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k; tEnd = tFirst + 1; tStep = 1;
        }
#endif
    }
#ifdef SG3_F23_STAMPS
    unsigned long long wgStart, wgLoop = 0, wgTiles = 0, wgPro = 0, wgEpi = 0, tTop = 0, tLoopEnd = 0;
    F23_STAMP(wgStart);
#endif
    for (int tile = tFirst; tile < tEnd; tile += tStep) {
#ifdef SG3_F23_STAMPS
    F23_STAMP(tTop);
#endif
    int bid = tile;
    const int mt = bid % p.mTiles; bid /= p.mTiles;
    const int xt = bid % p.xTiles; bid /= p.xTiles;
    const int yt = bid % p.yTiles; const int n = bid / p.yTiles;
    const int o0 = mt * 64, x0 = xt * 32, y0 = yt * (2 * TN);

    // ---- descriptors, as SGPR quads for the hand-issued loads below ----
    const unsigned HWb = (unsigned)(p.H * p.W) * (unsigned)EB;
    auto make_desc = [](const void* base, unsigned bytes) {
        const unsigned long long a = (unsigned long long)base;
        u32x4 d;
        d.x = __builtin_amdgcn_readfirstlane((unsigned)a);
        d.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);           // stride 0
        d.z = __builtin_amdgcn_readfirstlane(bytes);
        d.w = 0x00020000u;
        return d;
    };
    const u32x4 xd = make_desc(static_cast<const T*>(p.x) + (size_t)n * p.I * p.H * p.W, (unsigned)p.I * HWb);
    const u32x4 wd = make_desc(p.wp, (unsigned)p.mTiles * (unsigned)p.nch * (unsigned)CHUNKB);
    const u32x4 sd = make_desc(p.sIn + (size_t)n * p.I, (unsigned)p.I * 4u);

    // ---- staging task of this thread: (patch row, pair, channel half) ----
    const int sch = wave & 1;                                         // wave-uniform channel half
    const int srow = (wave >> 1) * 4 + (lane >> 4);
    const int spair = lane & 15;
    const bool sOk = srow < PR;
    unsigned g0, g1;                                                  // byte offsets of columns (2p, 2p+1) and (2p+2, 2p+3) in a channel plane
    {
        const int gy = y0 - p.pad + srow, gx = x0 - p.pad + 2 * spair;
        const bool rowOk = sOk && (unsigned)gy < (unsigned)p.H;
        g0 = rowOk && (unsigned)gx < (unsigned)p.W ? (unsigned)(gy * p.W + gx) * (unsigned)EB : 0x80000000u;
        // a thread requests its own column pair and the pair to its right (the neighbour's own pair: an L1 hit of the same
        // instruction's lines): the same 16 requests per chunk as a form that passes the neighbour's pair through DPP and lets only
        // the last lane of a row fetch its halo pair, but without the 16 v_mov_b32_dpp per thread and chunk
        g1 = rowOk && (unsigned)(gx + 2) < (unsigned)p.W ? (unsigned)(gy * p.W + gx + 2) * (unsigned)EB : 0x80000000u;
    }
    const int sL = srow * 256 + spair * 16 + sch * PLANE;             // + (xi * 2 + part) * 2 * PLANE (split) | + xi * 2 * PLANE (fp16)
    // ---- A: this wave's six fragments of a chunk ----
    const unsigned aG = (unsigned)((mb * 4 + xi) * NF) * FRAG + (unsigned)lane * 16u;
    const unsigned aS = (unsigned)mt * (unsigned)p.nch * (unsigned)CHUNKB;             // + chunk * CHUNKB + fragment * FRAG (scalar offset)
    // ---- B fragment reads ----
    const int bR = (xi * (SPLIT ? 4 : 2) + lh) * PLANE + (rg * TN) * 256 + pq * 16;  // (hi) plane of this lane's channel half; split: lo at + 2 PLANE

    // the kernel's accumulator-register allocation covers what the asm statements name (the highest register named as a clobber sizes
    // it); the compiler itself stays within its arch VGPRs (AUDIT: no v_accvgpr_* outside the asm statements)
    if constexpr (TN == 7) asm volatile("" ::: "a159");
    else if constexpr (TN == 5) asm volatile("" ::: "a127");
    else asm volatile("" ::: "a111");
    static_assert(TN == 4 || TN == 5 || TN == 7, "accumulator-register reservation is written for these tile heights");
    F23Seq<48, 16 * TN>::zero();

    // All vector-memory loads of the K loop are issued and waited for BY HAND (inline asm): the two request streams -- A fragments,
    // input samples -- are consumed in a different order than they are issued, and hipcc, merging the loop's paths, drains the whole
    // queue (s_waitcnt vmcnt(0)) in front of each consumer, i.e. waits for the loads it has just issued: 1800-2700 cycles per staging
    // block instead of ~600 (in-kernel stamps, profiles/r03_f23_stamps.txt).  The counter is in order, so "all but the N youngest" is
    // exact: N = the loads issued after the group that is needed.
    //
    // A fragments live in ACCUMULATOR registers that only this file's asm statements name: a[0:23] hold the six fragments of the even
    // chunks, a[24:47] those of the odd chunks (fragment f = 2 ky + (hi | lo) in a[AB + 4f : AB + 4f+3]); the matrix phase exists
    // twice, once per set.  A request leads its use by a whole iteration (the loads take 2-3 us under this kernel's own traffic) and
    // goes to the set that the chunk two back has finished reading; F23_LAND only waits for it (an earlier form kept one "current"
    // set and moved every request down with 24 v_accvgpr_mov per wave and chunk: 3 % of the kernel's vector issue).  The compiler
    // never sees these values, so it cannot copy, spill or reuse a register a load is still in flight to -- which it did, in three
    // different ways, while the fragments were ordinary asm outputs (AUDIT below).
    // The input samples (rb, rsc) stay compiler-allocated outputs of their requests; every wait for them names them as read-write
    // operands, which keeps consumers behind the wait and the registers reserved until then.
    typedef typename std::conditional<SPLIT, f32x2, unsigned>::type RB;   // a column pair of one channel: two floats | two halfs
    RB rb[8][2];
    u32x8 rsc;                                                        // the chunk's eight style scales of this wave's channel half: scalar registers
    constexpr int NB = 16, NA = NF;
#define F23_NB 16
#define F23_CNT_(N) #N
#define F23_CNT(N) F23_CNT_(N)
    // Each request is ONE asm statement that opens with "s_nop 4": hipcc may reload a spilled SGPR with v_readlane right in front
    // of the statement, and an SGPR written by a VALU instruction needs 5 wait states before a vector-memory instruction reads it
    // (descriptor or scalar offset) -- the compiler pads that hazard for its own instructions, not for ones inside an asm string.
    // (Seen with the persistent tile loop, which pushed the kernel into SGPR spills: stale descriptors, wrong samples staged.)
    auto fetch_a = [&](int ch) {
        const unsigned so = aS + (unsigned)ch * CHUNKB;               // wave-uniform
        const unsigned vo = ch < p.nch ? aG : 0x80000000u;            // beyond the last chunk: out of range, answered with zeros
        // chunk parity picks the register set: the set of chunk ch was last read by the MFMAs of chunk ch - 2, issued long before
        if constexpr (!SPLIT) {
            // fp16 form: three fragments (one per filter row) in a[0:11] | a[24:35]
            if (ch & 1)
                asm volatile("s_nop 4\n\t"
                             "buffer_load_dwordx4 a[24:27], %0, %1, %2 offen\n\t"
                             "buffer_load_dwordx4 a[28:31], %0, %1, %2 offen offset:1024\n\t"
                             "buffer_load_dwordx4 a[32:35], %0, %1, %2 offen offset:2048"
                             :: "v"(vo), "s"(wd), "s"(so)
                             : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35");
            else
                asm volatile("s_nop 4\n\t"
                             "buffer_load_dwordx4 a[0:3], %0, %1, %2 offen\n\t"
                             "buffer_load_dwordx4 a[4:7], %0, %1, %2 offen offset:1024\n\t"
                             "buffer_load_dwordx4 a[8:11], %0, %1, %2 offen offset:2048"
                             :: "v"(vo), "s"(wd), "s"(so)
                             : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11");
        } else
        if (ch & 1)
            asm volatile("s_nop 4\n\t"
                         "buffer_load_dwordx4 a[24:27], %0, %1, %2 offen\n\t"
                         "buffer_load_dwordx4 a[28:31], %0, %1, %2 offen offset:1024\n\t"
                         "buffer_load_dwordx4 a[32:35], %0, %1, %2 offen offset:2048\n\t"
                         "buffer_load_dwordx4 a[36:39], %0, %1, %3 offen\n\t"
                         "buffer_load_dwordx4 a[40:43], %0, %1, %3 offen offset:1024\n\t"
                         "buffer_load_dwordx4 a[44:47], %0, %1, %3 offen offset:2048"
                         :: "v"(vo), "s"(wd), "s"(so), "s"(so + 3 * FRAG)
                         : "memory", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39",
                           "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47");
        else
            asm volatile("s_nop 4\n\t"
                         "buffer_load_dwordx4 a[0:3], %0, %1, %2 offen\n\t"
                         "buffer_load_dwordx4 a[4:7], %0, %1, %2 offen offset:1024\n\t"
                         "buffer_load_dwordx4 a[8:11], %0, %1, %2 offen offset:2048\n\t"
                         "buffer_load_dwordx4 a[12:15], %0, %1, %3 offen\n\t"
                         "buffer_load_dwordx4 a[16:19], %0, %1, %3 offen offset:1024\n\t"
                         "buffer_load_dwordx4 a[20:23], %0, %1, %3 offen offset:2048"
                         :: "v"(vo), "s"(wd), "s"(so), "s"(so + 3 * FRAG)
                         : "memory", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15",
                           "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23");
    };
    // wait until all but the N youngest loads have landed: the fragments requested one iteration ago are in their register set (no
    // landing copies: even and odd chunks own a set each and the MFMA statements name the set of their chunk)
#define F23_LAND(N) asm volatile("s_waitcnt vmcnt(" F23_CNT(N) ")" ::: "memory")
    auto fetch_b = [&](int ch) {
        const bool in = ch < p.nch;
        unsigned cf[8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const int ci = ch * 16 + sch * 8 + c;                       // wave-uniform
            cf[c] = ci < p.I ? (unsigned)ci * HWb : 0u;                 // padded channels alias channel 0 and meet a zero scale
        }
        // the eight style scales of (chunk, channel half) arrive through the SCALAR cache, straight into scalar registers (a vector load
        // + v_readlane per channel costs 15-29 cycles per readlane beside the partner wave's matrix instructions, profiles/r04_mfma_valu_coissue.txt);
        // the descriptor covers this sample's I scales and the range check is per dword: channels beyond I (they alias channel 0 of the
        // input) and chunks beyond the last read zeros
        const unsigned v0 = in ? g0 : 0x80000000u, v1 = in ? g1 : 0x80000000u;
        const unsigned ss = (unsigned)(ch * 16 + sch * 8) * 4u;        // wave-uniform
        // split: a column pair is 8 bytes (dwordx2); fp16: 4 bytes (dword) -- same sixteen requests, same wait counts
#define F23_REQ_B(LD) \
        asm volatile("s_nop 4\n\t" \
                     "s_buffer_load_dwordx8 %0, %9, %10\n\t" \
                     LD " %1, %11, %13, %14 offen\n\t" \
                     LD " %2, %12, %13, %14 offen\n\t" \
                     LD " %3, %11, %13, %15 offen\n\t" \
                     LD " %4, %12, %13, %15 offen\n\t" \
                     LD " %5, %11, %13, %16 offen\n\t" \
                     LD " %6, %12, %13, %16 offen\n\t" \
                     LD " %7, %11, %13, %17 offen\n\t" \
                     LD " %8, %12, %13, %17 offen" \
                     : "=&s"(rsc), "=&v"(rb[0][0]), "=&v"(rb[0][1]), "=&v"(rb[1][0]), "=&v"(rb[1][1]), "=&v"(rb[2][0]), "=&v"(rb[2][1]), "=&v"(rb[3][0]), "=&v"(rb[3][1]) \
                     : "s"(sd), "s"(ss), "v"(v0), "v"(v1), "s"(xd), "s"(cf[0]), "s"(cf[1]), "s"(cf[2]), "s"(cf[3]) : "memory"); \
        asm volatile("s_nop 4\n\t" \
                     LD " %0, %8, %10, %11 offen\n\t" \
                     LD " %1, %9, %10, %11 offen\n\t" \
                     LD " %2, %8, %10, %12 offen\n\t" \
                     LD " %3, %9, %10, %12 offen\n\t" \
                     LD " %4, %8, %10, %13 offen\n\t" \
                     LD " %5, %9, %10, %13 offen\n\t" \
                     LD " %6, %8, %10, %14 offen\n\t" \
                     LD " %7, %9, %10, %14 offen" \
                     : "=&v"(rb[4][0]), "=&v"(rb[4][1]), "=&v"(rb[5][0]), "=&v"(rb[5][1]), "=&v"(rb[6][0]), "=&v"(rb[6][1]), "=&v"(rb[7][0]), "=&v"(rb[7][1]) \
                     : "v"(v0), "v"(v1), "s"(xd), "s"(cf[4]), "s"(cf[5]), "s"(cf[6]), "s"(cf[7]) : "memory")
        if constexpr (SPLIT) { F23_REQ_B("buffer_load_dwordx2"); } else { F23_REQ_B("buffer_load_dword"); }
#undef F23_REQ_B
    };
    // AUDIT after every edit (tools/audit_f23_asm.py on the -save-temps .s): between a hand-issued load and the wait that covers it
    // hipcc must not read or copy the destination registers (it treats them as written when the load is issued).
    // the scalar request shares its counter with the LDS instructions and may return out of order: lgkmcnt(0) (nothing else of this
    // wave is in flight there when a staging block starts)
#define F23_WAIT_B(N) asm volatile("s_waitcnt vmcnt(" F23_CNT(N) ") lgkmcnt(0)" : "+s"(rsc), "+v"(rb[0][0]), "+v"(rb[0][1]), "+v"(rb[1][0]), "+v"(rb[1][1]), "+v"(rb[2][0]), "+v"(rb[2][1]), \
        "+v"(rb[3][0]), "+v"(rb[3][1]), "+v"(rb[4][0]), "+v"(rb[4][1]), "+v"(rb[5][0]), "+v"(rb[5][1]), "+v"(rb[6][0]), "+v"(rb[6][1]), "+v"(rb[7][0]), "+v"(rb[7][1]) :: "memory")
    static_assert(NB == F23_NB && NA == (SPLIT ? 6 : 3), "the wait counts below are written for these request sizes");
    // "all but the A request's loads": six (split) | three (fp16) younger loads stay out
#define F23_WAIT_B_NA() do { if constexpr (SPLIT) { F23_WAIT_B(6); } else { F23_WAIT_B(3); } } while (0)

    auto stage = [&](int buf) {
        unsigned char* dst = sm + buf * BUF + sL;
        if constexpr (SPLIT) {
        // Per channel: t2|3 = s d2|3;  V0 = s d0 - t2,  V1 = s d1 + t2,  V2 = t2 - s d1,  V3 = s d1 - t3 (the style scale rides in the
        // multiplies).  Per transform point and channel pair: hi = the two values truncated to fp16 (v_cvt_pkrtz), lo = fp16(value - hi)
        // by v_fma_mixlo / mixhi: three more.
        // SINGLE-ISSUE fp32 instructions only, each spelled as asm so that hipcc's SLP vectoriser cannot pair them into v_pk_*_f32:
        // packed fp32 is the one vector class that does not execute beside the SIMD's other wave's matrix instructions (each costs
        // its full ~10 cycles of matrix-pipe time), v_mul / v_fma / v_cvt_pkrtz / v_fma_mix vanish there (tools/microbench_coissue.hip,
        // profiles/r04_mfma_valu_coissue.txt).
        float vt[4][8];
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const unsigned sc = rsc[c];
            const float d0 = rb[c][0].x, d1 = rb[c][0].y, d2 = rb[c][1].x, d3 = rb[c][1].y;
            float t2, t3;
            asm("v_mul_f32 %0, %1, %2" : "=v"(t2) : "s"(sc), "v"(d2));
            asm("v_mul_f32 %0, %1, %2" : "=v"(t3) : "s"(sc), "v"(d3));
            asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(vt[0][c]) : "v"(d0), "s"(sc), "v"(t2));     // V0 = s d0 - s d2
            asm("v_fma_f32 %0, %1, %2, %3" : "=v"(vt[1][c]) : "v"(d1), "s"(sc), "v"(t2));      // V1 = s d1 + s d2
            asm("v_fma_f32 %0, -%1, %2, %3" : "=v"(vt[2][c]) : "v"(d1), "s"(sc), "v"(t2));     // V2 = s d2 - s d1
            asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(vt[3][c]) : "v"(d1), "s"(sc), "v"(t3));     // V3 = s d1 - s d3
        }
#pragma unroll
        for (int t = 0; t < 4; t++) {
            u32x4 hv, lv;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float x0 = vt[t][2 * c], x1 = vt[t][2 * c + 1];
                const unsigned h = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(x0, x1));
                unsigned l;
                asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(h), "v"(x0));
                asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(x1));
                hv[c] = h; lv[c] = l;
            }
            if (sOk) {
                *reinterpret_cast<u32x4*>(dst + (t * 4) * PLANE) = hv;
                *reinterpret_cast<u32x4*>(dst + (t * 4 + 2) * PLANE) = lv;
            }
        }
        } else {
        // fp16 tensors: a request register holds two halfs (d0 | d1 << 16), (d2 | d3 << 16).  The mixed-precision FMA reads them where
        // they lie (op_sel picks the half, op_sel_hi marks the operand as fp16), multiplies by the fp32 style scale, adds the fp32
        // partner term and rounds ONCE to fp16 (round to nearest even, the mode register's default) into the low | high half of the
        // packed result: six single-issue instructions per channel, no separate conversion or split.
        unsigned hw[4][4];                                               // [transform point][channel pair]
#pragma unroll
        for (int c = 0; c < 8; c++) {
            const unsigned sc = rsc[c];
            const unsigned own = rb[c][0], right = rb[c][1];
            float t2, t3;
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(t2) : "v"(right), "s"(sc));                       // s d2
            asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t3) : "v"(right), "s"(sc));         // s d3
            unsigned& o0 = hw[0][c >> 1]; unsigned& o1 = hw[1][c >> 1]; unsigned& o2 = hw[2][c >> 1]; unsigned& o3 = hw[3][c >> 1];
            if ((c & 1) == 0) {
                asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel_hi:[1,0,0]" : "=&v"(o0) : "v"(own), "s"(sc), "v"(t2));                   // V0 = s d0 - s d2
                asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(o1) : "v"(own), "s"(sc), "v"(t2));     // V1 = s d1 + s d2
                asm("v_fma_mixlo_f16 %0, -%1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(o2) : "v"(own), "s"(sc), "v"(t2));    // V2 = s d2 - s d1
                asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(o3) : "v"(own), "s"(sc), "v"(t3));    // V3 = s d1 - s d3
            } else {
                asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel_hi:[1,0,0]" : "+v"(o0) : "v"(own), "s"(sc), "v"(t2));
                asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(o1) : "v"(own), "s"(sc), "v"(t2));
                asm("v_fma_mixhi_f16 %0, -%1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(o2) : "v"(own), "s"(sc), "v"(t2));
                asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(o3) : "v"(own), "s"(sc), "v"(t3));
            }
        }
        if (sOk) {
#pragma unroll
            for (int t = 0; t < 4; t++) *reinterpret_cast<u32x4*>(dst + (t * 2) * PLANE) = (u32x4){hw[t][0], hw[t][1], hw[t][2], hw[t][3]};
        }
        }
    };
    struct BFrag { v8h h, l; };
    auto load_b = [&](BFrag& f, int buf, int q) {
        const unsigned char* src = sm + buf * BUF + bR + q * 256;
        f.h = *reinterpret_cast<const v8h*>(src);
        if constexpr (SPLIT) f.l = *reinterpret_cast<const v8h*>(src + 2 * PLANE);
    };
    // an M block of pure channel padding (O = 203: channels 224..255 of the fourth tile) stages and synchronises but issues no MFMAs
    const bool active = o0 + mb * 32 < p.O;                                  // wave-uniform
    auto mfma_chunk = [&](auto aset, int buf) {
        constexpr int AB = decltype(aset)::value;                                // 0 | 24: register set of this chunk's A fragments
        if (!active) return;
#if F23_PRIO_MODE == 1
        __builtin_amdgcn_s_setprio(1);
#elif F23_PRIO_MODE == 3
        __builtin_amdgcn_s_setprio(0);
#endif
        BFrag b0, b1;
        load_b(b0, buf, 0);
        // patch rows two at a time: the next row's fragments are requested before the current row's nine products are issued
#define F23_ROWPAIR(Q) \
        if constexpr ((Q) < TN + 2) { \
            if constexpr ((Q) + 1 < TN + 2) load_b(b1, buf, (Q) + 1); \
            __builtin_amdgcn_sched_barrier(0); \
            if constexpr (SPLIT) f23_mfma_row<AB, TN, (Q)>(b0.h, b0.l); else f23_mfma_row_f16<AB, TN, (Q)>(b0.h); \
            __builtin_amdgcn_sched_barrier(0); \
            if constexpr ((Q) + 1 < TN + 2) { \
                if constexpr ((Q) + 2 < TN + 2) load_b(b0, buf, (Q) + 2); \
                __builtin_amdgcn_sched_barrier(0); \
                if constexpr (SPLIT) f23_mfma_row<AB, TN, (Q) + 1>(b1.h, b1.l); else f23_mfma_row_f16<AB, TN, (Q) + 1>(b1.h); \
                __builtin_amdgcn_sched_barrier(0); \
            } \
        }
        F23_ROWPAIR(0) F23_ROWPAIR(2) F23_ROWPAIR(4) F23_ROWPAIR(6) F23_ROWPAIR(8)
#undef F23_ROWPAIR
        static_assert(TN + 2 <= 10, "row pairs spelled out for up to ten patch rows");
#if F23_PRIO_MODE == 1
        __builtin_amdgcn_s_setprio(0);
#elif F23_PRIO_MODE == 3
        __builtin_amdgcn_s_setprio(1);
#endif
    };

    // Chunk body: MFMA loop on chunk ch (B image and A register set ch & 1) and the staging block
    //   H(k, j) = wait B(k) | stage(k) into the other image | request B(k+1) | wait A(j) | request A(j+1)
    //   waves 0-3 ("early"):  MFMA(ch) | H(ch+1, ch+1) | barrier        -- lands what the NEXT iteration multiplies with
    //   waves 4-7 ("late"):   H(ch+1, ch) | MFMA(ch) | barrier          -- lands what THIS iteration multiplies with
    // so on every SIMD one wave's staging arithmetic runs beside the other's matrix instructions, and every request (A: 6 loads,
    // B: 17) is a whole iteration old when it is waited for.  Load queue, oldest first, when an iteration starts:
    // early B(ch+1), A(ch+1);  late B(ch+1), A(ch).  Inside H: wait B with 6 younger loads (the A request), land A with 17 (the B
    // request just issued).  Only the late waves' very first wait finds a different queue (the common prologue leaves B(1) alone
    // behind it): an extra operand-less s_waitcnt makes it exact.
    const bool late = wave >= 4;                                             // wave-uniform: second wave of its SIMD
    const int nch = p.nch;
#if F23_PRIO_MODE == 2
    if (late) __builtin_amdgcn_s_setprio(1);
#elif F23_PRIO_MODE == 4
    if (!late) __builtin_amdgcn_s_setprio(1);
#elif F23_PRIO_MODE == 3
    __builtin_amdgcn_s_setprio(1);
#endif
    fetch_a(0);
    fetch_b(0);
    F23_WAIT_B(0);                                                           // everything, A(0) included
    stage(0);
    fetch_b(1);
    if (!late) { F23_LAND(F23_NB); fetch_a(1); }                                 // the late waves land A(0) in their first H
    __syncthreads();
#ifdef SG3_F23_STAMPS
    unsigned long long tA = 0, tB = 0, tC = 0, tD = 0, tE = 0, sPre = 0, sMfma = 0, sPost = 0, sBar = 0, tStart, rStart;
    unsigned long long h1 = 0, h2 = 0, h3 = 0, h4 = 0, sH[5] = {0, 0, 0, 0, 0};
    F23_STAMP(tStart);
    rStart = __builtin_amdgcn_s_memrealtime();
#endif
    for (int ch = 0; ch < nch; ch++) {
        const int buf = ch & 1;
        const bool more1 = ch + 1 < nch;
        F23_STAMP(tA);
        // ONE MFMA site in the loop (the accumulators must not become a phi of two branches: hipcc then keeps two copies of them)
        if (late) {
            if (ch == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            F23_WAIT_B_NA();
            F23_STAMP(h1);
            if (more1) stage(buf ^ 1);
            F23_STAMP(h2);
            fetch_b(ch + 2);
            F23_STAMP(h3);
            F23_LAND(F23_NB);
            F23_STAMP(h4);
            fetch_a(ch + 1);
        }
        F23_STAMP(tB);
        // two copies of the matrix phase, one per register set (the accumulators are asm-named registers, not compiler values: no merge cost)
        if (buf) mfma_chunk(std::integral_constant<int, 24>{}, 1); else mfma_chunk(std::integral_constant<int, 0>{}, 0);
        F23_STAMP(tC);
        if (!late) {
            F23_WAIT_B_NA();
            F23_STAMP(h1);
            if (more1) stage(buf ^ 1);
            F23_STAMP(h2);
            fetch_b(ch + 2);
            F23_STAMP(h3);
            F23_LAND(F23_NB);
            F23_STAMP(h4);
            fetch_a(ch + 2);
        }
        F23_STAMP(tD);
        __syncthreads();
#ifdef SG3_F23_STAMPS
        F23_STAMP(tE);
        sPre += tB - tA; sMfma += tC - tB; sPost += tD - tC; sBar += tE - tD;
        { const unsigned long long h0 = late ? tA : tC, hE = late ? tB : tD;
          sH[0] += h1 - h0; sH[1] += h2 - h1; sH[2] += h3 - h2; sH[3] += h4 - h3; sH[4] += hE - h4; }
#endif
    }
    // The last requests (beyond the last chunk) are never consumed, but the destination registers of the input samples stay RESERVED
    // until they have landed: the wait names them.  (An operand-less s_waitcnt here let hipcc reuse them for the epilogue's addresses
    // before the wait, and the returning zeros overwrote those.)  The matrix instructions were issued from asm statements: the
    // compiler does not know their latency, so the accumulators are given the wait states an XDL result needs before they are read.
    F23_WAIT_B(0);
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
#ifdef SG3_F23_STAMPS
    if (p.stamps && lane == 0) {
        unsigned long long tEnd; F23_STAMP(tEnd);
        const unsigned long long rEnd = __builtin_amdgcn_s_memrealtime();
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[0] = sPre; o[1] = sMfma; o[2] = sPost; o[3] = sBar; o[4] = tEnd - tStart; o[5] = rEnd - rStart;
        wgLoop += tEnd - tStart; wgTiles++; wgPro += tStart - tTop; tLoopEnd = tEnd;
        unsigned long long* o2 = p.stamps + (1u << 22) + ((size_t)blockIdx.x * 8 + wave) * 8;      // second half of the buffer: the staging block's parts
        for (int k = 0; k < 5; k++) o2[k] = sH[k];
    }
#endif

    // ---- output transform: the four xi-waves of an M block exchange their accumulators through LDS, one tile row pair at a time ----
    // wave (xi, mb) finishes accumulator registers 4 xi .. 4 xi + 3 = channels o0 + 32 mb + 8 xi + 4 lh + j of all 32 columns
    float* X = reinterpret_cast<float*>(sm);                                  // [2][8 waves][16 registers][64 lanes]
    const int oc = o0 + mb * 32 + 8 * xi + 4 * lh;
    float d[4];
    {
        const __amdgpu_buffer_rsrc_t dr = __builtin_amdgcn_make_buffer_rsrc((void*)(p.dcoef + (size_t)n * p.O), (short)0, p.O * 4, 0x00020000);
#pragma unroll
        for (int j = 0; j < 4; j++) d[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(dr, (oc + j) * 4, 0, 0));
    }
    const unsigned planeB = (unsigned)(p.outH * p.outPitch) * (unsigned)EB;
    const __amdgpu_buffer_rsrc_t orr = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(static_cast<T*>(p.out) + (size_t)n * p.O * p.outH * p.outPitch), (short)0, (int)((unsigned)p.O * planeB), 0x00020000);
    const int gx = x0 + 2 * pq;
    const unsigned colOff = gx < p.outW ? (unsigned)oc * planeB + (unsigned)gx * (unsigned)EB : 0x80000000u;
    auto finish_row = [&](auto bc) {
        constexpr int b = decltype(bc)::value;
        float accv[16];
        F23Seq<48 + 16 * b, 16>::read(accv);
        float* Xb = X + (b & 1) * (8 * 16 * 64);
#pragma unroll
        for (int r = 0; r < 16; r++) Xb[(wave * 16 + r) * 64 + lane] = accv[r];
        __syncthreads();
        const int gy = y0 + rg * TN + b;
        const unsigned rowOff = gy < p.outH ? colOff + (unsigned)(gy * p.outPitch) * (unsigned)EB : 0x80000000u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const float* src = Xb + ((mb * 4) * 16 + 4 * xi + j) * 64 + lane;
            const float m0 = src[0], m1 = src[16 * 64], m2 = src[2 * 16 * 64], m3 = src[3 * 16 * 64];
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const float ya = (m0 + m1 + m2) * d[j], yb = (m1 - m2 - m3) * d[j];
            if constexpr (SPLIT) {
                const u32x2 v = {__builtin_bit_cast(unsigned, ya), __builtin_bit_cast(unsigned, yb)};
                __builtin_amdgcn_raw_buffer_store_b64(v, orr, (int)(rowOff + (unsigned)j * planeB), 0, 0);
            } else {
                const v2h h = round2(ya, yb);                              // fp16 output, round to nearest even (the reference's fp16 convolution result)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, h), orr, (int)(rowOff + (unsigned)j * planeB), 0, 0);
            }
        }
    };
    finish_row(std::integral_constant<int, 0>{}); finish_row(std::integral_constant<int, 1>{});
    finish_row(std::integral_constant<int, 2>{}); finish_row(std::integral_constant<int, 3>{});
    if constexpr (TN > 4) finish_row(std::integral_constant<int, (TN > 4 ? 4 : 0)>{});
    if constexpr (TN > 5) finish_row(std::integral_constant<int, (TN > 5 ? 5 : 0)>{});
    if constexpr (TN > 6) finish_row(std::integral_constant<int, (TN > 6 ? 6 : 0)>{});
    __syncthreads();                                        // the next tile's first staging block writes where this exchange was read
#ifdef SG3_F23_STAMPS
    if (p.stamps && lane == 0) {        // whole-workgroup clock so far, K-loop share of it, tiles walked
        unsigned long long now; F23_STAMP(now);
        unsigned long long* o = p.stamps + ((size_t)blockIdx.x * 8 + wave) * 8;
        o[6] = now - wgStart; o[7] = wgLoop;
        wgEpi += now - tLoopEnd;
        p.stamps[(1u << 22) + ((size_t)blockIdx.x * 8 + wave) * 8 + 6] = wgPro;
        p.stamps[(1u << 22) + ((size_t)blockIdx.x * 8 + wave) * 8 + 7] = wgEpi;
        p.stamps[(1u << 22) + ((size_t)blockIdx.x * 8 + wave) * 8 + 5] = wgTiles;
    }
#endif
    }   // tiles of this workgroup
}

int64_t f23_packed_floats(int O, int I, bool split) {
    return (int64_t)ceil_div(O, 64) * ceil_div(I, 16) * (2 * 4 * (split ? 6 : 3) * 1024 / 4);
}

bool f23_supported(int dtype, int I, int O, int H, int W, int k, int pad, int outRowStride) {
    if (k != 3 || (dtype != SG3_F32 && dtype != SG3_F16)) return false;
    const long long eb = dtype == SG3_F32 ? 4 : 2;
    if ((W & 1) || (pad & 1)) return false;                                    // aligned column pairs (8 bytes | 4 bytes)
    const int outH = H + 2 * pad - 2, outW = W + 2 * pad - 2;
    if (outH <= 0 || outW <= 0) return false;
    const long long pitch = outRowStride > 0 ? outRowStride : outW;
    if ((pitch & 1) != 0) return false;
    // every store offset a lane can form (padded channels included) stays below 2^31 -- also on the 128-byte aligned row pitch the
    // inference path may choose AFTER the weights were packed in this kernel's order (the plan-time query passes outRowStride = 0)
    const long long worst = std::max<long long>(pitch, (outW + 31) / 32 * 32);
    if ((long long)(ceil_div(O, 64) * 64 + 32) * outH * worst * eb >= 0x7fffffffLL) return false;
    if ((long long)I * H * W * eb >= 0x7fffffffLL) return false;
    if ((long long)ceil_div(O, 64) * ceil_div(I, 16) * 49152 >= 0x7fffffffLL) return false;
    return true;
}

template <int TN, typename T>
static int launch_f23(const sg3_modconv_params& q, hipStream_t st) {
    constexpr size_t imageBytes = (size_t)2 * (sizeof(T) == 4 ? 16 : 8) * (2 * TN + 2) * 256;
    constexpr size_t ldsBytes = imageBytes > 65536 ? imageBytes : 65536;       // double-buffered B image, at least the 64 KB exchange area
    static_assert(ldsBytes <= 160 * 1024, "LDS");
    F23Params p;
    p.x = q.x; p.wp = q.wPacked; p.sIn = q.sIn; p.dcoef = q.dcoef; p.out = q.out;
    p.N = q.N; p.I = q.I; p.O = q.O; p.H = q.H; p.W = q.W; p.pad = q.pad;
    p.outH = q.H + 2 * q.pad - 2; p.outW = q.W + 2 * q.pad - 2;
    p.nch = ceil_div(q.I, 16);
    p.xTiles = ceil_div(p.outW, 32); p.yTiles = ceil_div(p.outH, 2 * TN); p.mTiles = ceil_div(q.O, 64);
    const long long total = (long long)p.xTiles * p.yTiles * p.mTiles * q.N;
    if (total > 0x7fffffffLL) { set_error("modulated_conv2d: grid too large"); return SG3_BAD_ARG; }
    p.totalBlocks = (int)total;
    p.outPitch = q.outRowStride > 0 ? q.outRowStride : p.outW;
#ifdef SG3_F23_STAMPS
    p.stamps = g_f23_stamps;
#endif
    auto kern = modconv_f23_kernel<TN, T>;
    // per device, once: the CU count (grid of persistent workgroups) and this instantiation's dynamic-LDS limit
    struct DevState { int cus; bool attr; };
    static DevState devs[64] = {};
    int dev = 0;
    SG3_HIP_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) { set_error("modulated_conv2d: device index %d out of range", dev); return SG3_BAD_ARG; }
    DevState& ds = devs[dev];
    if (!ds.attr) {
        SG3_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes));
        int n = 0;
        ds.cus = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
        ds.attr = true;
    }
    const int cus = ds.cus;
#ifdef SG3_F23_ONE_TILE
    const unsigned grid = (unsigned)total;
#else
    const unsigned grid = (unsigned)std::min<long long>(total, cus);       // one resident workgroup per CU walks the tiles
#endif
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), ldsBytes, st, p);
    SG3_LAUNCH_CHECK("modconv_f23_kernel");
    return SG3_OK;
}

// forced rows per wave: seeded ONCE from the environment (SG3_F23_TN), changed only by an explicit sg3_modconv_f23_force_rows call
static std::atomic<int> g_f23_rows{[] { const char* fe = getenv("SG3_F23_TN"); const int v = fe ? atoi(fe) : 0; return (v == 4 || v == 5 || v == 7) ? v : 0; }()};
int f23_force_rows(int rows) { return g_f23_rows.exchange((rows == 4 || rows == 5 || rows == 7) ? rows : 0); }

int launch_conv_f23(const sg3_modconv_params& q, hipStream_t st) {
    // rows per wave: one workgroup per CU, so the time goes with (rounds of 256 workgroups) x (rows per wave + per-chunk overhead)
    const int forced = g_f23_rows.load(std::memory_order_relaxed);          // sg3_modconv_f23_force_rows (tests, A/B timing); 0 = cost model
    const int outH = q.H + 2 * q.pad - 2, outW = q.W + 2 * q.pad - 2;
    const long long per = (long long)q.N * ceil_div(q.O, 64) * ceil_div(outW, 32);
    int best = 7; double bestCost = 1e300;
    const int cands[3] = {7, 5, 4};
    for (int c = 0; c < 3; c++) {
        const int tn = cands[c];
        const long long wgs = per * ceil_div(outH, 2 * tn);
        const double rounds = wgs <= 1024 ? (double)ceil_div64(wgs, 256) : wgs / 256.0;
        const double cost = rounds * (tn + 0.6);
        if (cost < bestCost * 0.999) { bestCost = cost; best = tn; }
    }
    if (forced == 4 || forced == 5 || forced == 7) best = forced;
    if (q.dtype == SG3_F16) {
        switch (best) {
            case 4: return launch_f23<4, _Float16>(q, st);
            case 5: return launch_f23<5, _Float16>(q, st);
            default: return launch_f23<7, _Float16>(q, st);
        }
    }
    switch (best) {
        case 4: return launch_f23<4, float>(q, st);
        case 5: return launch_f23<5, float>(q, st);
        default: return launch_f23<7, float>(q, st);
    }
}

} // namespace sg3
