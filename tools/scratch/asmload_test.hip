// isolates the hand-issued dwordx4 loads of modconv_f23_kernel: asm loads + asm wait vs builtin loads, same descriptor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const void* wp, unsigned bytes, unsigned* out, int mode, int nch) {
    const int lane = threadIdx.x & 63;
    const unsigned long long a = (unsigned long long)wp;
    u32x4 d;
    d.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    d.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    d.z = __builtin_amdgcn_readfirstlane(bytes);
    d.w = 0x00020000u;
    const unsigned aG = (unsigned)lane * 16u;
    u32x4 af[6];
    u32x4 acc = {0, 0, 0, 0};
    for (int ch = 0; ch < nch; ch++) {
        const unsigned so = (unsigned)ch * 6144u;
        const unsigned vo = ch < nch ? aG : 0x80000000u;
        if (mode == 0) {
            const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc((void*)wp, (short)0, (int)bytes, 0x00020000);
#pragma unroll
            for (int f = 0; f < 6; f++) af[f] = __builtin_amdgcn_raw_buffer_load_b128(wr, (int)vo, (int)(so + f * 1024), 0);
        } else {
#pragma unroll
            for (int f = 0; f < 6; f++)
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(af[f]) : "v"(vo), "s"(d), "s"(so + f * 1024) : "memory");
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(af[3]), "+v"(af[4]), "+v"(af[5]) :: "memory");
        }
#pragma unroll
        for (int f = 0; f < 6; f++) acc += af[f] * (unsigned)(f + 1 + ch);
    }
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 0] = acc.x; out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 1] = acc.y;
    out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 2] = acc.z; out[(blockIdx.x * blockDim.x + threadIdx.x) * 4 + 3] = acc.w;
}
int main() {
    const int nch = 4; const unsigned bytes = nch * 6144;
    std::vector<unsigned> h(bytes / 4); for (size_t i = 0; i < h.size(); i++) h[i] = (unsigned)(i * 2654435761u);
    void* w; unsigned *o0, *o1; hipMalloc(&w, bytes); hipMalloc(&o0, 64 * 16); hipMalloc(&o1, 64 * 16);
    hipMemcpy(w, h.data(), bytes, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, w, bytes, o0, 0, nch);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, w, bytes, o1, 1, nch);
    std::vector<unsigned> a(256), b(256); hipMemcpy(a.data(), o0, 1024, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o1, 1024, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 256; i++) bad += a[i] != b[i];
    printf("asm vs builtin: %d of 256 words differ (first: %08x vs %08x)\n", bad, a[0], b[0]);
    return 0;
}
