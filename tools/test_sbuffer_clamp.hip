// Is the range check of s_buffer_load_dwordx8 applied per dword on gfx950?  (raw descriptor, stride 0, num_records = 12 bytes:
// dwords 0..2 in range, 3..7 out of range.)  Prints the eight values a wave reads; expected with a per-dword check: 1 2 3 0 0 0 0 0.
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/test_sbuffer_clamp.hip -o /tmp/sbc && /tmp/sbc
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x8 __attribute__((ext_vector_type(8)));
__global__ void k(const float* src, float* dst, unsigned bytes, unsigned off) {
    const unsigned long long a = (unsigned long long)src;
    u32x4 d;
    d.x = __builtin_amdgcn_readfirstlane((unsigned)a);
    d.y = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    d.z = __builtin_amdgcn_readfirstlane(bytes);
    d.w = 0x00020000u;
    const unsigned so = __builtin_amdgcn_readfirstlane(off);
    u32x8 v;
    asm volatile("s_nop 4\n\ts_buffer_load_dwordx8 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) : "s"(d), "s"(so) : "memory");
    unsigned e[8] = {v.s0, v.s1, v.s2, v.s3, v.s4, v.s5, v.s6, v.s7};      // element copies first: hipcc 7.2 miscompiles a bit cast of a vector element expression
    if (threadIdx.x == 0) for (int i = 0; i < 8; i++) dst[i] = __builtin_bit_cast(float, e[i]);
}
int main() {
    float h[16]; for (int i = 0; i < 16; i++) h[i] = (float)(i + 1);
    float *s, *d; hipMalloc(&s, 64); hipMalloc(&d, 32); hipMemcpy(s, h, 64, hipMemcpyHostToDevice);
    const unsigned cases[][2] = {{12, 0}, {44, 32}, {64, 32}, {12, 16}, {30, 16}};
    for (auto& c : cases) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, s, d, c[0], c[1]);
        float o[8]; hipMemcpy(o, d, 32, hipMemcpyDeviceToHost);
        printf("num_records %2u offset %2u:", c[0], c[1]); for (int i = 0; i < 8; i++) printf(" %g", o[i]); printf("\n");
    }
    return 0;
}
