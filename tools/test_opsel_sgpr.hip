#include <hip/hip_runtime.h>
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float sg(float v){ return __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, v))); }
// acc += a * splat(hi half of SGPR pair)
__device__ __forceinline__ v2f fma_hi(v2f a, v2f tp, v2f acc) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(a), "s"(tp), "v"(acc));
    return r;
}
__device__ __forceinline__ v2f fma_lo(v2f a, v2f tp, v2f acc) {
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(a), "s"(tp), "v"(acc));
    return r;
}
__global__ void k(const float* t, v2f* io) {
    v2f tp[6];
#pragma unroll
    for (int i=0;i<6;i++) tp[i] = (v2f){sg(t[2*i]), sg(t[2*i+1])};
    v2f a = io[threadIdx.x], acc = {0,0};
#pragma unroll
    for (int i=0;i<6;i++) {
        acc = fma_lo(a, tp[i], acc);
        a = fma_hi(acc, tp[i], a);
    }
    io[threadIdx.x] = acc + a;
}
int main(){ float ht[12]; for(int i=0;i<12;i++) ht[i]=0.1f*(i+1); float* dt; v2f* dio; hipMalloc(&dt,48); hipMalloc(&dio,64*8); hipMemcpy(dt,ht,48,hipMemcpyHostToDevice);
 v2f h[64]; for(int i=0;i<64;i++) h[i]=(v2f){1.f+i, 2.f}; hipMemcpy(dio,h,512,hipMemcpyHostToDevice); hipLaunchKernelGGL(k,1,64,0,0,dt,dio); hipMemcpy(h,dio,512,hipMemcpyDeviceToHost);
 // host ref
 for (int l=0;l<2;l++){ float ax=1.f+l, ay=2.f, cx=0, cy=0; for(int i=0;i<6;i++){ cx=fmaf(ax,ht[2*i],cx); cy=fmaf(ay,ht[2*i],cy); ax=fmaf(cx,ht[2*i+1],ax); ay=fmaf(cy,ht[2*i+1],ay);} printf("lane %d gpu %g %g ref %g %g\n", l, h[l].x, h[l].y, cx+ax, cy+ay);} return 0; }
