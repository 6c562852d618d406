// VERDICT r3 item 6 (whole-layer fusion): a measured LOWER BOUND for fusing L13's filtered_lrelu with the
// ToRGB convolution behind it (reference networks_stylegan3.py:335-368).  Today: L13 filtered_lrelu writes [8,32,1024,1024] fp32 that
// the ToRGB 1x1 kernel reads back.  A fused kernel (16 waves = 16 channels of one strip per workgroup, weighted RGB partial sums
// exchanged through LDS, two partial RGB images stored) must do at least what these diagnostic builds of the product kernel do:
//     build 0  the product kernel (stores its 32 planes)
//     build 1  -DSG3_FUSION_BOUND=1: no store at all
//     build 2  -DSG3_FUSION_BOUND=2: no store; per lane and output row three packed multiplies + three 8-byte LDS writes (the wave's
//              share of the partial sums) -- WITHOUT the cross-wave exchange (>= 16 LDS reads + 15 adds + a store per (plane, pixel
//              pair), a workgroup barrier per row group) and without the pass that adds the two partial images, bias and clamp
//              (>= 0.3 GB of traffic, ~0.07 ms).
// Go / no-go of the VERDICT: fused filtered_lrelu + what is left of ToRGB <= 0.75 ms (0.686 + 0.207 = 0.89 ms today), i.e. the fused
// filtered_lrelu itself <= ~0.68 ms.  If build 2 is not clearly below that, no fused kernel can be.
// Build + run on the GPU box (three binaries):
//   for b in 0 1 2; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans $([ $b -gt 0 ] && echo -DSG3_FUSION_BOUND=$b) tools/flrelu_fusion_bound.hip -o /tmp/fb$b && /tmp/fb$b; done
#include "../stylegan3-editing_amd/csrc/sg3_filtered_lrelu.hip"
#include <vector>

namespace sg3 { void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); } }

static std::vector<float> lowpass(int taps, int up) {
    std::vector<float> f(taps);
    double s = 0;
    for (int k = 0; k < taps; k++) {
        const double m = k - (taps - 1) / 2.0, c = 0.8 / up;
        const double sinc = m == 0 ? 1.0 : sin(M_PI * c * m) / (M_PI * c * m);
        f[k] = (float)(sinc * (0.5 + 0.5 * cos(2 * M_PI * m / taps)));
        s += f[k];
    }
    for (auto& v : f) v = (float)(v / s);
    return f;
}

int main() {
    const int N = 8, C = 32, X = 1046, up = 2, pad0 = -11, pad1 = -12;       // L13 of T-1024: 1046^2 -> 1024^2
    const size_t elems = (size_t)N * C * 1056 * 1056 + 4096;
    float *x, *y, *b, *fu, *fd;
    hipMalloc(&x, elems * 4); hipMalloc(&y, elems * 4); hipMalloc(&b, 512 * 4); hipMalloc(&fu, 24 * 4); hipMalloc(&fd, 12 * 4);
    {
        std::vector<float> h(elems);
        unsigned s = 12345u;
        for (auto& v : h) { float a = 0; for (int i = 0; i < 4; i++) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; } v = a * 1.7f; }
        hipMemcpy(x, h.data(), elems * 4, hipMemcpyHostToDevice); hipMemcpy(b, h.data(), 512 * 4, hipMemcpyHostToDevice);
    }
    auto hu = lowpass(6 * up, up), hd = lowpass(12, 2);
    hipMemcpy(fu, hu.data(), hu.size() * 4, hipMemcpyHostToDevice); hipMemcpy(fd, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
    int yH, yW;
    if (sg3_filtered_lrelu_shape(X, X, up, 2, 6 * up, 0, 12, 0, pad0, pad1, pad0, pad1, &yH, &yW, nullptr, nullptr, nullptr) != 0) return 1;
    sg3_filtered_lrelu_params p = {};
    p.x = x; p.y = y; p.b = b; p.fu = fu; p.fd = fd; p.dtype = SG3_F32; p.N = N; p.C = C; p.xH = p.xW = X; p.yH = yH; p.yW = yW;
    p.xStride[3] = 1; p.xStride[2] = X; p.xStride[1] = (int64_t)X * X; p.xStride[0] = p.xStride[1] * C;
    p.yStride[3] = 1; p.yStride[2] = yW; p.yStride[1] = (int64_t)yH * yW; p.yStride[0] = p.yStride[1] * C;
    p.bStride = 1; p.up = up; p.down = 2; p.fuW = 6 * up; p.fdW = 12; p.px0 = p.py0 = pad0;
    p.gain = 1.41421356f; p.slope = 0.2f; p.clamp = 256.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0; double total = 0; std::vector<double> us;
    while (total < 1500.0) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) if (sg3_filtered_lrelu(&p, nullptr) != 0) return 2;
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1); total += ms; us.push_back(ms / 20 * 1e3);
    }
#ifdef SG3_FUSION_BOUND
    const int build = SG3_FUSION_BOUND;
#else
    const int build = 0;
#endif
    printf("L13 filtered_lrelu [8,32,1046,1046] -> [8,32,%d,%d], build %d (%s): %.1f us (last of %zu groups of 20 launches; first %.1f)\n", yH, yW, build,
           build == 0 ? "product kernel" : (build == 1 ? "no store" : "no store, RGB partial sums written to LDS per wave"), us.back(), us.size(), us.front());
    return 0;
}
