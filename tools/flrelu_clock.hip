// In-kernel clock of flrelu_stream_kernel on the T-1024 layer geometries (diagnostic build: the kernel source is compiled
// here with -DSG3_STAMPS, which makes every wave stamp s_memtime / s_memrealtime at its start and end; the product library
// carries no stamps).  clock = d(shader cycles) / d(100 MHz ticks), median over the waves of the last launch after ~1.5 s of
// back-to-back launches on random data (MI355X_MICROARCH.md, "DVFS give-back" item 6).  Three memory variants per layer:
//   hbm     every plane has its own input and output (the product's traffic)
//   in$     all planes read ONE input plane (cache resident), outputs go to HBM
//   in$out$ all planes also write ONE output plane
//   xpitch / xypitch   the product's traffic with the input rows / input and output rows on a 128-byte pitch
// so the time and the clock with and without the HBM stream can be told apart (the row pitch makes no difference to this kernel:
// L11 1208 / 1209 / 1216 us; it does to the convolution that writes those rows).
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-honor-nans -DSG3_STAMPS tools/flrelu_clock.hip -o /tmp/fc && /tmp/fc
#include "../stylegan3-editing_amd/csrc/sg3_filtered_lrelu.hip"
#include <algorithm>
#include <cstdlib>
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

struct Layer { const char* name; int C, x, up, pad0, pad1; };

int main() {
    const int N = 8;
    const Layer layers[] = {{"L6  512ch 150 up2", 512, 150, 2, 9, 8}, {"L9  128ch 278 up4", 128, 278, 4, -6, -9}, {"L10  81ch 534 up4", 81, 534, 4, -6, -9},
                            {"L11  51ch 1046 up2", 51, 1046, 2, 9, 8}, {"L13  32ch 1046 up2", 32, 1046, 2, -11, -12}};
    const size_t maxElems = (size_t)N * 81 * 1056 * 1056 + 4096;
    float *x, *y, *b, *fu, *fd; unsigned long long* stamps;
    const size_t stampSlots = 4u << 20;
    hipMalloc(&x, maxElems * 4); hipMalloc(&y, maxElems * 4); hipMalloc(&b, 512 * 4); hipMalloc(&fu, 24 * 4); hipMalloc(&fd, 12 * 4);
    hipMalloc(&stamps, stampSlots * 16);
    {
        std::vector<float> h(maxElems);
        unsigned s = 12345u;
        for (auto& v : h) {     // sum of 4 uniforms: roughly normal, random significands
            float a = 0; for (int i = 0; i < 4; i++) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; }
            v = a * 1.7f;
        }
        hipMemcpy(x, h.data(), maxElems * 4, hipMemcpyHostToDevice);
        hipMemcpy(b, h.data(), 512 * 4, hipMemcpyHostToDevice);
    }
    sg3::g_stamps = stamps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-20s %-8s %9s %9s %9s %10s %9s\n", "layer", "memory", "us", "alg TB/s", "GHz med", "GHz p10-p90", "wave us");
    for (const Layer& L : layers) {
        auto hu = lowpass(6 * L.up, L.up), hd = lowpass(12, 2);
        hipMemcpy(fu, hu.data(), hu.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(fd, hd.data(), hd.size() * 4, hipMemcpyHostToDevice);
        int yH, yW;
        if (sg3_filtered_lrelu_shape(L.x, L.x, L.up, 2, 6 * L.up, 0, 12, 0, L.pad0, L.pad1, L.pad0, L.pad1, &yH, &yW, nullptr, nullptr, nullptr) != 0) return 1;
        for (int variant = 0; variant < 5; variant++) {          // 3: input rows on a 128-byte pitch; 4: input and output rows
            sg3_filtered_lrelu_params p = {};
            p.x = x; p.y = y; p.b = b; p.fu = fu; p.fd = fd; p.dtype = SG3_F32; p.N = N; p.C = L.C; p.xH = p.xW = L.x; p.yH = yH; p.yW = yW;
            const int xp = variant >= 3 ? (L.x + 31) / 32 * 32 : L.x, yp = variant >= 4 ? (yW + 31) / 32 * 32 : yW;
            p.xStride[3] = 1; p.xStride[2] = xp; p.xStride[1] = (variant == 1 || variant == 2) ? 0 : (int64_t)L.x * xp; p.xStride[0] = p.xStride[1] * L.C;
            p.yStride[3] = 1; p.yStride[2] = yp; p.yStride[1] = variant == 2 ? 0 : (int64_t)yH * yp; p.yStride[0] = p.yStride[1] * L.C;
            p.bStride = 1; p.up = L.up; p.down = 2; p.fuW = 6 * L.up; p.fdW = 12; p.px0 = p.py0 = L.pad0;
            p.gain = 1.41421356f; p.slope = 0.2f; p.clamp = 256.f;
            float ms = 0; int reps = 0; double total = 0;
            while (total < 1500.0) {      // ~1.5 s back to back, timed in groups of 20 launches
                hipEventRecord(e0);
                for (int i = 0; i < 20; i++) if (sg3_filtered_lrelu(&p, nullptr) != 0) return 2;
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1); total += ms; reps++;
            }
            const double us = ms / 20 * 1e3;
            int ns, tw, nc, ch;
            sg3::stream_grid(N, L.C, yH, yW, 2, ns, tw, nc, ch);
            const size_t blocks = (size_t)N * L.C * ns * nc;
            if (blocks > stampSlots) return 3;
            std::vector<unsigned long long> h(blocks * 2);
            hipMemcpy(h.data(), stamps, blocks * 16, hipMemcpyDeviceToHost);
            std::vector<double> ghz; double life = 0;
            for (size_t i = 0; i < blocks; i++) if (h[2 * i + 1] > 50) { ghz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1); life += h[2 * i + 1] * 0.01; }
            std::sort(ghz.begin(), ghz.end());
            const double bytes = (double)N * L.C * ((double)L.x * L.x + (double)yH * yW) * 4;
            printf("%-20s %-8s %9.1f %9.2f %9.3f %5.2f-%4.2f %9.1f\n", L.name, variant == 0 ? "hbm" : (variant == 1 ? "in$" : (variant == 2 ? "in$out$" : (variant == 3 ? "xpitch" : "xypitch"))), us,
                   bytes / us * 1e-6, ghz.empty() ? 0.0 : ghz[ghz.size() / 2], ghz.empty() ? 0.0 : ghz[ghz.size() / 10], ghz.empty() ? 0.0 : ghz[ghz.size() * 9 / 10],
                   ghz.empty() ? 0.0 : life / ghz.size());
            fflush(stdout);
        }
    }
    return 0;
}
