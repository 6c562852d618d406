// Where a chunk of modconv_f23_kernel spends its cycles (diagnostic build: the kernel source is compiled here with
// -DSG3_F23_STAMPS; the product library carries no stamps).  Per wave and 16-channel chunk, s_memtime sums of the four
// segments of the loop body -- [late waves: stage + input requests] [MFMA loop] [A request; early waves: stage + input requests]
// [barrier] -- averaged over workgroups, for the early (0-3) and late (4-7) waves, plus the in-kernel clock.
// The stamped build serialises what the real kernel overlaps: read the SHARES, not the length.
// Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DSG3_F23_STAMPS tools/f23_stamps.hip -o /tmp/f23s && /tmp/f23s [TN]
#include "../stylegan3-editing_amd/csrc/sg3_modconv_f23.hip"
#include <algorithm>
#include <vector>

namespace sg3 { void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); } }

struct Layer { const char* name; int I, O, H; };

int main(int argc, char** argv) {
    if (argc > 1) sg3::f23_force_rows(atoi(argv[1]));
    const int N = 8;
    const Layer layers[] = {{"L5 512->512 @84", 512, 512, 84}, {"L6 512->512 @148", 512, 512, 148}, {"L8 323->203 @276", 323, 203, 276}, {"L9 203->128 @532", 203, 128, 532}};
    size_t maxIn = 0, maxOut = 0, maxW = 0;
    for (const Layer& L : layers) {
        maxIn = std::max(maxIn, (size_t)N * L.I * L.H * L.H); maxOut = std::max(maxOut, (size_t)N * L.O * (L.H + 2) * (L.H + 2));
        maxW = std::max(maxW, (size_t)sg3::f23_packed_floats(L.O, L.I, true));
    }
    float *x, *out, *sIn, *dcoef, *wp; unsigned long long* stamps;
    const size_t stampWgs = 1u << 16;                 // 64 u64 per workgroup in each half of the buffer
    hipMalloc(&x, maxIn * 4); hipMalloc(&out, maxOut * 4); hipMalloc(&sIn, N * 512 * 4); hipMalloc(&dcoef, N * 512 * 4); hipMalloc(&wp, maxW * 4);
    hipMalloc(&stamps, (size_t)2 * stampWgs * 64 * 8);
    {
        std::vector<float> h(maxIn);
        unsigned s = 12345u;
        auto rnd = [&]() { float a = 0; for (int i = 0; i < 4; i++) { s = s * 1664525u + 1013904223u; a += (float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f; } return a * 1.7f; };
        for (auto& v : h) v = rnd() * 2000.f;                       // |x * sIn| up to ~2^13 as in the product after the power-of-two rescale
        hipMemcpy(x, h.data(), maxIn * 4, hipMemcpyHostToDevice);
        std::vector<float> sc(N * 512); for (auto& v : sc) v = 1.f + 0.3f * rnd();
        hipMemcpy(sIn, sc.data(), sc.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dcoef, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
        std::vector<_Float16> hw(maxW * 2);
        for (size_t i = 0; i < hw.size(); i++) hw[i] = (_Float16)(((i / 512) & 1) ? rnd() * 1e-3f : rnd());     // hi | lo fragments alternate
        hipMemcpy(wp, hw.data(), maxW * 4, hipMemcpyHostToDevice);
    }
    sg3::g_f23_stamps = stamps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("%-20s %8s %6s | per chunk and wave, cycles: %-28s | %-28s | GHz\n", "layer", "us", "wgs", "early: pre mfma post barrier", "late: pre mfma post barrier");
    for (const Layer& L : layers) {
        sg3_modconv_params q = {};
        q.x = x; q.wPacked = wp; q.sIn = sIn; q.dcoef = dcoef; q.out = out; q.dtype = SG3_F32;
        q.N = N; q.I = L.I; q.O = L.O; q.H = q.W = L.H; q.k = 3; q.pad = 2; q.precision = SG3_CONV_F16X3_F23;
        float ms = 0; double total = 0;
        while (total < 1000.0) {
            hipEventRecord(e0);
            for (int i = 0; i < 10; i++) if (sg3::launch_conv_f23(q, nullptr) != 0) return 2;
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); total += ms;
        }
        hipDeviceSynchronize();
        // the grid of the launch: recompute as launch_conv_f23 does is not needed -- count the workgroups that wrote a clock
        std::vector<unsigned long long> h((size_t)2 * stampWgs * 64);
        hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
        hipMemset(stamps, 0, (size_t)2 * stampWgs * 64 * 8);
        double sum[2][4] = {}, hs[2][5] = {}; size_t cnt[2] = {}; std::vector<double> ghz;
        double wgTotal = 0, wgLoop = 0, wgTiles = 0, wgPro = 0, wgEpi = 0;
        const int nch = (L.I + 15) / 16;
        for (size_t wg = 0; wg < stampWgs; wg++)
            for (int w = 0; w < 8; w++) {
                const unsigned long long* o = &h[(wg * 8 + w) * 8];
                if (o[5] < 50) continue;
                for (int k = 0; k < 4; k++) sum[w >> 2][k] += (double)o[k] / nch;
                for (int k = 0; k < 5; k++) hs[w >> 2][k] += (double)h[(1u << 22) + (wg * 8 + w) * 8 + k] / nch;
                cnt[w >> 2]++; ghz.push_back((double)o[4] / (double)o[5] * 0.1);
                wgTotal += (double)o[6]; wgLoop += (double)o[7]; wgTiles += (double)h[(1u << 22) + (wg * 8 + w) * 8 + 5];
                wgPro += (double)h[(1u << 22) + (wg * 8 + w) * 8 + 6]; wgEpi += (double)h[(1u << 22) + (wg * 8 + w) * 8 + 7];
            }
        std::sort(ghz.begin(), ghz.end());
        printf("%-20s %8.1f %6zu | %6.0f %6.0f %6.0f %6.0f     | %6.0f %6.0f %6.0f %6.0f     | %.3f\n", L.name, ms / 10 * 1e3, cnt[0] / 4,
               sum[0][0] / cnt[0], sum[0][1] / cnt[0], sum[0][2] / cnt[0], sum[0][3] / cnt[0],
               sum[1][0] / cnt[1], sum[1][1] / cnt[1], sum[1][2] / cnt[1], sum[1][3] / cnt[1], ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
        printf("    staging block (wait B | stage | request B | land A | request A): early %5.0f %5.0f %5.0f %5.0f %5.0f   late %5.0f %5.0f %5.0f %5.0f %5.0f\n",
               hs[0][0] / cnt[0], hs[0][1] / cnt[0], hs[0][2] / cnt[0], hs[0][3] / cnt[0], hs[0][4] / cnt[0],
               hs[1][0] / cnt[1], hs[1][1] / cnt[1], hs[1][2] / cnt[1], hs[1][3] / cnt[1], hs[1][4] / cnt[1]);
        printf("    per tile and wave: %.0f cycles, of which the K loop %.0f (%.1f %%), the prologue (tile decode, first requests, first staging) %.0f, the output transform and stores %.0f\n",
               wgTotal / wgTiles, wgLoop / wgTiles, 100.0 * wgLoop / wgTotal, wgPro / wgTiles, wgEpi / wgTiles);
        fflush(stdout);
    }
    return 0;
}
