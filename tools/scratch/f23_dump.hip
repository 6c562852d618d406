// diagnostic: do the hand-issued A-fragment loads of modconv_f23_kernel deliver the packed weights?  (-DSG3_F23_STAMPS -DSG3_F23_DUMPA)
#include "../../stylegan3-editing_amd/csrc/sg3_modconv_f23.hip"
#include <vector>
namespace sg3 { void set_error(const char* fmt, ...) { va_list a; va_start(a, fmt); vfprintf(stderr, fmt, a); va_end(a); fputc('\n', stderr); } }
int main() {
    const int N = 1, I = 64, O = 64, H = 30;
    const int nch = 4;
    const size_t wFloats = (size_t)sg3::f23_packed_floats(O, I);
    float *x, *out, *sIn, *dcoef; unsigned* wp; unsigned long long* dbg;
    hipMalloc(&x, (size_t)N * I * H * H * 4); hipMalloc(&out, (size_t)N * O * (H + 2) * (H + 2) * 4); hipMalloc(&sIn, N * I * 4); hipMalloc(&dcoef, N * O * 4);
    hipMalloc(&wp, wFloats * 4); hipMalloc(&dbg, 1 << 22);
    hipMemset(x, 0, (size_t)N * I * H * H * 4); hipMemset(sIn, 0, N * I * 4); hipMemset(dcoef, 0, N * O * 4); hipMemset(dbg, 0xff, 1 << 22);
    std::vector<unsigned> hw(wFloats); for (size_t i = 0; i < hw.size(); i++) hw[i] = (unsigned)i;
    hipMemcpy(wp, hw.data(), wFloats * 4, hipMemcpyHostToDevice);
    sg3::g_f23_stamps = dbg;
    sg3_modconv_params q = {};
    q.x = x; q.wPacked = (const float*)wp; q.sIn = sIn; q.dcoef = dcoef; q.out = out; q.dtype = SG3_F32;
    q.N = N; q.I = I; q.O = O; q.H = q.W = H; q.k = 3; q.pad = 2; q.precision = SG3_CONV_F16X3_F23;
    setenv("SG3_F23_TN", "4", 1);
    if (sg3::launch_conv_f23(q, nullptr) != 0) return 2;
    hipDeviceSynchronize();
    std::vector<unsigned> h(nch * 8 * 6 * 64 * 4);
    hipMemcpy(h.data(), (unsigned*)dbg + 4096, h.size() * 4, hipMemcpyDeviceToHost);
    int bad = 0; int per[4][8][6] = {};
    for (int ch = 0; ch < nch; ch++) for (int w = 0; w < 8; w++) for (int f = 0; f < 6; f++) for (int l = 0; l < 64; l++) for (int e = 0; e < 4; e++) {
        const unsigned got = h[(((ch * 8 + w) * 6 + f) * 64 + l) * 4 + e];
        const int xi = w & 3, mb = w >> 2;
        const unsigned want = (unsigned)(((((size_t)ch * 2 + mb) * 4 + xi) * 6 + f) * 256 + l * 4 + e);     // dword index in the packed buffer (M tile 0)
        if (got != want) { per[ch][w][f]++; if (bad < 24) printf("ch %d wave %d frag %d lane %d e %d: got %u (0x%x) want %u\n", ch, w, f, l, e, got, got, want); bad++; }
    }
    printf("fragment dump: %d mismatches\n", bad);
    for (int ch = 0; ch < nch; ch++) for (int w = 0; w < 8; w++) for (int f = 0; f < 6; f++) if (per[ch][w][f]) printf("  ch %d wave %d frag %d: %d dwords\n", ch, w, f, per[ch][w][f]);
    return 0;
}
