// sg3_fourier.hip -- the Fourier features of SynthesisInput as one kernel, written channels-first.
//
// Reference (models/stylegan3/networks_stylegan3.py:236-241):
//     x = (grid.unsqueeze(3) @ freqs.permute(0, 2, 1).unsqueeze(1).unsqueeze(2)).squeeze(3)      [N,H,W,C], K = 2
//     x = x + phases.unsqueeze(1).unsqueeze(2)
//     x = torch.sin(x * (np.pi * 2))
//     x = x * amplitudes.unsqueeze(1).unsqueeze(2)
// followed by the channel mix, which wants [N,C,H,W].  The K = 2 product is the worst case for a BLAS library (250 us for
// 5.3 M outputs); here every output is  sin((fma(gy, f1, gx * f0) + phase) * 2pi) * amp  in exactly that order: the fp32
// matrix instruction the library uses accumulates k = 0 then k = 1 with one rounding per step, which is this fma chain,
// and the remaining steps are the reference's elementwise ops one for one -- so the features agree with the torch path
// bit for bit (tests/test_gpu_ops.py::test_fourier_features_match_torch_ops).
#include "sg3_common.h"
#include <cmath>

namespace sg3 {

__global__ void __launch_bounds__(256)
fourier_features_kernel(sg3_fourier_params p, int chunks) {
    const int HW = p.H * p.W;
    const int plane = blockIdx.x / chunks, chunk = blockIdx.x % chunks;       // plane = n * C + c
    const float f0 = p.freqs[2 * plane], f1 = p.freqs[2 * plane + 1];
    const float ph = p.phases[plane], amp = p.amps[plane];
    float* out = p.out + (size_t)plane * HW;
    for (int i = chunk * 256 + threadIdx.x; i < HW; i += chunks * 256) {
        const float gx = p.grid[2 * i], gy = p.grid[2 * i + 1];
        const float t = gx * f0;
        const float x = __builtin_fmaf(gy, f1, t);
        const float u = (x + ph) * 6.283185307179586f;
        out[i] = sinf(u) * amp;
    }
}

} // namespace sg3

extern "C" int sg3_fourier_features(const sg3_fourier_params* p, void* stream) {
    using namespace sg3;
    SG3_REQUIRE(p && p->grid && p->freqs && p->phases && p->amps && p->out, "fourier_features: null tensor");
    SG3_REQUIRE(p->N > 0 && p->C > 0 && p->H > 0 && p->W > 0, "fourier_features: empty tensor");
    const int chunks = std::min(8, ceil_div(p->H * p->W, 256));
    const long long blocks = (long long)p->N * p->C * chunks;
    SG3_REQUIRE(blocks <= 0x7fffffffLL, "fourier_features: grid too large");
    hipLaunchKernelGGL(fourier_features_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, *p, chunks);
    SG3_LAUNCH_CHECK("fourier_features_kernel");
    return SG3_OK;
}
