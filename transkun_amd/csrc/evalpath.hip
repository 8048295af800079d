// evalpath.hip -- unnormalised path score and its gradient (NeuralSemiCRFInterval.py:508-550).
// The reference builds four Python index lists per call and issues three gathers and a
// scatter_add; here the host packs the interval lists once into CSR form (pairs, offsets) and
// a single kernel does prefix sums of the noise and the per-chain gather/accumulate.
#include "common.h"

namespace semicrf {

// One thread per chain: lanes are consecutive chains, so noise[t][c] loads are coalesced.
// cum [B][T] scratch: cum[c][t] = fp32( sum_{u<t} noise[u][c] accumulated in double ), which is
// what torch's CPU cumsum produces (double accumulator, fp32 store).
__global__ __launch_bounds__(64) void eval_path_kernel(const float* __restrict__ score,
                                                        const float* __restrict__ noise, int T, int B,
                                                        const int* __restrict__ pairs,
                                                        const int* __restrict__ offsets,
                                                        float* __restrict__ cum, float* __restrict__ out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    float* cc = cum + (size_t)c * T;
    double acc = 0.0;
    cc[0] = 0.0f;
    for (int t = 1; t < T; ++t) {
        acc += (double)noise[(size_t)(t - 1) * B + c];
        cc[t] = (float)acc;
    }
    float r = 0.0f;
    const int k0 = offsets[c], k1 = offsets[c + 1];
    for (int k = k0; k < k1; ++k) {
        const int i = pairs[2 * k], j = pairs[2 * k + 1];
        r += score[((size_t)j * T + i) * B + c] - (cc[j] - cc[i]);
    }
    out[c] = r + cc[T - 1];
}

__global__ __launch_bounds__(64) void eval_path_bwd_kernel(const float* __restrict__ gout, int T, int B,
                                                            const int* __restrict__ pairs,
                                                            const int* __restrict__ offsets,
                                                            float* dScore, float* dNoise)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const float g = gout[c];
    const int k0 = offsets[c], k1 = offsets[c + 1];
    if (dNoise) {
        for (int t = 0; t < T - 1; ++t) dNoise[(size_t)t * B + c] += g;      // d cum[T-1]
    }
    for (int k = k0; k < k1; ++k) {
        const int i = pairs[2 * k], j = pairs[2 * k + 1];
        if (dScore) dScore[((size_t)j * T + i) * B + c] += g;
        if (dNoise)
            for (int t = i; t < j; ++t) dNoise[(size_t)t * B + c] -= g;      // -(cum[j]-cum[i])
    }
}

void launch_eval_path(const float* score, const float* noise, int T, int B, const int* pairs,
                      const int* offsets, float* cum, float* out, hipStream_t stream)
{
    hipLaunchKernelGGL(eval_path_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, score, noise, T, B, pairs,
                       offsets, cum, out);
}

void launch_eval_path_bwd(const float* gout, int T, int B, const int* pairs, const int* offsets,
                          float* dScore, float* dNoise, hipStream_t stream)
{
    hipLaunchKernelGGL(eval_path_bwd_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, gout, T, B, pairs, offsets,
                       dScore, dNoise);
}

}  // namespace semicrf
