// marginals.hip -- dense marginals / gradient of logZ given alpha (v), beta (q) and logZ.
// NeuralSemiCRFInterval.py:424-447 and ComputeLogZFasterGrad.backward :469-472 in one pass:
// reads the lower triangle of score once, writes the full [T][T][B] gradient once
// (zeros above the diagonal are part of the reference's contract).
#include "common.h"

namespace semicrf {

// One workgroup per (row e, slab of 256*VEC consecutive (b,c) elements).
__global__ __launch_bounds__(256) void marginals_kernel(const float* __restrict__ score,
                                                         const float* __restrict__ v,
                                                         const float* __restrict__ q,
                                                         const float* __restrict__ logZ,
                                                         const float* __restrict__ gout, int T, int B,
                                                         float* __restrict__ dScore, int gstride, float gscale)
{
    const int e = blockIdx.y;
    const size_t rowlen = (size_t)T * B;
    const size_t base = (size_t)e * rowlen;
    const size_t lim = (size_t)(e + 1) * B;  // elements with b <= e
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < rowlen;
         idx += (size_t)gridDim.x * blockDim.x) {
        float g = 0.0f;
        if (idx < lim) {
            const int b = (int)(idx / B);
            const int c = (int)(idx - (size_t)b * B);
            const float s = score[base + idx];
            float a = v[(size_t)b * B + c] + ((q[(size_t)e * B + c] - logZ[c]) + s);
            if (b == e) a -= 2.0f * softplus_f(s);
            g = gscale * gout[(size_t)c * gstride] * expf(a);
        }
        dScore[base + idx] = g;
    }
}

__global__ __launch_bounds__(256) void noise_grad_kernel(const float* __restrict__ noise,
                                                          const float* __restrict__ v,
                                                          const float* __restrict__ q,
                                                          const float* __restrict__ logZ,
                                                          const float* __restrict__ gout, int T, int B,
                                                          float* __restrict__ dNoise, int gstride, float gscale)
{
    const size_t n = (size_t)(T - 1) * B;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(idx / B);
        const int c = (int)(idx - (size_t)t * B);
        const float a = ((v[idx] + q[(size_t)(t + 1) * B + c]) + noise[idx]) - logZ[c];
        dNoise[idx] = gscale * gout[(size_t)c * gstride] * expf(a);
    }
}

void launch_marginals(const float* score, const float* noise, const float* v, const float* q,
                      const float* logZ, const float* gout, int T, int B, float* dScore, float* dNoise,
                      hipStream_t stream, int gstride, float gscale)
{
    const size_t rowlen = (size_t)T * B;
    int gx = (int)((rowlen + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(marginals_kernel, dim3(gx, T), dim3(256), 0, stream, score, v, q, logZ, gout, T, B, dScore, gstride, gscale);
    if (T > 1 && dNoise) {
        const size_t n = (size_t)(T - 1) * B;
        int g2 = (int)((n + 255) / 256);
        if (g2 > 2048) g2 = 2048;
        hipLaunchKernelGGL(noise_grad_kernel, dim3(g2), dim3(256), 0, stream, noise, v, q, logZ, gout, T, B, dNoise, gstride, gscale);
    }
}

}  // namespace semicrf
