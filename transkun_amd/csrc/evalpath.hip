// evalpath.hip -- unnormalised path score and its gradient (NeuralSemiCRFInterval.py:508-550).
// The reference builds four Python index lists per call and issues three gathers and a
// scatter_add; here the host packs the interval lists once into CSR form (pairs, offsets).
//
//   out[c] = sum_{(b,e) in path_c} ( s[e,b,c] - (cum[e]-cum[b]) ) + cum[T-1],  cum = prefix sums of noise
//          = sum_path s[e,b,c]  -  sum_path sum_{t=b}^{e-1} noise[t,c]  +  sum_t noise[t,c]
// The second form needs no prefix array: intervals are short and never overlap, so the inner sums touch
// at most T-1 values per chain.  Sums are accumulated in double (torch's CPU cumsum does the same), in an order
// that is fixed by the thread layout: evalPath is bit-reproducible run to run.
#include "common.h"

namespace semicrf {

// One 256-thread workgroup per chain: the threads stride over the chain's noise column and over its intervals (one
// s[e,b,c] gather and the short covered-noise sum each), then a fixed-order tree reduction in double -- the result does
// not depend on scheduling (no atomics), needs no zeroed output, and the whole call is one launch.
constexpr int EP_THREADS = 256;

__global__ __launch_bounds__(EP_THREADS) void eval_path_kernel(const float* __restrict__ score,
                                                               const float* __restrict__ noise, int T, int B, int K,
                                                               const int* __restrict__ pairs,
                                                               const int* __restrict__ offsets, float* __restrict__ out,
                                                               const float* __restrict__ sub)
{
    __shared__ double red[EP_THREADS];
    const int c = blockIdx.x, tid = threadIdx.x;
    double acc = 0.0;
    for (int t = tid; t < T - 1; t += EP_THREADS) acc += (double)noise[(size_t)t * B + c];          // cum[T-1]
    const int k0 = offsets[c], k1 = offsets[c + 1];
    for (int k = k0 + tid; k < k1 && k < K; k += EP_THREADS) {
        const int b = pairs[2 * k], e = pairs[2 * k + 1];
        double covered = 0.0;
        for (int t = b; t < e; ++t) covered += (double)noise[(size_t)t * B + c];
        acc += (double)score[((size_t)e * T + b) * B + c] - covered;
    }
    red[tid] = acc;
    __syncthreads();
    for (int d = EP_THREADS / 2; d > 0; d >>= 1) {
        if (tid < d) red[tid] += red[tid + d];
        __syncthreads();
    }
    if (tid == 0) out[c] = sub ? (float)red[0] - sub[c] : (float)red[0];       // sub = logZ: logProb (reference :587-588), one fp32 subtraction
}

// dNoise[t][c] += gout[c]  (d cum[T-1] / d noise): fully parallel
__global__ __launch_bounds__(256) void eval_path_bwd_noise_kernel(const float* __restrict__ gout, int T, int B,
                                                                  float* __restrict__ dNoise, int gstride, float gscale)
{
    const size_t n = (size_t)(T - 1) * B;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dNoise[i] += gscale * gout[(i % B) * gstride];
}

// One thread per interval: dScore[e][b][c] += g; dNoise[t][c] -= g for the covered gaps t in [b, e).
// Atomics keep the (rare, caller-error) duplicate/overlapping intervals exact and let the stores pipeline.
__global__ __launch_bounds__(256) void eval_path_bwd_pairs_kernel(const float* __restrict__ gout, int T, int B, int K,
                                                                  const int* __restrict__ pairs,
                                                                  const int* __restrict__ offsets,
                                                                  float* dScore, float* dNoise, int gstride, float gscale)
{
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= K) return;
    // chain of interval k: largest c with offsets[c] <= k
    int lo = 0, hi = B;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= k) lo = mid; else hi = mid;
    }
    const int c = lo;
    const float g = gscale * gout[(size_t)c * gstride];
    const int b = pairs[2 * k], e = pairs[2 * k + 1];
    if (dScore) atomicAdd(dScore + ((size_t)e * T + b) * B + c, g);
    if (dNoise)
        for (int t = b; t < e; ++t) atomicAdd(dNoise + (size_t)t * B + c, -g);
}

void launch_eval_path(const float* score, const float* noise, int T, int B, int K, const int* pairs,
                      const int* offsets, float* out, hipStream_t stream, const float* sub)
{
    hipLaunchKernelGGL(eval_path_kernel, dim3(B), dim3(EP_THREADS), 0, stream, score, noise, T, B, K, pairs, offsets, out, sub);
}

void launch_eval_path_bwd(const float* gout, int T, int B, int K, const int* pairs, const int* offsets,
                          float* dScore, float* dNoise, hipStream_t stream, int gstride, float gscale)
{
    if (dNoise && T > 1) {
        const size_t n = (size_t)(T - 1) * B;
        int g = (int)((n + 255) / 256);
        if (g > 2048) g = 2048;
        hipLaunchKernelGGL(eval_path_bwd_noise_kernel, dim3(g), dim3(256), 0, stream, gout, T, B, dNoise, gstride, gscale);
    }
    if (K > 0)
        hipLaunchKernelGGL(eval_path_bwd_pairs_kernel, dim3((K + 255) / 256), dim3(256), 0, stream, gout, T, B, K,
                           pairs, offsets, dScore, dNoise, gstride, gscale);
}

}  // namespace semicrf
