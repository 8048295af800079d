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
#include "pathscore.h"

namespace semicrf {

// One wave per chain (four chains per 256-thread workgroup): pathscore.h -- the same function the forward sweep's path role runs.
constexpr int EP_THREADS = 256;

__global__ __launch_bounds__(EP_THREADS) void eval_path_kernel(const float* __restrict__ score,
                                                               const float* __restrict__ noise, int T, int B, int K,
                                                               const int* __restrict__ pairs,
                                                               const int* __restrict__ offsets, float* __restrict__ out,
                                                               const float* __restrict__ sub)
{
    const int c = blockIdx.x * (EP_THREADS / 64) + (threadIdx.x >> 6);
    if (c >= B) return;
    const double r = path_score_wave(score, noise, T, B, K, pairs, offsets, c);
    if ((threadIdx.x & 63) == 0) out[c] = sub ? (float)r - sub[c] : (float)r;       // sub = logZ: logProb (reference :587-588), one fp32 subtraction
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
    hipLaunchKernelGGL(eval_path_kernel, dim3((B + EP_THREADS / 64 - 1) / (EP_THREADS / 64)), dim3(EP_THREADS), 0, stream, score, noise, T, B, K, pairs, offsets, out, sub);
}

void launch_eval_path_bwd(const float* gout, int T, int B, int K, const int* pairs, const int* offsets,
                          float* dScore, float* dNoise, hipStream_t stream, int gstride, float gscale, int noise_term_done)
{
    // noise_term_done: the caller's gradient sweep has already added d cum[T-1] / d noise = gout to every gap (persist.hip: noiseAdd)
    if (dNoise && T > 1 && !noise_term_done) {
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
