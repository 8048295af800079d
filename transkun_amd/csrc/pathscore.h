// pathscore.h -- the unnormalised path score of ONE chain by ONE wave (NeuralSemiCRFInterval.py:508-550), shared by the stand-alone
// evalPath kernel (evalpath.hip) and the path role of the forward sweep (persist.hip: semicrf_logprob_fwd as one launch), so that
// logProb == evalPath - logZ bit for bit whichever way it is computed.
//   out[c] = sum_{(b,e) in path_c} ( s[e,b,c] - sum_{t=b}^{e-1} noise[t,c] ) + sum_t noise[t,c]
// The lanes stride over the chain's noise column and over its intervals; sums in double (torch's CPU cumsum does the same), combined
// by a butterfly over the 64 lanes: a fixed order, the result is bit-reproducible run to run and the same in every lane.
#pragma once
#include "common.h"

namespace semicrf {

__device__ __forceinline__ double path_score_wave(const float* __restrict__ score, const float* __restrict__ noise, int T, int B, int K,
                                                  const int* __restrict__ pairs, const int* __restrict__ offsets, int c)
{
    const int lane = threadIdx.x & 63;
    double acc = 0.0;
    for (int t = lane; t < T - 1; t += 64) acc += (double)noise[(size_t)t * B + c];          // cum[T-1]
    const int k0 = offsets[c], k1 = offsets[c + 1];
    for (int k = k0 + lane; k < k1 && k < K; k += 64) {
        const int b = pairs[2 * k], e = pairs[2 * k + 1];
        double covered = 0.0;
        for (int t = b; t < e; ++t) covered += (double)noise[(size_t)t * B + c];
        acc += (double)score[((size_t)e * T + b) * B + c] - covered;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) acc += __shfl_xor(acc, d, 64);
    return acc;
}

}  // namespace semicrf
