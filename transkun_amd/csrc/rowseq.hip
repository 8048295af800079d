// rowseq.hip -- row-sequential reference kernels (impl 1).
//
// One workgroup owns 16 consecutive chains and walks the T positions in order; its 256
// threads are 16 column slots x 16 chains, so a wave reads 4 cells of 64 contiguous bytes.
// Simple and valid for every T and B; used for odd shapes and as the on-GPU cross-check of
// the blocked kernels.  HBM-bound work but only B/16 workgroups are alive, so it reaches a few
// percent of the roofline -- the blocked kernels in blocked.hip are the product path for the
// sizes that matter.
//
// Recurrences (NeuralSemiCRFInterval.py):  LSE/FWD :402-410 (alpha), LSE/BWD the flipped half
// :386-414 (beta), MAX/BWD viterbiBackward :27-51, MAX/FWD viterbi :122-144.
#include "common.h"

namespace semicrf {

constexpr int RS_G = 16;  // chains per workgroup
constexpr int RS_Q = 16;  // column slots per chain

// MODE 0 = log-sum-exp semiring, 1 = max semiring with argmax codes.
// DIR 0 = ascending positions (alpha / viterbi), 1 = descending (beta / viterbiBackward).
// u [T][B]: the DP values.  code [B][T] (MODE 1): key+1 of the winning candidate (0 = skip, else
// absolute index of the other endpoint + 1), plus bit 30 when s[t,t] > 0.
template <int MODE, int DIR>
__global__ __launch_bounds__(256) void rowseq_sweep_kernel(const float* __restrict__ score,
                                                            const float* __restrict__ noise, int T, int B,
                                                            float* u, int* __restrict__ code,
                                                            float* __restrict__ out_last)
{
    const int tid = threadIdx.x;
    const int cq = tid & (RS_G - 1);
    const int q = tid >> 4;
    const int c = blockIdx.x * RS_G + cq;
    const bool valid = c < B;
    const size_t Bs = (size_t)B;

    __shared__ float redA[RS_Q][RS_G];
    __shared__ float redB[RS_Q][RS_G];

    for (int p = 0; p < T; ++p) {
        const int r = DIR == 0 ? p : T - 1 - p;
        float accM = SEMICRF_NEG_INF, accS = 0.0f;
        int key = 0x7fffffff;
        if (valid) {
            for (int pp = q; pp < p; pp += RS_Q) {
                const int rr = DIR == 0 ? pp : T - 1 - pp;
                // FWD: cell [end=r][begin=rr];  BWD: cell [end=rr][begin=r]
                const size_t cell = DIR == 0 ? ((size_t)r * T + rr) : ((size_t)rr * T + r);
                const float t = u[(size_t)rr * Bs + c] + score[cell * Bs + c];
                if (MODE == 0) lse_push(accM, accS, t);
                else max_push(accM, key, t, rr);
            }
        }
        redA[q][cq] = accM;
        redB[q][cq] = MODE == 0 ? accS : __int_as_float(key);
        __syncthreads();
        if (q == 0 && valid) {
            const float d = score[((size_t)r * T + r) * Bs + c];
            float res;
            if (MODE == 0) {
                float M = SEMICRF_NEG_INF, S = 0.0f;
#pragma unroll
                for (int k = 0; k < RS_Q; ++k) lse_merge(M, S, redA[k][cq], redB[k][cq]);
                if (p > 0) {
                    const int rp = DIR == 0 ? r - 1 : r + 1;        // previous position
                    const int gap = DIR == 0 ? r - 1 : r;           // noise index between r and rp
                    lse_push(M, S, u[(size_t)rp * Bs + c] + noise[(size_t)gap * Bs + c]);
                    res = M + logf(S) + softplus_f(d);
                } else {
                    res = softplus_f(d);
                }
            } else {
                float best = SEMICRF_NEG_INF;
                int bk = 0x7fffffff;
                if (p > 0) {
                    const int rp = DIR == 0 ? r - 1 : r + 1;
                    const int gap = DIR == 0 ? r - 1 : r;
                    best = u[(size_t)rp * Bs + c] + noise[(size_t)gap * Bs + c];  // skip: first candidate
                    bk = -1;
#pragma unroll
                    for (int k = 0; k < RS_Q; ++k) max_push(best, bk, redA[k][cq], __float_as_int(redB[k][cq]));
                    res = d > 0.0f ? best + d : best;
                } else {
                    res = d > 0.0f ? d : 0.0f;
                    bk = -1;
                }
                code[(size_t)c * T + r] = (bk + 1) | (d > 0.0f ? 0x40000000 : 0);
            }
            u[(size_t)r * Bs + c] = res;
            if (p == T - 1 && out_last) out_last[c] = res;
        }
        __syncthreads();
    }
}

void launch_rowseq_sweep(int mode, int dir, const float* score, const float* noise, int T, int B,
                         float* u, int* code, float* out_last, hipStream_t stream)
{
    dim3 grid((B + RS_G - 1) / RS_G), block(256);
    if (mode == 0 && dir == 0)
        hipLaunchKernelGGL((rowseq_sweep_kernel<0, 0>), grid, block, 0, stream, score, noise, T, B, u, code, out_last);
    else if (mode == 0 && dir == 1)
        hipLaunchKernelGGL((rowseq_sweep_kernel<0, 1>), grid, block, 0, stream, score, noise, T, B, u, code, out_last);
    else if (mode == 1 && dir == 0)
        hipLaunchKernelGGL((rowseq_sweep_kernel<1, 0>), grid, block, 0, stream, score, noise, T, B, u, code, out_last);
    else
        hipLaunchKernelGGL((rowseq_sweep_kernel<1, 1>), grid, block, 0, stream, score, noise, T, B, u, code, out_last);
}

}  // namespace semicrf
