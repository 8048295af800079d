// decode.hip -- on-device backtrack of the Viterbi pointer table and packing of the result.
// The reference copies the [T-1][B] pointer table and the diagonal mask to the host and walks
// them in TorchScript (NeuralSemiCRFInterval.py:56-102 / :150-199).  Here one thread per chain
// walks its own contiguous code row code[c][0..T) (written transposed by the DP kernels so the
// walk is cache-line local), then the per-chain lists are packed so that only
// O(#intervals) int32 cross PCIe.
//
// code[c][t] = (key+1) | (s[t,t]>0 ? 1<<30 : 0), key = -1 (skip) or the absolute index of the
// other endpoint chosen at t.
#include "common.h"

namespace semicrf {

constexpr int CODE_DIAG = 0x40000000;
constexpr int CODE_MASK = 0x3fffffff;

// region: [B][2T][2] int32, counts: [B]
__global__ __launch_bounds__(64) void backtrack_kernel(const int* __restrict__ code, int T, int B,
                                                        const int* __restrict__ start, int forward,
                                                        int* __restrict__ region, int* __restrict__ counts)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const int* cc = code + (size_t)c * T;
    int* out = region + (size_t)c * (size_t)(2 * T) * 2;
    int n = 0;
    if (!forward) {
        // viterbiBackward walk, :70-98
        int j = start ? start[c] : 0;
        while (j < T - 1) {
            const int w = cc[j];
            if (w & CODE_DIAG) { out[2 * n] = j; out[2 * n + 1] = j; ++n; }
            const int key = (w & CODE_MASK) - 1;
            if (key < 0) j += 1;
            else { out[2 * n] = j; out[2 * n + 1] = key; ++n; j = key; }
        }
        if (cc[T - 1] & CODE_DIAG) { out[2 * n] = T - 1; out[2 * n + 1] = T - 1; ++n; }
    } else {
        // viterbi walk, :164-196; emitted descending here, reversed by the pack kernel
        int j = start ? start[c] : T - 1;
        while (j > 0) {
            const int w = cc[j];
            if (w & CODE_DIAG) { out[2 * n] = j; out[2 * n + 1] = j; ++n; }
            const int key = (w & CODE_MASK) - 1;
            if (key < 0) j -= 1;
            else { out[2 * n] = key; out[2 * n + 1] = j; ++n; j = key; }
        }
        if (cc[0] & CODE_DIAG) { out[2 * n] = 0; out[2 * n + 1] = 0; ++n; }
    }
    counts[c] = n;
}

// exclusive prefix sum of counts[B] -> offsets[B+1]; single workgroup, any B
__global__ __launch_bounds__(256) void offsets_kernel(const int* __restrict__ counts, int B,
                                                       int* __restrict__ offsets)
{
    __shared__ int part[256];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 256) {
        const int i = base + threadIdx.x;
        const int x = i < B ? counts[i] : 0;
        part[threadIdx.x] = x;
        __syncthreads();
        for (int d = 1; d < 256; d <<= 1) {  // Hillis-Steele inclusive scan
            int y = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += y;
            __syncthreads();
        }
        if (i < B) offsets[i] = carry + part[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[B] = carry;
}

__global__ __launch_bounds__(256) void pack_kernel(const int* __restrict__ region,
                                                    const int* __restrict__ counts,
                                                    const int* __restrict__ offsets, int T, int B,
                                                    int forward, int* __restrict__ pairs, long long cap)
{
    const int c = blockIdx.x;
    const int n = counts[c];
    const long long off = offsets[c];
    const int* src = region + (size_t)c * (size_t)(2 * T) * 2;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const int s = forward ? (n - 1 - k) : k;
        const long long d = off + k;
        if (d < cap) {
            pairs[2 * d] = src[2 * s];
            pairs[2 * d + 1] = src[2 * s + 1];
        }
    }
}

void launch_backtrack(const int* code, int T, int B, const int* start, int forward, int* region,
                      int* counts, int* pairs, long long cap, int* offsets, hipStream_t stream)
{
    hipLaunchKernelGGL(backtrack_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, code, T, B, start, forward,
                       region, counts);
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(256), 0, stream, counts, B, offsets);
    hipLaunchKernelGGL(pack_kernel, dim3(B), dim3(256), 0, stream, region, counts, offsets, T, B, forward, pairs,
                       cap);
}

}  // namespace semicrf
