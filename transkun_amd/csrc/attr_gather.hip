// attr_gather.hip -- interval features for the attribute heads, straight from the packed decode output
// (replaces TransKun.fetchIntervalFeaturesBatch, ModelTransformer.py:501-532, and the concatenation
// that feeds the velocity / onset-offset predictors, :578-582).
//
// The reference walks the decoded Python lists per segment, builds index tensors on the host, copies them to the
// device and runs two index_selects plus a cat per segment.  Here the intervals never leave the device: for interval i
// of chain c = n * nSym + sym (packed (begin, end) pairs + offsets[C+1], the layout semicrf_viterbi leaves in HBM)
//   out[i] = [ ctx[c, begin, :] | ctx[c, end, :] | ctx[c, begin, :] * ctx[c, end, :] ]      (3 D floats)
//   symIdx[i] = c % nSym,   scatterIdx[i] = c      (the reference's symIdx_all / scatterIdx_all, int64)
// One wave per interval, 16-byte accesses; HBM-bound (reads 2 D floats, writes 3 D per interval).
#include "common.h"

namespace semicrf {

__device__ __forceinline__ int chain_of_interval(const int* __restrict__ offsets, int C, int i)
{
    int lo = 0, hi = C;                       // largest c with offsets[c] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void interval_features_kernel(const float* __restrict__ ctx, int C, int T, int D,
                                                                long long ldc, const int* __restrict__ pairs, int K,
                                                                const int* __restrict__ offsets, int nSym,
                                                                float* __restrict__ out, long long* __restrict__ symIdx,
                                                                long long* __restrict__ scatterIdx)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= K) return;
    const int c = chain_of_interval(offsets, C, i);
    const int b = pairs[2 * i], e = pairs[2 * i + 1];
    const float* pa = ctx + ((size_t)c * T + b) * ldc;
    const float* pb = ctx + ((size_t)c * T + e) * ldc;
    float* o = out + (size_t)i * 3 * D;
    const bool vec = (D & 3) == 0 && (ldc & 3) == 0 && (((uintptr_t)ctx | (uintptr_t)out) & 15) == 0;
    if (vec) {
        for (int d = lane * 4; d < D; d += 256) {
            const float4 a = *(const float4*)(pa + d), bb = *(const float4*)(pb + d);
            *(float4*)(o + d) = a;
            *(float4*)(o + D + d) = bb;
            *(float4*)(o + 2 * D + d) = make_float4(a.x * bb.x, a.y * bb.y, a.z * bb.z, a.w * bb.w);
        }
    } else {
        for (int d = lane; d < D; d += 64) {
            const float a = pa[d], bb = pb[d];
            o[d] = a; o[D + d] = bb; o[2 * D + d] = a * bb;
        }
    }
    if (lane == 0) {
        if (symIdx) symIdx[i] = c % nSym;
        if (scatterIdx) scatterIdx[i] = c;
    }
}

// backward: dctx[c, begin, :] += g_a + g_ab * ctx[c, end, :];  dctx[c, end, :] += g_b + g_ab * ctx[c, begin, :]
// (atomics: a frame may begin one interval and end another, and begin == end for singletons)
__global__ __launch_bounds__(256) void interval_features_bwd_kernel(const float* __restrict__ gout, const float* __restrict__ ctx,
                                                                    int C, int T, int D, long long ldc,
                                                                    const int* __restrict__ pairs, int K,
                                                                    const int* __restrict__ offsets, float* dctx,
                                                                    long long lddc)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= K) return;
    const int c = chain_of_interval(offsets, C, i);
    const int b = pairs[2 * i], e = pairs[2 * i + 1];
    const float* pa = ctx + ((size_t)c * T + b) * ldc;
    const float* pb = ctx + ((size_t)c * T + e) * ldc;
    const float* g = gout + (size_t)i * 3 * D;
    float* da = dctx + ((size_t)c * T + b) * lddc;
    float* db = dctx + ((size_t)c * T + e) * lddc;
    for (int d = lane; d < D; d += 64) {
        const float gab = g[2 * D + d];
        atomicAdd(da + d, g[d] + gab * pb[d]);
        atomicAdd(db + d, g[D + d] + gab * pa[d]);
    }
}

void launch_interval_features(const float* ctx, int C, int T, int D, long long ldc, const int* pairs, int K,
                              const int* offsets, int nSym, float* out, long long* symIdx, long long* scatterIdx,
                              hipStream_t stream)
{
    if (K <= 0) return;
    hipLaunchKernelGGL(interval_features_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, ctx, C, T, D, ldc, pairs, K, offsets,
                       nSym, out, symIdx, scatterIdx);
}

void launch_interval_features_bwd(const float* gout, const float* ctx, int C, int T, int D, long long ldc, const int* pairs,
                                  int K, const int* offsets, float* dctx, long long lddc, hipStream_t stream)
{
    if (K <= 0) return;
    hipLaunchKernelGGL(interval_features_bwd_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, gout, ctx, C, T, D, ldc, pairs, K,
                       offsets, dctx, lddc);
}

}  // namespace semicrf
