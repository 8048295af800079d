// scorer.hip -- scaled-inner-product interval scores (LayersTransformer.py:406-441).
//   S[e,b,c] = (sum_d (q[c,e,d]*qscale) * k[c,b,d]) * len(|e-b|)  (+ diag[c,e] when e == b)
// Output layout [T][T][C] with the chain axis contiguous: exactly what the CRF kernels stream,
// so the reference's permute(2,3,0,1).contiguous() copy (:439) never happens.
//
// impl 1 (this kernel): one thread per output element, fp32 FMA chain over d.  Valid for any
// shape; the MFMA kernel in scorer_mfma.hip is the product path when D % 64 == 0.
#include "common.h"

namespace semicrf {

__device__ __forceinline__ float len_scale(int len, int mode)
{
    if (mode == SEMICRF_LEN_LINEAR) return (float)len;
    if (mode == SEMICRF_LEN_SQRT) return sqrtf((float)len);
    return 1.0f;
}

__global__ __launch_bounds__(256) void interval_score_naive_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ diag, int C, int T,
    int D, long long ldq, long long ldk, long long ldd, float qscale, int mode, int full,
    float* __restrict__ S)
{
    const int e = blockIdx.y;
    const size_t rowlen = (size_t)T * C;
    const size_t lim = full ? rowlen : (size_t)(e + 1) * C;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < lim;
         idx += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(idx / C);
        const int c = (int)(idx - (size_t)b * C);
        const float* qr = q + ((size_t)c * T + e) * ldq;
        const float* kr = k + ((size_t)c * T + b) * ldk;
        float acc = 0.0f;
        for (int d = 0; d < D; ++d) acc = fmaf(qr[d] * qscale, kr[d], acc);
        const int len = e > b ? e - b : b - e;
        acc *= len_scale(len, mode);
        if (e == b) acc += diag[((size_t)c * T + e) * ldd];
        S[(size_t)e * rowlen + idx] = acc;
    }
}

void launch_interval_score_naive(const float* q, const float* k, const float* diag, int C, int T, int D,
                                 long long ldq, long long ldk, long long ldd, float qscale, int mode, int full,
                                 float* S, hipStream_t stream)
{
    const size_t rowlen = (size_t)T * C;
    int gx = (int)((rowlen + 255) / 256);
    if (gx > 256) gx = 256;
    hipLaunchKernelGGL(interval_score_naive_kernel, dim3(gx, T), dim3(256), 0, stream, q, k, diag, C, T, D, ldq,
                       ldk, ldd, qscale, mode, full, S);
}

}  // namespace semicrf
