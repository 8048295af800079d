// segment.hip -- the device side of the transcription segment loop.
//
// After decode, the reference (TransKun.transcribeFrames, ModelTransformer.py:549-725) walks the Python interval lists on
// the host: optional onset-bound filter (:554-555), one Note per interval with refined onset/offset times (:684-697), the
// last confirmed offset per symbol (`lastP`, :712-718), and TransKun.transcribe turns that into the next segment's
// forcedStartPos (:789-791).  Here these steps consume the packed decode output where semicrf_viterbi left it in HBM:
//   segment_onset_filter -- keeps the intervals with begin < bound (lists are ascending in begin: a per-chain cut) and
//                           re-packs them;
//   segment_events       -- per chain, in list order: the event times in double precision with the reference's
//                           operation order (no contraction: bit-identical to the Python floats), the hasOnset /
//                           hasOffset flags, lastP, and the next forced start max(lastP - stepFrames, 0) -- an int32
//                           vector that the next semicrf_viterbi takes as `start` without a host round trip.
// One thread per chain: the per-symbol `lastEnd` clamp is a sequential recurrence over at most 2T short intervals.
#include "common.h"

namespace semicrf {

__global__ __launch_bounds__(256) void onset_count_kernel(const int* __restrict__ pairs, const int* __restrict__ offsets, int B,
                                                          int bound, int* __restrict__ counts)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    int lo = offsets[c];
    const int hi0 = offsets[c + 1];
    int hi = hi0;
    // the reference filters `e[0] < onsetBound` over an ascending list: everything before the first begin >= bound.
    // Singletons and touching intervals share begins, so the predicate is monotone in list order.
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (pairs[2 * mid] < bound) lo = mid + 1; else hi = mid;
    }
    counts[c] = lo - offsets[c];
}

__global__ __launch_bounds__(256) void onset_pack_kernel(const int* __restrict__ pairs, const int* __restrict__ offsets,
                                                         const int* __restrict__ counts, const int* __restrict__ new_offsets, int B,
                                                         int* __restrict__ out, long long cap)
{
    const int c = blockIdx.x;
    const int n = counts[c];
    const long long src = offsets[c], dst = new_offsets[c];
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        if (dst + i < cap) {
            out[2 * (dst + i)] = pairs[2 * (src + i)];
            out[2 * (dst + i) + 1] = pairs[2 * (src + i) + 1];
        }
}

// src_total: the total of the offsets the counts were derived from (or null): a negative one is semicrf_viterbi's "a wait timed
// out" marker and is handed on instead of the sum
__global__ __launch_bounds__(256) void offsets_scan_kernel(const int* __restrict__ counts, int B, int* __restrict__ offsets,
                                                           const int* __restrict__ src_total)
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
        for (int d = 1; d < 256; d <<= 1) {
            const int y = threadIdx.x >= d ? part[threadIdx.x - d] : 0;
            __syncthreads();
            part[threadIdx.x] += y;
            __syncthreads();
        }
        if (i < B) offsets[i] = carry + part[threadIdx.x] - x;
        __syncthreads();
        if (threadIdx.x == 255) carry += part[255];
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[B] = (src_total && *src_total < 0) ? *src_total : carry;
}

void launch_onset_filter(const int* pairs, const int* offsets, int B, int bound, int* pairs_out, long long cap, int* offsets_out,
                         int* counts, hipStream_t stream)
{
    hipLaunchKernelGGL(onset_count_kernel, dim3((B + 255) / 256), dim3(256), 0, stream, pairs, offsets, B, bound, counts);
    hipLaunchKernelGGL(offsets_scan_kernel, dim3(1), dim3(256), 0, stream, counts, B, offsets_out, offsets + B);
    hipLaunchKernelGGL(onset_pack_kernel, dim3(B), dim3(64), 0, stream, pairs, offsets, counts, offsets_out, B, pairs_out, cap);
}

// ModelTransformer.py:684-718 per chain c = segment * nSym + symbol, then :794-800 (shift by the segment's begin time)
__global__ __launch_bounds__(128) void segment_events_kernel(const int* __restrict__ pairs, const int* __restrict__ offsets, int B, int nSym,
                                                             const float* __restrict__ ofValue, const unsigned char* __restrict__ ofPresence,
                                                             int lastFrameIdx, double frameDur, const double* __restrict__ beginTime,
                                                             int stepFrames, double* __restrict__ times, unsigned char* __restrict__ flags,
                                                             int* __restrict__ lastP, int* __restrict__ nextStart)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= B) return;
    const double bt = beginTime[c / nSym];
    double lastEnd = 0.0;                                   // :678
    int curLastP = 0;                                       // :679
    for (int i = offsets[c]; i < offsets[c + 1]; ++i) {
        const int b = pairs[2 * i], e = pairs[2 * i + 1];
        double start = ((double)b + (double)ofValue[2 * i]) * frameDur;            // :684
        double end = ((double)e + (double)ofValue[2 * i + 1]) * frameDur;          // :685
        const bool hasOnset = b > 0 || ofPresence[2 * i] != 0;                      // :689
        const bool hasOffset = e < lastFrameIdx || ofPresence[2 * i + 1] != 0;      // :690
        start = start > lastEnd ? start : lastEnd;                                  // :694
        const double e2 = start + 1e-8;
        end = end > e2 ? end : e2;                                                  // :695
        lastEnd = end;                                                              // :696
        if (hasOffset) curLastP = e;                                                // :708-709
        // TransKun.transcribe :794-800: shift by the segment's begin time, clamp at 0
        double s2 = start + bt, en2 = end + bt;
        s2 = s2 > 0.0 ? s2 : 0.0;
        en2 = en2 > s2 ? en2 : s2;
        times[2 * i] = s2; times[2 * i + 1] = en2;
        flags[2 * i] = hasOnset ? 1 : 0; flags[2 * i + 1] = hasOffset ? 1 : 0;
    }
    lastP[c] = curLastP;                                    // :718
    const int ns = curLastP - stepFrames;                   // transcribe :789-791
    nextStart[c] = ns > 0 ? ns : 0;
}

void launch_segment_events(const int* pairs, const int* offsets, int B, int nSym, const float* ofValue, const unsigned char* ofPresence,
                           int lastFrameIdx, double frameDur, const double* beginTime, int stepFrames, double* times, unsigned char* flags,
                           int* lastP, int* nextStart, hipStream_t stream)
{
    hipLaunchKernelGGL(segment_events_kernel, dim3((B + 127) / 128), dim3(128), 0, stream, pairs, offsets, B, nSym, ofValue, ofPresence,
                       lastFrameIdx, frameDur, beginTime, stepFrames, times, flags, lastP, nextStart);
}

}  // namespace semicrf
