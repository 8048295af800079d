// decode.hip -- on-device backtrack of the Viterbi pointer table and packing of the result.
// The reference copies the [T-1][B] pointer table and the diagonal mask to the host and walks
// them in TorchScript (NeuralSemiCRFInterval.py:56-102 / :150-199).  Here one workgroup per chain
// stages its contiguous code row code[c][0..T) (written transposed by the DP kernels) in LDS and finds the
// visited frames there by pointer doubling (serial walk for very long rows); the per-chain lists are then
// packed so that only O(#intervals) int32 cross PCIe.
//
// code[c][t] = (key+1) | (s[t,t]>0 ? 1<<30 : 0), key = -1 (skip) or the absolute index of the
// other endpoint chosen at t.
#include <atomic>
#include "common.h"

namespace semicrf {

constexpr int CODE_DIAG = 0x40000000;
constexpr int CODE_MASK = 0x3fffffff;

// region: [B][2T][2] int32, counts: [B]
// One 64-thread workgroup per chain: the chain's code row is first copied into LDS with coalesced loads
// (T*4 bytes; rows longer than BT_LDS_MAX ints are walked in global memory instead), then lane 0 walks it:
// a dependent LDS read per step (~100 cycles) instead of a dependent global load (~1 us).
constexpr int BT_LDS_MAX = 16384;

__global__ __launch_bounds__(64) void backtrack_kernel(const int* __restrict__ code, int T, int B,
                                                        const int* __restrict__ start, int forward,
                                                        int* __restrict__ region, int* __restrict__ counts)
{
    extern __shared__ int s_code[];
    const int c = blockIdx.x;
    const int* cc = code + (size_t)c * T;
    const bool in_lds = T <= BT_LDS_MAX;
    if (in_lds) {
        for (int t = threadIdx.x; t < T; t += 64) s_code[t] = cc[t];
        __syncthreads();
    }
    if (threadIdx.x != 0) return;
    auto rd = [&](int t) { return in_lds ? s_code[t] : cc[t]; };
    int* out = region + (size_t)c * (size_t)(2 * T) * 2;
    int n = 0;
    if (!forward) {
        // viterbiBackward walk, :70-98
        int j = start ? start[c] : 0;
        while (j < T - 1) {
            const int w = rd(j);
            if (w & CODE_DIAG) { out[2 * n] = j; out[2 * n + 1] = j; ++n; }
            const int key = (w & CODE_MASK) - 1;
            if (key < 0) j += 1;
            else { out[2 * n] = j; out[2 * n + 1] = key; ++n; j = key; }
        }
        if (rd(T - 1) & CODE_DIAG) { out[2 * n] = T - 1; out[2 * n + 1] = T - 1; ++n; }
    } else {
        // viterbi walk, :164-196; emitted descending here, reversed by the pack kernel
        int j = start ? start[c] : T - 1;
        while (j > 0) {
            const int w = rd(j);
            if (w & CODE_DIAG) { out[2 * n] = j; out[2 * n + 1] = j; ++n; }
            const int key = (w & CODE_MASK) - 1;
            if (key < 0) j -= 1;
            else { out[2 * n] = key; out[2 * n + 1] = j; ++n; j = key; }
        }
        if (rd(0) & CODE_DIAG) { out[2 * n] = 0; out[2 * n + 1] = 0; ++n; }
    }
    counts[c] = n;
}

// Parallel backtrack: the walk is a linked list (next(j) = j +- 1 for "skip", else the chosen other endpoint), so the set
// of visited frames follows from pointer doubling -- mark the start; in round r every marked frame marks the frame 2^r
// links ahead, then every frame squares its jump -- in ceil(log2 T) rounds instead of up to T dependent LDS reads
// (T=2048: ~2 us instead of 290 us per chain).  The emissions (0..2 per visited frame, the reference's order) are then
// placed by a block-wide exclusive scan.  One 256-thread workgroup per chain, everything in LDS: code, two jump arrays,
// marks (16 bytes per frame: T <= 8192; longer rows take the serial kernel above).
constexpr int BT_PAR_MAX = 8192;
constexpr int BT_PAR_THREADS = 256;

__global__ __launch_bounds__(BT_PAR_THREADS) void backtrack_par_kernel(const int* __restrict__ code, int T, int B,
                                                                       const int* __restrict__ start, int forward,
                                                                       int* __restrict__ region, int* __restrict__ counts)
{
    extern __shared__ int s_bt[];
    int* const s_code = s_bt;
    int* jump = s_bt + T;
    int* jump2 = s_bt + 2 * T;
    int* const mark = s_bt + 3 * T;
    __shared__ int s_scan[BT_PAR_THREADS];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int* cc = code + (size_t)c * T;
    const int term = forward ? 0 : T - 1;                  // the walk stops there (its own emission is unconditional)
    int st = start ? start[c] : (forward ? T - 1 : 0);
    st = st < 0 ? 0 : (st > T - 1 ? T - 1 : st);
    for (int t = tid; t < T; t += BT_PAR_THREADS) {
        const int w = cc[t];
        s_code[t] = w;
        const int key = (w & CODE_MASK) - 1;
        int nx;
        if (t == term) nx = term;                          // self loop
        else if (key < 0) nx = forward ? t - 1 : t + 1;
        else nx = key;
        jump[t] = nx;
        mark[t] = t == st ? 1 : 0;
    }
    __syncthreads();
    for (int span = 1; span < T; span <<= 1) {
        for (int t = tid; t < T; t += BT_PAR_THREADS) {
            const int j = jump[t];
            if (mark[t]) mark[j] = 1;                      // racing writers all store 1; a frame marked early in the round
                                                           // only marks further frames of the same path
            jump2[t] = jump[j];
        }
        __syncthreads();
        int* tmp = jump; jump = jump2; jump2 = tmp;
    }
    // emissions per frame: visited frames other than the terminal emit (t,t) if the diagonal is on, and the chosen
    // interval; the terminal emits (term,term) if its diagonal is on -- whether visited or not (:97-98 / :195-196)
    auto count_of = [&](int t) -> int {
        const int w = s_code[t];
        if (t == term) return (w & CODE_DIAG) ? 1 : 0;
        if (!mark[t]) return 0;
        return ((w & CODE_DIAG) ? 1 : 0) + (((w & CODE_MASK) - 1) >= 0 ? 1 : 0);
    };
    // each thread owns a contiguous run of frames, in WALK order (ascending for the backward walk, descending for the
    // forward one -- whose list the pack kernel reverses, like the serial kernel's)
    const int L = (T + BT_PAR_THREADS - 1) / BT_PAR_THREADS;
    const int i0 = tid * L, i1 = (i0 + L < T) ? i0 + L : T;      // walk-order indices
    int mine = 0;
    for (int i = i0; i < i1; ++i) mine += count_of(forward ? T - 1 - i : i);
    s_scan[tid] = mine;
    __syncthreads();
    for (int d = 1; d < BT_PAR_THREADS; d <<= 1) {          // Hillis-Steele inclusive scan
        const int y = tid >= d ? s_scan[tid - d] : 0;
        __syncthreads();
        s_scan[tid] += y;
        __syncthreads();
    }
    int n = s_scan[tid] - mine;                             // exclusive prefix of this thread's run
    int* out = region + (size_t)c * (size_t)(2 * T) * 2;
    for (int i = i0; i < i1; ++i) {
        const int t = forward ? T - 1 - i : i;
        const int w = s_code[t];
        if (t == term) {
            if (w & CODE_DIAG) { out[2 * n] = t; out[2 * n + 1] = t; ++n; }
        } else if (mark[t]) {
            if (w & CODE_DIAG) { out[2 * n] = t; out[2 * n + 1] = t; ++n; }
            const int key = (w & CODE_MASK) - 1;
            if (key >= 0) {
                out[2 * n] = forward ? key : t; out[2 * n + 1] = forward ? t : key; ++n;
            }
        }
    }
    if (tid == BT_PAR_THREADS - 1) counts[c] = s_scan[tid];
}

// exclusive prefix sum of counts[B] -> offsets[B+1]; single workgroup, any B
// err: the sweep's error words (one per chain chunk, `nerr` of them `err_stride` words apart; 0xffffffff = fine) or
// nullptr.  A sweep that gave up on a bounded wait leaves garbage codes: the total comes back as -1 instead.
__global__ __launch_bounds__(256) void offsets_kernel(const int* __restrict__ counts, int B,
                                                       int* __restrict__ offsets, const unsigned* __restrict__ err,
                                                       int nerr, int err_stride)
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
    if (threadIdx.x == 0) {
        bool bad = false;
        for (int i = 0; err && i < nerr; ++i) bad = bad || err[(size_t)i * err_stride] != 0xffffffffu;
        offsets[B] = bad ? -1 : carry;
    }
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
                      int* counts, int* pairs, long long cap, int* offsets, hipStream_t stream,
                      const unsigned* err, int nerr, int err_stride)
{
    const size_t lds = T <= BT_LDS_MAX ? (size_t)T * sizeof(int) : 0;
    // function attributes are per device (a process may drive several)
    static std::atomic<bool> attr_set[64], par_attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load()) {
        (void)hipFuncSetAttribute((const void*)backtrack_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  BT_LDS_MAX * (int)sizeof(int));
        attr_set[dev].store(true);
    }
    if (T <= BT_PAR_MAX && T >= 2) {
        if (!par_attr_set[dev].load()) {
            (void)hipFuncSetAttribute((const void*)backtrack_par_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      4 * BT_PAR_MAX * (int)sizeof(int));
            par_attr_set[dev].store(true);
        }
        hipLaunchKernelGGL(backtrack_par_kernel, dim3(B), dim3(BT_PAR_THREADS), (size_t)4 * T * sizeof(int), stream, code, T, B,
                           start, forward, region, counts);
    } else {
        hipLaunchKernelGGL(backtrack_kernel, dim3(B), dim3(64), lds, stream, code, T, B, start, forward, region, counts);
    }
    hipLaunchKernelGGL(offsets_kernel, dim3(1), dim3(256), 0, stream, counts, B, offsets, err, nerr, err_stride);
    hipLaunchKernelGGL(pack_kernel, dim3(B), dim3(256), 0, stream, region, counts, offsets, T, B, forward, pairs,
                       cap);
}

}  // namespace semicrf
