// scorer_bwd.hip -- backward of the scaled-inner-product interval scores on the CDNA4 matrix cores
// (the autograd of LayersTransformer.py:406-441 after the Linear map).
//
//   S[e,b,c] = qscale * <q[c,e,:], k[c,b,:]> * len(|e-b|)  (+ diag[c,e] when e == b),   dS given as [T][T][C], e >= b
//   G[e,b,c] = dS[e,b,c] * qscale * len(|e-b|) for e >= b, 0 above the diagonal (at e == b: len(0) = 0 unless scaling is off)
//   dq[c,e,:] = sum_{b<=e} G[e,b,c] k[c,b,:]     dk[c,b,:] = sum_{e>=b} G[e,b,c] q[c,e,:]     ddiag[c,t] = dS[t,t,c]
//
// One kernel, two instantiations: ROWS_E (dq: a workgroup owns 32 end frames and walks the begin tiles to the
// left of the diagonal) and its transpose (dk: owns 32 begin frames, walks the end tiles below).  A workgroup
// is 8 waves = 8 chains; per step the 32x32x8-chain block of dS is read as 32-byte runs, scaled and transposed
// through LDS into per-chain [e][b] tiles (double buffered), and each wave accumulates its chain's 32 x D
// result with exact-fp32 v_mfma_f32_32x32x2_f32 (A = the G tile from LDS, B = the rows of k or q straight from
// global memory, 128-byte runs).  Every output element is written once (no atomics): results are deterministic.
#include "common.h"

namespace semicrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BT = 32;            // tile edge (frames)
constexpr int BC = 8;             // chains per workgroup (one per wave)
constexpr int BPAD = BT + 1;      // LDS row pitch: conflict-free transposed reads
constexpr int BND_MAX = 8;        // D / 32 <= 8

__device__ __forceinline__ float len_scale_bwd(int len, int mode)
{
    if (mode == SEMICRF_LEN_LINEAR) return (float)len;
    if (mode == SEMICRF_LEN_SQRT) return sqrtf((float)len);
    return 1.0f;
}

// FUSED: dS is not read; the cotangent is built on the fly from the CRF's own quantities (the dense
// [T][T][C] gradient is never written):  dS[e,b,c] = gout[c] * marginal[e,b,c],
//   marginal = exp(alpha[b,c] + beta[e,c] + S[e,b,c] - logZ[c])                       (e > b)
//            = exp(alpha[t,c] + beta[t,c] + S[t,t,c] - 2 softplus(S[t,t,c]) - logZ[c])  (e == b == t)
// (NeuralSemiCRFInterval.py:424-447); `dS` then points at S itself.
struct FusedArgs {
    const float* alpha;   // [T][C] by frame
    const float* beta;    // [T][C] by frame
    const float* logZ;    // [C]
    const float* gout;    // [C]
};

__device__ __forceinline__ float marginal_of(float s, float a, float b, float lz, bool diag)
{
    float x = a + b + s - lz;
    if (diag) x -= 2.0f * softplus_f(s);
    return __expf(x);
}

template <bool ROWS_E, int ND, bool FUSED>
__global__ __launch_bounds__(64 * BC) void interval_score_bwd_kernel(
    const float* __restrict__ dS, const float* __restrict__ other, float* __restrict__ out, int C, int T,
    long long ldo, long long ldout, float qscale, int mode, FusedArgs F)
{
    extern __shared__ __attribute__((aligned(16))) float g_lds[];     // [buffer][chain][e][b], 2 x 8 x 32 x 33 floats
    auto G = [&](int buf, int ch) -> float* { return g_lds + (buf * BC + ch) * (BT * BPAD); };
    const int nt = (T + BT - 1) / BT;
    const int rt = blockIdx.x;                     // row tile (end frames for dq, begin frames for dk)
    const int cg = blockIdx.y * BC;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int col = lane & 31, half = lane >> 5;
    const int c = cg + wave < C ? cg + wave : C - 1;
    // dq walks the begin tiles 0 .. rt, dk the end tiles rt .. nt-1
    const int nstep = ROWS_E ? rt + 1 : nt - rt;
    auto tile_of = [&](int s) { return ROWS_E ? s : rt + s; };

    // stage the (e-tile, b-tile) block of dS for 8 chains: 1024 cells x 32 bytes, two float4 per cell
    const bool vec = (C % 4 == 0) && (((uintptr_t)dS & 15) == 0);      // uniform: 16-byte loads are aligned
    auto stage = [&](int buf, int ct) __attribute__((always_inline)) {
        const int e0 = (ROWS_E ? rt : ct) * BT, b0 = (ROWS_E ? ct : rt) * BT;
        constexpr int NS = (BT * BT * 2) / (64 * BC);
        float4 sv[NS];
#pragma unroll
        for (int it = 0; it < NS; ++it) {
            const int idx = tid + it * 64 * BC;
            const int cell = idx >> 1, quad = idx & 1;
            const int el = cell >> 5, bl = cell & 31;
            const int e = e0 + el, b = b0 + bl;
            const int cc = cg + quad * 4;
            sv[it] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < T && b <= e) {          // e == b: len(0) = 0 unless length scaling is off
                const float* src = dS + ((size_t)e * T + b) * C + cc;
                if (vec) {
                    if (cc < C) sv[it] = *(const float4*)src;
                } else {
                    if (cc + 0 < C) sv[it].x = src[0];
                    if (cc + 1 < C) sv[it].y = src[1];
                    if (cc + 2 < C) sv[it].z = src[2];
                    if (cc + 3 < C) sv[it].w = src[3];
                }
            }
        }
        if (FUSED) {
            // this thread's four chains are the same in every iteration (cc depends on tid only): logZ and gout once
            const int ccq = cg + (tid & 1) * 4;
            float lz[4], gz[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                lz[i] = ccq + i < C ? F.logZ[ccq + i] : 0.f;
                gz[i] = ccq + i < C ? F.gout[ccq + i] : 0.f;
            }
            float4 av[NS], bw[NS];
#pragma unroll
            for (int it = 0; it < NS; ++it) {
                const int idx = tid + it * 64 * BC;
                const int cell = idx >> 1;
                const int el = cell >> 5, bl = cell & 31;
                const int e = e0 + el < T ? e0 + el : T - 1, b = b0 + bl < T ? b0 + bl : T - 1;
                const float* ap = F.alpha + (size_t)b * C + ccq;
                const float* bp = F.beta + (size_t)e * C + ccq;
                if (vec && ccq < C) { av[it] = *(const float4*)ap; bw[it] = *(const float4*)bp; }
                else {
                    av[it] = make_float4(ccq < C ? ap[0] : 0.f, ccq + 1 < C ? ap[1] : 0.f, ccq + 2 < C ? ap[2] : 0.f, ccq + 3 < C ? ap[3] : 0.f);
                    bw[it] = make_float4(ccq < C ? bp[0] : 0.f, ccq + 1 < C ? bp[1] : 0.f, ccq + 2 < C ? bp[2] : 0.f, ccq + 3 < C ? bp[3] : 0.f);
                }
            }
#pragma unroll
            for (int it = 0; it < NS; ++it) {
                const int idx = tid + it * 64 * BC;
                const int cell = idx >> 1;
                const int el = cell >> 5, bl = cell & 31;
                const int e = e0 + el, b = b0 + bl;
                if (e < T && b <= e) {
                    const bool dg = e == b;
                    sv[it].x = gz[0] * marginal_of(sv[it].x, av[it].x, bw[it].x, lz[0], dg);
                    sv[it].y = gz[1] * marginal_of(sv[it].y, av[it].y, bw[it].y, lz[1], dg);
                    sv[it].z = gz[2] * marginal_of(sv[it].z, av[it].z, bw[it].z, lz[2], dg);
                    sv[it].w = gz[3] * marginal_of(sv[it].w, av[it].w, bw[it].w, lz[3], dg);
                }
            }
        }
#pragma unroll
        for (int it = 0; it < NS; ++it) {
            const int idx = tid + it * 64 * BC;
            const int cell = idx >> 1, quad = idx & 1;
            const int el = cell >> 5, bl = cell & 31;
            const int d = (e0 + el) - (b0 + bl);
            const float sc = d >= 0 ? qscale * len_scale_bwd(d, mode) : 0.0f;
            G(buf, quad * 4 + 0)[el * BPAD + bl] = sv[it].x * sc;
            G(buf, quad * 4 + 1)[el * BPAD + bl] = sv[it].y * sc;
            G(buf, quad * 4 + 2)[el * BPAD + bl] = sv[it].z * sc;
            G(buf, quad * 4 + 3)[el * BPAD + bl] = sv[it].w * sc;
        }
    };

    f32x16 acc[ND];
#pragma unroll
    for (int n = 0; n < ND; ++n)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[n][i] = 0.0f;

    stage(0, tile_of(0));
    __syncthreads();
    for (int s = 0, buf = 0; s < nstep; ++s, buf ^= 1) {
        const int ct = tile_of(s);
        if (s + 1 < nstep) stage(buf ^ 1, tile_of(s + 1));      // the next block lands while this one is multiplied
        // A operand: 32 rows x 32 contraction steps of this wave's chain; lane = (row, contraction parity)
        float a[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int kk = 2 * i + half;
            a[i] = ROWS_E ? G(buf, wave)[col * BPAD + kk] : G(buf, wave)[kk * BPAD + col];
        }
        // B operand: rows kk of the other factor (k for dq, q for dk) of the walked tile, 32 columns per block
        const int r0 = ct * BT;
        const float* ob = other + (size_t)c * T * ldo;
        // (scheduling fences keep the 16 loads of the next block together and ahead of this block's 16 matrix
        // instructions: left alone the compiler pairs every load with an immediate wait)
        float bv[2][16];
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int rr = r0 + 2 * i + half;
            bv[0][i] = ob[(size_t)(rr < T ? rr : T - 1) * ldo + col];            // rows >= T meet G == 0
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int n = 0; n < ND; ++n) {
            if (n + 1 < ND) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int rr = r0 + 2 * i + half;
                    bv[(n + 1) & 1][i] = ob[(size_t)(rr < T ? rr : T - 1) * ldo + (n + 1) * 32 + col];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], bv[n & 1][i], acc[n], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }

    // C/D layout of 32x32: col j = lane & 31, row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    if (cg + wave < C) {
        float* ob = out + (size_t)c * T * ldout;
#pragma unroll
        for (int n = 0; n < ND; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = rt * BT + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (row < T) ob[(size_t)row * ldout + n * 32 + col] = acc[n][r];
            }
    }
}

// ddiag[c][t] = dS[t][t][c]
// (Cs, SL: the slot layout of dS's chain axis, common.h; Cs == C and group == pitch: chains and slots coincide)
__global__ __launch_bounds__(256) void interval_score_bwd_diag_kernel(const float* __restrict__ dS, float* __restrict__ ddiag,
                                                                       int C, int T, long long ldd, int Cs, ChainSlots SL)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    ddiag[(size_t)c * T * ldd + (size_t)t * ldd] = dS[((size_t)t * T + t) * Cs + slot_of_chain(SL, c)];
}

// FUSED ddiag[c][t] = gout[c] * marginal[t,t,c]
__global__ __launch_bounds__(256) void interval_score_bwd_diag_fused_kernel(const float* __restrict__ S, FusedArgs F,
                                                                             float* __restrict__ ddiag, int C, int T, long long ldd,
                                                                             int Cs, ChainSlots SL)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * C) return;
    const int t = (int)(i / C), c = (int)(i % C);
    const int sl = slot_of_chain(SL, c);
    const float s = S[((size_t)t * T + t) * Cs + sl];
    ddiag[((size_t)c * T + t) * ldd] = F.gout[sl] * marginal_of(s, F.alpha[(size_t)t * Cs + sl], F.beta[(size_t)t * Cs + sl], F.logZ[sl], true);
}

// Gradient of the merged projection's per-(chain, end) constant (interval_score_fwd_p, rowc: S[e,b] = qscale (q_e.k_b + rowc_e) len(e-b)):
//   drowc[c][e] = qscale * sum_{b <= e} dS[e,b,slot(c)] * len(e-b)                  (FUSED: dS = gout * marginal, built on the fly)
// One thread per (end, chain), chains fastest (coalesced over the slot axis), the begins summed in order: deterministic.
template <bool FUSED>
__global__ __launch_bounds__(256) void interval_score_bwd_rowsum_kernel(const float* __restrict__ dS, FusedArgs F,
                                                                         float* __restrict__ drowc, long long ldrc, int C, int T,
                                                                         float qscale, int mode, int Cs, ChainSlots SL)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= (size_t)T * C) return;
    const int e = (int)(i / C), c = (int)(i % C);
    const int sl = slot_of_chain(SL, c);
    float lz = 0.f, gz = 0.f, be = 0.f;
    if (FUSED) { lz = F.logZ[sl]; gz = F.gout[sl]; be = F.beta[(size_t)e * Cs + sl]; }
    const float* row = dS + (size_t)e * T * Cs + sl;
    float acc = 0.0f;
    for (int b = 0; b <= e; ++b) {
        float x = row[(size_t)b * Cs];
        if (FUSED) x = gz * marginal_of(x, F.alpha[(size_t)b * Cs + sl], be, lz, b == e);
        acc += x * len_scale_bwd(e - b, mode);
    }
    drowc[((size_t)c * T + e) * ldrc] = qscale * acc;
}

void launch_interval_score_bwd_rowsum(const float* dS, const float* const* fused, float* drowc, long long ldrc, int C, int T,
                                      float qscale, int mode, int group, int pitch, hipStream_t stream)
{
    const size_t n = (size_t)T * C;
    const int Cs = (C / group) * pitch;
    const ChainSlots SL{group, pitch};
    if (fused) {
        const FusedArgs F{fused[0], fused[1], fused[2], fused[3]};
        hipLaunchKernelGGL(interval_score_bwd_rowsum_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dS, F, drowc, ldrc,
                           C, T, qscale, mode, Cs, SL);
    } else {
        const FusedArgs F{nullptr, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(interval_score_bwd_rowsum_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dS, F, drowc, ldrc,
                           C, T, qscale, mode, Cs, SL);
    }
}

// The evalPath part of logProb's gradient pushed through the scorer: every interval (b, e) of chain c contributes
// w = gout[c] * qscale * len(e-b) to d S[e,b,c], i.e.  dq[c,e,:] += w k[c,b,:],  dk[c,b,:] += w q[c,e,:],
// ddiag[c,e] += gout[c] when b == e.  One wave per interval; atomics keep duplicate intervals (a caller error) exact.
__global__ __launch_bounds__(256) void interval_score_path_bwd_kernel(
    const float* __restrict__ gout, const int* __restrict__ pairs, int K, const int* __restrict__ offsets,
    const float* __restrict__ q, const float* __restrict__ k, int C, int T, int D, long long ldq, long long ldk, float qscale,
    int mode, float* dq, float* dk, float* ddiag, long long lddq, long long lddk, long long lddd, int Cs, ChainSlots SL,
    float* drowc, long long lddrc)
{
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= K) return;
    int lo = 0, hi = Cs;                      // slot of interval i: largest s with offsets[s] <= i (ghost slots hold no interval)
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (offsets[mid] <= i) lo = mid; else hi = mid;
    }
    const int c = chain_of_slot(SL, lo);
    if (c < 0) return;                        // an interval filed under a ghost slot (a caller error): ignored
    const int b = pairs[2 * i], e = pairs[2 * i + 1];
    const float g = gout[lo];
    const float w = g * qscale * len_scale_bwd(e - b, mode);
    const float* qe = q + ((size_t)c * T + e) * ldq;
    const float* kb = k + ((size_t)c * T + b) * ldk;
    for (int d = lane; d < D; d += 64) {
        if (dq) atomicAdd(dq + ((size_t)c * T + e) * lddq + d, w * kb[d]);
        if (dk) atomicAdd(dk + ((size_t)c * T + b) * lddk + d, w * qe[d]);
    }
    if (ddiag && b == e && lane == 0) atomicAdd(ddiag + ((size_t)c * T + e) * lddd, g);
    if (drowc && lane == 0) atomicAdd(drowc + ((size_t)c * T + e) * lddrc, w);        // the merged projection's row constant
}

void launch_interval_score_path_bwd(const float* gout, const int* pairs, int K, const int* offsets, const float* q,
                                    const float* k, int C, int T, int D, long long ldq, long long ldk, float qscale, int mode,
                                    float* dq, float* dk, float* ddiag, long long lddq, long long lddk, long long lddd,
                                    hipStream_t stream, int group, int pitch, float* drowc, long long lddrc)
{
    if (K <= 0) return;
    hipLaunchKernelGGL(interval_score_path_bwd_kernel, dim3((K + 3) / 4), dim3(256), 0, stream, gout, pairs, K, offsets, q, k, C,
                       T, D, ldq, ldk, qscale, mode, dq, dk, ddiag, lddq, lddk, lddd, (C / group) * pitch, ChainSlots{group, pitch},
                       drowc, lddrc);
}

// ddiag alone, for the packed path (which produces dq and dk): plain (F == nullptr) or fused
void launch_interval_score_bwd_diag(const float* dS, const float* const* fused, float* ddiag, int C, int T, long long lddd,
                                    int group, int pitch, hipStream_t stream)
{
    const size_t n = (size_t)T * C;
    const int Cs = (C / group) * pitch;
    const ChainSlots SL{group, pitch};
    if (fused) {
        const FusedArgs F{fused[0], fused[1], fused[2], fused[3]};
        hipLaunchKernelGGL(interval_score_bwd_diag_fused_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dS, F, ddiag,
                           C, T, lddd, Cs, SL);
    } else {
        hipLaunchKernelGGL(interval_score_bwd_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dS, ddiag, C, T,
                           lddd, Cs, SL);
    }
}

bool interval_score_bwd_supported(int C, int T, int D) { return D % 32 == 0 && D >= 32 && D <= 32 * BND_MAX && T >= 1 && C >= 1; }

template <bool ROWS_E, bool FUSED>
static void launch_bwd_pass(const float* dS, const float* other, float* out, int C, int T, int D, long long ldo,
                            long long ldout, float qscale, int mode, FusedArgs F, hipStream_t stream)
{
    const dim3 grid((T + BT - 1) / BT, (C + BC - 1) / BC), block(64 * BC);
    const size_t lds = (size_t)2 * BC * BT * BPAD * sizeof(float);
    switch (D / 32) {
#define SEMICRF_BWD_CASE(N)                                                                                             \
    case N: {                                                                                                           \
        static PerDeviceOnce attr_once;                                                                                   \
        if (attr_once.first()) {                                                                                                \
            (void)hipFuncSetAttribute((const void*)interval_score_bwd_kernel<ROWS_E, N, FUSED>,                                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        }                                                                                                               \
        hipLaunchKernelGGL((interval_score_bwd_kernel<ROWS_E, N, FUSED>), grid, block, lds, stream, dS, other, out, C, T,  \
                           ldo, ldout, qscale, mode, F);                                                                        \
        break;                                                                                                          \
    }
        SEMICRF_BWD_CASE(1) SEMICRF_BWD_CASE(2) SEMICRF_BWD_CASE(3) SEMICRF_BWD_CASE(4)
        SEMICRF_BWD_CASE(5) SEMICRF_BWD_CASE(6) SEMICRF_BWD_CASE(7) SEMICRF_BWD_CASE(8)
#undef SEMICRF_BWD_CASE
    }
}

void launch_interval_score_bwd(const float* dS, const float* q, const float* k, int C, int T, int D, long long ldq,
                               long long ldk, float qscale, int mode, float* dq, float* dk, float* ddiag,
                               long long lddq, long long lddk, long long lddd, hipStream_t stream)
{
    const FusedArgs F{nullptr, nullptr, nullptr, nullptr};
    if (dq) launch_bwd_pass<true, false>(dS, k, dq, C, T, D, ldk, lddq, qscale, mode, F, stream);
    if (dk) launch_bwd_pass<false, false>(dS, q, dk, C, T, D, ldq, lddk, qscale, mode, F, stream);
    if (ddiag) {
        const size_t n = (size_t)T * C;
        hipLaunchKernelGGL(interval_score_bwd_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dS, ddiag,
                           C, T, lddd, C, ChainSlots{C, C});
    }
}

void launch_interval_score_bwd_fused(const float* S, const float* alpha, const float* beta, const float* logZ,
                                     const float* gout, const float* q, const float* k, int C, int T, int D,
                                     long long ldq, long long ldk, float qscale, int mode, float* dq, float* dk,
                                     float* ddiag, long long lddq, long long lddk, long long lddd, hipStream_t stream)
{
    const FusedArgs F{alpha, beta, logZ, gout};
    if (dq) launch_bwd_pass<true, true>(S, k, dq, C, T, D, ldk, lddq, qscale, mode, F, stream);
    if (dk) launch_bwd_pass<false, true>(S, q, dk, C, T, D, ldq, lddk, qscale, mode, F, stream);
    if (ddiag) {
        const size_t n = (size_t)T * C;
        hipLaunchKernelGGL(interval_score_bwd_diag_fused_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, S, F,
                           ddiag, C, T, lddd, C, ChainSlots{C, C});
    }
}

}  // namespace semicrf
