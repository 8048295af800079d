// scorer_mfma.hip -- scaled-inner-product interval scores on the CDNA4 matrix cores
// (LayersTransformer.py:406-441 after the Linear map).
//
//   S[e,b,c] = qscale * <q[c,e,:], k[c,b,:]> * len(|e-b|)  (+ diag[c,e] when e == b),   layout [T][T][C]
//
// The contraction is exact fp32: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, a k-ordered fmaf chain; no
// xf32/TF32 exists on gfx950), so scores match the reference's fp32 einsum to round-off and the CRF decode
// that consumes them stays comparable.  One workgroup = one 32x32 (end, begin) tile for 16 chains:
// each of the 4 waves runs the 32x32xD MFMA chain for 4 chains in turn (lane = one row of q / one row of k,
// the two half-waves take the two halves of the d axis -- any d permutation applied to both operands is
// legal), results are scaled, transposed through a padded LDS tile and written chain-contiguous (64-byte
// segments), lower triangle only unless the caller asks for the full square.
#include "common.h"

namespace semicrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int ST = 32;        // tile edge (positions)
constexpr int SC = 16;        // chains per workgroup
constexpr int SPAD = SC + 1;  // LDS row padding: conflict-free column writes

__device__ __forceinline__ float len_scale_mfma(int len, int mode)
{
    if (mode == SEMICRF_LEN_LINEAR) return (float)len;
    if (mode == SEMICRF_LEN_SQRT) return sqrtf((float)len);
    return 1.0f;
}

// tile list: blockIdx.x enumerates (et, bt) with bt <= et (or the full square), blockIdx.y the chain group
// NCH = chunks of 32 contraction values per half-wave (D / 64) when known at compile time (<= 4), else 0: with a
// compile-time trip count the two-stage load/multiply pipeline unrolls without branches and every wait is exact
template <bool ALIGNED, int NCH>
__global__ __launch_bounds__(256) void interval_score_mfma_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ diag, int C, int T, int D,
    long long ldq, long long ldk, long long ldd, float qscale, int mode, int full, float* __restrict__ S)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [ST*ST][SPAD]
    const int nt = (T + ST - 1) / ST;
    int et, bt;
    if (full) {
        et = blockIdx.x / nt; bt = blockIdx.x % nt;
    } else {
        // invert t = et*(et+1)/2 + bt
        int t = blockIdx.x;
        et = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
        while (et * (et + 1) / 2 > t) --et;
        while ((et + 1) * (et + 2) / 2 <= t) ++et;
        bt = t - et * (et + 1) / 2;
    }
    const int e0 = et * ST, b0 = bt * ST;
    const int cg = blockIdx.y * SC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane & 31, half = lane >> 5;
    const int Dh = D >> 1;                                  // d range of this half-wave: [half*Dh, half*Dh + Dh)
    const int er = e0 + row < T ? e0 + row : T - 1;         // clamped rows (masked at the write)
    const int br = b0 + row < T ? b0 + row : T - 1;

    for (int round = 0; round < SC / 4; ++round) {
        const int ci = wave + 4 * round;
        const int c = cg + ci < C ? cg + ci : C - 1;
        const float* qp = q + ((size_t)c * T + er) * ldq + (size_t)half * Dh;
        const float* kp = k + ((size_t)c * T + br) * ldk + (size_t)half * Dh;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        // chunks of 32 d-values per half-wave: 8 float4 of q and of k per lane; the next chunk is requested
        // before the 32 MFMAs of the current one so that the loads hide under the matrix pipe
        auto load_chunk = [&](float4 (&qa)[8], float4 (&ka)[8], int d0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float* qq = qp + d0 + 4 * m;
                const float* kk = kp + d0 + 4 * m;
                if (ALIGNED) {
                    qa[m] = *(const float4*)qq;
                    ka[m] = *(const float4*)kk;
                } else {                      // rows only 4-byte aligned (packed Linear output, ld = 2D+1)
                    qa[m] = make_float4(qq[0], qq[1], qq[2], qq[3]);
                    ka[m] = make_float4(kk[0], kk[1], kk[2], kk[3]);
                }
            }
        };
        auto mma_chunk = [&](const float4 (&qa)[8], const float4 (&ka)[8]) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].x, ka[m].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].y, ka[m].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].z, ka[m].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].w, ka[m].w, acc, 0, 0, 0);
            }
        };
        float4 qa0[8], ka0[8], qa1[8], ka1[8];
        load_chunk(qa0, ka0, 0);
        if (NCH > 0) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch & 1) {
                    if (ch + 1 < NCH) load_chunk(qa0, ka0, (ch + 1) * 32);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_chunk(qa1, ka1);
                } else {
                    if (ch + 1 < NCH) load_chunk(qa1, ka1, (ch + 1) * 32);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_chunk(qa0, ka0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int d0 = 0; d0 < Dh; d0 += 64) {
                if (d0 + 32 < Dh) load_chunk(qa1, ka1, d0 + 32);
                mma_chunk(qa0, ka0);
                if (d0 + 32 >= Dh) break;
                if (d0 + 64 < Dh) load_chunk(qa0, ka0, d0 + 64);
                mma_chunk(qa1, ka1);
            }
        }
        // C/D layout of 32x32: col j = lane & 31, row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        const int bj = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ei = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int e = e0 + ei, b = b0 + bj;
            float v = acc[r] * qscale;
            const int len = e > b ? e - b : b - e;
            v *= len_scale_mfma(len, mode);
            if (e == b && e < T) v += diag[((size_t)c * T + e) * ldd];
            tile[(ei * ST + bj) * SPAD + ci] = v;
        }
    }
    __syncthreads();
    // write out: 4 threads per cell (4 chains each), chain axis contiguous
    const int nq = SC / 4;
    for (int idx = threadIdx.x; idx < ST * ST * nq; idx += 256) {
        const int cell = idx / nq, qd = idx % nq;
        const int ei = cell / ST, bj = cell % ST;
        const int e = e0 + ei, b = b0 + bj;
        if (e >= T || b >= T || (!full && b > e)) continue;
        float* dst = S + ((size_t)e * T + b) * C + cg + qd * 4;
        const float* src = tile + cell * SPAD + qd * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (cg + qd * 4 + i < C) dst[i] = src[i];
    }
}

bool interval_score_mfma_supported(int C, int T, int D) { return D % 64 == 0 && T >= 1 && C >= 1; }

void launch_interval_score_mfma(const float* q, const float* k, const float* diag, int C, int T, int D,
                                long long ldq, long long ldk, long long ldd, float qscale, int mode, int full,
                                float* S, hipStream_t stream)
{
    const int nt = (T + ST - 1) / ST;
    const int ntiles = full ? nt * nt : nt * (nt + 1) / 2;
    const size_t lds = (size_t)ST * ST * SPAD * sizeof(float);
    const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ldq % 4 == 0 && ldk % 4 == 0;
    const int nch = (D / 64 <= 4) ? D / 64 : 0;
    const dim3 grid(ntiles, (C + SC - 1) / SC), block(256);
#define SEMICRF_FWD_LAUNCH(A, N)                                                                                        \
    do {                                                                                                                \
        static bool attr_set = false;                                                                                   \
        if (!attr_set) {                                                                                                \
            (void)hipFuncSetAttribute((const void*)interval_score_mfma_kernel<A, N>,                                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
            attr_set = true;                                                                                            \
        }                                                                                                               \
        hipLaunchKernelGGL((interval_score_mfma_kernel<A, N>), grid, block, lds, stream, q, k, diag, C, T, D, ldq, ldk, \
                           ldd, qscale, mode, full, S);                                                                 \
    } while (0)
#define SEMICRF_FWD_DISPATCH(A)                                                                                         \
    switch (nch) {                                                                                                      \
    case 1: SEMICRF_FWD_LAUNCH(A, 1); break;                                                                            \
    case 2: SEMICRF_FWD_LAUNCH(A, 2); break;                                                                            \
    case 3: SEMICRF_FWD_LAUNCH(A, 3); break;                                                                            \
    case 4: SEMICRF_FWD_LAUNCH(A, 4); break;                                                                            \
    default: SEMICRF_FWD_LAUNCH(A, 0); break;                                                                           \
    }
    if (aligned) { SEMICRF_FWD_DISPATCH(true) } else { SEMICRF_FWD_DISPATCH(false) }
#undef SEMICRF_FWD_DISPATCH
#undef SEMICRF_FWD_LAUNCH
}

}  // namespace semicrf
