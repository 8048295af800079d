// persist.hip -- persistent blocked sweep for the semi-CRF recurrences on gfx950 (impl 0 / auto).
//
// One launch computes, for every chain, the T-long dependent recurrence
//     u[p] = finalize( (+)_{j<p} ( u[j] (x) cell(p,j) ), skip(p) )        p = 0..T-1 (position order)
// in the (logsumexp,+) semiring (alpha/beta sweeps, NeuralSemiCRFInterval.py:402-410) or the
// (max,+) semiring with argmax (viterbi / viterbiBackward, :27-51, :122-144).  Positions run over
// frames ascending (DIR 0) or descending (DIR 1); cell(p,j) is score[end][begin] of the two frames.
//
// Work decomposition (positions in blocks of 16):
//   * SPINE workgroup, one per 16 chains: four waves form a ring, wave w owns position blocks
//     k = w, w+4, ...; a lane is (row r of the block, 4 chains as one float4).  Every wave applies each
//     newly finished u[j] to its own block's rows (band = the current block and the next three), the
//     owner of the current block finalises one position per step and publishes it through LDS to its
//     ring mates and through HBM to the panels.  The T-step dependent chain never leaves one CU.
//   * PANEL workgroup, one per (position block k >= 4, 32 chains): streams the far field -- all cells
//     (p in block k, j < 16(k-3)) -- tile by tile as the spine publishes u, keeps the partial
//     accumulators in registers and hands ONE number per (position, chain) to the spine.
//   * Hand-offs are 8-byte {tag, value} granules written with relaxed agent-scope atomic stores and
//     polled with relaxed agent-scope atomic loads (data is the flag; no fences, placement independent).
//     Roles are drawn from an atomic ticket so that a workgroup only ever waits on lower tickets; every
//     spin is bounded and raises the error word instead of hanging.
//
// HBM traffic: every lower-triangle cell is read exactly once (128-byte lines in the panels, 64-byte
// segments in the band).  Algorithmic bytes per sweep: 4*B*(T(T+1)/2 + T-1).
#include <atomic>
#include "common.h"

namespace semicrf {

constexpr int PB = 16;             // positions per block
constexpr int RING = 4;            // spine waves; band = RING-1 off-diagonal blocks + the diagonal block
constexpr int GS = 16;             // chains per spine workgroup
constexpr int GP = 32;             // chains per panel workgroup
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int SPIN_LIMIT = 1 << 20;       // global-memory polls (with s_sleep 2..8): ~0.3 s
constexpr int SPIN_LIMIT_LDS = 1 << 24;   // LDS polls (s_sleep 1): ~0.5 s
constexpr float RESCALE_THR = 64.0f;

typedef unsigned long long u64;
template <int V>
struct IC { static constexpr int value = V; };

struct SweepParams {
    const float* score;
    const float* noise;
    int T, B, K;
    int nSpine, nPanelGroups;
    unsigned tag;          // nonzero launch epoch
    unsigned* ctrl;        // [0] ticket, [1] error
    u64* ug;               // [T][B] granules of u (position-major: index p*B + c)
    u64* farg;             // [T][B] granules of far-field partials
    float* u_out;          // [T][B] by FRAME (natural-log units for LSE) or nullptr
    float* last_out;       // [B] value at the last position (logZ for DIR 0) or nullptr
    int* code;             // MAX: [B][T] backtrack codes by frame
};

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }

__device__ __forceinline__ u64 make_granule(unsigned tag, float v)
{
    return ((u64)tag << 32) | (u64)__float_as_uint(v);
}
__device__ __forceinline__ void store_granule(u64* p, u64 g)
{
    __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 load_granule(const u64* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int DIR>
__device__ __forceinline__ int frame_of(int p, int T) { return DIR == 0 ? p : T - 1 - p; }

// element offset (without the chain) of cell(pi, pj), pj < pi, in score [T][T][B]
template <int DIR>
__device__ __forceinline__ size_t cell_index(int pi, int pj, int T)
{
    return DIR == 0 ? (size_t)pi * T + pj : (size_t)(T - 1 - pj) * T + (T - 1 - pi);
}
// noise row between positions p-1 and p
template <int DIR>
__device__ __forceinline__ int gap_of(int p, int T) { return DIR == 0 ? p - 1 : T - 1 - p; }

// log2-domain lazily rescaled accumulator: value = M + log2(S); empty = (-inf, 0)
__device__ __forceinline__ void acc_push(float& M, float& S, float t)
{
    if (t > M + RESCALE_THR) {
        S = S * fexp2(M - t);   // M = -inf: exp2(-inf) = 0 and S = 0
        M = t;
    }
    S += fexp2(t - M);
}
__device__ __forceinline__ void acc_merge(float& M, float& S, float M2, float S2)
{
    if (S2 == 0.0f) return;
    if (M2 > M) {
        S = S * fexp2(M - M2) + S2;   // M = -inf -> S*0 (S = 0)
        M = M2;
    } else {
        S += S2 * fexp2(M2 - M);
    }
}

// sticky device-side status word (0 = fine); read and cleared by semicrf_debug_device_status()
__device__ unsigned g_dev_status = 0;

__device__ __forceinline__ void set_error(unsigned* ctrl, unsigned code)
{
    __hip_atomic_store(ctrl + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&g_dev_status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Bounded waiting: returns true when the caller must give up.  The first waiter to exceed its limit
// raises the error word; everybody else notices it within a few hundred polls and drains, so a
// protocol bug costs well under a second instead of hanging the GPU.
__device__ __forceinline__ bool spin_abort(unsigned* ctrl, int& spins, int limit, unsigned code)
{
    ++spins;
    if (spins > limit) { set_error(ctrl, code); return true; }
    if ((spins & 255) == 0 &&
        __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return true;
    return false;
}

// ---------------------------------------------------------------------------------------------
// SPINE role
// ---------------------------------------------------------------------------------------------
template <int MODE, int DIR>
__device__ void spine_role(const SweepParams& P, int sg, float (*ubuf)[GS], int* done_ptr)
{
    const int T = P.T, B = P.B;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int r = lane >> 2, qd = lane & 3;
    const int c = sg * GS + qd * 4;
    const bool cvalid = c < B;
    const size_t Bs = (size_t)B;
    const float* __restrict__ score = P.score;
    const float* __restrict__ noise = P.noise;
    const unsigned tag = P.tag;

    for (int k = wave; k < P.K; k += RING) {
        const int prow = k * PB + r;
        const bool rvalid = cvalid && prow < T;
        const int frow = frame_of<DIR>(prow, T);

        // diagonal cell and the singleton factor
        float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rvalid) d = *(const float4*)(score + ((size_t)frow * T + frow) * Bs + c);
        float sp[4];
        if (MODE == 0) {
            sp[0] = softplus_f(d.x) * LOG2E; sp[1] = softplus_f(d.y) * LOG2E;
            sp[2] = softplus_f(d.z) * LOG2E; sp[3] = softplus_f(d.w) * LOG2E;
        } else {
            sp[0] = d.x; sp[1] = d.y; sp[2] = d.z; sp[3] = d.w;
        }
        // skip weight between prow-1 and prow
        float nz[4] = {0.f, 0.f, 0.f, 0.f};
        if (rvalid && prow >= 1) {
            const float4 n4 = *(const float4*)(noise + (size_t)gap_of<DIR>(prow, T) * Bs + c);
            nz[0] = n4.x; nz[1] = n4.y; nz[2] = n4.z; nz[3] = n4.w;
        }

        float aM[4], aS[4];
        int aK[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { aM[i] = SEMICRF_NEG_INF; aS[i] = 0.f; aK[i] = 0x7fffffff; }

        const int jbeg = (k - (RING - 1)) > 0 ? (k - (RING - 1)) * PB : 0;
        const int jend = (k * PB + PB < T ? k * PB + PB : T);   // exclusive
        const int own0 = k * PB;

        // register double buffer of cells: chunk of 8 columns
        float xs[2][8][4];
        auto load_chunk = [&](auto bufc, int j0) {
            constexpr int buf = decltype(bufc)::value;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
                if (rvalid && j < prow) x = *(const float4*)(score + cell_index<DIR>(prow, j, T) * Bs + c);
                if (MODE == 0) {
                    float v0 = x.x * LOG2E, v1 = x.y * LOG2E, v2 = x.z * LOG2E, v3 = x.w * LOG2E;
                    if (j == prow - 1) {
                        // fold the skip term into the first sub-diagonal cell: log2(2^s + 2^n)
                        const float xv[4] = {v0, v1, v2, v3};
                        float wv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const float n2 = nz[i] * LOG2E;
                            const float m = fmaxf(xv[i], n2);
                            wv[i] = m + flog2(1.0f + fexp2(-fabsf(xv[i] - n2)));
                        }
                        v0 = wv[0]; v1 = wv[1]; v2 = wv[2]; v3 = wv[3];
                    }
                    xs[buf][u][0] = v0; xs[buf][u][1] = v1; xs[buf][u][2] = v2; xs[buf][u][3] = v3;
                } else {
                    xs[buf][u][0] = x.x; xs[buf][u][1] = x.y; xs[buf][u][2] = x.z; xs[buf][u][3] = x.w;
                }
            }
        };

        auto process_chunk = [&](auto bufc, int j0) {
            constexpr int buf = decltype(bufc)::value;
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + u;
                if (j >= jend) break;               // wave-uniform
                float uj[4];
                if (j < own0) {
                    // shadow phase: wait for the ring mate that owns block j/PB to publish u[j]
                    int spins = 0;
                    while (__hip_atomic_load(done_ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= j) {
                        __builtin_amdgcn_s_sleep(1);
                        if (spin_abort(P.ctrl, spins, SPIN_LIMIT_LDS, 2)) break;
                    }
                    const float4 uv = *(const float4*)&ubuf[j & 127][qd * 4];
                    uj[0] = uv.x; uj[1] = uv.y; uj[2] = uv.z; uj[3] = uv.w;
                } else {
                    // diagonal phase: this wave finalises position j = own0 + jj
                    const int jj = j - own0;
                    if (jj == 0 && k >= RING) {
                        // merge the far-field partial handed over by the panel workgroup
                        if (rvalid) {
                            int spins = 0;
                            u64 g[4];
                            bool ok = false;
                            while (!ok) {
                                ok = true;
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    g[i] = load_granule(P.farg + (size_t)prow * Bs + c + i);
                                    if (MODE == 0) ok = ok && ((unsigned)(g[i] >> 32) == tag);
                                    else ok = ok && ((unsigned)(g[i] >> 48) == (tag & 0xffffu));
                                }
                                if (!ok) {
                                    __builtin_amdgcn_s_sleep(2);
                                    if (spin_abort(P.ctrl, spins, SPIN_LIMIT, 3)) break;
                                }
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float fv = __uint_as_float((unsigned)g[i]);
                                if (MODE == 0) acc_push(aM[i], aS[i], fv);
                                else max_push(aM[i], aK[i], fv, (int)((g[i] >> 32) & 0xffffu));
                            }
                        }
                    }
                    float res[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (MODE == 0) {
                            res[i] = (prow == 0) ? sp[i] : (aM[i] + flog2(aS[i]) + sp[i]);
                        } else {
                            const float best = (prow == 0) ? 0.0f : aM[i];
                            res[i] = sp[i] > 0.0f ? best + sp[i] : best;
                        }
                    }
                    const int src = (jj << 2) | qd;
#pragma unroll
                    for (int i = 0; i < 4; ++i) uj[i] = __shfl(res[i], src);
                    if (r == jj && rvalid) {
                        // publish: LDS for the ring, granules for the panels, plain arrays for the caller
                        *(float4*)&ubuf[j & 127][qd * 4] = make_float4(res[0], res[1], res[2], res[3]);
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            store_granule(P.ug + (size_t)j * Bs + c + i, make_granule(tag, res[i]));
                        if (P.u_out) {
                            const float sc = MODE == 0 ? LN2 : 1.0f;
                            *(float4*)(P.u_out + (size_t)frow * Bs + c) =
                                make_float4(res[0] * sc, res[1] * sc, res[2] * sc, res[3] * sc);
                        }
                        if (P.last_out && j == T - 1) {
                            const float sc = MODE == 0 ? LN2 : 1.0f;
                            *(float4*)(P.last_out + c) = make_float4(res[0] * sc, res[1] * sc, res[2] * sc, res[3] * sc);
                        }
                        if (MODE == 1) {
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const int key = (prow == 0) ? -1 : aK[i];
                                P.code[(size_t)(c + i) * T + frow] = (key + 1) | (sp[i] > 0.0f ? 0x40000000 : 0);
                            }
                        }
                    }
                    // make the LDS data visible before the progress counter moves (in-order DS queue)
                    if (lane == (jj << 2))
                        __hip_atomic_store(done_ptr, j + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                // apply u[j] to this lane's row
                if (rvalid && j < prow) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (MODE == 0) {
                            acc_push(aM[i], aS[i], uj[i] + xs[buf][u][i]);
                        } else {
                            const int key = frame_of<DIR>(j, T);
                            if (j == prow - 1) max_push(aM[i], aK[i], uj[i] + nz[i], -1);
                            max_push(aM[i], aK[i], uj[i] + xs[buf][u][i], key);
                        }
                    }
                }
            }
        };

        int j0 = jbeg;
        load_chunk(IC<0>{}, j0);
        while (j0 < jend) {
            load_chunk(IC<1>{}, j0 + 8);
            process_chunk(IC<0>{}, j0);
            j0 += 8;
            if (j0 >= jend) break;
            load_chunk(IC<0>{}, j0 + 8);
            process_chunk(IC<1>{}, j0);
            j0 += 8;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PANEL role
// ---------------------------------------------------------------------------------------------
// Workgroup = 4 waves, position block k (16 positions), 32 chains.  lane = slot*8 + quad8:
// quad8 selects 4 of the 32 chains, so 8 consecutive lanes read one 128-byte line.
//   DIR 0: wave w owns positions 16k+4w+r (r<4); per tile a lane holds columns pj = 16m+slot+8h.
//   DIR 1: wave w owns tile rows pj = 16m+4w+r; a lane holds positions pi = 16k+slot+8h.
template <int MODE, int DIR>
__device__ void panel_role(const SweepParams& P, int pidx, float* lds)
{
    const int T = P.T, B = P.B;
    const int k = RING + pidx / P.nPanelGroups;
    const int g = pidx % P.nPanelGroups;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int slot = lane >> 3, q8 = lane & 7;
    const int c = g * GP + q8 * 4;
    const bool cvalid = c < B;
    const size_t Bs = (size_t)B;
    const float* __restrict__ score = P.score;
    const unsigned tag = P.tag;
    const int nTiles = k - (RING - 1);          // tiles m = 0 .. k-RING

    // accumulators: DIR 0 -> [r][chain], DIR 1 -> [h][chain] (only h < 2 used)
    float aM[4][4], aS[4][4];
    int aK[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int i = 0; i < 4; ++i) { aM[a][i] = SEMICRF_NEG_INF; aS[a][i] = 0.f; aK[a][i] = 0x7fffffff; }

    float4 x[2][4][2];   // [buffer][r][h]
    auto pi_of = [&](int rr, int h) { return DIR == 0 ? k * PB + wave * 4 + rr : k * PB + slot + 8 * h; };
    auto pj_of = [&](int m, int rr, int h) { return DIR == 0 ? m * PB + slot + 8 * h : m * PB + wave * 4 + rr; };
    auto load_tile = [&](auto bufc, int m) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pi = pi_of(rr, h), pj = pj_of(m, rr, h);
                float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                if (cvalid && pi < T) v = *(const float4*)(score + cell_index<DIR>(pi, pj, T) * Bs + c);
                x[buf][rr][h] = v;
            }
    };

    auto process_tile = [&](auto bufc, int m) {
        constexpr int buf = decltype(bufc)::value;
        // wait for the spine(s) of these 32 chains to publish block m (poll its last position), then
        // read the granules this lane needs; every granule carries its own tag.
        const int plast = m * PB + PB - 1;
        if (lane == 0) {
            int spins = 0;
            for (int half = 0; half < 2; ++half) {
                const int cc = g * GP + half * GS;
                if (cc >= B) break;
                while ((unsigned)(load_granule(P.ug + (size_t)plast * Bs + cc) >> 32) != tag) {
                    __builtin_amdgcn_s_sleep(8);
                    if (spin_abort(P.ctrl, spins, SPIN_LIMIT, 4)) break;
                }
            }
        }
        // lanes reconverge here; now fetch u for the columns/rows of this tile
        constexpr int NU = DIR == 0 ? 2 : 4;
        float uv[NU][4];
        if (cvalid) {
            int spins = 0;
            bool ok = false;
            while (!ok) {
                ok = true;
#pragma unroll
                for (int a = 0; a < NU; ++a) {
                    const int pj = DIR == 0 ? m * PB + slot + 8 * a : m * PB + wave * 4 + a;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const u64 gr = load_granule(P.ug + (size_t)pj * Bs + c + i);
                        ok = ok && ((unsigned)(gr >> 32) == tag);
                        uv[a][i] = __uint_as_float((unsigned)gr);
                    }
                }
                if (!ok) {
                    __builtin_amdgcn_s_sleep(2);
                    if (spin_abort(P.ctrl, spins, SPIN_LIMIT, 5)) break;
                }
            }
        } else {
#pragma unroll
            for (int a = 0; a < NU; ++a)
#pragma unroll
                for (int i = 0; i < 4; ++i) uv[a][i] = 0.f;
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pi = pi_of(rr, h);
                if (!(cvalid && pi < T)) continue;
                const int ai = DIR == 0 ? rr : h;
                const int ui = DIR == 0 ? h : rr;
                const float4 xv = x[buf][rr][h];
                const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (MODE == 0) {
                        acc_push(aM[ai][i], aS[ai][i], fmaf(xe[i], LOG2E, uv[ui][i]));
                    } else {
                        const int pj = pj_of(m, rr, h);
                        max_push(aM[ai][i], aK[ai][i], uv[ui][i] + xe[i], frame_of<DIR>(pj, T));
                    }
                }
            }
    };

    if (nTiles > 0) {
        int m = 0;
        load_tile(IC<0>{}, 0);
        while (m < nTiles) {
            if (m + 1 < nTiles) load_tile(IC<1>{}, m + 1);
            process_tile(IC<0>{}, m);
            ++m;
            if (m >= nTiles) break;
            if (m + 1 < nTiles) load_tile(IC<0>{}, m + 1);
            process_tile(IC<1>{}, m);
            ++m;
        }
    }

    // ---- reduce the partials and hand them to the spine ------------------------------------------
    if (DIR == 0) {
        // across the 8 column slots of the wave (lane bits 3..5)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int off = 8; off < 64; off <<= 1) {
                    const float oM = __shfl_xor(aM[rr][i], off);
                    if (MODE == 0) {
                        const float oS = __shfl_xor(aS[rr][i], off);
                        acc_merge(aM[rr][i], aS[rr][i], oM, oS);
                    } else {
                        const int oK = __shfl_xor(aK[rr][i], off);
                        max_push(aM[rr][i], aK[rr][i], oM, oK);
                    }
                }
            }
        if (slot == 0 && cvalid) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int pi = k * PB + wave * 4 + rr;
                if (pi >= T) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u64 gr;
                    if (MODE == 0) gr = make_granule(tag, aM[rr][i] + flog2(aS[rr][i]));
                    else gr = ((u64)(((tag & 0xffffu) << 16) | ((unsigned)aK[rr][i] & 0xffffu)) << 32) |
                              (u64)__float_as_uint(aM[rr][i]);
                    store_granule(P.farg + (size_t)pi * Bs + c + i, gr);
                }
            }
        }
    } else {
        // across the 4 waves through LDS: lds[wave][h][slot][q8*4+i] x {M, S/K}
        float* lm = lds;
        float* ls = lds + 4 * 2 * 8 * 32;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int idx = ((wave * 2 + h) * 8 + slot) * 32 + q8 * 4 + i;
                lm[idx] = aM[h][i];
                ls[idx] = MODE == 0 ? aS[h][i] : __int_as_float(aK[h][i]);
            }
        __syncthreads();
        // 16 positions x 32 chains = 512 results, 2 per thread
        for (int e = tid; e < 2 * 8 * 32; e += 256) {
            const int ch = e & 31, sl = (e >> 5) & 7, h = e >> 8;
            float M = SEMICRF_NEG_INF, S = 0.f;
            int Kk = 0x7fffffff;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                const int idx = ((w * 2 + h) * 8 + sl) * 32 + ch;
                if (MODE == 0) acc_merge(M, S, lm[idx], ls[idx]);
                else max_push(M, Kk, lm[idx], __float_as_int(ls[idx]));
            }
            const int pi = k * PB + sl + 8 * h;
            const int cc = g * GP + ch;
            if (pi < T && cc < B) {
                u64 gr;
                if (MODE == 0) gr = make_granule(tag, M + flog2(S));
                else gr = ((u64)(((tag & 0xffffu) << 16) | ((unsigned)Kk & 0xffffu)) << 32) | (u64)__float_as_uint(M);
                store_granule(P.farg + (size_t)pi * Bs + cc, gr);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
template <int MODE, int DIR>
__global__ __launch_bounds__(256, 2) void persist_sweep_kernel(SweepParams P)
{
    __shared__ float s_ubuf[128][GS];     // spine: ring of the last 128 published positions
    __shared__ float s_red[2 * 4 * 2 * 8 * 32];   // panel DIR 1 reduction
    __shared__ int s_ticket;
    __shared__ int s_done;
    if (threadIdx.x == 0) {
        s_ticket = (int)atomicAdd(P.ctrl, 1u);
        s_done = 0;
    }
    __syncthreads();
    const int ticket = s_ticket;
    if (ticket < P.nSpine) spine_role<MODE, DIR>(P, ticket, s_ubuf, &s_done);
    else panel_role<MODE, DIR>(P, ticket - P.nSpine, s_red);
}

size_t persist_workspace_bytes(int T, int B)
{
    return align_up(256) + 2 * align_up((size_t)T * B * sizeof(u64));
}

bool persist_supported(int T, int B) { return (B % 4 == 0) && T >= 1 && T < 65535; }

static unsigned next_tag()
{
    static std::atomic<unsigned> counter{0};
    const unsigned lo = (counter.fetch_add(1) % 65535u) + 1u;   // 1..65535
    return (lo << 16) | lo;                                      // both 16-bit halves nonzero
}

// mode 0 = LSE, 1 = MAX.  ws must hold persist_workspace_bytes().  Enqueues a memset + one kernel.
int launch_persist_sweep(int mode, int dir, const float* score, const float* noise, int T, int B, float* u_out,
                         float* last_out, int* code, void* ws, hipStream_t stream)
{
    SweepParams P;
    P.score = score; P.noise = noise; P.T = T; P.B = B; P.K = (T + PB - 1) / PB;
    P.nSpine = (B + GS - 1) / GS;
    P.nPanelGroups = (B + GP - 1) / GP;
    P.tag = next_tag();
    char* w = (char*)ws;
    P.ctrl = (unsigned*)w;
    P.ug = (u64*)(w + align_up(256));
    P.farg = (u64*)(w + align_up(256) + align_up((size_t)T * B * sizeof(u64)));
    P.u_out = u_out; P.last_out = last_out; P.code = code;
    const size_t zbytes = persist_workspace_bytes(T, B);
    if (hipMemsetAsync(ws, 0, zbytes, stream) != hipSuccess) return 1;
    const int nPanelBlocks = P.K > RING ? P.K - RING : 0;
    const int grid = P.nSpine + nPanelBlocks * P.nPanelGroups;
    dim3 g(grid), b(256);
    if (mode == 0 && dir == 0) hipLaunchKernelGGL((persist_sweep_kernel<0, 0>), g, b, 0, stream, P);
    else if (mode == 0 && dir == 1) hipLaunchKernelGGL((persist_sweep_kernel<0, 1>), g, b, 0, stream, P);
    else if (mode == 1 && dir == 0) hipLaunchKernelGGL((persist_sweep_kernel<1, 0>), g, b, 0, stream, P);
    else hipLaunchKernelGGL((persist_sweep_kernel<1, 1>), g, b, 0, stream, P);
    return 0;
}

int read_and_clear_device_status()
{
    unsigned v = 0, z = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_dev_status), sizeof(v)) != hipSuccess) return -1;
    if (v != 0 && hipMemcpyToSymbol(HIP_SYMBOL(g_dev_status), &z, sizeof(z)) != hipSuccess) return -1;
    return (int)v;
}

}  // namespace semicrf
