// persist.hip -- persistent blocked sweep for the semi-CRF recurrences on gfx950 (impl 0 / auto).
//
// One launch computes, for every chain, the T-long dependent recurrence
//     u[p] = finalize( (+)_{j<p} ( u[j] (x) cell(p,j) ), skip(p) )        p = 0..T-1 (position order)
// in the (logsumexp,+) semiring (alpha/beta sweeps, NeuralSemiCRFInterval.py:402-410) or the
// (max,+) semiring with argmax (viterbi / viterbiBackward, :27-51, :122-144).  Positions run over
// frames ascending (DIR 0) or descending (DIR 1); cell(p,j) is score[end][begin] of the two frames.
//
// Work decomposition (positions in blocks of 16):
//   * SPINE workgroup, one per 8 chains: four waves form a ring, wave w owns position blocks
//     k = w, w+4, ...; a lane is (row r of the block, 2 chains).  Every wave applies each newly
//     finished u[j] to its own block's rows (band = the current block and the next three); the owner
//     of the current block finalises one position per step (a ~35-instruction dependent step: a lone
//     wave issues one instruction per ~4 cycles, so the step is kept that small) and publishes it
//     through LDS to its ring mates; once per block it publishes 16 positions to HBM for the panels.
//     The T-step dependent chain never leaves one CU.  Band cells are prefetched 16..32 steps ahead
//     in registers.
//   * PANEL workgroup, one per (position block k >= 4, 32 chains): streams the far field -- all cells
//     (p in block k, j < 16(k-3)) -- tile by tile as the spine publishes u, keeps the partial
//     accumulators in registers and hands ONE number per (position, chain) to the spine.
//   * Hand-offs are 8-byte {tag, value} granules written with relaxed agent-scope atomic stores and
//     polled with relaxed agent-scope atomic loads (data is the flag; no fences, placement independent).
//     Roles are drawn from an atomic ticket so that a workgroup only ever waits on lower tickets; every
//     spin is bounded and raises the error word instead of hanging.
//
// HBM traffic: every lower-triangle cell is read exactly once (128-byte lines in the panels, 32-byte
// segments in the band).  Algorithmic bytes per sweep: 4*B*(T(T+1)/2 + T-1).
#include <atomic>
#include <stdlib.h>
#include "common.h"

#ifndef SEMICRF_ABL
#define SEMICRF_ABL 0      // timing-ablation bits for the LSE diagonal step (development only)
#endif

namespace semicrf {

constexpr int PB = 16;             // positions per block
constexpr int RING = 4;            // spine waves; band = RING-1 off-diagonal blocks + the diagonal block
constexpr int GS = 8;              // chains per spine workgroup (2 per lane)
constexpr int GP = 32;             // chains per panel workgroup (4 per lane)
constexpr int TPT = 16;            // tiles (column blocks) per panel task
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int SPIN_LIMIT = 1 << 20;       // global-memory polls (with s_sleep): ~0.3 s
constexpr int SPIN_LIMIT_LDS = 1 << 24;   // LDS polls (s_sleep 1): ~0.5 s
constexpr float RESCALE_THR = 64.0f;

typedef unsigned long long u64;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
template <int V>
struct IC { static constexpr int value = V; };

struct SweepParams {
    const float* score;
    const float* noise;
    int T, B, K;
    int nSpine, nPanelGroups;
    int nTasks;            // panel tasks (k ascending, then column part, then chain group)
    unsigned tag;          // nonzero launch epoch
    unsigned dbg;          // SEMICRF_DEBUG_FLAGS (timing experiments only; results are wrong when set):
                           // 1 spine ignores far partials, 2 panels exit at once, 4 panels do not wait for u,
                           // 8 spine exits at once, 16 spine 0 records per-step timestamps,
                           // 32 panels only stream their cells (no granules, no math)
    unsigned* ctrl;        // [0] ticket, [1] error, [2] panel task queue head, [3] panels that stepped aside,
                           // [64 .. 64+4096) one flag per compute unit: a spine lives here
    u64* ts;               // [T] debug timestamps of the diagonal steps of spine 0 (dbg & 16)
    u64* ug;               // [T][B] granules of u (position-major: index p*B + c)
    u64* farg;             // [parts][T][B] granules of far-field partials (part = column range of TPT tiles)
    float* u_out;          // [T][B] by FRAME (natural-log units for LSE) or nullptr
    float* last_out;       // [B] value at the last position (logZ for DIR 0) or nullptr
    int* code;             // MAX: [B][T] backtrack codes by frame
    // GRAD (LSE, DIR 1 only): marginals are a by-product of the beta sweep (NeuralSemiCRFInterval.py:424-447, :469-472)
    const float* vfwd;     // [T][B] alpha values by frame (natural log)
    const float* logZ;     // [B]
    const float* gout;     // [B] upstream gradient
    float* dScore;         // [T][T][B]: lower triangle + diagonal written here (the upper triangle by zero_upper_kernel)
    float* dNoise;         // [T-1][B]
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

// log2-domain accumulator, exact running max, one exp per push: value = M + log2(S); empty = (-inf, 0)
__device__ __forceinline__ void acc_push1(float& M, float& S, float t)
{
    const float d = t - M;                       // M = -inf -> +inf
    const float e = fexp2(-fabsf(d));
    const bool up = d > 0.0f;
    S = up ? fmaf(S, e, 1.0f) : S + e;
    M = up ? t : M;
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
// softplus in log2 units: log2(1 + 2^(x*log2e)), linear above the reference's threshold (20)
__device__ __forceinline__ float softplus2(float x)
{
    const float x2 = x * LOG2E;
    return x > 20.0f ? x2 : flog2(1.0f + fexp2(x2));
}

// identifies the compute unit this wave runs on: (XCC id, SE/SH/CU ids of HW_ID) -- used only to keep panel
// workgroups off the CUs that host a spine (they would steal issue slots from the latency-critical wave)
__device__ __forceinline__ unsigned cu_key()
{
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);       // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // HW_REG_XCC_ID
    return ((xcc & 15u) << 8) | ((hw >> 8) & 255u);
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
// LDS ring: 128 positions x 4 chain pairs x 16 bytes {u0, u1, seq = position+1, pad}.  A position is
// published with ONE 16-byte DS write per lane by the four lanes r == 0 of the owning wave (after the
// broadcast every lane holds u[j] of its pair); consumers read their pair's 16 bytes and check seq.
//
// A lone wave issues about one instruction per 4 cycles, so the T-step dependent chain is bounded by the
// instruction count of a step; the step bodies below are written to stay near 30 instructions:
//   * cell buffers rotate A/B/C over the four blocks of the band (fully unrolled, no register moves),
//   * the diagonal step needs no lane predicates: the broadcast value is what gets published and the
//     next broadcast simply reads the lanes of the next row,
//   * row jj+1 receives its last term through logaddexp2(Vp, u + W) where Vp (everything but that term)
//     is refreshed one step ahead by the lazily rescaled (M,S) push that runs beside it,
//   * far-field partials are requested one block ahead of the diagonal phase.
constexpr int FAR_PREFETCH = 4;     // parts whose far granules are requested ahead of time

struct SpineBlk { float2 v[PB]; };

template <int MODE, int DIR, bool GRAD>
__device__ __forceinline__ void spine_role(const SweepParams& P, int sg, float* ring, float* dummy)
{
    // kernel arguments are copied into locals: lambdas that capture the struct by reference make the
    // compiler spill it to scratch and reload fields inside the step loops
    const int T = P.T, B = P.B, K = P.K;
    const unsigned dbg = P.dbg;
    unsigned* const ctrl = P.ctrl;
    u64* const ts = P.ts;
    u64* const ug = P.ug;
    const u64* const farg = P.farg;
    float* const u_out = P.u_out;
    float* const last_out = P.last_out;
    int* const code = P.code;
    const float* const vfwd = P.vfwd;
    const float* const logZp = P.logZ;
    const float* const goutp = P.gout;
    float* const dScore = P.dScore;
    float* const dNoise = P.dNoise;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int lane = threadIdx.x & 63;
    const int r = lane >> 2, pr = lane & 3;
    const int c = sg * GS + pr * 2;
    const bool cvalid = c < B;
    const size_t Bs = (size_t)B;
    const float* __restrict__ score = P.score;
    const float* __restrict__ noise = P.noise;
    const unsigned tag = P.tag;
    const long long stride = DIR == 0 ? (long long)B : -(long long)T * B;   // floats per +1 in j
    const bool trace = (dbg & 16u) && sg == 0 && lane == 0;
    if (threadIdx.x == 0) __hip_atomic_store(ctrl + 64 + cu_key(), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const float* rd_base = ring + pr * 4;                                   // + (j & 127) * 16 floats
    float* wr_base = r == 0 ? ring + pr * 4 : dummy + lane * 4;             // writers: the four lanes of row 0
    const int bp_addr = pr << 2;                                            // ds_bpermute byte address of lane pr

    for (int k = wave; k < K; k += RING) {
        u64* ev = ts + T + (size_t)k * 8;          // debug events of this block (8 slots)
        if (trace) ev[0] = __builtin_readcyclecounter();
        const int prow = k * PB + r;
        const bool rvalid = cvalid && prow < T;
        const int frow = frame_of<DIR>(prow < T ? prow : T - 1, T);
        const int own0 = k * PB;
        const float* rowp = score + cell_index<DIR>(rvalid ? prow : 0, 0, T) * Bs + (cvalid ? c : 0);

        // loads are unconditional (addresses clamped into the tensor) so that all 16 are issued back to back;
        // cells with j >= prow are only ever applied AFTER this lane's row has been finalised, and invalid
        // lanes never publish, so what they accumulate is never read.
        auto load_block = [&](int b) -> SpineBlk {
            SpineBlk o;
            const int j0 = b * PB;
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int j = j0 + u < T ? j0 + u : T - 1;
                o.v[u] = *(const float2*)(rowp + (long long)j * stride);
            }
            return o;
        };
        SpineBlk A, Bk, C;
        if (k >= 3) A = load_block(k - 3);
        if (k >= 2) Bk = load_block(k - 2);
        if (k >= 1) C = load_block(k - 1);

        // ---- per-row constants ----------------------------------------------------------------
        float sp[2] = {0.f, 0.f};   // LSE: softplus2(diag); MAX: diag
        float nz[2] = {0.f, 0.f};   // MAX: noise between prow-1 and prow
        float wl[2] = {0.f, 0.f};   // LSE: log2(2^s[prow][prow-1] + 2^n): first sub-diagonal cell with the skip folded in
        if (rvalid) {
            const float2 d = *(const float2*)(score + ((size_t)frow * T + frow) * Bs + c);
            if (MODE == 0) { sp[0] = softplus2(d.x); sp[1] = softplus2(d.y); }
            else { sp[0] = d.x; sp[1] = d.y; }
            if (prow >= 1) {
                const float2 n2 = *(const float2*)(noise + (size_t)gap_of<DIR>(prow, T) * Bs + c);
                nz[0] = n2.x; nz[1] = n2.y;
                if (MODE == 0) {
                    const float2 s1 = *(const float2*)(rowp + (long long)(prow - 1) * stride);
                    const float a0 = s1.x * LOG2E, b0 = n2.x * LOG2E, a1 = s1.y * LOG2E, b1 = n2.y * LOG2E;
                    wl[0] = fmaxf(a0, b0) + flog2(1.0f + fexp2(-fabsf(a0 - b0)));
                    wl[1] = fmaxf(a1, b1) + flog2(1.0f + fexp2(-fabsf(a1 - b1)));
                }
            }
        }
        // GRAD: marginal(prow, j) = gz * exp2(t + arow) with t = u[j] + cell*log2e, arow = (alpha[frame] - logZ)*log2e
        float arow[2] = {0.f, 0.f}, gz[2] = {0.f, 0.f}, draw[2] = {0.f, 0.f};
        float* const growp = GRAD ? dScore + (rowp - score) : nullptr;
        if (GRAD && rvalid) {
            const float2 vv = *(const float2*)(vfwd + (size_t)frow * Bs + c);
            const float2 lz = *(const float2*)(logZp + c);
            const float2 go = *(const float2*)(goutp + c);
            arow[0] = (vv.x - lz.x) * LOG2E; arow[1] = (vv.y - lz.y) * LOG2E;
            gz[0] = go.x; gz[1] = go.y;
            const float2 d = *(const float2*)(score + ((size_t)frow * T + frow) * Bs + c);
            draw[0] = d.x * LOG2E; draw[1] = d.y * LOG2E;
        }
        if (trace) ev[1] = __builtin_readcyclecounter();

        float aM[2] = {SEMICRF_NEG_INF, SEMICRF_NEG_INF}, aS[2] = {0.f, 0.f};
        int aK[2] = {0x7fffffff, 0x7fffffff};

        // lazily rescaled LSE push of (p0,p1) into (aM,aS): one exp per chain, rare wave-uniform slow path
        auto lse_push2 = [&](float p0, float p1) {
            const float d0 = p0 - aM[0], d1 = p1 - aM[1];       // M = -inf -> +inf
            if (__any(fmaxf(d0, d1) > RESCALE_THR)) {
                if (d0 > RESCALE_THR) { aS[0] = aS[0] * fexp2(-d0) + 1.0f; aM[0] = p0; } else aS[0] += fexp2(d0);
                if (d1 > RESCALE_THR) { aS[1] = aS[1] * fexp2(-d1) + 1.0f; aM[1] = p1; } else aS[1] += fexp2(d1);
            } else {
                aS[0] += fexp2(d0);
                aS[1] += fexp2(d1);
            }
        };

        // wait for position j in the ring and return this lane's pair
        auto ring_get = [&](int j) -> float2 {
            const float4* e = (const float4*)(rd_base + (j & 127) * 16);
            float4 v = *e;
            if (!__all(__float_as_int(v.z) == j + 1)) {
                int spins = 0;
                while (true) {
                    __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");           // force a fresh LDS read
                    v = *e;
                    if (__all(__float_as_int(v.z) == j + 1)) break;
                    if (spin_abort(ctrl, spins, SPIN_LIMIT_LDS, 2)) break;
                }
            }
            return make_float2(v.x, v.y);
        };

        // ---------------- shadow phase: apply a block published by a ring mate ---------------------
        // `last`: block k-1, whose final column own0-1 is the first sub-diagonal cell of row 0 (skip folded in)
        auto shadow = [&](const SpineBlk& X, int b, bool last) {
            if (trace) ev[3 + (b - (k - 3))] = __builtin_readcyclecounter();
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                const int j = b * PB + u;
                const float2 uv = ring_get(j);
                if (MODE == 0) {
                    float p0 = fmaf(X.v[u].x, LOG2E, uv.x), p1 = fmaf(X.v[u].y, LOG2E, uv.y);
                    if (GRAD && rvalid && j < prow)
                        *(float2*)(growp + (long long)j * stride) =
                            make_float2(gz[0] * fexp2(p0 + arow[0]), gz[1] * fexp2(p1 + arow[1]));
                    if (last && u == PB - 1 && r == 0) {
                        if (GRAD && rvalid)       // noise marginal of the gap between prow-1 and prow
                            *(float2*)(dNoise + (size_t)gap_of<DIR>(prow, T) * Bs + c) =
                                make_float2(gz[0] * fexp2(uv.x + nz[0] * LOG2E + arow[0]),
                                            gz[1] * fexp2(uv.y + nz[1] * LOG2E + arow[1]));
                        p0 = uv.x + wl[0]; p1 = uv.y + wl[1];
                    }
                    lse_push2(p0, p1);
                } else {
                    const int key = frame_of<DIR>(j, T);
                    if (last && u == PB - 1 && r == 0) {         // the skip candidate goes first (key -1)
                        max_push(aM[0], aK[0], uv.x + nz[0], -1);
                        max_push(aM[1], aK[1], uv.y + nz[1], -1);
                    }
                    max_push(aM[0], aK[0], uv.x + X.v[u].x, key);
                    max_push(aM[1], aK[1], uv.y + X.v[u].y, key);
                }
            }
        };

        if (k >= 3) shadow(A, k - 3, false);
        A = load_block(k);                               // own block: two blocks of lead
        if (k >= 2) shadow(Bk, k - 2, false);

        // far-field partials of this block: request them one block ahead of the diagonal phase
        const int nparts = k >= RING ? (k - RING) / TPT + 1 : 0;
        u64 fg[FAR_PREFETCH][2];
#pragma unroll
        for (int part = 0; part < FAR_PREFETCH; ++part) {
            fg[part][0] = 0; fg[part][1] = 0;
            if (part < nparts && rvalid && !(dbg & 1u)) {
                const u64* fp = farg + ((size_t)part * T + prow) * Bs + c;
                fg[part][0] = load_granule(fp);
                fg[part][1] = load_granule(fp + 1);
            }
        }
        if (k >= 1) shadow(C, k - 1, true);

        // ---------------- diagonal phase: finalise the 16 positions of block k ------------------------
        if (trace) ev[6] = __builtin_readcyclecounter();
        if (nparts > 0 && !(dbg & 1u) && rvalid) {
            for (int part = 0; part < nparts; ++part) {
                const u64* fp = farg + ((size_t)part * T + prow) * Bs + c;
                u64 g0 = 0, g1 = 0;
#pragma unroll
                for (int q = 0; q < FAR_PREFETCH; ++q) if (q == part) { g0 = fg[q][0]; g1 = fg[q][1]; }
                int spins = 0;
                while (true) {
                    bool ok;
                    if (MODE == 0) ok = (unsigned)(g0 >> 32) == tag && (unsigned)(g1 >> 32) == tag;
                    else ok = (unsigned)(g0 >> 48) == (tag & 0xffffu) && (unsigned)(g1 >> 48) == (tag & 0xffffu);
                    if (ok) break;
                    __builtin_amdgcn_s_sleep(2);
                    if (spin_abort(ctrl, spins, SPIN_LIMIT, 3)) break;
                    g0 = load_granule(fp);
                    g1 = load_granule(fp + 1);
                }
                if (MODE == 0) {
                    acc_push1(aM[0], aS[0], __uint_as_float((unsigned)g0));
                    acc_push1(aM[1], aS[1], __uint_as_float((unsigned)g1));
                } else {
                    max_push(aM[0], aK[0], __uint_as_float((unsigned)g0), (int)((g0 >> 32) & 0xffffu));
                    max_push(aM[1], aK[1], __uint_as_float((unsigned)g1), (int)((g1 >> 32) & 0xffffu));
                }
            }
        }

        float* const wr = wr_base + (own0 & 127) * 16;       // ring entry of position own0 (+16 floats per step)
        int mykey[2] = {-1, -1};
        if (MODE == 0) {
            const float W[2] = {wl[0] + sp[0], wl[1] + sp[1]};
            float Vp[2] = {aM[0] + flog2(aS[0]) + sp[0], aM[1] + flog2(aS[1]) + sp[1]};
            float cv[2] = {prow == 0 ? sp[0] : Vp[0], prow == 0 ? sp[1] : Vp[1]};      // value of row 0 (lanes r == 0)
#pragma unroll
            for (int jj = 0; jj < PB; ++jj) {
                const int j = own0 + jj;
                // broadcast u[j] from the lanes of row jj to everybody
                const float u0 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr + (jj << 4), __float_as_int(cv[0])));
                const float u1 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr + (jj << 4), __float_as_int(cv[1])));
                // publish it: data, then sequence number (in-order DS queue)
                *(float4*)(wr + jj * 16) = make_float4(u0, u1, __int_as_float(j + 1), 0.0f);   // one DS write: data + seq
                // critical chain: value of row jj+1 = logaddexp2(Vp, u + W) (only its lanes matter)
                const float t0 = u0 + W[0], t1 = u1 + W[1];
                cv[0] = fmaxf(Vp[0], t0) + flog2(1.0f + fexp2(-fabsf(Vp[0] - t0)));
                cv[1] = fmaxf(Vp[1], t1) + flog2(1.0f + fexp2(-fabsf(Vp[1] - t1)));
                // generic push for the rows further down (branch-free exact-max form: the whole step stays one
                // basic block so that the scheduler can overlap it with the critical chain), then refresh Vp
                {
                    const float p0 = fmaf(A.v[jj].x, LOG2E, u0), p1 = fmaf(A.v[jj].y, LOG2E, u1);
                    if (GRAD) {
                        if (rvalid && r > jj)
                            *(float2*)(growp + (long long)j * stride) =
                                make_float2(gz[0] * fexp2(p0 + arow[0]), gz[1] * fexp2(p1 + arow[1]));
                        if (rvalid && r == jj + 1)
                            *(float2*)(dNoise + (size_t)gap_of<DIR>(prow, T) * Bs + c) =
                                make_float2(gz[0] * fexp2(u0 + nz[0] * LOG2E + arow[0]),
                                            gz[1] * fexp2(u1 + nz[1] * LOG2E + arow[1]));
                    }
                    const float n0 = fmaxf(aM[0], p0), n1 = fmaxf(aM[1], p1);
                    aS[0] = fmaf(aS[0], fexp2(aM[0] - n0), fexp2(p0 - n0));      // M = -inf: exp2(-inf) = 0, S = 0
                    aS[1] = fmaf(aS[1], fexp2(aM[1] - n1), fexp2(p1 - n1));
                    aM[0] = n0; aM[1] = n1;
                }
                Vp[0] = aM[0] + flog2(aS[0]) + sp[0];
                Vp[1] = aM[1] + flog2(aS[1]) + sp[1];
            }
        } else {
            // (max,+): after the push of u[jj] the accumulator of row jj+1 is complete
            float cv[2];
            {
                const float b0 = prow == 0 ? 0.0f : aM[0], b1 = prow == 0 ? 0.0f : aM[1];
                cv[0] = sp[0] > 0.0f ? b0 + sp[0] : b0;
                cv[1] = sp[1] > 0.0f ? b1 + sp[1] : b1;
                if (r == 0) { mykey[0] = prow == 0 ? -1 : aK[0]; mykey[1] = prow == 0 ? -1 : aK[1]; }
            }
#pragma unroll
            for (int jj = 0; jj < PB; ++jj) {
                const int j = own0 + jj;
                const float u0 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr + (jj << 4), __float_as_int(cv[0])));
                const float u1 = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr + (jj << 4), __float_as_int(cv[1])));
                *(float4*)(wr + jj * 16) = make_float4(u0, u1, __int_as_float(j + 1), 0.0f);
                const int key = frame_of<DIR>(j < T ? j : T - 1, T);
                if (r == jj + 1) {                               // the skip candidate goes first (key -1)
                    max_push(aM[0], aK[0], u0 + nz[0], -1);
                    max_push(aM[1], aK[1], u1 + nz[1], -1);
                }
                max_push(aM[0], aK[0], u0 + A.v[jj].x, key);
                max_push(aM[1], aK[1], u1 + A.v[jj].y, key);
                cv[0] = sp[0] > 0.0f ? aM[0] + sp[0] : aM[0];
                cv[1] = sp[1] > 0.0f ? aM[1] + sp[1] : aM[1];
                if (r == jj + 1) { mykey[0] = aK[0]; mykey[1] = aK[1]; }
            }
        }
        if (trace) ev[7] = __builtin_readcyclecounter();

        // ---- once per block: publish the 16 finished positions to HBM -----------------------------
        if (rvalid) {
            const float2 mine = *(const float2*)(rd_base + (prow & 127) * 16);
            store_granule(ug + (size_t)prow * Bs + c, make_granule(tag, mine.x));
            store_granule(ug + (size_t)prow * Bs + c + 1, make_granule(tag, mine.y));
            const float sc = MODE == 0 ? LN2 : 1.0f;
            if (u_out) *(float2*)(u_out + (size_t)frow * Bs + c) = make_float2(mine.x * sc, mine.y * sc);
            if (last_out && prow == T - 1) *(float2*)(last_out + c) = make_float2(mine.x * sc, mine.y * sc);
            if (GRAD)      // diagonal: gout * exp(alpha + beta - logZ + s - 2 softplus(s))
                *(float2*)(dScore + ((size_t)frow * T + frow) * Bs + c) =
                    make_float2(gz[0] * fexp2(arow[0] + mine.x + draw[0] - 2.0f * sp[0]),
                                gz[1] * fexp2(arow[1] + mine.y + draw[1] - 2.0f * sp[1]));
            if (MODE == 1) {
                code[(size_t)c * T + frow] = (mykey[0] + 1) | (sp[0] > 0.0f ? 0x40000000 : 0);
                code[(size_t)(c + 1) * T + frow] = (mykey[1] + 1) | (sp[1] > 0.0f ? 0x40000000 : 0);
            }
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
// Cells and u-granules of tile m+1 are requested before tile m is processed.
template <int MODE, int DIR, bool GRAD>
__device__ __forceinline__ void panel_role(const SweepParams& P, float* lds, int* s_task)
{
    const int T = P.T, B = P.B;
    const unsigned dbg = P.dbg;
    unsigned* const ctrl = P.ctrl;
    u64* const farg = P.farg;
    const int nTasks = P.nTasks, nPanelGroups = P.nPanelGroups;
    const int maxAside = P.nSpine;
    const float* const vfwd = P.vfwd;
    const float* const logZp = P.logZ;
    const float* const goutp = P.gout;
    float* const dScore = P.dScore;
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int slot = lane >> 3, q8 = lane & 7;
    const size_t Bs = (size_t)B;
    const float* __restrict__ score = P.score;
    const unsigned tag = P.tag;
    constexpr int NA = DIR == 0 ? 4 : 2;        // accumulators per chain: rows (DIR 0) or column halves (DIR 1)
    constexpr int NU = DIR == 0 ? 2 : 4;        // u positions a lane needs per tile
    constexpr int NL = DIR == 0 ? 2 : 1;        // u positions a lane LOADS per tile (DIR 1: position slot&3, shared by shuffles)
    const auto ursrc = __builtin_amdgcn_make_buffer_rsrc((void*)P.ug, 0, (int)((size_t)T * Bs * 8), 0x00020000);

    if (dbg & 128u) {
        // power probe: burn ALU cycles for ~300 us without touching memory (is the spine slowed by clocks or by traffic?)
        float a = (float)threadIdx.x, b = 1.0001f;
        for (int i = 0; i < 20000; ++i) { a = fmaf(a, b, 0.5f); b = fmaf(b, 0.9999f, 0.0001f); }
        if (a == 12345.678f) ctrl[5] = 1;
        return;
    }
    while (true) {
        // ---- next task: (k, part, g), ordered so that a task only waits on spine progress below k-3 ----
        __syncthreads();
        if (tid == 0) {
            int t = -1;
            // leave the CU to the spine if one lives here (at most maxAside panels do so, the rest keep working)
            if (!(dbg & 64u) && __hip_atomic_load(ctrl + 64 + cu_key(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u &&
                atomicAdd(ctrl + 3, 1u) < (unsigned)maxAside) t = 0x7fffffff;
            if (t < 0) t = (int)atomicAdd(ctrl + 2, 1u);
            *s_task = t;
        }
        __syncthreads();
        const int task = *s_task;
        if (task >= nTasks) break;
        const int g = task % nPanelGroups;
        int tt = task / nPanelGroups;
        int a = 0;
        while (tt >= TPT * (a + 1) * (a + 2) / 2) ++a;          // group a: blocks with a+1 parts
        tt -= TPT * a * (a + 1) / 2;
        const int q = a * TPT + tt / (a + 1);
        const int part = tt % (a + 1);
        const int k = RING + q;
        const int m0 = part * TPT;
        const int m1 = (m0 + TPT < q + 1) ? m0 + TPT : q + 1;   // tiles m0 .. m1-1 of the q+1 far tiles of block k
        const int c = g * GP + q8 * 4;
        const bool cvalid = c < B;

        float aM[NA][4], aS[NA][4];
        int aK[NA][4];
#pragma unroll
        for (int ai = 0; ai < NA; ++ai)
#pragma unroll
            for (int i = 0; i < 4; ++i) { aM[ai][i] = SEMICRF_NEG_INF; aS[ai][i] = 0.f; aK[ai][i] = 0x7fffffff; }

        // GRAD (DIR 1): marginal(pi, pj) = gz * exp2(t + arow[h]), arow = (alpha[frame(pi)] - logZ) * log2e
        float arow[2][4], gz[4];
        if (GRAD) {
            float lz[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = c + i < B;
                lz[i] = ok ? logZp[c + i] : 0.f;
                gz[i] = ok ? goutp[c + i] : 0.f;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pi = k * PB + slot + 8 * h;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float vv = (c + i < B && pi < T) ? vfwd[(size_t)frame_of<DIR>(pi, T) * Bs + c + i] : 0.f;
                    arow[h][i] = (vv - lz[i]) * LOG2E;
                }
            }
        }
        float4 x[2][4][2];    // [buffer][r][h] cells
        v4u gq[2][NL][2];     // [buffer][loaded u position][chain pair] granules (2 granules per 16-byte load)
        auto pi_of = [&](int rr, int h) { return DIR == 0 ? k * PB + wave * 4 + rr : k * PB + slot + 8 * h; };
        auto pj_of = [&](int m, int rr, int h) { return DIR == 0 ? m * PB + slot + 8 * h : m * PB + wave * 4 + rr; };
        auto pu_of = [&](int m, int ai) { return DIR == 0 ? m * PB + slot + 8 * ai : m * PB + wave * 4 + ai; };

        auto load_tile = [&](auto bufc, int m) {
            constexpr int buf = decltype(bufc)::value;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pi = pi_of(rr, h), pj = pj_of(m, rr, h);
                    const size_t ci = cell_index<DIR>(pi < T ? pi : T - 1, pj, T);
                    x[buf][rr][h] = *(const float4*)(score + ci * Bs + (cvalid ? c : 0));
                }
        };
        auto load_gran = [&](auto bufc, int m) {
            constexpr int buf = decltype(bufc)::value;
            if (dbg & 32u) return;
#pragma unroll
            for (int ai = 0; ai < NL; ++ai) {
                const int pu = DIR == 0 ? pu_of(m, ai) : m * PB + wave * 4 + (slot & 3);
                const int off = (int)(((size_t)pu * Bs + (cvalid ? c : 0)) * 8);
                gq[buf][ai][0] = __builtin_amdgcn_raw_buffer_load_b128(ursrc, off, 0, 16);        // sc1
                gq[buf][ai][1] = __builtin_amdgcn_raw_buffer_load_b128(ursrc, off + 16, 0, 16);
            }
        };

        auto process_tile = [&](auto bufc, int m) {
            constexpr int buf = decltype(bufc)::value;
            if (dbg & 32u) {          // streaming probe: touch the data, nothing else
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        aS[0][0] += x[buf][rr][h].x + x[buf][rr][h].y + x[buf][rr][h].z + x[buf][rr][h].w;
                return;
            }
            // every granule carries its own tag: retry until the spine has published block m
            if (!(dbg & 4u)) {
                int spins = 0;
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int ai = 0; ai < NL; ++ai)
#pragma unroll
                        for (int hh = 0; hh < 2; ++hh)
                            ok = ok && (gq[buf][ai][hh].y == tag || c + 2 * hh >= B) &&
                                 (gq[buf][ai][hh].w == tag || c + 2 * hh + 1 >= B);
                    if (__all(ok || !cvalid)) break;
                    __builtin_amdgcn_s_sleep(16);
                    if (spin_abort(ctrl, spins, SPIN_LIMIT, 5)) break;
                    load_gran(bufc, m);
                }
            }
            float uv[NU][4];
            if (DIR == 0) {
#pragma unroll
                for (int ai = 0; ai < NL; ++ai) {
                    uv[ai][0] = __uint_as_float(gq[buf][ai][0].x); uv[ai][1] = __uint_as_float(gq[buf][ai][0].z);
                    uv[ai][2] = __uint_as_float(gq[buf][ai][1].x); uv[ai][3] = __uint_as_float(gq[buf][ai][1].z);
                }
            } else {
                // position a lives in the lanes with slot == a (and a+4): fetch it from lane (a << 3) | q8
                const float own[4] = {__uint_as_float(gq[buf][0][0].x), __uint_as_float(gq[buf][0][0].z),
                                      __uint_as_float(gq[buf][0][1].x), __uint_as_float(gq[buf][0][1].z)};
#pragma unroll
                for (int ai = 0; ai < NU; ++ai)
#pragma unroll
                    for (int i = 0; i < 4; ++i) uv[ai][i] = __shfl(own[i], (ai << 3) | q8);
            }

            if (MODE == 0) {
                // t = u + cell*log2e; lazily rescaled accumulators: one exp per cell, rare rescale branch
                float t[4][2][4];
                float exc = 0.0f;
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 xv = x[buf][rr][h];
                        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
                        const int ai = DIR == 0 ? rr : h, ui = DIR == 0 ? h : rr;
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            t[rr][h][i] = fmaf(xe[i], LOG2E, uv[ui][i]);
                            exc = fmaxf(exc, t[rr][h][i] - (aM[ai][i] + RESCALE_THR));
                        }
                    }
                if (GRAD) {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const int pi = pi_of(rr, h), pj = pj_of(m, rr, h);
                            if (cvalid && pi < T) {
                                float* dst = dScore + cell_index<DIR>(pi, pj, T) * Bs + c;
                                const float g0 = gz[0] * fexp2(t[rr][h][0] + arow[h][0]), g1 = gz[1] * fexp2(t[rr][h][1] + arow[h][1]);
                                const float g2 = gz[2] * fexp2(t[rr][h][2] + arow[h][2]), g3 = gz[3] * fexp2(t[rr][h][3] + arow[h][3]);
                                if (c + 3 < B) *(float4*)dst = make_float4(g0, g1, g2, g3);
                                else *(float2*)dst = make_float2(g0, g1);            // B % 4 == 2: last pair only
                            }
                        }
                }
                if (__any(exc > 0.0f)) {
                    // some accumulator's reference point is too low (always on the first tile): move it up
#pragma unroll
                    for (int ai = 0; ai < NA; ++ai)
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            float mx = aM[ai][i];
#pragma unroll
                            for (int e = 0; e < (DIR == 0 ? 2 : 4); ++e)
                                mx = fmaxf(mx, DIR == 0 ? t[ai][e][i] : t[e][ai][i]);
                            if (mx > aM[ai][i] + RESCALE_THR) {
                                aS[ai][i] = aS[ai][i] * fexp2(aM[ai][i] - mx);     // -inf - mx -> exp2 = 0, S = 0
                                aM[ai][i] = mx;
                            }
                        }
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int ai = DIR == 0 ? rr : h;
#pragma unroll
                        for (int i = 0; i < 4; ++i) aS[ai][i] += fexp2(t[rr][h][i] - aM[ai][i]);
                    }
            } else {
#pragma unroll
                for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 xv = x[buf][rr][h];
                        const float xe[4] = {xv.x, xv.y, xv.z, xv.w};
                        const int ai = DIR == 0 ? rr : h, ui = DIR == 0 ? h : rr;
                        const int key = frame_of<DIR>(pj_of(m, rr, h), T);
#pragma unroll
                        for (int i = 0; i < 4; ++i) max_push(aM[ai][i], aK[ai][i], uv[ui][i] + xe[i], key);
                    }
            }
        };

        {
            int m = m0;
            load_tile(IC<0>{}, m);
            load_gran(IC<0>{}, m);
            while (m < m1) {
                if (m + 1 < m1) { load_tile(IC<1>{}, m + 1); load_gran(IC<1>{}, m + 1); }
                process_tile(IC<0>{}, m);
                ++m;
                if (m >= m1) break;
                if (m + 1 < m1) { load_tile(IC<0>{}, m + 1); load_gran(IC<0>{}, m + 1); }
                process_tile(IC<1>{}, m);
                ++m;
            }
        }

        // ---- reduce the partials and hand them to the spine ------------------------------------------
        u64* fbase = farg + (size_t)part * T * Bs;
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
                        if (c + i >= B) continue;
                        u64 gr;
                        if (MODE == 0) gr = make_granule(tag, aM[rr][i] + flog2(aS[rr][i]));
                        else gr = ((u64)(((tag & 0xffffu) << 16) | ((unsigned)aK[rr][i] & 0xffffu)) << 32) |
                                  (u64)__float_as_uint(aM[rr][i]);
                        store_granule(fbase + (size_t)pi * Bs + c + i, gr);
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
                    store_granule(fbase + (size_t)pi * Bs + cc, gr);
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
template <int MODE, int DIR, bool GRAD>
__global__ __launch_bounds__(256, 2) void persist_sweep_kernel(SweepParams P)
{
    __shared__ __attribute__((aligned(16))) float s_ring[128 * 16];            // spine: ring of the last 128 published positions
    __shared__ __attribute__((aligned(16))) float s_dummy[64 * 4 + 16 * 16];   // spine: sink of the non-writer lanes' stores
    __shared__ float s_red[2 * 4 * 2 * 8 * 32];   // panel DIR 1 reduction
    __shared__ int s_ticket;
    __shared__ int s_task;
    if (threadIdx.x == 0) s_ticket = (int)atomicAdd(P.ctrl, 1u);
    for (int i = threadIdx.x; i < 128 * 16; i += 256) s_ring[i] = 0.0f;      // sequence numbers start at 0
    __syncthreads();
    const int ticket = s_ticket;
    if (ticket < P.nSpine) {
        if (!(P.dbg & 8u)) spine_role<MODE, DIR, GRAD>(P, ticket, s_ring, s_dummy);
    } else {
        if (!(P.dbg & 2u)) panel_role<MODE, DIR, GRAD>(P, s_red, &s_task);
    }
}

constexpr size_t CTRL_BYTES = (64 + 4096) * sizeof(unsigned);

static int max_parts(int T)
{
    const int K = (T + PB - 1) / PB;
    return K > RING ? (K - 1 - RING) / TPT + 1 : 1;
}

// writes the exact zeros of the upper triangle (begin > end) of the dense gradient: row e, columns e+1..T-1
__global__ __launch_bounds__(256) void zero_upper_kernel(float* __restrict__ dScore, int T, int B)
{
    const int e = blockIdx.y;
    const size_t n = (size_t)(T - 1 - e) * B;                 // floats to clear in this row
    float* rowp = dScore + ((size_t)e * T + e + 1) * B;
    const size_t n4 = ((uintptr_t)rowp & 15) == 0 ? n / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        ((float4*)rowp)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) rowp[i] = 0.f;
}

size_t persist_workspace_bytes(int T, int B)
{
    return CTRL_BYTES + align_up((size_t)2 * T * sizeof(u64)) +
           (size_t)(1 + max_parts(T)) * align_up((size_t)T * B * sizeof(u64));
}

// B even: the spine reads chain pairs; a panel lane reads 4 chains and masks the ones past B (B % 4 == 2: the
// 16-byte loads are then only 8-byte aligned, which global memory accepts)
bool persist_supported(int T, int B) { return (B % 2 == 0) && T >= 1 && T < 65535 && (long long)T * B * 8 < (1ll << 31); }

static unsigned next_tag()
{
    static std::atomic<unsigned> counter{0};
    const unsigned lo = (counter.fetch_add(1) % 65535u) + 1u;   // 1..65535
    return (lo << 16) | lo;                                      // both 16-bit halves nonzero
}

// mode 0 = LSE, 1 = MAX.  ws must hold persist_workspace_bytes().  Enqueues a memset + one kernel.
struct GradArgs {
    const float* vfwd; const float* logZ; const float* gout; float* dScore; float* dNoise;
};

static int launch_persist_sweep_impl(int mode, int dir, const float* score, const float* noise, int T, int B,
                                     float* u_out, float* last_out, int* code, void* ws, hipStream_t stream,
                                     const GradArgs* grad)
{
    SweepParams P;
    P.vfwd = nullptr; P.logZ = nullptr; P.gout = nullptr; P.dScore = nullptr; P.dNoise = nullptr;
    if (grad) { P.vfwd = grad->vfwd; P.logZ = grad->logZ; P.gout = grad->gout; P.dScore = grad->dScore; P.dNoise = grad->dNoise; }
    P.score = score; P.noise = noise; P.T = T; P.B = B; P.K = (T + PB - 1) / PB;
    P.nSpine = (B + GS - 1) / GS;
    P.nPanelGroups = (B + GP - 1) / GP;
    P.tag = next_tag();
    const char* dbg = getenv("SEMICRF_DEBUG_FLAGS");
    P.dbg = dbg ? (unsigned)atoi(dbg) : 0u;
    char* w = (char*)ws;
    P.ctrl = (unsigned*)w;
    P.ts = (u64*)(w + CTRL_BYTES);
    const size_t ts_bytes = align_up((size_t)2 * T * sizeof(u64));
    P.ug = (u64*)(w + CTRL_BYTES + ts_bytes);
    P.farg = (u64*)(w + CTRL_BYTES + ts_bytes + align_up((size_t)T * B * sizeof(u64)));
    P.u_out = u_out; P.last_out = last_out; P.code = code;
    const size_t zbytes = persist_workspace_bytes(T, B);
    if (hipMemsetAsync(ws, 0, zbytes, stream) != hipSuccess) return 1;
    // panel tasks: block k = RING + q has q/TPT + 1 column parts
    long long ntask = 0;
    for (int q = 0; q < P.K - RING; ++q) ntask += (q / TPT + 1);
    P.nTasks = (int)(ntask * P.nPanelGroups);
    // persistent panel workgroups: fill the chip at the kernel's occupancy (2 workgroups per CU)
    int dev = 0, ncu = 256;
    if (hipGetDevice(&dev) == hipSuccess) {
        int v = 0;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    }
    // Panel workgroups.  More of them means more loads in flight than the ~8 MB that saturate HBM, i.e. only
    // longer queues -- and the spine's band loads and hand-offs wait in the same queues.  Measured at T=1024,
    // NBatch=352: forward 292 us with 1 panel workgroup per CU vs 336 us with 2; the gradient sweep (which also
    // stores a tile per tile loaded) is best around 1.25 per CU.
    // The far field grows with T^2 and the spine's chain with T, so longer sequences get more panel workgroups.
    float per_cu = (float)T / 1024.0f;
    per_cu = per_cu < 0.75f ? 0.75f : (per_cu > 2.0f ? 2.0f : per_cu);
    if (grad) per_cu *= 1.25f;
    int nPanelWG = (int)(per_cu * ncu);
    if (nPanelWG > 2 * ncu - P.nSpine) nPanelWG = 2 * ncu - P.nSpine;
    if (const char* e = getenv("SEMICRF_PANEL_WGS")) { const int v = atoi(e); if (v > 0) nPanelWG = v; }   // tuning knob
    if (nPanelWG < ncu / 2) nPanelWG = ncu / 2;
    if (nPanelWG > P.nTasks) nPanelWG = P.nTasks;
    const int grid = P.nSpine + nPanelWG;
    dim3 g(grid), b(256);
    if (grad) {
        if (T > 1) {
            int gx = (int)(((size_t)T * B / 4 + 255) / 256);
            if (gx > 8) gx = 8;
            hipLaunchKernelGGL(zero_upper_kernel, dim3(gx, T - 1), dim3(256), 0, stream, grad->dScore, T, B);
        }
        hipLaunchKernelGGL((persist_sweep_kernel<0, 1, true>), g, b, 0, stream, P);
    } else if (mode == 0 && dir == 0) hipLaunchKernelGGL((persist_sweep_kernel<0, 0, false>), g, b, 0, stream, P);
    else if (mode == 0 && dir == 1) hipLaunchKernelGGL((persist_sweep_kernel<0, 1, false>), g, b, 0, stream, P);
    else if (mode == 1 && dir == 0) hipLaunchKernelGGL((persist_sweep_kernel<1, 0, false>), g, b, 0, stream, P);
    else hipLaunchKernelGGL((persist_sweep_kernel<1, 1, false>), g, b, 0, stream, P);
    return 0;
}

int launch_persist_sweep(int mode, int dir, const float* score, const float* noise, int T, int B, float* u_out,
                         float* last_out, int* code, void* ws, hipStream_t stream)
{
    return launch_persist_sweep_impl(mode, dir, score, noise, T, B, u_out, last_out, code, ws, stream, nullptr);
}

// Fused backward: beta sweep + marginals (dScore fully written incl. the zero upper triangle, dNoise).
int launch_persist_logz_bwd(const float* score, const float* noise, const float* v, const float* logZ,
                            const float* gout, int T, int B, float* dScore, float* dNoise, float* q_out, void* ws,
                            hipStream_t stream)
{
    GradArgs ga{v, logZ, gout, dScore, dNoise};
    return launch_persist_sweep_impl(0, 1, score, noise, T, B, q_out, nullptr, nullptr, ws, stream, &ga);
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
