// common.h -- shared helpers for the gfx950 semi-CRF kernels (internal; the public ABI is include/semicrf_hip.h)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>
#include <stdio.h>
#include <math.h>
#include "../../include/semicrf_hip.h"

namespace semicrf {

constexpr int WAVE = 64;
#define SEMICRF_NEG_INF (-__builtin_huge_valf())

void set_error(const char* fmt, ...);

#define SEMICRF_CHECK_ARG(cond, ...)                      \
    do {                                                  \
        if (!(cond)) {                                    \
            ::semicrf::set_error(__VA_ARGS__);            \
            return SEMICRF_EINVAL;                        \
        }                                                 \
    } while (0)

#define SEMICRF_CHECK_LAUNCH(what)                                                            \
    do {                                                                                      \
        hipError_t e__ = hipGetLastError();                                                   \
        if (e__ != hipSuccess) {                                                              \
            ::semicrf::set_error("%s: %s", what, hipGetErrorString(e__));                     \
            return SEMICRF_ELAUNCH;                                                           \
        }                                                                                     \
    } while (0)

static inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

// "Once per DEVICE" latch for per-device settings such as hipFuncSetAttribute (a process may drive several GPUs).
struct PerDeviceOnce {
    std::atomic<bool> done[64];
    bool first()
    {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
        if (done[dev].load()) return false;
        done[dev].store(true);
        return true;
    }
};

// ---- slot layout of the chain axis (interval_score_*_p, include/semicrf_hip.h) ---------------------------------
// The C chains come in groups of `group` (the symbols of one segment); every group owns `pitch` >= group slots of the
// chain axis of S / dS / alpha / beta / logZ / gout / interval offsets; q, k, diag and their gradients are indexed by chain.
// group == pitch: slots and chains coincide.
struct ChainSlots { int group, pitch; };
__host__ __device__ inline int slot_of_chain(ChainSlots s, int c) { return (c / s.group) * s.pitch + c % s.group; }
__host__ __device__ inline int chain_of_slot(ChainSlots s, int slot)      // -1: a ghost slot
{
    const int p = slot % s.pitch;
    return p < s.group ? (slot / s.pitch) * s.group + p : -1;
}

// ---- device math ---------------------------------------------------------------------------
// F.softplus(beta=1, threshold=20): NeuralSemiCRFInterval.py:218,232,395,427
__device__ __forceinline__ float softplus_f(float x) { return x > 20.0f ? x : log1pf(expf(x)); }

// online log-sum-exp accumulator update in natural-log domain: (M,S) <- (M,S) (+) t
__device__ __forceinline__ void lse_push(float& M, float& S, float t)
{
    float nm = fmaxf(M, t);
    S = S * expf(M - nm) + expf(t - nm);
    M = nm;
}
// merge two accumulators
__device__ __forceinline__ void lse_merge(float& M, float& S, float M2, float S2)
{
    float nm = fmaxf(M, M2);
    // (-inf) - (-inf) would be NaN: an empty accumulator has S == 0, guard it
    float a = (S == 0.0f) ? 0.0f : S * expf(M - nm);
    float b = (S2 == 0.0f) ? 0.0f : S2 * expf(M2 - nm);
    S = a + b;
    M = nm;
}

// Viterbi candidate merge: larger value wins; on ties the smaller key (candidate order) wins.
__device__ __forceinline__ void max_push(float& best, int& key, float t, int k)
{
    if (t > best || (t == best && k < key)) { best = t; key = k; }
}

}  // namespace semicrf
