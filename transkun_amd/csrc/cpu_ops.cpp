// cpu_ops.cpp -- the semi-CRF entry points for CPU tensors (dispatch key CPU of torch.ops.semicrf.*).
//
// The reference class runs wherever its tensors live (NeuralSemiCRFInterval.py:553-588; crfMinimalExample.py:28-38 is
// BASELINE config #1, "plumbing, no GPU").  These are the product's own host kernels for that case: plain C++, written
// for this layout ([T][T][B], chain axis contiguous: every inner loop runs over chains), OpenMP over chain blocks.  They
// are NOT a fallback: a GPU tensor never reaches them (the dispatcher selects by device), and nothing here includes or
// links the test suite's CPU checker (tests/test_abi.py::test_product_never_imports_oracle).
//
// Arithmetic: fp32 values as in the reference, sums of exponentials accumulated in double; decode is exact: one fp32 add
// per candidate, first maximum in the reference's candidate order.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "cpu_ops.h"

namespace semicrf_cpu {

namespace {
constexpr int CB = 16;                          // chains per block: one 64-byte line of every cell (a thread owns whole lines of what it writes)

inline float softplus(float x) { return x > 20.0f ? x : log1pf(expf(x)); }      // F.softplus, threshold 20 (reference :218,:395)
inline float relu_sel(float x) { return x > 0.0f ? x : 0.0f; }                   // s * (s > 0), reference :29, :49-51

struct Lse {                                    // running log-sum-exp with exact maximum
    float m;
    double s;
    inline void push(float x)
    {
        if (x == -INFINITY) return;             // a masked cell adds nothing (and -inf - -inf would poison the sum with NaN)
        if (x <= m) s += (double)expf(x - m);
        else { s = s * (double)expf(m - x) + 1.0; m = x; }
    }
    inline float value() const { return s == 0.0 ? -INFINITY : m + (float)log(s); }   // nothing pushed: logsumexp of the empty set
};
}  // namespace

// alpha sweep (computeLogZ :207-246; forward_backward :394-410, :417): v [T][B], logZ [B]
void logz_fwd(const float* score, const float* noise, int T, int B, float* logZ, float* v)
{
    // Row by row with the chain blocks spread over the threads INSIDE a row: all threads stream the same 4 B T bytes of row i at
    // the same time.  (A thread per chain block walking the whole tensor on its own -- 64 bytes out of every 4 B -- touched a new
    // page every third cell: 4.9 s per step at T=1024 x 352 on 16 threads where the torch op loop takes 2.3.)
    const size_t Bs = (size_t)B;
    const int nblk = (B + CB - 1) / CB;
#pragma omp parallel
    for (int i = 0; i < T; ++i) {
        const float* row = score + (size_t)i * T * Bs;
#pragma omp for schedule(static) nowait
        for (int blk = 0; blk < nblk; ++blk) {
            const int c0 = blk * CB;
            const int nc = B - c0 < CB ? B - c0 : CB;
            float* vi = v + (size_t)i * Bs + c0;
            if (i == 0) {
                for (int c = 0; c < nc; ++c) vi[c] = softplus(row[c0 + c]);
            } else {
                const float* vp = v + (size_t)(i - 1) * Bs + c0;
                const float* nz = noise + (size_t)(i - 1) * Bs + c0;
                float m[CB];
                double s[CB];
                for (int c = 0; c < nc; ++c) m[c] = vp[c] + nz[c];
                for (int j = 0; j < i; ++j) {
                    const float* vj = v + (size_t)j * Bs + c0;
                    const float* cell = row + (size_t)j * Bs + c0;
                    for (int c = 0; c < nc; ++c) { const float x = vj[c] + cell[c]; m[c] = x > m[c] ? x : m[c]; }
                }
                // every candidate -inf (masked cells, torch.logsumexp gives -inf): the reference point is 0, every term exp(-inf) = 0
                float mm[CB];
                for (int c = 0; c < nc; ++c) mm[c] = m[c] == -INFINITY ? 0.0f : m[c];
                for (int c = 0; c < nc; ++c) s[c] = (double)expf(vp[c] + nz[c] - mm[c]);
                for (int j = 0; j < i; ++j) {
                    const float* vj = v + (size_t)j * Bs + c0;
                    const float* cell = row + (size_t)j * Bs + c0;
                    for (int c = 0; c < nc; ++c) s[c] += (double)expf(vj[c] + cell[c] - mm[c]);
                }
                const float* dg = row + (size_t)i * Bs + c0;
                for (int c = 0; c < nc; ++c) vi[c] = (s[c] == 0.0 ? -INFINITY : mm[c] + (float)log(s[c])) + softplus(dg[c]);
            }
            if (i == T - 1)
                for (int c = 0; c < nc; ++c) logZ[c0 + c] = vi[c];
        }                                               // (no barrier: the blocks are independent; a static schedule keeps a thread on ITS blocks, and threads with
                                                        // equal work stay within a few rows of each other -- a barrier per row cost seconds in the autograd thread)
    }
}

// beta sweep by frames (the flipped half of forward_backward :386-414) and, when dScore is given, the marginals
// (:424-447) times gout (:469-472).  Row e is visited once, right after q[e] is final: its cells are pushed into the
// accumulators of the frames t < e and turned into marginals in the same pass.
void logz_bwd(const float* score, const float* noise, const float* v, const float* logZ, const float* gout, int T, int B,
              float* dScore, float* dNoise, float* q)
{
    const size_t Bs = (size_t)B;
    const int nblk = (B + CB - 1) / CB;
    std::vector<Lse> acc_all((size_t)nblk * T * CB, Lse{-INFINITY, 0.0});      // per chain block: the accumulators of the frames t < e
#pragma omp parallel
    for (int e = T - 1; e >= 0; --e) {                    // row by row, the chain blocks spread over the threads inside a row (see logz_fwd)
        const float* row = score + (size_t)e * T * Bs;
#pragma omp for schedule(static) nowait
        for (int blk = 0; blk < nblk; ++blk) {
            const int c0 = blk * CB;
            const int nc = B - c0 < CB ? B - c0 : CB;
            Lse* const acc = acc_all.data() + (size_t)blk * T * CB;
            const float* dg = row + (size_t)e * Bs + c0;
            float* qe = q + (size_t)e * Bs + c0;
            if (e == T - 1) {
                for (int c = 0; c < nc; ++c) qe[c] = softplus(dg[c]);
            } else {
                const float* qn = q + (size_t)(e + 1) * Bs + c0;
                const float* nz = noise + (size_t)e * Bs + c0;
                for (int c = 0; c < nc; ++c) {
                    Lse a = acc[(size_t)e * CB + c];
                    a.push(qn[c] + nz[c]);
                    qe[c] = a.value() + softplus(dg[c]);
                }
                if (dNoise) {
                    const float* ve = v + (size_t)e * Bs + c0;
                    for (int c = 0; c < nc; ++c)
                        dNoise[(size_t)e * Bs + c0 + c] = gout[c0 + c] * expf(ve[c] + qn[c] + nz[c] - logZ[c0 + c]);
                }
            }
            float* drow = dScore ? dScore + (size_t)e * T * Bs : nullptr;
            for (int t = 0; t < e; ++t) {
                const float* cell = row + (size_t)t * Bs + c0;
                Lse* a = &acc[(size_t)t * CB];
                for (int c = 0; c < nc; ++c) a[c].push(qe[c] + cell[c]);
                if (drow) {
                    const float* vt = v + (size_t)t * Bs + c0;
                    float* d = drow + (size_t)t * Bs + c0;
                    for (int c = 0; c < nc; ++c) d[c] = gout[c0 + c] * expf(vt[c] + qe[c] + cell[c] - logZ[c0 + c]);
                }
            }
            if (drow) {
                const float* ve = v + (size_t)e * Bs + c0;
                float* d = drow + (size_t)e * Bs + c0;
                for (int c = 0; c < nc; ++c)
                    d[c] = gout[c0 + c] * expf(ve[c] + qe[c] + dg[c] - 2.0f * softplus(dg[c]) - logZ[c0 + c]);
                for (int t = e + 1; t < T; ++t) memset(drow + (size_t)t * Bs + c0, 0, (size_t)nc * sizeof(float));   // begin > end: exact zeros
            }
        }
    }
}

// Viterbi + backtrack (viterbiBackward :13-104, forward == 0; viterbi :107-202, forward == 1).  Candidates are single fp32
// adds; ties: skip first, then the nearest frame (forward == 0: smallest end; forward == 1: smallest begin) -- the first
// maximum of the reference's concatenation.  pairs [cap][2] chain-major, offsets [B+1].
void viterbi(const float* score, const float* noise, int T, int B, const int32_t* start, int forward, int32_t* pairs, int64_t cap,
             int32_t* offsets)
{
    const size_t Bs = (size_t)B;
    std::vector<int32_t> ptr((size_t)T * B, -1);           // [frame][chain]: -1 skip, else the partner frame
    std::vector<float> u((size_t)T * B);
    if (forward) {
#pragma omp parallel for schedule(dynamic, 1)
        for (int c0 = 0; c0 < B; c0 += CB) {
            const int nc = B - c0 < CB ? B - c0 : CB;
            for (int i = 0; i < T; ++i) {
                const float* row = score + (size_t)i * T * Bs;
                const float* dg = row + (size_t)i * Bs + c0;
                float* ui = &u[(size_t)i * Bs + c0];
                if (i == 0) { for (int c = 0; c < nc; ++c) ui[c] = relu_sel(dg[c]); continue; }
                float m[CB];
                int32_t a[CB];
                for (int c = 0; c < nc; ++c) { m[c] = u[(size_t)(i - 1) * Bs + c0 + c] + noise[(size_t)(i - 1) * Bs + c0 + c]; a[c] = -1; }
                for (int j = 0; j < i; ++j) {
                    const float* uj = &u[(size_t)j * Bs + c0];
                    const float* cell = row + (size_t)j * Bs + c0;
                    for (int c = 0; c < nc; ++c) { const float x = uj[c] + cell[c]; if (x > m[c]) { m[c] = x; a[c] = j; } }
                }
                for (int c = 0; c < nc; ++c) { ui[c] = m[c] + relu_sel(dg[c]); ptr[(size_t)i * Bs + c0 + c] = a[c]; }
            }
        }
    } else {
#pragma omp parallel for schedule(dynamic, 1)
        for (int c0 = 0; c0 < B; c0 += CB) {
            const int nc = B - c0 < CB ? B - c0 : CB;
            std::vector<float> m((size_t)T * CB, -INFINITY);
            std::vector<int32_t> a((size_t)T * CB, -1);
            for (int e = T - 1; e >= 0; --e) {
                const float* row = score + (size_t)e * T * Bs;
                const float* dg = row + (size_t)e * Bs + c0;
                float* ue = &u[(size_t)e * Bs + c0];
                if (e == T - 1) {
                    for (int c = 0; c < nc; ++c) ue[c] = relu_sel(dg[c]);
                } else {
                    for (int c = 0; c < nc; ++c) {
                        // ends arrived farthest first: a later equal candidate (nearer end) won with >=; the skip goes in front of all
                        const float xs = u[(size_t)(e + 1) * Bs + c0 + c] + noise[(size_t)e * Bs + c0 + c];
                        float mm = m[(size_t)e * CB + c];
                        int32_t aa = a[(size_t)e * CB + c];
                        if (xs >= mm) { mm = xs; aa = -1; }
                        ue[c] = mm + relu_sel(dg[c]);
                        ptr[(size_t)e * Bs + c0 + c] = aa;
                    }
                }
                for (int t = 0; t < e; ++t) {
                    const float* cell = row + (size_t)t * Bs + c0;
                    float* mt = &m[(size_t)t * CB];
                    int32_t* at = &a[(size_t)t * CB];
                    for (int c = 0; c < nc; ++c) { const float x = ue[c] + cell[c]; if (x >= mt[c]) { mt[c] = x; at[c] = e; } }
                }
            }
        }
    }
    // backtrack, chain by chain (:61-102, :161-196)
    std::vector<std::vector<int32_t>> out((size_t)B);
#pragma omp parallel for schedule(dynamic, 8)
    for (int c = 0; c < B; ++c) {
        std::vector<int32_t>& o = out[(size_t)c];
        auto diag = [&](int t) { return score[((size_t)t * T + t) * Bs + c] > 0.0f; };
        if (!forward) {
            int j = start ? start[c] : 0;
            while (j < T - 1) {
                if (diag(j)) { o.push_back(j); o.push_back(j); }
                const int32_t p = ptr[(size_t)j * Bs + c];
                if (p < 0) ++j;
                else { o.push_back(j); o.push_back(p); j = p; }
            }
            if (diag(T - 1)) { o.push_back(T - 1); o.push_back(T - 1); }
        } else {
            int j = start ? start[c] : T - 1;
            std::vector<int32_t> rev;
            while (j > 0) {
                if (diag(j)) { rev.push_back(j); rev.push_back(j); }
                const int32_t p = ptr[(size_t)j * Bs + c];
                if (p < 0) --j;
                else { rev.push_back(p); rev.push_back(j); j = p; }
            }
            if (diag(0)) { rev.push_back(0); rev.push_back(0); }
            for (size_t i = rev.size(); i >= 2; i -= 2) { o.push_back(rev[i - 2]); o.push_back(rev[i - 1]); }
        }
    }
    int64_t k = 0;
    for (int c = 0; c < B; ++c) {
        offsets[c] = (int32_t)k;
        const std::vector<int32_t>& o = out[(size_t)c];
        for (size_t i = 0; i + 1 < o.size(); i += 2, ++k)
            if (k < cap) { pairs[2 * k] = o[i]; pairs[2 * k + 1] = o[i + 1]; }
    }
    offsets[B] = (int32_t)k;
}

// evalPath (:508-550): sum of the path's interval scores plus the noise of every gap no interval covers
void eval_path(const float* score, const float* noise, int T, int B, const int32_t* pairs, const int32_t* offsets, float* out)
{
    const size_t Bs = (size_t)B;
#pragma omp parallel for schedule(dynamic, 8)
    for (int c = 0; c < B; ++c) {
        std::vector<double> cum((size_t)T, 0.0);
        for (int t = 1; t < T; ++t) cum[(size_t)t] = cum[(size_t)t - 1] + (double)noise[(size_t)(t - 1) * Bs + c];
        double acc = T > 0 ? cum[(size_t)T - 1] : 0.0;
        for (int i = offsets[c]; i < offsets[c + 1]; ++i) {
            const int b = pairs[2 * i], e = pairs[2 * i + 1];
            acc += (double)score[((size_t)e * T + b) * Bs + c] - (cum[(size_t)e] - cum[(size_t)b]);
        }
        out[c] = (float)acc;
    }
}

// gradient of sum_c gout[c] evalPath[c], ADDED to dScore / dNoise (either may be null)
void eval_path_bwd(const float* gout, int T, int B, const int32_t* pairs, const int32_t* offsets, float* dScore, float* dNoise)
{
    const size_t Bs = (size_t)B;
#pragma omp parallel for schedule(dynamic, 8)
    for (int c = 0; c < B; ++c) {
        // the derivative of eval_path above, which is LINEAR in the noise: every gap counts once (cum[T-1]) and once less per
        // interval that covers it -- also for overlapping intervals, where a 0/1 "covered" flag is not the derivative of the
        // forward (the reference's gathers :540-548 and evalpath.hip's eval_path_bwd_pairs_kernel agree)
        if (dNoise)
            for (int t = 0; t + 1 < T; ++t) dNoise[(size_t)t * Bs + c] += gout[c];
        for (int i = offsets[c]; i < offsets[c + 1]; ++i) {
            const int b = pairs[2 * i], e = pairs[2 * i + 1];
            if (dScore) dScore[((size_t)e * T + b) * Bs + c] += gout[c];
            if (dNoise)
                for (int t = b; t < e; ++t) dNoise[(size_t)t * Bs + c] -= gout[c];
        }
    }
}

}  // namespace semicrf_cpu
