/*
 * semicrf_oracle.c -- TEST INFRASTRUCTURE ONLY (not the product path).
 *
 * Plain-C, scalar, single-threaded CPU restatement of the algorithms of the
 * reference's Neural Semi-CRF interval layer and interval scorer.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library; the product (transkun_amd) never does and fails loudly when
 * the HIP library is missing.
 *
 * Parity status: PINNED.  tools/make_golden.py imports the reference itself
 * (/root/reference/transkun/CRF, torch CPU) in the build container and writes
 * tests/golden/ (npz files); tests/test_oracle_golden.py checks every function here
 * against those vectors (decode: identical lists; fp32: <= 2e-6 relative).
 *
 * Reference citations are to /root/reference/transkun/...
 *   CRF/NeuralSemiCRFInterval.py   (NSCI below)
 *   LayersTransformer.py           (LT below)
 *
 * Layouts (all fp32, C-contiguous):
 *   score [T][T][B]   indexed [end][begin][chain]; only end >= begin is read
 *   noise [T-1][B]    score of "no event between frame t and t+1"
 *
 * Arithmetic is fp32 with the reference's operation order wherever the order
 * is observable (Viterbi candidates are single fp32 adds; see NSCI:38-39,49-51).
 * Build with -ffp-contract=off (see Makefile) so no FMA contraction happens.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define S_(e, b, c) score[((size_t)(e) * (size_t)T + (size_t)(b)) * (size_t)B + (size_t)(c)]
#define N_(t, c) noise[(size_t)(t) * (size_t)B + (size_t)(c)]

/* ---- floating-point sweeps: instantiated twice ------------------------------------------
 * semicrf_oracle_fp.inc is the restatement proper.  REAL = float gives the reference's fp32
 * arithmetic (exported names oracle_alpha, ...); REAL = double gives a higher-precision truth
 * (names ..._f64) that tests use to tell kernel bugs from fp32 round-off noise. */
#define REAL float
#define FN(name) name
#define M_EXP expf
#define M_LOG logf
#define M_LOG1P log1pf
#define M_FABS fabsf
#include "semicrf_oracle_fp.inc"
#undef REAL
#undef FN
#undef M_EXP
#undef M_LOG
#undef M_LOG1P
#undef M_FABS
#define REAL double
#define FN(name) name##_f64
#define M_EXP exp
#define M_LOG log
#define M_LOG1P log1p
#define M_FABS fabs
#include "semicrf_oracle_fp.inc"
#undef REAL
#undef FN

/* relu-select of the singleton score: s * (s > 0)  (NSCI:29,49-51,122,142-144). */
static float singleton_term(float s) { return s > 0.0f ? s : s * 0.0f; }

/*
 * viterbiBackward (NSCI:13-104), the default decode.
 * start: NULL (=> 0 for every chain, NSCI:61-62) or B ints.
 * Output: pairs (begin,end) appended per chain into pairs[2*k], offsets[B+1] (prefix counts).
 * Returns the total number of intervals, or -1 if cap (in intervals) would be exceeded.
 * T == 1: the reference raises (torch.stack of an empty list); here the obvious answer.
 */
long oracle_viterbi_backward(const float* score, const float* noise, int T, int B,
                             const int* start, int* pairs, long cap, long* offsets)
{
    float* q = (float*)malloc(sizeof(float) * (size_t)T);
    int* ptr = (int*)malloc(sizeof(int) * (size_t)(T > 1 ? T - 1 : 1));
    long total = 0;
    for (int c = 0; c < B; ++c) {
        offsets[c] = total;
        q[T - 1] = singleton_term(S_(T - 1, T - 1, c));
        for (int t = T - 2; t >= 0; --t) {
            /* candidates in the order [skip, e=t+1, t+2, ...]; first maximum wins (NSCI:36-46) */
            float best = q[t + 1] + N_(t, c);
            int sel = -1;
            for (int e = t + 1; e < T; ++e) {
                float cand = q[e] + S_(e, t, c);
                if (cand > best) { best = cand; sel = e - (t + 1); }
            }
            ptr[t] = sel; /* reference stores ptr[T-t-2] = sel (NSCI:46,79) */
            q[t] = best + singleton_term(S_(t, t, c));
        }
        int j = start ? start[c] : 0;
        while (j < T - 1) { /* NSCI:77-94 */
            int sel = ptr[j];
            if (S_(j, j, c) > 0.0f) {
                if (total >= cap) goto overflow;
                pairs[2 * total] = j; pairs[2 * total + 1] = j; ++total;
            }
            if (sel < 0) j += 1;
            else {
                int i = sel + j + 1;
                if (total >= cap) goto overflow;
                pairs[2 * total] = j; pairs[2 * total + 1] = i; ++total;
                j = i;
            }
        }
        if (S_(T - 1, T - 1, c) > 0.0f) { /* NSCI:97-98: regardless of start */
            if (total >= cap) goto overflow;
            pairs[2 * total] = T - 1; pairs[2 * total + 1] = T - 1; ++total;
        }
    }
    offsets[B] = total;
    free(q); free(ptr);
    return total;
overflow:
    free(q); free(ptr);
    return -1;
}

/*
 * viterbi (NSCI:107-202), the forward=True variant.  start: NULL (=> T-1) or B ints
 * (meaning the END position to backtrack from).  The per-chain list is reversed to
 * ascending order (NSCI:196).
 */
long oracle_viterbi_forward(const float* score, const float* noise, int T, int B,
                            const int* start, int* pairs, long cap, long* offsets)
{
    float* v = (float*)malloc(sizeof(float) * (size_t)T);
    int* ptr = (int*)malloc(sizeof(int) * (size_t)(T > 1 ? T - 1 : 1));
    long total = 0;
    for (int c = 0; c < B; ++c) {
        offsets[c] = total;
        long first = total;
        v[0] = singleton_term(S_(0, 0, c));
        for (int i = 1; i < T; ++i) {
            /* candidates [skip, j=0, 1, ..., i-1]; first maximum wins (NSCI:129-139) */
            float best = v[i - 1] + N_(i - 1, c);
            int sel = -1;
            for (int j = 0; j < i; ++j) {
                float cand = v[j] + S_(i, j, c);
                if (cand > best) { best = cand; sel = j; }
            }
            ptr[i - 1] = sel;
            v[i] = best + singleton_term(S_(i, i, c));
        }
        int j = start ? start[c] : T - 1;
        while (j > 0) { /* NSCI:172-190 */
            int sel = ptr[j - 1];
            if (S_(j, j, c) > 0.0f) {
                if (total >= cap) goto overflow;
                pairs[2 * total] = j; pairs[2 * total + 1] = j; ++total;
            }
            if (sel < 0) j -= 1;
            else {
                if (total >= cap) goto overflow;
                pairs[2 * total] = sel; pairs[2 * total + 1] = j; ++total;
                j = sel;
            }
        }
        if (S_(0, 0, c) > 0.0f) { /* NSCI:192-193 */
            if (total >= cap) goto overflow;
            pairs[2 * total] = 0; pairs[2 * total + 1] = 0; ++total;
        }
        /* reverse this chain's list (NSCI:196) */
        for (long a = first, b = total - 1; a < b; ++a, --b) {
            int t0 = pairs[2 * a], t1 = pairs[2 * a + 1];
            pairs[2 * a] = pairs[2 * b]; pairs[2 * a + 1] = pairs[2 * b + 1];
            pairs[2 * b] = t0; pairs[2 * b + 1] = t1;
        }
    }
    offsets[B] = total;
    free(v); free(ptr);
    return total;
overflow:
    free(v); free(ptr);
    return -1;
}

/*
 * evalPath (NSCI:508-550): unnormalised path score per chain
 *   out[c] = sum_{(i,j) in path_c} ( s[j,i,c] - (cum[j]-cum[i]) ) + cum[T-1]
 * cum = cumsum(pad(noise)) along time; torch's CPU cumsum accumulates fp32 input in
 * double and rounds each prefix to fp32, restated here.  The scatter_add (NSCI:547)
 * accumulates in fp32 in list order.
 */
void oracle_eval_path(const float* score, const float* noise, int T, int B,
                      const int* pairs, const long* offsets, float* out)
{
    float* cum = (float*)malloc(sizeof(float) * (size_t)T);
    for (int c = 0; c < B; ++c) {
        double acc = 0.0;
        cum[0] = 0.0f;
        for (int t = 1; t < T; ++t) { acc += (double)N_(t - 1, c); cum[t] = (float)acc; }
        float r = 0.0f;
        for (long k = offsets[c]; k < offsets[c + 1]; ++k) {
            int i = pairs[2 * k], j = pairs[2 * k + 1];
            float g = S_(j, i, c) - (cum[j] - cum[i]);
            r += g;
        }
        out[c] = r + cum[T - 1];
    }
    free(cum);
}

/*
 * ScaledInnerProductIntervalScorer.forward after the Linear map (LT:406-441).
 * q [C][T][D] (NOT yet divided), k [C][T][D], diag [C][T]; C = N*P chains.
 *   qs = q / sqrt(D)                              (LT:410)
 *   S[e,b,c] = (sum_d qs[c,e,d]*k[c,b,d]) * lenscale(|e-b|)   (LT:413-427)
 *   S[t,t,c] += diag[c,t]                         (LT:431-433)
 * length_scaling: 0 = "linear", 1 = "sqrt", 2 = "none".
 * Output S [T][T][C] (LT:439), full square like the reference.  fp32 sequential-d dot.
 */
void oracle_interval_score(const float* q, const float* k, const float* diag, int C, int T, int D,
                           int length_scaling, float* S)
{
    float sq = sqrtf((float)D);
    float* qs = (float*)malloc(sizeof(float) * (size_t)D);
    for (int c = 0; c < C; ++c)
        for (int e = 0; e < T; ++e) {
            for (int d = 0; d < D; ++d) qs[d] = q[((size_t)c * T + e) * D + d] / sq;
            for (int b = 0; b < T; ++b) {
                float acc = 0.0f;
                const float* kr = k + ((size_t)c * T + b) * D;
                for (int d = 0; d < D; ++d) acc += qs[d] * kr[d];
                int len = e > b ? e - b : b - e;
                if (length_scaling == 0) acc = acc * (float)len;
                else if (length_scaling == 1) acc = acc * sqrtf((float)len);
                if (e == b) acc = acc + diag[(size_t)c * T + e];
                S[((size_t)e * T + b) * C + c] = acc;
            }
        }
    free(qs);
}
