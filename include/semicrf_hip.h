/*
 * semicrf_hip.h -- C ABI of the MI355X (gfx950) Neural Semi-CRF interval layer.
 *
 * This is the drop-in boundary: plain pointers, sizes and a HIP stream handle; no torch
 * types.  The reference (Yujia-Yan/Transkun) has no FFI of its own -- its boundary is the
 * pure-Python class transkun/CRF/NeuralSemiCRFInterval.py:553-588 -- so each entry point
 * below cites the reference *function* it replaces; transkun_amd/CRF (ctypes) is the
 * host-side mirror of that class, and INTEGRATION.md shows the binding a Transkun
 * maintainer would add.
 *
 * Conventions
 *   - All pointers are DEVICE pointers (HBM) unless named h_*.  fp32 everywhere.
 *   - score  [T][T][B]  C-contiguous, indexed [end][begin][chain]; only end >= begin is read.
 *   - noise  [T-1][B]   score of "no event between frames t and t+1".
 *   - stream is a hipStream_t passed as void*; every call only ENQUEUES work on it
 *     (no host synchronisation), so calls compose with torch's current stream.
 *   - ws / ws_bytes: caller-owned scratch of at least semicrf_workspace_bytes(op,T,B) bytes,
 *     256-byte aligned; contents are undefined afterwards.
 *   - Return value: SEMICRF_OK or an error code; semicrf_last_error() gives the message of the
 *     last failing call on this thread.  Nothing is written on SEMICRF_EINVAL.
 */
#ifndef SEMICRF_HIP_H
#define SEMICRF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI history.  1: rounds 1-4.  2 (round 6; the changes themselves are round 5's): new entry points scorer_proj_nn3 /
 * scorer_proj_nn3_workspace_bytes and the flag bits SEMICRF_LEN_BF16X3, SEMICRF_PROJ_TN_BF16X3; semicrf_workspace_bytes(LOGZ_FWD /
 * LOGZ_BWD / VITERBI) grew by a second u buffer (leased workspaces alternate), B path-score granules and -- for launches of at most
 * 192 chains -- the spine-major band copy K * ceil(B/4) * 4 * 4 KB + its flag words (tens to ~200 MB at T = 1024..2048): callers
 * must size workspaces with semicrf_workspace_bytes of THIS library, never with a constant from an older one.  No entry point of
 * version 1 changed its signature or meaning. */
#define SEMICRF_ABI_VERSION 2

#define SEMICRF_OK 0
#define SEMICRF_EINVAL 1      /* bad shape / null pointer / unsupported size */
#define SEMICRF_EWORKSPACE 2  /* ws_bytes too small */
#define SEMICRF_ELAUNCH 3     /* HIP reported an error at enqueue time */
#define SEMICRF_ETIMEOUT 4    /* an EARLIER sweep gave up on a bounded wait on the device (see semicrf_async_error); nothing enqueued */

/* op ids for semicrf_workspace_bytes */
#define SEMICRF_OP_LOGZ_FWD 0
#define SEMICRF_OP_LOGZ_BWD 1
#define SEMICRF_OP_VITERBI 2
#define SEMICRF_OP_EVAL_PATH 3
#define SEMICRF_OP_INTERVAL_SCORE 4

/* length scaling of the interval scorer (LayersTransformer.py:416-427) */
#define SEMICRF_LEN_LINEAR 0
#define SEMICRF_LEN_SQRT 1
#define SEMICRF_LEN_NONE 2

/* interval_score_fwd, full_square bit 2 (OR it to 0 / 1 / 2): opt-in contraction on the bf16 matrix instructions.  Default
 * (bit clear) is the exact fp32 contraction.  With the bit every operand is split exactly into three bf16 limbs and six of
 * the nine limb products are accumulated in fp32: |S - S_exact| <= 2^-21 * qscale * len * sum_d |q_d k_d| (measured
 * <= 2^-22; tests/test_gpu_parity.py::test_scorer_bf16x3), i.e. fp32-grade but NOT bit-identical to the default, 1.3-1.4x
 * faster.  Operands must be finite with |x| < 2^127 (the first limb rounds to nearest: larger values round to infinity);
 * limbs below the bf16 normal range (|x| < 2^-110) may be flushed to zero by the matrix instruction. */
#define SEMICRF_SCORE_BF16X3 4
/* The same for the backward: OR it to length_scaling of interval_score_bwd_ws* / interval_score_bwd_fused_ws* (bit 4).  The two
 * products dq = G k, dk = G^T q run on the bf16 matrix instructions with every operand (the scaled cotangent G as well as k and q)
 * split exactly into three limbs: |dq - dq_exact| <= 2^-21 * sum_b |G[e,b] k[b,d]| per element (likewise dk), fp32-grade, not
 * bit-identical to the default.  Honoured where the packed path runs (workspace, D in {64,128,256}, T >= 64, aligned rows);
 * the direct kernels ignore it (exact fp32).  ddiag and drowc are exact fp32 sums either way (drowc in a different fixed order). */
#define SEMICRF_LEN_BF16X3 16

typedef void* semicrf_stream_t;

int semicrf_abi_version(void);
const char* semicrf_last_error(void);
size_t semicrf_workspace_bytes(int op, int T, int B);

/* Optional: LEASED workspaces.  The sweeps (semicrf_logz_fwd / _logz_bwd / _beta / _viterbi) need their scratch to read
 * 0xff when they start and therefore fill a caller-owned buffer in front of every launch (17 MB, 6.8 us at T=1024,
 * NBatch=352).  A buffer the caller registers is filled once; every launch leaves it the way the fill would.  The caller
 * promises, until semicrf_workspace_unregister:
 *   - nothing but this library's sweeps writes to the buffer (pass the registered base pointer as `ws`, unchanged);
 *   - at most one stream uses it at a time (launches into one workspace are ordered by the stream they are enqueued on);
 *   - the calling thread's current device is the buffer's device when it registers.
 * (Stream capture: a sweep enqueued on a capturing stream takes the ordinary path -- its fill becomes a node of the graph --
 *  whatever the workspace; every entry point of this library can be captured into a HIP graph and replayed.)
 * A change of (operation, T, B) costs one fill.  A launch that aborted (see below: NaN outputs / negative decode total)
 * raises a pinned host word; the next launch of ANY lease is preceded by a fill again.  semicrf_workspace_register may
 * synchronise the device (once per device); call it at set-up time.  (No counterpart in the reference: its
 * NeuralSemiCRFInterval.py:207-246, :386-414 allocate their temporaries per call.) */
int semicrf_workspace_register(void* ws, size_t ws_bytes);
int semicrf_workspace_unregister(void* ws);

/* Select kernel implementation: 0 = auto (fastest valid), 1 = row-sequential reference kernels.
 * Process-wide; meant for tests and A/B benchmarking. */
void semicrf_set_impl(int impl);
int semicrf_get_impl(void);

/* Debug/test hook, SYNCHRONISES the device: returns and clears the sticky device-side status word.
 * 0 = no kernel ever gave up on a bounded spin; 2..12 = a hand-off wait timed out (results invalid);
 * -1 = HIP error.  The persistent kernels never hang: every wait is bounded (0.3 - 2 s).
 *
 * What a caller sees WITHOUT this hook when a wait does time out (e.g. the GPU is shared and part of the persistent
 * kernel was not resident in time): the calls still return SEMICRF_OK -- nothing synchronises the host -- but the results
 * are poisoned, not silently wrong: logZ, the last row of v / q_out / beta and the gradient's last diagonal cells are NaN
 * for the chains of every workgroup that saw a timeout, and semicrf_viterbi writes offsets[B] = -1. */
int semicrf_debug_device_status(void);

/* The asynchronous error word, WITHOUT synchronising: nonzero (the device's code, 2..13) when a sweep enqueued earlier by this
 * process has given up on a bounded hand-off wait since the last look; reading clears it.  Every sweep entry point
 * (semicrf_logz_fwd / _logz_bwd / _beta / _viterbi / _logprob_*) looks first and returns SEMICRF_ETIMEOUT -- enqueueing nothing -- so
 * that a time-out is an ERROR on the caller's next call at the latest, not only NaN-poisoned outputs; a caller that synchronises
 * (reads results on the host) can ask here right after its synchronisation.  The word is raised by the device with a
 * system-scope store when the wait gives up, i.e. before the poisoned outputs are written.  (Error convention of the boundary:
 * SURVEY.md 8b -- the reference raises on every failure; it has no asynchronous ones.) */
int semicrf_async_error(void);

/* Host-side view of the sweeps' workgroup -> role map (test hook, no device work): the role ticket of workgroup `block`
 * in a launch of `grid` workgroups of which `n_spine` are ring workgroups.  Tickets < n_spine are rings (ticket = the
 * 4-chain group); the eight rings of a 32-chain panel group get workgroup indices that are equal modulo 8 (one XCD). */
int semicrf_debug_wg_ticket(int n_spine, int grid, int block);

/* Test hook, process-wide: force one of interval_score_fwd's kernels where it applies (0 = register loads, 32 = streaming,
 * 2 = the 64 x 128 tiles with the epilogue inside the contraction loop; 64 / 128 = the earlier shared-operand tiles, compiled into
 * the DEBUG library only -- libsemicrf_hip_debug.so, the same ABI built with -DSEMICRF_DEBUG_BUILD=1, which the parity tests load
 * through ctypes as the bit-level reference; the release library ignores 64 / 128 -- all give bit-identical scores); -1 =
 * automatic choice (the default).  The release library reads no environment variables. */
void semicrf_debug_score_variant(int variant);

/*
 * Log-partition, forward (alpha) sweep.
 * Replaces: computeLogZ (NeuralSemiCRFInterval.py:207-246) and the un-flipped half of
 * forward_backward (:394-410,:417).
 *   v[0] = softplus(s[0,0]);  v[i] = logaddexp(v[i-1]+n[i-1], logsumexp_{j<i}(v[j]+s[i,j])) + softplus(s[i,i])
 * Outputs: logZ [B] (= v[T-1]);  v [T][B] (may be NULL when no backward will follow).
 */
int semicrf_logz_fwd(const float* score, const float* noise, int T, int B,
                     float* logZ, float* v, void* ws, size_t ws_bytes, semicrf_stream_t stream);

/*
 * Gradient of sum_c gout[c]*logZ[c] w.r.t. score and noise: the backward (beta) sweep fused with
 * the marginals.  Replaces: the flipped half of forward_backward (:386-414), the marginals
 * (:424-447) and ComputeLogZFasterGrad.backward (:469-472).
 *   dScore[e,b,c] = gout[c] * exp(v[b] + q[e] - logZ + s[e,b])                     e > b
 *   dScore[t,t,c] = gout[c] * exp(v[t] + q[t] - logZ + s[t,t] - 2 softplus(s[t,t]))
 *   dScore[e,b,c] = 0 exactly                                                       e < b
 *   dNoise[t,c]   = gout[c] * exp(v[t] + q[t+1] + n[t] - logZ)
 * Inputs v, logZ come from semicrf_logz_fwd on the same score/noise.  dScore [T][T][B] is fully
 * written (including the zeros).  q_out [T][B] may be NULL.
 */
int semicrf_logz_bwd(const float* score, const float* noise, const float* v, const float* logZ,
                     const float* gout, int T, int B, float* dScore, float* dNoise, float* q_out,
                     void* ws, size_t ws_bytes, semicrf_stream_t stream);

/*
 * Viterbi decode.  Replaces: viterbiBackward (:13-104, forward=0, the default of .decode) and
 * viterbi (:107-202, forward=1), including the backtrack that the reference runs on the host.
 *   start: NULL or B ints (forcedStartPos; for forward=1 it is the END position, :161-165).
 *   pairs: int32 [cap][2] (begin,end), chain-major, ascending within a chain (packed).
 *   offsets: int32 [B+1] prefix counts; offsets[B] = total.  If total > cap the pairs content is
 *   truncated but offsets are still exact (callers allocate cap = B*(2T) to be safe, or retry).
 * Decoded indices are bit-identical to the reference's CPU path (first-maximum tie-break in the
 * candidate order [skip, nearest, ..., farthest], single fp32 add per candidate).
 */
int semicrf_viterbi(const float* score, const float* noise, int T, int B, const int32_t* start,
                    int forward, int32_t* pairs, int64_t cap, int32_t* offsets,
                    void* ws, size_t ws_bytes, semicrf_stream_t stream);

/*
 * Unnormalised path score.  Replaces: evalPath (:508-550).
 *   pairs int32 [K][2] (begin,end), offsets int32 [B+1] (chain c owns pairs[offsets[c]:offsets[c+1]]).
 *   out[c] = sum_path ( s[end,begin,c] - (cum[end]-cum[begin]) ) + cum[T-1],  cum = prefix sums of noise.
 *   K = number of intervals in pairs (= offsets[B]).
 */
int semicrf_eval_path(const float* score, const float* noise, int T, int B,
                      const int32_t* pairs, int64_t K, const int32_t* offsets, float* out,
                      void* ws, size_t ws_bytes, semicrf_stream_t stream);

/*
 * Gradient of sum_c gout[c]*evalPath[c], ACCUMULATED (+=) into dScore / dNoise (autograd of the
 * gathers at :540-548).  dScore[end,begin,c] += gout[c] per path interval;
 * dNoise[t,c] += gout[c] * [gap t not covered by an interval of the path].
 * K = number of intervals in pairs (= offsets[B], known to the host that packed them).
 * Either output may be NULL.  Callers that want a fresh gradient zero the buffers first.
 */
int semicrf_eval_path_bwd(const float* gout, int T, int B, const int32_t* pairs, int64_t K,
                          const int32_t* offsets, float* dScore, float* dNoise, semicrf_stream_t stream);

/*
 * logProb as ONE call each way.  Replaces: NeuralSemiCRFInterval.logProb (:587-588: evalPath(intervals) - computeLogZ())
 * and its backward (the autograd of :540-548 plus ComputeLogZFasterGrad.backward :469-472).
 *   semicrf_logprob_fwd: logProb[c] = evalPath[c] - logZ[c]; also leaves logZ [B] and (when non-NULL) v [T][B] for the
 *     backward.  Workspace: semicrf_workspace_bytes(SEMICRF_OP_LOGZ_FWD, T, B).
 *   semicrf_logprob_bwd: gradient of sum_c g[c] * logProb[c] with g[c] = gout[c * gout_stride]; gout_stride is 1 or 0 --
 *     0 reads ONE value for every chain, which is what the loss -logProb.sum() / n hands down (an expanded scalar): no
 *     [B] copy of it and no negated copy are made.  dScore [T][T][B] is fully written (marginals times -g, +g on the path
 *     cells, exact zeros for begin > end), dNoise [T-1][B] likewise.  Workspace: SEMICRF_OP_LOGZ_BWD.
 */
int semicrf_logprob_fwd(const float* score, const float* noise, int T, int B, const int32_t* pairs, int64_t K,
                        const int32_t* offsets, float* logProb, float* logZ, float* v, void* ws, size_t ws_bytes,
                        semicrf_stream_t stream);
int semicrf_logprob_bwd(const float* score, const float* noise, const float* v, const float* logZ, const float* gout,
                        int gout_stride, int T, int B, const int32_t* pairs, int64_t K, const int32_t* offsets,
                        float* dScore, float* dNoise, void* ws, size_t ws_bytes, semicrf_stream_t stream);

/*
 * The two gradient entry points with flags (replacing the same reference lines as semicrf_logz_bwd / semicrf_logprob_bwd).
 */
/* flags of semicrf_logz_bwd_f / semicrf_logprob_bwd_f */
#define SEMICRF_GRAD_UPPER_IS_ZERO 1   /* the caller's promise: every cell begin > end of dScore already holds +0.0f; the call does
                                        * not write them (a third of the bytes the gradient sweep moves).  transkun_amd keeps a pool
                                        * of gradient buffers whose upper triangle this library zeroed and nobody has written since
                                        * (transkun_amd/CRF: _GradPool); the result is the same dense tensor with an exactly-zero
                                        * upper triangle that NeuralSemiCRFInterval.py:436-440, :469-472 hand to autograd. */
int semicrf_logz_bwd_f(const float* score, const float* noise, const float* v, const float* logZ,
                       const float* gout, int T, int B, float* dScore, float* dNoise, float* q_out, int flags,
                       void* ws, size_t ws_bytes, semicrf_stream_t stream);
int semicrf_logprob_bwd_f(const float* score, const float* noise, const float* v, const float* logZ, const float* gout,
                          int gout_stride, int T, int B, const int32_t* pairs, int64_t K, const int32_t* offsets,
                          float* dScore, float* dNoise, int flags, void* ws, size_t ws_bytes, semicrf_stream_t stream);

/*
 * Interval-score construction.  Replaces: ScaledInnerProductIntervalScorer.forward after the
 * Linear map (LayersTransformer.py:406-441).
 *   q,k: [C][T][D] with row strides ldq/ldk (in floats; >= D); diag: [C][T] with stride ldd between
 *   consecutive t (so the packed Linear output [C][T][2D+1] can be passed without a split copy).
 *   S[e,b,c] = (sum_d (q[c,e,d]*qscale) * k[c,b,d]) * len(|e-b|)  (+ diag[c,t] on e==b)
 *   S is [T][T][C] (chain axis contiguous, the CRF's layout).  full_square: 0 = e >= b is computed and the cells e < b
 *   are set to zero (S needs no initialisation by the caller); 1 = the full square, as the reference materialises it;
 *   2 = e >= b only, the rest of S is left untouched -- for callers that hand S to the sweeps of this library only, which
 *   never read e < b (tests/test_gpu_parity.py::test_upper_triangle_is_never_read): saves the 2 T^2 C bytes of zeros.
 *   | SEMICRF_SCORE_BF16X3: three-limb bf16 contraction (above); honoured where the LDS-tiled kernels run (16-byte
 *   aligned rows, D % 64 == 0), the exact fp32 contraction otherwise.
 *   noise_out [T-1][C] is zero-filled when non-NULL (:436-437).
 */
int interval_score_fwd(const float* q, const float* k, const float* diag, int C, int T, int D,
                       int64_t ldq, int64_t ldk, int64_t ldd, float qscale, int length_scaling,
                       int full_square, float* S, float* noise_out, semicrf_stream_t stream);

/*
 * Backward of interval_score_fwd.  Replaces: the autograd of LayersTransformer.py:410-433 (scale, einsum,
 * length scaling, diag_embed).  dS is [T][T][C] (the CRF's gradient layout; only e >= b is read: the counterpart of
 * interval_score_fwd with full_square == 0):
 *   dq[c,e,:] = qscale * sum_{b<=e} dS[e,b,c] len(e-b) k[c,b,:]
 *   dk[c,b,:] = qscale * sum_{e>=b} dS[e,b,c] len(e-b) q[c,e,:]
 *   ddiag[c,t] = dS[t,t,c]
 * dq/dk: [C][T][D] with row strides lddq/lddk; ddiag: [C][T] with stride lddd.  Any output may be NULL.
 * Requires D % 32 == 0 and D <= 256 (SEMICRF_EINVAL otherwise; the Python mirror then differentiates with torch).
 */
int interval_score_bwd(const float* dS, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                       int64_t ldk, float qscale, int length_scaling, float* dq, float* dk, float* ddiag,
                       int64_t lddq, int64_t lddk, int64_t lddd, semicrf_stream_t stream);

/*
 * The same with a workspace: the cotangent is first repacked into per-chain matrices (scaled, zero above the diagonal)
 * and dq/dk become two batched triangular GEMMs with LDS-shared operands (scorer_bwd_gemm.hip), about 1.7x faster at
 * T=1024, C=352, D=256.  interval_score_bwd_workspace_bytes returns 0 when the packed path does not apply (D not in
 * {64, 128, 256}, T < 64); with ws == NULL, too few bytes or q/k rows that are not 16-byte aligned the call runs
 * exactly interval_score_bwd.
 */
size_t interval_score_bwd_workspace_bytes(int C, int T, int D);
int interval_score_bwd_ws(const float* dS, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                          int64_t ldk, float qscale, int length_scaling, float* dq, float* dk, float* ddiag,
                          int64_t lddq, int64_t lddk, int64_t lddd, void* ws, size_t ws_bytes, semicrf_stream_t stream);

/*
 * SLOT LAYOUT of the chain axis (the *_p entry points): a private layout between this library's scorer and its CRF kernels.
 *
 * The model's chains come in groups of `group` symbols per segment (90: ModelTransformer.py:97); the CRF kernels stream the
 * score tensor in 32-chain pieces of 128 bytes, and a chain count that is not a multiple of 32 makes most pieces straddle two
 * lines (T=691: 90 / 360 chains run 22 - 32 % slower than 96 / 384).  With the slot layout every group owns `pitch` >= group
 * SLOTS of the chain axis: chain c = g * group + p lives in slot g * pitch + p; the slots p = group .. pitch-1 are ghosts.
 *   - S (and dS) is [T][T][Cs], Cs = (C / group) * pitch; the scorer multiplies only the C real chains and writes exact zeros
 *     into the ghost slots (cells e >= b; with full_square == 0 also the zeros above the diagonal);
 *   - the CRF entry points are called with B = Cs: a ghost slot is an ordinary chain of all-zero scores whose results the
 *     caller drops; noise, alpha, beta, logZ, gout and the interval offsets ([Cs + 1], ghost slots empty) are slot-indexed;
 *   - q, k, diag and their gradients stay chain-indexed ([C][T][..]).
 * group == pitch (== any divisor layout) is the plain contiguous layout; a padded pitch must be a multiple of 4 and needs the
 * LDS-tiled kernels (16-byte aligned rows, D % 64 == 0, T >= 128; backward: the workspace path) -- SEMICRF_EINVAL otherwise.
 * The reference has no counterpart: its scorer ends with permute(2,3,0,1).contiguous() (LayersTransformer.py:439) and the
 * glue flattens (N, P) (ModelTransformer.py:215-216); transkun_amd/fused.py and transcribe.py use the slot layout inside and
 * hand out chain-indexed results.
 */
int interval_score_fwd_p(const float* q, const float* k, const float* diag, int C, int T, int D,
                         int64_t ldq, int64_t ldk, int64_t ldd, float qscale, int length_scaling,
                         int full_square, int group, int pitch, float* S, float* noise_out, semicrf_stream_t stream);
int interval_score_bwd_ws_p(const float* dS, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                            int64_t ldk, float qscale, int length_scaling, int group, int pitch, float* dq, float* dk,
                            float* ddiag, int64_t lddq, int64_t lddk, int64_t lddd, void* ws, size_t ws_bytes,
                            semicrf_stream_t stream);
int interval_score_bwd_fused_ws_p(const float* S, const float* alpha, const float* beta, const float* logZ,
                                  const float* gout, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                                  int64_t ldk, float qscale, int length_scaling, int group, int pitch, float* dq, float* dk,
                                  float* ddiag, int64_t lddq, int64_t lddk, int64_t lddd, void* ws, size_t ws_bytes,
                                  semicrf_stream_t stream);
int interval_score_path_bwd_p(const float* gout, const int32_t* pairs, int64_t K, const int32_t* offsets, const float* q,
                              const float* k, int C, int T, int D, int64_t ldq, int64_t ldk, float qscale, int length_scaling,
                              int group, int pitch, float* dq, float* dk, float* ddiag, int64_t lddq, int64_t lddk,
                              int64_t lddd, semicrf_stream_t stream);

/*
 * MERGED PROJECTION (the *_pc entry points): the scorer with a per-(chain, end) constant inside the contraction,
 *   S[e,b,c] = qscale * ( <q[c,e,:], k[c,b,:]> + rowc[c,e] ) * len(|e-b|)  (+ diag[c,e] on e == b).
 * Why: the reference projects ctx twice, q = ctx Wq^T + bq and k = ctx Wk^T + bk (LayersTransformer.py:388-397, :406-410), and
 * contracts <q_e, k_b>.  Algebraically <q_e, k_b> = <ctx_e A + v, ctx_b> + c_e with A = Wq^T Wk (256 x 256), v = bq Wk,
 * c_e = <ctx_e, Wq^T bk> + <bq, bk>: ONE 256 -> 256 projection z = ctx A + v, the contraction's second operand is ctx ITSELF
 * (no k tensor), and c is a matrix-vector product -- half the Linear's flops forward and backward.  The result is fp32-grade
 * but NOT bit-identical to the reference's operation order (a reassociation; tests/test_gpu_parity.py::test_merged_projection
 * holds it to the scorer tolerance and the segment goldens to logProb 2e-5), so only this package's own fused route and
 * transcription step use it (transkun_amd/fused.py); ScaledInnerProductIntervalScorer.forward keeps the two projections.
 * rowc / drowc: [C][T] with stride ldrc / lddrc between consecutive frames, or NULL (then these are the *_p entry points).
 * drowc[c,e] = qscale * sum_{b<=e} dS[e,b,c] len(e-b) is WRITTEN by the bwd entry points (summed in a fixed order) and ADDED
 * to by interval_score_path_bwd_pc.  Needs the LDS-tiled kernels like the slot layout.
 */
int interval_score_fwd_pc(const float* q, const float* k, const float* diag, const float* rowc, int C, int T, int D,
                          int64_t ldq, int64_t ldk, int64_t ldd, int64_t ldrc, float qscale, int length_scaling,
                          int full_square, int group, int pitch, float* S, float* noise_out, semicrf_stream_t stream);
int interval_score_bwd_ws_pc(const float* dS, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                             int64_t ldk, float qscale, int length_scaling, int group, int pitch, float* dq, float* dk,
                             float* ddiag, float* drowc, int64_t lddq, int64_t lddk, int64_t lddd, int64_t lddrc, void* ws,
                             size_t ws_bytes, semicrf_stream_t stream);
int interval_score_bwd_fused_ws_pc(const float* S, const float* alpha, const float* beta, const float* logZ,
                                   const float* gout, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                                   int64_t ldk, float qscale, int length_scaling, int group, int pitch, float* dq, float* dk,
                                   float* ddiag, float* drowc, int64_t lddq, int64_t lddk, int64_t lddd, int64_t lddrc,
                                   void* ws, size_t ws_bytes, semicrf_stream_t stream);
int interval_score_path_bwd_pc(const float* gout, const int32_t* pairs, int64_t K, const int32_t* offsets, const float* q,
                               const float* k, int C, int T, int D, int64_t ldq, int64_t ldk, float qscale,
                               int length_scaling, int group, int pitch, float* dq, float* dk, float* ddiag, float* drowc,
                               int64_t lddq, int64_t lddk, int64_t lddd, int64_t lddrc, semicrf_stream_t stream);

/*
 * The scorer's projection.  Replaces: the nn.Linear of ScaledInnerProductIntervalScorer (LayersTransformer.py:388-397: `self.map`,
 * applied at :406-410) and its autograd, as exact-fp32 matrix-core GEMMs of this library (csrc/proj_gemm.hip: an fp32 fmaf chain per
 * output, v_mfma_f32_32x32x2_f32) -- in the packed forms this package uses: y = [q | diag | 0 0 0] or [z | c | diag | 0 0], i.e. N
 * "main" columns (N in {64, 128, 256}) followed by two extra columns and zero padding.
 *
 *   scorer_proj_nn:  out[M][ldout] (+)= A[M][lda] (K columns used) * B[Kpad][ldb] (N columns)  (+ bias[N])
 *                    B is row-major with the contraction index as ROW and holds whole chunks of 32 rows, zero beyond K (Kpad =
 *                    K rounded up to 32: the caller pads; K % 4 == 0).  Forward: A = x, B = W[:N]^T.  Input gradient: A = dy (K =
 *                    the packed width), B = W (its N = the Linear's input size), accumulate = 1 adds to what out holds.
 *                    w2 != NULL (forward): two more output columns out[m][N + j] = <A[m], w2[j]> + b2[j] (w2 [2][K]), and
 *                    zero_cols columns of zeros behind them.
 *   scorer_proj_tn:  dW[total_rows][lddw] (N columns) = dy[M][lddy]^T x[M][ldx], db[total_rows] = column sums of dy: the R main
 *                    columns of dy through the matrix cores, the two columns extra_col0, extra_col0 + 1 (or -1: none) as dot
 *                    products on the side, rows / entries beyond them zero.  The contraction over M is cut into slices whose
 *                    partial results go through `ws` (scorer_proj_tn_workspace_bytes) and are summed in a fixed order.
 * Rows must be 16-byte aligned (pointers and leading dimensions), M * ld * 4 < 2^31.  SEMICRF_EINVAL for anything else (the
 * Python mirror then uses torch's GEMM).
 */
int scorer_proj_nn(const float* A, int64_t lda, int64_t M, int K, const float* B, int64_t ldb, int N, float* out, int64_t ldout,
                   const float* bias, const float* w2, const float* b2, int zero_cols, int accumulate, semicrf_stream_t stream);
/* scorer_proj_nn with the three-limb bf16 contraction (csrc/proj_gemm3.hip; opt-in, fp32-grade: |out - exact| <= 2^-21 sum_k |A[m][k] B[k][n]|
 * per element, not bit-identical): B is split once per call into `ws` (scorer_proj_nn3_workspace_bytes; 16-byte aligned), A in the loop.
 * N == 256 only; any other shape (or ws == NULL / too small) runs scorer_proj_nn's exact kernel.  The two extra columns are fp32 dot
 * products as before (in another fixed order). */
size_t scorer_proj_nn3_workspace_bytes(int K, int N);
int scorer_proj_nn3(const float* A, int64_t lda, int64_t M, int K, const float* B, int64_t ldb, int N, float* out, int64_t ldout,
                    const float* bias, const float* w2, const float* b2, int zero_cols, int accumulate, void* ws, size_t ws_bytes,
                    semicrf_stream_t stream);
/* scorer_proj_tn, total_rows | SEMICRF_PROJ_TN_BF16X3: the matrix part of the weight gradient on the three-limb bf16 kernel (N == 256;
 * fp32-grade, not bit-identical; the bias gradient and the two extra rows stay exact fp32 sums).  Other widths ignore the bit. */
#define SEMICRF_PROJ_TN_BF16X3 0x40000000
size_t scorer_proj_tn_workspace_bytes(int64_t M, int R, int N);
int scorer_proj_tn(const float* dy, int64_t lddy, int64_t M, int R, int extra_col0, int total_rows, const float* x, int64_t ldx, int N,
                   float* dW, int64_t lddw, float* db, void* ws, size_t ws_bytes, semicrf_stream_t stream);

/*
 * The weights of the MERGED projection and their gradient.  Replaces: nothing the reference computes as such -- its two projections
 * q = x Wq^T + bq, k = x Wk^T + bk (LayersTransformer.py:392-397, applied at :406-410) enter the score only through
 * <q_e, k_b> = <x_e A + v, x_b> + c_e with A = Wq^T Wk, v = bq Wk, c_e = <x_e, Wq^T bk> + <bq, bk>, so ONE size -> size GEMM
 * [z | c | diag | 0 ..] = x Wm^T + bm does (transkun_amd.fused.merged_weights).  W [2 D + 1][size] and bias [2 D + 1] are the
 * Linear's parameters (rows Wq, Wk, the diagonal row); Wm [rows][size], bm [rows] with rows >= size + 2:
 *   Wm[i] = sum_r Wk1[r][i] Wq[r] (i <= size, Wk1 = [Wk | bk]),  Wm[size + 1] = the diagonal row,  zero rows behind;  bm alike.
 * _bwd: dW [2 D + 1][size], dbias [2 D + 1] from dWm, dbm (the autograd of the forward).  size <= 256, contiguous rows.
 */
int scorer_merge_weights_fwd(const float* W, const float* bias, int D, int size, int rows, float* Wm, float* bm, float* WmT,
                             semicrf_stream_t stream);    /* WmT (or NULL): [size][size], WmT[k][n] = Wm[n][k] for n < size -- scorer_proj_nn's B */
/* The Linear's parameters (W [2 D + 1][size], bias) in the layouts scorer_proj_nn reads, one launch: BT [size][2 D] (BT[k][n] = W[n][k] for
 * the q and k rows: B of the two forward products, ldb = 2 D), Wqd [rows_pad][size] = [Wq; diagonal row; zero rows] (B of the input
 * gradient through [q | diag | 0 ..]), w2 [2][size] = [diagonal row; 0], b2 [2] = [its bias, 0] (the forward's two extra columns). */
int scorer_stage_linear(const float* W, const float* bias, int D, int size, int rows_pad, float* BT, float* Wqd, float* w2, float* b2,
                        semicrf_stream_t stream);
size_t scorer_merge_weights_bwd_workspace_bytes(int size);           /* the transpose of dWm's first size + 1 rows */
int scorer_merge_weights_bwd(const float* W, const float* bias, const float* dWm, const float* dbm, int D, int size, int rows, float* dW,
                             float* dbias, void* ws, size_t ws_bytes, semicrf_stream_t stream);

/*
 * Backward-direction values only (the beta half of forward_backward, NeuralSemiCRFInterval.py:386-414, without the
 * marginals): beta[t][c] by frame, natural log.  Workspace: semicrf_workspace_bytes(SEMICRF_OP_LOGZ_FWD, T, B).
 * Used by interval_score_bwd_fused, which rebuilds the marginals tile by tile instead of reading a dense gradient.
 */
int semicrf_beta(const float* score, const float* noise, int T, int B, float* beta, void* ws, size_t ws_bytes,
                 semicrf_stream_t stream);

/*
 * Loss gradient fused into the scorer backward: interval_score_bwd with the cotangent
 *   dS[e,b,c] = gout[c] * marginal[e,b,c]      (marginal as in NeuralSemiCRFInterval.py:424-440, e >= b)
 * built on the fly from S (= score [T][T][C]), alpha (= v of semicrf_logz_fwd), beta (semicrf_beta) and logZ --
 * the dense [T][T][C] gradient of ComputeLogZFasterGrad.backward (:469-472) is never written or read.
 * Outputs as interval_score_bwd.  The evalPath part of logProb's gradient (one-hot on the path cells) is sparse and
 * is added by the caller (transkun_amd/fused.py).
 */
int interval_score_bwd_fused(const float* S, const float* alpha, const float* beta, const float* logZ,
                             const float* gout, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                             int64_t ldk, float qscale, int length_scaling, float* dq, float* dk, float* ddiag,
                             int64_t lddq, int64_t lddk, int64_t lddd, semicrf_stream_t stream);

/*
 * The evalPath half of logProb's gradient (one-hot on the path cells, NeuralSemiCRFInterval.py:540-548) pushed through
 * the scorer, ADDED to dq/dk/ddiag: for every interval (b, e) of chain c (pairs/offsets as in semicrf_eval_path)
 *   dq[c,e,:] += w k[c,b,:],  dk[c,b,:] += w q[c,e,:],  w = gout[c] qscale len(e-b);  ddiag[c,e] += gout[c] if b == e.
 * Completes interval_score_bwd_fused[_ws] to the gradient of logProb = evalPath - logZ.
 */
int interval_score_path_bwd(const float* gout, const int32_t* pairs, int64_t K, const int32_t* offsets, const float* q,
                            const float* k, int C, int T, int D, int64_t ldq, int64_t ldk, float qscale, int length_scaling,
                            float* dq, float* dk, float* ddiag, int64_t lddq, int64_t lddk, int64_t lddd,
                            semicrf_stream_t stream);

/*
 * interval_score_bwd_fused on the packed path (workspace as interval_score_bwd_ws): the repack kernel evaluates the
 * marginals while it builds the per-chain matrices, the two GEMMs are the same.  Falls back to
 * interval_score_bwd_fused exactly like interval_score_bwd_ws falls back to interval_score_bwd.
 */
int interval_score_bwd_fused_ws(const float* S, const float* alpha, const float* beta, const float* logZ,
                                const float* gout, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                                int64_t ldk, float qscale, int length_scaling, float* dq, float* dk, float* ddiag,
                                int64_t lddq, int64_t lddk, int64_t lddd, void* ws, size_t ws_bytes,
                                semicrf_stream_t stream);

/*
 * Interval features for the attribute heads.  Replaces: TransKun.fetchIntervalFeaturesBatch
 * (ModelTransformer.py:501-532) and the concatenation that feeds the velocity / onset-offset predictors (:578-582),
 * consuming the packed decode output on the device (pairs [K][2], offsets [C+1] as written by semicrf_viterbi; chain
 * c = segment * nSym + symbol) instead of Python lists:
 *   out[i] = [ ctx[c,begin,:] | ctx[c,end,:] | ctx[c,begin,:] * ctx[c,end,:] ]   ([K][3D]; ctx is [C][T][D], row stride ldc)
 *   symIdx[i] = c % nSym, scatterIdx[i] = c                                       (int64; either may be NULL)
 * The backward ADDS into dctx ([C][T][D], row stride lddc; zero it first for a fresh gradient).
 */
int interval_features_gather(const float* ctx, int C, int T, int D, int64_t ldc, const int32_t* pairs, int64_t K,
                             const int32_t* offsets, int nSym, float* out, int64_t* symIdx, int64_t* scatterIdx,
                             semicrf_stream_t stream);
int interval_features_gather_bwd(const float* gout, const float* ctx, int C, int T, int D, int64_t ldc, const int32_t* pairs,
                                 int64_t K, const int32_t* offsets, float* dctx, int64_t lddc, semicrf_stream_t stream);

/*
 * Transcription segment loop, on the packed decode output in HBM.
 *
 * segment_onset_filter.  Replaces: the onsetBound filter of TransKun.transcribeFrames (ModelTransformer.py:554-555),
 *   `path = [[e for e in _ if e[0] < onsetBound] for _ in path]`: pairs/offsets (as written by semicrf_viterbi) ->
 *   pairs_out [cap][2] / offsets_out [B+1]; counts_ws: B ints of scratch.  offsets_out[B] is exact even if it exceeds cap.
 *
 * segment_events.  Replaces: the per-interval event assembly of transcribeFrames (:672-718) and the hand-off of
 *   TransKun.transcribe (:789-800), for chains c = segment * nSym + symbol in list order:
 *     start = (begin + ofValue[i][0]) * frameDur, end = (end + ofValue[i][1]) * frameDur   (double, the reference's order)
 *     hasOnset = begin > 0 || ofPresence[i][0];  hasOffset = end < lastFrameIdx || ofPresence[i][1]
 *     start = max(start, lastEnd); end = max(end, start + 1e-8); lastEnd = end            (per chain)
 *     times[i] = the two shifted by beginTime[segment], clamped (start >= 0, end >= start);  flags[i] = {hasOnset, hasOffset}
 *     lastP[c] = end frame of the chain's last interval with hasOffset (0 if none);  nextStart[c] = max(lastP[c] - stepFrames, 0)
 *   ofValue: float [K][2] (already (mean - 0.5) / 0.99 clamped, :650-653), ofPresence: bytes [K][2] (logit > 0, :655),
 *   beginTime: double [B / nSym].  nextStart is what the next semicrf_viterbi takes as `start`.
 */
int segment_onset_filter(const int32_t* pairs, const int32_t* offsets, int B, int bound, int32_t* pairs_out, int64_t cap,
                         int32_t* offsets_out, int32_t* counts_ws, semicrf_stream_t stream);
int segment_events(const int32_t* pairs, int64_t K, const int32_t* offsets, int B, int nSym, const float* ofValue,
                   const unsigned char* ofPresence, int lastFrameIdx, double frameDur, const double* beginTime, int stepFrames,
                   double* times, unsigned char* flags, int32_t* lastP, int32_t* nextStart, semicrf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SEMICRF_HIP_H */
