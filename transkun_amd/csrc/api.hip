// api.hip -- extern "C" entry points declared in include/semicrf_hip.h: argument checks,
// workspace carving and dispatch to the kernels.  Nothing here synchronises the host.
#include <stdarg.h>
#include <string.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include "common.h"

namespace semicrf {

static thread_local char g_err[512] = "";
static std::atomic<int> g_impl{0};

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// kernels (defined in the other translation units)
void launch_rowseq_sweep(int mode, int dir, const float* score, const float* noise, int T, int B, float* u,
                         int* code, float* out_last, hipStream_t stream);
void launch_marginals(const float* score, const float* noise, const float* v, const float* q,
                      const float* logZ, const float* gout, int T, int B, float* dScore, float* dNoise,
                      hipStream_t stream, int gstride = 1, float gscale = 1.0f);
void launch_backtrack(const int* code, int T, int B, const int* start, int forward, int* region, int* counts,
                      int* pairs, long long cap, int* offsets, hipStream_t stream, const unsigned* err, int nerr, int err_stride);
const unsigned* persist_error_words(void* pws, int* n, int* stride);
void launch_onset_filter(const int* pairs, const int* offsets, int B, int bound, int* pairs_out, long long cap, int* offsets_out,
                         int* counts, hipStream_t stream);
void launch_segment_events(const int* pairs, const int* offsets, int B, int nSym, const float* ofValue, const unsigned char* ofPresence,
                           int lastFrameIdx, double frameDur, const double* beginTime, int stepFrames, double* times, unsigned char* flags,
                           int* lastP, int* nextStart, hipStream_t stream);
void launch_eval_path(const float* score, const float* noise, int T, int B, int K, const int* pairs,
                      const int* offsets, float* out, hipStream_t stream, const float* sub = nullptr);
void launch_eval_path_bwd(const float* gout, int T, int B, int K, const int* pairs, const int* offsets,
                          float* dScore, float* dNoise, hipStream_t stream, int gstride = 1, float gscale = 1.0f, int noise_term_done = 0);
void launch_interval_score_naive(const float* q, const float* k, const float* diag, int C, int T, int D,
                                 long long ldq, long long ldk, long long ldd, float qscale, int mode, int full,
                                 float* S, hipStream_t stream);

bool interval_score_mfma_supported(int C, int T, int D);
bool interval_score_bwd_supported(int C, int T, int D);
void launch_interval_score_bwd_fused(const float* S, const float* alpha, const float* beta, const float* logZ,
                                     const float* gout, const float* q, const float* k, int C, int T, int D,
                                     long long ldq, long long ldk, float qscale, int mode, float* dq, float* dk,
                                     float* ddiag, long long lddq, long long lddk, long long lddd, hipStream_t stream);
void launch_interval_score_bwd(const float* dS, const float* q, const float* k, int C, int T, int D, long long ldq,
                               long long ldk, float qscale, int mode, float* dq, float* dk, float* ddiag,
                               long long lddq, long long lddk, long long lddd, hipStream_t stream);
void launch_zero_upper(float* X, int T, int B, hipStream_t stream);
void launch_interval_features(const float* ctx, int C, int T, int D, long long ldc, const int* pairs, int K,
                              const int* offsets, int nSym, float* out, long long* symIdx, long long* scatterIdx,
                              hipStream_t stream);
void launch_interval_features_bwd(const float* gout, const float* ctx, int C, int T, int D, long long ldc, const int* pairs,
                                  int K, const int* offsets, float* dctx, long long lddc, hipStream_t stream);
void launch_interval_score_path_bwd(const float* gout, const int* pairs, int K, const int* offsets, const float* q,
                                    const float* k, int C, int T, int D, long long ldq, long long ldk, float qscale, int mode,
                                    float* dq, float* dk, float* ddiag, long long lddq, long long lddk, long long lddd,
                                    hipStream_t stream, int group, int pitch, float* drowc = nullptr, long long lddrc = 1);
void launch_interval_score_bwd_rowsum(const float* dS, const float* const* fused, float* drowc, long long ldrc, int C, int T,
                                      float qscale, int mode, int group, int pitch, hipStream_t stream);
void launch_interval_score_bwd_diag(const float* dS, const float* const* fused, float* ddiag, int C, int T, long long lddd,
                                    int group, int pitch, hipStream_t stream);
bool interval_score_slots_supported(int C, int T, int D, const float* q, const float* k, long long ldq, long long ldk);
size_t interval_score_bwd_ws_bytes(int C, int T, int D);
bool launch_interval_score_bwd_packed(const float* dS, const float* q, const float* k, int C, int T, int D, long long ldq,
                                      long long ldk, float qscale, int mode, float* dq, float* dk, long long lddq,
                                      long long lddk, void* ws, size_t ws_bytes, hipStream_t stream,
                                      const float* const* fused, int group, int pitch, float* drowc = nullptr, long long lddrc = 1,
                                      int prec = 0);
int launch_interval_score_mfma(const float* q, const float* k, const float* diag, int C, int T, int D,
                                long long ldq, long long ldk, long long ldd, float qscale, int mode, int full,
                                float* S, hipStream_t stream, int prec, int group, int pitch, const float* rowc = nullptr,
                                long long ldrc = 1);
int launch_proj_nn(const float* A, long long lda, long long M, int K, const float* B, long long ldb, int N, float* out, long long ldout,
                   const float* bias, const float* w2, const float* b2, int zero_cols, int accumulate, hipStream_t stream);
size_t proj_tn_workspace_bytes(long long M, int R, int N);
size_t proj_nn3_workspace_bytes(int K, int N);
int launch_proj_nn3(const float* A, long long lda, long long M, int K, const float* B, long long ldb, int N, float* out, long long ldout,
                    const float* bias, const float* w2, const float* b2, int zero_cols, int accumulate, void* ws, size_t ws_bytes,
                    hipStream_t stream);
void launch_merge_weights_fwd(const float* W, const float* bias, int D, int size, int rows, float* Wm, float* bm, float* WmT, hipStream_t stream);
void launch_merge_weights_bwd(const float* W, const float* bias, const float* dWm, const float* dbm, int D, int size, float* dW, float* dbias,
                              float* ws, hipStream_t stream);
size_t merge_weights_bwd_workspace_bytes(int size);
void launch_stage_linear(const float* W, const float* bias, int D, int size, int rows_pad, float* BT, float* Wqd, float* w2, float* b2,
                         hipStream_t stream);
int launch_proj_tn(const float* A, long long lda, long long M, int R, int extra_col0, int total_rows, const float* X, long long ldx, int N,
                   float* dW, long long lddw, float* db, void* ws, size_t ws_bytes, hipStream_t stream);
size_t persist_workspace_bytes(int T, int B);
bool persist_supported(int T, int B);
int launch_persist_sweep(int mode, int dir, const float* score, const float* noise, int T, int B, float* u_out,
                         float* last_out, int* code, void* ws, hipStream_t stream, int lease, unsigned lease_tag);
int persist_set_host_abort_word(unsigned* devptr);
int persist_wg_ticket(int nSpine, int grid, int b);
void set_score_variant(int v);

int read_and_clear_device_status();

int launch_persist_logz_bwd(const float* score, const float* noise, const float* v, const float* logZ,
                            const float* gout, int T, int B, float* dScore, float* dNoise, float* q_out, void* ws,
                            hipStream_t stream, int lease, unsigned lease_tag, int gstride = 1, float gscale = 1.0f, int keep_upper = 0,
                            float noise_add = 0.0f);
int launch_persist_logprob_fwd(const float* score, const float* noise, int T, int B, float* u_out, float* logZ, const int* pairs,
                               int K, const int* offsets, float* logProb, void* ws, hipStream_t stream, int lease, unsigned lease_tag);

static bool use_persist(int T, int B) { return g_impl.load() == 0 && persist_supported(T, B); }

struct Carver {
    char* p;
    size_t left;
    bool ok = true;
    Carver(void* ws, size_t n) : p((char*)ws), left(n) {}
    template <typename U>
    U* take(size_t count)
    {
        size_t bytes = align_up(count * sizeof(U));
        if (bytes > left) { ok = false; return nullptr; }
        U* r = (U*)p;
        p += bytes;
        left -= bytes;
        return r;
    }
};

static size_t tb(int T, int B) { return align_up((size_t)T * (size_t)B * 4); }

// ---- leased workspaces (semicrf_workspace_register) ------------------------------------------------------------------
// A sweep needs its scratch to read 0xff everywhere when it starts: u values are their own "not published yet" flag.  For a
// caller-owned buffer that means one fill (17 MB at T=1024, NBatch=352: 6.8 us) in front of every launch.  A workspace
// that the caller REGISTERS -- promising that nothing but sweeps of this library writes to it, that at most one stream
// uses it at a time -- is filled once; every launch then leaves it the way the fill would (persist.hip: the rings put
// their u values back, the last wave resets the control words, far partials carry the workspace's own launch count).
// A launch that timed out raises a pinned host word; the host then distrusts every lease and fills again.
struct Lease { size_t bytes; int op, T, B; bool clean; unsigned count; };
static std::mutex g_lease_mu;
static std::unordered_map<void*, Lease> g_leases;
static unsigned* g_abort_word = nullptr;                 // pinned + mapped: the device address equals the host address
static std::atomic<bool> g_abort_on_device[64];          // the kernels of device d know the word's address

// The pinned abort word exists for EVERY caller of the sweeps (round 4; until then only for leased workspaces): a bounded wait
// that timed out on the device raises it with a system-scope store, and the NEXT sweep entry point of this library reports it as
// SEMICRF_ETIMEOUT instead of launching -- a time-out is loud on every path (the poisoned outputs of the launch that aborted have
// been handed out by then: nothing synchronises the host; semicrf_async_error lets a caller that does synchronise ask earlier).
static int ensure_abort_word(hipStream_t stream)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (g_abort_on_device[dev].load()) return 0;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) return 0;   // not now: the set-up synchronises
    {
        std::lock_guard<std::mutex> lk(g_lease_mu);
        if (!g_abort_word) {
            void* p = nullptr;
            if (hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess) return 1;
            memset(p, 0, 64);
            g_abort_word = (unsigned*)p;
        }
    }
    void* dp = nullptr;
    if (hipHostGetDevicePointer(&dp, g_abort_word, 0) != hipSuccess || persist_set_host_abort_word((unsigned*)dp)) return 1;
    g_abort_on_device[dev].store(true);
    return 0;
}

// the code a device-side time-out left behind (0: none); clears it and distrusts every lease
static unsigned take_async_error()
{
    if (!g_abort_word) return 0u;
    const unsigned code = __atomic_exchange_n(g_abort_word, 0u, __ATOMIC_RELAXED);
    if (code != 0u) {
        std::lock_guard<std::mutex> lk(g_lease_mu);
        for (auto& kv : g_leases) kv.second.clean = false;          // some launch gave up: its workspace is in an unknown state
    }
    return code;
}

// first thing in every sweep entry point
static int sweep_prologue(const char* what, hipStream_t stream)
{
    if (ensure_abort_word(stream)) { set_error("%s: could not set up the device's abort word", what); return SEMICRF_ELAUNCH; }
    if (const unsigned code = take_async_error()) {
        set_error("%s: an earlier sweep on this GPU gave up on a bounded hand-off wait (device code %u: the GPU is shared with work that "
                  "kept part of the persistent kernel from running, or a CU mask hides compute units); the results of that launch "
                  "and of launches enqueued behind it are invalid (NaN-poisoned).  Nothing was enqueued by this call.", what, code);
        return SEMICRF_ETIMEOUT;
    }
    return SEMICRF_OK;
}

// what the next sweep into `ws` has to do: 0 ordinary, 1 fill + self-clean, 2 clean already; tag = the lease's launch count
static int lease_acquire(void* ws, int op, int T, int B, unsigned* tag, hipStream_t stream)
{
    *tag = 0;
    std::lock_guard<std::mutex> lk(g_lease_mu);
    if (g_leases.empty()) return 0;
    auto it = g_leases.find(ws);
    if (it == g_leases.end()) return 0;
    {
        // A launch that is being CAPTURED into a graph is replayed with the parameters of the capture -- the same granule tag
        // every time, which a lease must never repeat (the rings would run ahead on the previous replay's partials and clear u
        // under the panels).  Captured launches take the ordinary path: the fill is a node of the graph and resets everything
        // on every replay; the lease no longer counts as clean.
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
            it->second.clean = false;
            return 0;
        }
    }
    // (an abort word raised since the entry point's prologue is left for the next prologue: it reports AND distrusts the leases)
    Lease& L = it->second;
    const bool clean = L.clean && L.op == op && L.T == T && L.B == B;
    L.op = op; L.T = T; L.B = B; L.clean = true;                    // the launch enqueued next leaves it clean
    *tag = ++L.count;
    if (L.count == 0u) { *tag = ++L.count; return 1; }              // the count wrapped: the kernel's generation check expects count - 1
    return clean ? 2 : 1;
}
// a sweep that could not be enqueued leaves nothing behind, and a launch of the row-sequential kernels (impl 1) carves its
// scratch out of the same buffer: either way the lease no longer counts as clean
static void lease_failed(void* ws)
{
    std::lock_guard<std::mutex> lk(g_lease_mu);
    auto it = g_leases.find(ws);
    if (it != g_leases.end()) it->second.clean = false;
}

}  // namespace semicrf

using namespace semicrf;

extern "C" {

int semicrf_abi_version(void) { return SEMICRF_ABI_VERSION; }
const char* semicrf_last_error(void) { return g_err; }
void semicrf_set_impl(int impl) { g_impl.store(impl); }
int semicrf_get_impl(void) { return g_impl.load(); }
int semicrf_debug_device_status(void) { return read_and_clear_device_status(); }
int semicrf_async_error(void) { return (int)take_async_error(); }
int semicrf_debug_wg_ticket(int n_spine, int grid, int block) { return persist_wg_ticket(n_spine, grid, block); }
void semicrf_debug_score_variant(int variant) { set_score_variant(variant); }

int semicrf_workspace_register(void* ws, size_t ws_bytes)
{
    SEMICRF_CHECK_ARG(ws != nullptr && ws_bytes > 0, "workspace is NULL or empty");
    if (ensure_abort_word(nullptr)) { set_error("could not hand the abort word to the device"); return SEMICRF_ELAUNCH; }
    std::lock_guard<std::mutex> lk(g_lease_mu);
    g_leases[ws] = Lease{ws_bytes, -1, 0, 0, false, 0u};
    return SEMICRF_OK;
}

int semicrf_workspace_unregister(void* ws)
{
    std::lock_guard<std::mutex> lk(g_lease_mu);
    g_leases.erase(ws);
    return SEMICRF_OK;
}

size_t semicrf_workspace_bytes(int op, int T, int B)
{
    if (T <= 0 || B <= 0) return 0;
    switch (op) {
        case SEMICRF_OP_LOGZ_FWD: return tb(T, B) + persist_workspace_bytes(T, B) + 4096;
        case SEMICRF_OP_LOGZ_BWD: return tb(T, B) + persist_workspace_bytes(T, B) + 4096;
        case SEMICRF_OP_VITERBI:
            // u [T][B] f32, code [B][T] i32, region [B][2T][2] i32, counts [B] i32, persistent-sweep scratch
            return tb(T, B) * 2 + align_up((size_t)B * 2 * T * 2 * 4) + align_up((size_t)B * 4) +
                   persist_workspace_bytes(T, B) + 4096;
        case SEMICRF_OP_EVAL_PATH: return 4096;
        case SEMICRF_OP_INTERVAL_SCORE: return 4096;
        default: return 0;
    }
}

static int check_common(const float* score, const float* noise, int T, int B)
{
    SEMICRF_CHECK_ARG(T >= 1 && B >= 1, "T=%d, B=%d must be >= 1", T, B);
    SEMICRF_CHECK_ARG((long long)T * T * B < (1ll << 40), "T*T*B too large");
    SEMICRF_CHECK_ARG(T < (1 << 29), "T too large");
    SEMICRF_CHECK_ARG(score != nullptr, "score is NULL");
    SEMICRF_CHECK_ARG(noise != nullptr || T == 1, "noise is NULL");
    return SEMICRF_OK;
}

// the intervals of semicrf_logprob_fwd, when the sweep's launch can compute the path scores on the side (*folded = 1)
struct PathFold { const int32_t* pairs; int64_t K; const int32_t* offsets; float* logProb; int folded; };

static int logz_fwd_impl(const float* score, const float* noise, int T, int B, float* logZ, float* v, void* ws,
                         size_t ws_bytes, semicrf_stream_t stream, PathFold* pf)
{
    if (int rc = check_common(score, noise, T, B)) return rc;
    SEMICRF_CHECK_ARG(logZ != nullptr, "logZ is NULL");
    if (int rc = sweep_prologue("semicrf_logz_fwd", (hipStream_t)stream)) return rc;
    Carver cv(ws, ws_bytes);
    // the sweep's own workspace comes FIRST: a leased ws is only clean where the previous launch left it clean, so its place
    // must not depend on which optional outputs the caller passes (v == NULL used to move it behind the scratch alpha: a
    // logProb without gradient followed by one with, same lease -> NaN)
    const bool fast = use_persist(T, B);
    void* pws = fast ? cv.take<char>(persist_workspace_bytes(T, B)) : nullptr;
    float* vv = v ? v : cv.take<float>((size_t)T * B);
    if (!cv.ok || !ws) { set_error("workspace too small for logz_fwd"); return SEMICRF_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    if (fast) {
        unsigned ltag = 0;
        const int lease = lease_acquire(ws, SEMICRF_OP_LOGZ_FWD, T, B, &ltag, st);
        const int rc = pf ? launch_persist_logprob_fwd(score, noise, T, B, vv, logZ, pf->pairs, (int)pf->K, pf->offsets, pf->logProb, pws, st, lease, ltag)
                          : launch_persist_sweep(0, 0, score, noise, T, B, vv, logZ, nullptr, pws, st, lease, ltag);
        if (rc) {
            lease_failed(ws);
            set_error("persistent sweep could not be enqueued (too many chain chunks for this device)"); return SEMICRF_ELAUNCH;
        }
        if (pf) pf->folded = 1;
    } else {
        lease_failed(ws);
        launch_rowseq_sweep(0, 0, score, noise, T, B, vv, nullptr, logZ, st);
    }
    SEMICRF_CHECK_LAUNCH("semicrf_logz_fwd");
    return SEMICRF_OK;
}

int semicrf_logz_fwd(const float* score, const float* noise, int T, int B, float* logZ, float* v, void* ws,
                     size_t ws_bytes, semicrf_stream_t stream)
{
    return logz_fwd_impl(score, noise, T, B, logZ, v, ws, ws_bytes, stream, nullptr);
}

static int logz_bwd_impl(const float* score, const float* noise, const float* v, const float* logZ,
                         const float* gout, int gstride, float gscale, int T, int B, float* dScore, float* dNoise, float* q_out,
                         int flags, void* ws, size_t ws_bytes, semicrf_stream_t stream, float noise_add = 0.0f, int* noise_folded = nullptr)
{
    if (int rc = check_common(score, noise, T, B)) return rc;
    SEMICRF_CHECK_ARG((flags & ~SEMICRF_GRAD_UPPER_IS_ZERO) == 0, "unknown flags %d", flags);
    if (int rc = sweep_prologue("semicrf_logz_bwd", (hipStream_t)stream)) return rc;
    const int keep_upper = (flags & SEMICRF_GRAD_UPPER_IS_ZERO) ? 1 : 0;
    SEMICRF_CHECK_ARG(v && logZ && gout && dScore, "v/logZ/gout/dScore must be non-NULL");
    SEMICRF_CHECK_ARG(dNoise != nullptr || T == 1, "dNoise is NULL");
    SEMICRF_CHECK_ARG(gstride == 0 || gstride == 1, "gout stride must be 0 (one value for all chains) or 1");
    Carver cv(ws, ws_bytes);
    const bool fast = use_persist(T, B);
    void* pws = fast ? cv.take<char>(persist_workspace_bytes(T, B)) : nullptr;      // first: see semicrf_logz_fwd
    float* q = q_out ? q_out : cv.take<float>((size_t)T * B);
    if (!cv.ok || !ws) { set_error("workspace too small for logz_bwd"); return SEMICRF_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    if (fast) {
        // beta sweep fused with the marginals: score is read once, dScore written once
        unsigned ltag = 0;
        const int lease = lease_acquire(ws, SEMICRF_OP_LOGZ_BWD, T, B, &ltag, st);
        if (launch_persist_logz_bwd(score, noise, v, logZ, gout, T, B, dScore, dNoise, q, pws, st, lease, ltag, gstride, gscale, keep_upper, noise_add)) {
            lease_failed(ws);
            set_error("persistent sweep could not be enqueued (too many chain chunks for this device)"); return SEMICRF_ELAUNCH;
        }
        if (noise_folded) *noise_folded = 1;        // the ring waves added noise_add * gout to every gap
    } else {
        lease_failed(ws);
        launch_rowseq_sweep(0, 1, score, noise, T, B, q, nullptr, nullptr, st);
        launch_marginals(score, noise, v, q, logZ, gout, T, B, dScore, dNoise, st, gstride, gscale);
    }
    return SEMICRF_OK;
}

int semicrf_logz_bwd_f(const float* score, const float* noise, const float* v, const float* logZ,
                       const float* gout, int T, int B, float* dScore, float* dNoise, float* q_out, int flags, void* ws,
                       size_t ws_bytes, semicrf_stream_t stream)
{
    if (int rc = logz_bwd_impl(score, noise, v, logZ, gout, 1, 1.0f, T, B, dScore, dNoise, q_out, flags, ws, ws_bytes, stream)) return rc;
    SEMICRF_CHECK_LAUNCH("semicrf_logz_bwd");
    return SEMICRF_OK;
}

int semicrf_logz_bwd(const float* score, const float* noise, const float* v, const float* logZ,
                     const float* gout, int T, int B, float* dScore, float* dNoise, float* q_out, void* ws,
                     size_t ws_bytes, semicrf_stream_t stream)
{
    return semicrf_logz_bwd_f(score, noise, v, logZ, gout, T, B, dScore, dNoise, q_out, 0, ws, ws_bytes, stream);
}

int semicrf_logprob_fwd(const float* score, const float* noise, int T, int B, const int32_t* pairs, int64_t K,
                        const int32_t* offsets, float* logProb, float* logZ, float* v, void* ws, size_t ws_bytes,
                        semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(offsets && logProb && logZ, "offsets/logProb/logZ must be non-NULL");
    SEMICRF_CHECK_ARG(K >= 0 && K < (1ll << 31) && (K == 0 || pairs), "bad interval count");
    // ONE launch where the persistent sweep runs (round 5): spare waves of the sweep's launch compute the path scores while it runs,
    // the ring wave that finalises logZ subtracts (persist.hip: path_role); the row-sequential fallback keeps the path kernel
    PathFold pf{pairs, K, offsets, logProb, 0};
    if (int rc = logz_fwd_impl(score, noise, T, B, logZ, v, ws, ws_bytes, stream, &pf)) return rc;
    if (!pf.folded) launch_eval_path(score, noise, T, B, (int)K, pairs, offsets, logProb, (hipStream_t)stream, logZ);
    SEMICRF_CHECK_LAUNCH("semicrf_logprob_fwd");
    return SEMICRF_OK;
}

int semicrf_logprob_bwd(const float* score, const float* noise, const float* v, const float* logZ, const float* gout,
                        int gout_stride, int T, int B, const int32_t* pairs, int64_t K, const int32_t* offsets, float* dScore,
                        float* dNoise, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    return semicrf_logprob_bwd_f(score, noise, v, logZ, gout, gout_stride, T, B, pairs, K, offsets, dScore, dNoise, 0, ws, ws_bytes, stream);
}

int semicrf_logprob_bwd_f(const float* score, const float* noise, const float* v, const float* logZ, const float* gout,
                          int gout_stride, int T, int B, const int32_t* pairs, int64_t K, const int32_t* offsets, float* dScore,
                          float* dNoise, int flags, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(offsets, "offsets must be non-NULL");
    SEMICRF_CHECK_ARG(K >= 0 && K < (1ll << 31) && (K == 0 || pairs), "bad interval count");
    // d logProb = d evalPath - d logZ: the marginals with -gout, then +gout on the path's cells and uncovered gaps
    // (round 5: the path score's `+ gout on every gap` is added by the gradient sweep's ring waves as they store the noise marginals;
    // what is left for a second launch is the scatter onto the path's own cells and covered gaps)
    int noise_folded = 0;
    if (int rc = logz_bwd_impl(score, noise, v, logZ, gout, gout_stride, -1.0f, T, B, dScore, dNoise, nullptr, flags, ws, ws_bytes, stream,
                               1.0f, &noise_folded)) return rc;
    launch_eval_path_bwd(gout, T, B, (int)K, pairs, offsets, dScore, dNoise, (hipStream_t)stream, gout_stride, 1.0f, noise_folded);
    SEMICRF_CHECK_LAUNCH("semicrf_logprob_bwd");
    return SEMICRF_OK;
}

int semicrf_beta(const float* score, const float* noise, int T, int B, float* beta, void* ws, size_t ws_bytes,
                 semicrf_stream_t stream)
{
    if (int rc = check_common(score, noise, T, B)) return rc;
    SEMICRF_CHECK_ARG(beta, "beta must be non-NULL");
    if (int rc = sweep_prologue("semicrf_beta", (hipStream_t)stream)) return rc;
    Carver cv(ws, ws_bytes);
    const bool fast = use_persist(T, B);
    void* pws = fast ? cv.take<char>(persist_workspace_bytes(T, B)) : nullptr;
    if (!cv.ok || (fast && !ws)) { set_error("workspace too small for beta"); return SEMICRF_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    if (fast) {
        unsigned ltag = 0;
        const int lease = lease_acquire(ws, SEMICRF_OP_LOGZ_FWD + 100, T, B, &ltag, st);
        if (launch_persist_sweep(0, 1, score, noise, T, B, beta, nullptr, nullptr, pws, st, lease, ltag)) {
            lease_failed(ws);
            set_error("persistent sweep could not be enqueued (too many chain chunks for this device)"); return SEMICRF_ELAUNCH;
        }
    } else {
        lease_failed(ws);
        launch_rowseq_sweep(0, 1, score, noise, T, B, beta, nullptr, nullptr, st);
    }
    SEMICRF_CHECK_LAUNCH("semicrf_beta");
    return SEMICRF_OK;
}

int semicrf_viterbi(const float* score, const float* noise, int T, int B, const int32_t* start, int forward,
                    int32_t* pairs, int64_t cap, int32_t* offsets, void* ws, size_t ws_bytes,
                    semicrf_stream_t stream)
{
    if (int rc = check_common(score, noise, T, B)) return rc;
    SEMICRF_CHECK_ARG(pairs && offsets && cap >= 0, "pairs/offsets must be non-NULL");
    SEMICRF_CHECK_ARG((long long)B * 2 * T < (1ll << 31), "B*2T exceeds int32 offsets");
    if (int rc = sweep_prologue("semicrf_viterbi", (hipStream_t)stream)) return rc;
    Carver cv(ws, ws_bytes);
    float* u = cv.take<float>((size_t)T * B);
    int* code = cv.take<int>((size_t)T * B);
    int* region = cv.take<int>((size_t)B * 2 * T * 2);
    int* counts = cv.take<int>((size_t)B);
    const bool fast = use_persist(T, B);
    void* pws = fast ? cv.take<char>(persist_workspace_bytes(T, B)) : nullptr;
    if (!cv.ok || !ws) { set_error("workspace too small for viterbi"); return SEMICRF_EWORKSPACE; }
    hipStream_t st = (hipStream_t)stream;
    if (fast) {
        unsigned ltag = 0;
        const int lease = lease_acquire(ws, SEMICRF_OP_VITERBI, T, B, &ltag, st);
        if (launch_persist_sweep(1, forward ? 0 : 1, score, noise, T, B, nullptr, nullptr, code, pws, st, lease, ltag)) {
            lease_failed(ws);
            set_error("persistent sweep could not be enqueued (too many chain chunks for this device)"); return SEMICRF_ELAUNCH;
        }
    } else {
        lease_failed(ws);
        launch_rowseq_sweep(1, forward ? 0 : 1, score, noise, T, B, u, code, nullptr, st);
    }
    int nerr = 0, estride = 0;
    const unsigned* err = fast ? persist_error_words(pws, &nerr, &estride) : nullptr;
    launch_backtrack(code, T, B, start, forward ? 1 : 0, region, counts, pairs, (long long)cap, offsets, st, err, nerr, estride);
    SEMICRF_CHECK_LAUNCH("semicrf_viterbi");
    return SEMICRF_OK;
}

int semicrf_eval_path(const float* score, const float* noise, int T, int B, const int32_t* pairs, int64_t K,
                      const int32_t* offsets, float* out, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    if (int rc = check_common(score, noise, T, B)) return rc;
    SEMICRF_CHECK_ARG(offsets && out, "offsets/out must be non-NULL");
    SEMICRF_CHECK_ARG(K >= 0 && K < (1ll << 31) && (K == 0 || pairs), "bad interval count");
    (void)ws; (void)ws_bytes;
    launch_eval_path(score, noise, T, B, (int)K, pairs, offsets, out, (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("semicrf_eval_path");
    return SEMICRF_OK;
}

int semicrf_eval_path_bwd(const float* gout, int T, int B, const int32_t* pairs, int64_t K, const int32_t* offsets,
                          float* dScore, float* dNoise, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(T >= 1 && B >= 1, "T=%d, B=%d must be >= 1", T, B);
    SEMICRF_CHECK_ARG(gout && offsets, "gout/offsets must be non-NULL");
    SEMICRF_CHECK_ARG(K >= 0 && K < (1ll << 31) && (K == 0 || pairs), "bad interval count");
    launch_eval_path_bwd(gout, T, B, (int)K, pairs, offsets, dScore, dNoise, (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("semicrf_eval_path_bwd");
    return SEMICRF_OK;
}

static int check_slots(int C, int group, int pitch)
{
    SEMICRF_CHECK_ARG(group >= 1 && pitch >= group && C % group == 0, "bad slot layout: C=%d group=%d pitch=%d", C, group, pitch);
    SEMICRF_CHECK_ARG(pitch == group || pitch % 4 == 0, "a padded slot pitch must be a multiple of 4 (pitch=%d)", pitch);
    SEMICRF_CHECK_ARG((long long)(C / group) * pitch < (1ll << 31), "too many slots");
    return SEMICRF_OK;
}

int interval_score_fwd_pc(const float* q, const float* k, const float* diag, const float* rowc, int C, int T, int D, int64_t ldq,
                          int64_t ldk, int64_t ldd, int64_t ldrc, float qscale, int length_scaling, int full_square, int group,
                          int pitch, float* S, float* noise_out, semicrf_stream_t stream);

int interval_score_fwd_p(const float* q, const float* k, const float* diag, int C, int T, int D, int64_t ldq,
                         int64_t ldk, int64_t ldd, float qscale, int length_scaling, int full_square, int group, int pitch,
                         float* S, float* noise_out, semicrf_stream_t stream)
{
    return interval_score_fwd_pc(q, k, diag, nullptr, C, T, D, ldq, ldk, ldd, 1, qscale, length_scaling, full_square, group, pitch, S,
                                 noise_out, stream);
}

int interval_score_fwd_pc(const float* q, const float* k, const float* diag, const float* rowc, int C, int T, int D, int64_t ldq,
                          int64_t ldk, int64_t ldd, int64_t ldrc, float qscale, int length_scaling, int full_square, int group,
                          int pitch, float* S, float* noise_out, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(!rowc || ldrc >= 1, "bad row-constant stride");
    SEMICRF_CHECK_ARG(C >= 1 && T >= 1 && D >= 1, "C=%d T=%d D=%d must be >= 1", C, T, D);
    SEMICRF_CHECK_ARG(q && k && diag && S, "q/k/diag/S must be non-NULL");
    SEMICRF_CHECK_ARG(ldq >= D && ldk >= D && ldd >= 1, "bad leading dimensions");
    SEMICRF_CHECK_ARG(length_scaling >= 0 && length_scaling <= 2, "bad length_scaling %d", length_scaling);
    SEMICRF_CHECK_ARG(full_square >= 0 && (full_square & 3) <= 2 && (full_square & ~7) == 0, "bad full_square %d", full_square);
    if (int rc = check_slots(C, group, pitch)) return rc;
    if (pitch == group) group = pitch = C;                     // no ghosts: ONE group (quads must not straddle a group's end)
    const bool slots = pitch != group || rowc != nullptr;
    const int Cs = (C / group) * pitch;
    if (slots)
        SEMICRF_CHECK_ARG(g_impl.load() == 0 && interval_score_slots_supported(C, T, D, q, k, ldq, ldk),
                          "a padded slot layout / a row constant needs the shared-operand kernels: 16-byte aligned rows, D %% 64 == 0, T >= 128");
    hipStream_t st = (hipStream_t)stream;
    const int prec = (full_square & SEMICRF_SCORE_BF16X3) ? 1 : 0;   // opt-in: three-limb bf16 contraction (scorer_mfma.hip)
    full_square &= 3;
    if (full_square == 0) launch_zero_upper(S, T, Cs, st);     // begin > end: defined (zero), half the bytes of a full fill
    const int full_kernel = full_square == 1 ? 1 : 0;           // 2: lower triangle only, the rest of S is left as it is
    if (g_impl.load() == 0 && interval_score_mfma_supported(C, T, D)) {
        const int lrc = launch_interval_score_mfma(q, k, diag, C, T, D, ldq, ldk, ldd, qscale, length_scaling, full_kernel, S, st, prec, group,
                                                   pitch, rowc, ldrc);
        if (lrc == 2) {
            set_error("interval_score_fwd: a padded slot layout / a row constant needs the tiled kernels (D <= 256, D %% 64 == 0, T >= 128, "
                      "16-byte aligned rows, 32 T Cs floats within 32-bit offsets)");
            return SEMICRF_EINVAL;
        }
        if (lrc != 0) {
            set_error("interval_score_fwd: work list allocation failed");
            return SEMICRF_ELAUNCH;
        }
    } else {
        SEMICRF_CHECK_ARG(!slots, "a padded slot layout needs the shared-operand kernels");
        launch_interval_score_naive(q, k, diag, C, T, D, ldq, ldk, ldd, qscale, length_scaling, full_kernel, S, st);
    }
    if (noise_out && T > 1) {
        if (hipMemsetAsync(noise_out, 0, (size_t)(T - 1) * Cs * sizeof(float), st) != hipSuccess) {
            set_error("hipMemsetAsync failed");
            return SEMICRF_ELAUNCH;
        }
    }
    SEMICRF_CHECK_LAUNCH("interval_score_fwd");
    return SEMICRF_OK;
}

int interval_score_fwd(const float* q, const float* k, const float* diag, int C, int T, int D, int64_t ldq,
                       int64_t ldk, int64_t ldd, float qscale, int length_scaling, int full_square, float* S,
                       float* noise_out, semicrf_stream_t stream)
{
    return interval_score_fwd_p(q, k, diag, C, T, D, ldq, ldk, ldd, qscale, length_scaling, full_square, C > 0 ? C : 1, C > 0 ? C : 1, S,
                                noise_out, stream);
}

int interval_score_bwd(const float* dS, const float* q, const float* k, int C, int T, int D, int64_t ldq, int64_t ldk,
                       float qscale, int length_scaling, float* dq, float* dk, float* ddiag, int64_t lddq,
                       int64_t lddk, int64_t lddd, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(C >= 1 && T >= 1 && D >= 1, "C=%d T=%d D=%d must be >= 1", C, T, D);
    SEMICRF_CHECK_ARG(dS && q && k, "dS/q/k must be non-NULL");
    SEMICRF_CHECK_ARG(ldq >= D && ldk >= D && (!dq || lddq >= D) && (!dk || lddk >= D) && (!ddiag || lddd >= 1),
                      "bad leading dimensions");
    SEMICRF_CHECK_ARG(length_scaling >= 0 && length_scaling <= 2, "bad length_scaling %d", length_scaling);
    SEMICRF_CHECK_ARG(interval_score_bwd_supported(C, T, D), "interval_score_bwd needs D %% 32 == 0 and D <= 256 (D=%d)", D);
    launch_interval_score_bwd(dS, q, k, C, T, D, ldq, ldk, qscale, length_scaling, dq, dk, ddiag, lddq, lddk, lddd,
                              (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("interval_score_bwd");
    return SEMICRF_OK;
}

size_t interval_score_bwd_workspace_bytes(int C, int T, int D) { return interval_score_bwd_ws_bytes(C, T, D); }

int interval_score_bwd_ws_pc(const float* dS, const float* q, const float* k, int C, int T, int D, int64_t ldq, int64_t ldk,
                             float qscale, int length_scaling, int group, int pitch, float* dq, float* dk, float* ddiag, float* drowc,
                             int64_t lddq, int64_t lddk, int64_t lddd, int64_t lddrc, void* ws, size_t ws_bytes, semicrf_stream_t stream);

int interval_score_bwd_ws_p(const float* dS, const float* q, const float* k, int C, int T, int D, int64_t ldq, int64_t ldk,
                            float qscale, int length_scaling, int group, int pitch, float* dq, float* dk, float* ddiag, int64_t lddq,
                            int64_t lddk, int64_t lddd, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    return interval_score_bwd_ws_pc(dS, q, k, C, T, D, ldq, ldk, qscale, length_scaling, group, pitch, dq, dk, ddiag, nullptr, lddq, lddk, lddd,
                                    1, ws, ws_bytes, stream);
}

int interval_score_bwd_ws_pc(const float* dS, const float* q, const float* k, int C, int T, int D, int64_t ldq, int64_t ldk,
                             float qscale, int length_scaling, int group, int pitch, float* dq, float* dk, float* ddiag, float* drowc,
                             int64_t lddq, int64_t lddk, int64_t lddd, int64_t lddrc, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(!drowc || lddrc >= 1, "bad row-constant stride");
    SEMICRF_CHECK_ARG(C >= 1 && T >= 1 && D >= 1, "C=%d T=%d D=%d must be >= 1", C, T, D);
    SEMICRF_CHECK_ARG(dS && q && k, "dS/q/k must be non-NULL");
    SEMICRF_CHECK_ARG(ldq >= D && ldk >= D && (!dq || lddq >= D) && (!dk || lddk >= D) && (!ddiag || lddd >= 1),
                      "bad leading dimensions");
    const int prec = (length_scaling & SEMICRF_LEN_BF16X3) ? 1 : 0;   // opt-in: the two products on the three-limb bf16 kernels (scorer_bwd_gemm.hip)
    length_scaling &= ~SEMICRF_LEN_BF16X3;
    SEMICRF_CHECK_ARG(length_scaling >= 0 && length_scaling <= 2, "bad length_scaling %d", length_scaling);
    SEMICRF_CHECK_ARG(interval_score_bwd_supported(C, T, D), "interval_score_bwd needs D %% 32 == 0 and D <= 256 (D=%d)", D);
    if (int rc = check_slots(C, group, pitch)) return rc;
    if (pitch == group) group = pitch = C;                     // no ghosts: ONE group (quads must not straddle a group's end)
    const bool slots = pitch != group;
    hipStream_t st = (hipStream_t)stream;
    if (g_impl.load() == 0 && (dq || dk) &&
        launch_interval_score_bwd_packed(dS, q, k, C, T, D, ldq, ldk, qscale, length_scaling, dq, dk, lddq, lddk, ws, ws_bytes, st, nullptr,
                                         group, pitch, drowc, lddrc, prec)) {
        if (ddiag) launch_interval_score_bwd_diag(dS, nullptr, ddiag, C, T, lddd, group, pitch, st);      // (drowc: out of the dq GEMM)
    } else {
        SEMICRF_CHECK_ARG(!slots, "a padded slot layout needs the packed path (workspace, D in {64,128,256}, T >= 64, aligned rows)");
        launch_interval_score_bwd(dS, q, k, C, T, D, ldq, ldk, qscale, length_scaling, dq, dk, ddiag, lddq, lddk, lddd, st);
        if (drowc) launch_interval_score_bwd_rowsum(dS, nullptr, drowc, lddrc, C, T, qscale, length_scaling, group, pitch, st);
    }
    SEMICRF_CHECK_LAUNCH("interval_score_bwd_ws");
    return SEMICRF_OK;
}

int interval_score_bwd_ws(const float* dS, const float* q, const float* k, int C, int T, int D, int64_t ldq, int64_t ldk,
                          float qscale, int length_scaling, float* dq, float* dk, float* ddiag, int64_t lddq,
                          int64_t lddk, int64_t lddd, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    return interval_score_bwd_ws_p(dS, q, k, C, T, D, ldq, ldk, qscale, length_scaling, C > 0 ? C : 1, C > 0 ? C : 1, dq, dk, ddiag, lddq, lddk,
                                   lddd, ws, ws_bytes, stream);
}

int interval_score_bwd_fused(const float* S, const float* alpha, const float* beta, const float* logZ,
                             const float* gout, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                             int64_t ldk, float qscale, int length_scaling, float* dq, float* dk, float* ddiag,
                             int64_t lddq, int64_t lddk, int64_t lddd, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(C >= 1 && T >= 1 && D >= 1, "C=%d T=%d D=%d must be >= 1", C, T, D);
    SEMICRF_CHECK_ARG(S && alpha && beta && logZ && gout && q && k, "S/alpha/beta/logZ/gout/q/k must be non-NULL");
    SEMICRF_CHECK_ARG(ldq >= D && ldk >= D && (!dq || lddq >= D) && (!dk || lddk >= D) && (!ddiag || lddd >= 1),
                      "bad leading dimensions");
    SEMICRF_CHECK_ARG(length_scaling >= 0 && length_scaling <= 2, "bad length_scaling %d", length_scaling);
    SEMICRF_CHECK_ARG(interval_score_bwd_supported(C, T, D), "interval_score_bwd_fused needs D %% 32 == 0 and D <= 256 (D=%d)", D);
    launch_interval_score_bwd_fused(S, alpha, beta, logZ, gout, q, k, C, T, D, ldq, ldk, qscale, length_scaling, dq, dk,
                                    ddiag, lddq, lddk, lddd, (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("interval_score_bwd_fused");
    return SEMICRF_OK;
}

int interval_score_bwd_fused_ws_pc(const float* S, const float* alpha, const float* beta, const float* logZ,
                                   const float* gout, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                                   int64_t ldk, float qscale, int length_scaling, int group, int pitch, float* dq, float* dk, float* ddiag,
                                   float* drowc, int64_t lddq, int64_t lddk, int64_t lddd, int64_t lddrc, void* ws, size_t ws_bytes,
                                   semicrf_stream_t stream);

int interval_score_bwd_fused_ws_p(const float* S, const float* alpha, const float* beta, const float* logZ,
                                  const float* gout, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                                  int64_t ldk, float qscale, int length_scaling, int group, int pitch, float* dq, float* dk, float* ddiag,
                                  int64_t lddq, int64_t lddk, int64_t lddd, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    return interval_score_bwd_fused_ws_pc(S, alpha, beta, logZ, gout, q, k, C, T, D, ldq, ldk, qscale, length_scaling, group, pitch, dq, dk,
                                          ddiag, nullptr, lddq, lddk, lddd, 1, ws, ws_bytes, stream);
}

int interval_score_bwd_fused_ws_pc(const float* S, const float* alpha, const float* beta, const float* logZ,
                                   const float* gout, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                                   int64_t ldk, float qscale, int length_scaling, int group, int pitch, float* dq, float* dk, float* ddiag,
                                   float* drowc, int64_t lddq, int64_t lddk, int64_t lddd, int64_t lddrc, void* ws, size_t ws_bytes,
                                   semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(!drowc || lddrc >= 1, "bad row-constant stride");
    SEMICRF_CHECK_ARG(C >= 1 && T >= 1 && D >= 1, "C=%d T=%d D=%d must be >= 1", C, T, D);
    SEMICRF_CHECK_ARG(S && alpha && beta && logZ && gout && q && k, "S/alpha/beta/logZ/gout/q/k must be non-NULL");
    SEMICRF_CHECK_ARG(ldq >= D && ldk >= D && (!dq || lddq >= D) && (!dk || lddk >= D) && (!ddiag || lddd >= 1),
                      "bad leading dimensions");
    const int prec = (length_scaling & SEMICRF_LEN_BF16X3) ? 1 : 0;   // opt-in: the two products on the three-limb bf16 kernels (scorer_bwd_gemm.hip)
    length_scaling &= ~SEMICRF_LEN_BF16X3;
    SEMICRF_CHECK_ARG(length_scaling >= 0 && length_scaling <= 2, "bad length_scaling %d", length_scaling);
    SEMICRF_CHECK_ARG(interval_score_bwd_supported(C, T, D), "interval_score_bwd_fused needs D %% 32 == 0 and D <= 256 (D=%d)", D);
    if (int rc = check_slots(C, group, pitch)) return rc;
    if (pitch == group) group = pitch = C;                     // no ghosts: ONE group (quads must not straddle a group's end)
    const bool slots = pitch != group;
    hipStream_t st = (hipStream_t)stream;
    const float* fused[4] = {alpha, beta, logZ, gout};
    if (g_impl.load() == 0 && (dq || dk) &&
        launch_interval_score_bwd_packed(S, q, k, C, T, D, ldq, ldk, qscale, length_scaling, dq, dk, lddq, lddk, ws, ws_bytes, st, fused,
                                         group, pitch, drowc, lddrc, prec)) {
        if (ddiag) launch_interval_score_bwd_diag(S, fused, ddiag, C, T, lddd, group, pitch, st);
    } else {
        SEMICRF_CHECK_ARG(!slots, "a padded slot layout needs the packed path (workspace, D in {64,128,256}, T >= 64, aligned rows)");
        launch_interval_score_bwd_fused(S, alpha, beta, logZ, gout, q, k, C, T, D, ldq, ldk, qscale, length_scaling, dq, dk, ddiag, lddq,
                                        lddk, lddd, st);
        if (drowc) launch_interval_score_bwd_rowsum(S, fused, drowc, lddrc, C, T, qscale, length_scaling, group, pitch, st);
    }
    SEMICRF_CHECK_LAUNCH("interval_score_bwd_fused_ws");
    return SEMICRF_OK;
}

int interval_score_bwd_fused_ws(const float* S, const float* alpha, const float* beta, const float* logZ,
                                const float* gout, const float* q, const float* k, int C, int T, int D, int64_t ldq,
                                int64_t ldk, float qscale, int length_scaling, float* dq, float* dk, float* ddiag,
                                int64_t lddq, int64_t lddk, int64_t lddd, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    return interval_score_bwd_fused_ws_p(S, alpha, beta, logZ, gout, q, k, C, T, D, ldq, ldk, qscale, length_scaling, C > 0 ? C : 1,
                                         C > 0 ? C : 1, dq, dk, ddiag, lddq, lddk, lddd, ws, ws_bytes, stream);
}

int interval_score_path_bwd_pc(const float* gout, const int32_t* pairs, int64_t K, const int32_t* offsets, const float* q,
                               const float* k, int C, int T, int D, int64_t ldq, int64_t ldk, float qscale, int length_scaling,
                               int group, int pitch, float* dq, float* dk, float* ddiag, float* drowc, int64_t lddq, int64_t lddk,
                               int64_t lddd, int64_t lddrc, semicrf_stream_t stream);

int interval_score_path_bwd_p(const float* gout, const int32_t* pairs, int64_t K, const int32_t* offsets, const float* q,
                              const float* k, int C, int T, int D, int64_t ldq, int64_t ldk, float qscale, int length_scaling,
                              int group, int pitch, float* dq, float* dk, float* ddiag, int64_t lddq, int64_t lddk, int64_t lddd,
                              semicrf_stream_t stream)
{
    return interval_score_path_bwd_pc(gout, pairs, K, offsets, q, k, C, T, D, ldq, ldk, qscale, length_scaling, group, pitch, dq, dk, ddiag,
                                      nullptr, lddq, lddk, lddd, 1, stream);
}

int interval_score_path_bwd_pc(const float* gout, const int32_t* pairs, int64_t K, const int32_t* offsets, const float* q,
                               const float* k, int C, int T, int D, int64_t ldq, int64_t ldk, float qscale, int length_scaling,
                               int group, int pitch, float* dq, float* dk, float* ddiag, float* drowc, int64_t lddq, int64_t lddk,
                               int64_t lddd, int64_t lddrc, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(!drowc || lddrc >= 1, "bad row-constant stride");
    SEMICRF_CHECK_ARG(C >= 1 && T >= 1 && D >= 1, "C=%d T=%d D=%d must be >= 1", C, T, D);
    SEMICRF_CHECK_ARG(gout && offsets && q && k, "gout/offsets/q/k must be non-NULL");
    SEMICRF_CHECK_ARG(K >= 0 && K < (1ll << 31) && (K == 0 || pairs), "bad interval count");
    SEMICRF_CHECK_ARG(ldq >= D && ldk >= D && (!dq || lddq >= D) && (!dk || lddk >= D) && (!ddiag || lddd >= 1),
                      "bad leading dimensions");
    SEMICRF_CHECK_ARG(length_scaling >= 0 && length_scaling <= 2, "bad length_scaling %d", length_scaling);
    if (int rc = check_slots(C, group, pitch)) return rc;
    if (pitch == group) group = pitch = C;
    launch_interval_score_path_bwd(gout, pairs, (int)K, offsets, q, k, C, T, D, ldq, ldk, qscale, length_scaling, dq, dk, ddiag,
                                   lddq, lddk, lddd, (hipStream_t)stream, group, pitch, drowc, lddrc);
    SEMICRF_CHECK_LAUNCH("interval_score_path_bwd");
    return SEMICRF_OK;
}

int interval_score_path_bwd(const float* gout, const int32_t* pairs, int64_t K, const int32_t* offsets, const float* q,
                            const float* k, int C, int T, int D, int64_t ldq, int64_t ldk, float qscale, int length_scaling,
                            float* dq, float* dk, float* ddiag, int64_t lddq, int64_t lddk, int64_t lddd,
                            semicrf_stream_t stream)
{
    return interval_score_path_bwd_p(gout, pairs, K, offsets, q, k, C, T, D, ldq, ldk, qscale, length_scaling, C > 0 ? C : 1, C > 0 ? C : 1,
                                     dq, dk, ddiag, lddq, lddk, lddd, stream);
}

int scorer_proj_nn(const float* A, int64_t lda, int64_t M, int K, const float* B, int64_t ldb, int N, float* out, int64_t ldout,
                   const float* bias, const float* w2, const float* b2, int zero_cols, int accumulate, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(A && B && out, "A/B/out must be non-NULL");
    SEMICRF_CHECK_ARG(M >= 1 && K >= 4 && lda >= K && ldb >= N && zero_cols >= 0 && (!w2 || b2), "bad sizes");
    SEMICRF_CHECK_ARG(ldout >= N + (w2 ? 2 + zero_cols : 0), "ldout too small for the packed output");
    SEMICRF_CHECK_ARG(launch_proj_nn(A, lda, M, K, B, ldb, N, out, ldout, bias, w2, b2, zero_cols, accumulate, (hipStream_t)stream) == 0,
                      "scorer_proj_nn: N must be 64, 128 or 256, K %% 4 == 0, rows 16-byte aligned, M * ld * 4 < 2^31 (N=%d K=%d)", N, K);
    SEMICRF_CHECK_LAUNCH("scorer_proj_nn");
    return SEMICRF_OK;
}

size_t scorer_proj_nn3_workspace_bytes(int K, int N) { return proj_nn3_workspace_bytes(K, N); }

int scorer_proj_nn3(const float* A, int64_t lda, int64_t M, int K, const float* B, int64_t ldb, int N, float* out, int64_t ldout,
                    const float* bias, const float* w2, const float* b2, int zero_cols, int accumulate, void* ws, size_t ws_bytes,
                    semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(A && B && out, "A/B/out must be non-NULL");
    SEMICRF_CHECK_ARG(M >= 1 && K >= 4 && lda >= K && ldb >= N && zero_cols >= 0 && (!w2 || b2), "bad sizes");
    SEMICRF_CHECK_ARG(ldout >= N + (w2 ? 2 + zero_cols : 0), "ldout too small for the packed output");
    if (launch_proj_nn3(A, lda, M, K, B, ldb, N, out, ldout, bias, w2, b2, zero_cols, accumulate, ws, ws_bytes, (hipStream_t)stream) != 0)
        return scorer_proj_nn(A, lda, M, K, B, ldb, N, out, ldout, bias, w2, b2, zero_cols, accumulate, stream);      // not this kernel's shape: exact
    SEMICRF_CHECK_LAUNCH("scorer_proj_nn3");
    return SEMICRF_OK;
}

size_t scorer_proj_tn_workspace_bytes(int64_t M, int R, int N) { return M >= 1 && R >= 1 ? proj_tn_workspace_bytes(M, R, N) : 0; }

int scorer_proj_tn(const float* dy, int64_t lddy, int64_t M, int R, int extra_col0, int total_rows, const float* x, int64_t ldx, int N,
                   float* dW, int64_t lddw, float* db, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(dy && x && dW, "dy/x/dW must be non-NULL");
    const int x3flag = total_rows & SEMICRF_PROJ_TN_BF16X3;       // opt-in: the matrix part on the three-limb bf16 kernel
    total_rows &= ~SEMICRF_PROJ_TN_BF16X3;
    SEMICRF_CHECK_ARG(M >= 1 && R >= 1 && total_rows >= R && lddy >= R && ldx >= N && lddw >= N, "bad sizes");
    SEMICRF_CHECK_ARG(extra_col0 < 0 || (extra_col0 >= R && extra_col0 + 2 <= total_rows && extra_col0 + 2 <= lddy), "bad extra columns");
    const int rc = launch_proj_tn(dy, lddy, M, R, extra_col0, total_rows | x3flag, x, ldx, N, dW, lddw, db, ws, ws_bytes, (hipStream_t)stream);
    if (rc == 2) { set_error("scorer_proj_tn: workspace missing or too small"); return SEMICRF_EWORKSPACE; }
    SEMICRF_CHECK_ARG(rc == 0, "scorer_proj_tn: N must be 64, 128 or 256, rows 16-byte aligned, M * ld * 4 < 2^31 (N=%d)", N);
    SEMICRF_CHECK_LAUNCH("scorer_proj_tn");
    return SEMICRF_OK;
}

int scorer_merge_weights_fwd(const float* W, const float* bias, int D, int size, int rows, float* Wm, float* bm, float* WmT,
                             semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(W && bias && Wm && bm, "W/bias/Wm/bm must be non-NULL");
    SEMICRF_CHECK_ARG(D >= 1 && size >= 1 && size <= 256 && rows >= size + 2, "scorer_merge_weights: D=%d size=%d (<= 256) rows=%d (>= size + 2)", D, size, rows);
    launch_merge_weights_fwd(W, bias, D, size, rows, Wm, bm, WmT, (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("scorer_merge_weights_fwd");
    return SEMICRF_OK;
}

int scorer_stage_linear(const float* W, const float* bias, int D, int size, int rows_pad, float* BT, float* Wqd, float* w2, float* b2,
                        semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(W && bias && BT && Wqd && w2 && b2, "W/bias/BT/Wqd/w2/b2 must be non-NULL");
    SEMICRF_CHECK_ARG(D >= 1 && size >= 1 && rows_pad >= D + 1 && rows_pad < (1 << 20), "scorer_stage_linear: D=%d size=%d rows_pad=%d (>= D + 1)", D, size, rows_pad);
    launch_stage_linear(W, bias, D, size, rows_pad, BT, Wqd, w2, b2, (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("scorer_stage_linear");
    return SEMICRF_OK;
}

size_t scorer_merge_weights_bwd_workspace_bytes(int size) { return size >= 1 ? merge_weights_bwd_workspace_bytes(size) : 0; }

int scorer_merge_weights_bwd(const float* W, const float* bias, const float* dWm, const float* dbm, int D, int size, int rows, float* dW,
                             float* dbias, void* ws, size_t ws_bytes, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(W && bias && dWm && dbm && dW && dbias, "W/bias/dWm/dbm/dW/dbias must be non-NULL");
    SEMICRF_CHECK_ARG(D >= 1 && size >= 1 && size <= 256 && rows >= size + 2, "scorer_merge_weights: D=%d size=%d (<= 256) rows=%d (>= size + 2)", D, size, rows);
    if (!ws || ws_bytes < merge_weights_bwd_workspace_bytes(size) || ((uintptr_t)ws & 3)) { set_error("scorer_merge_weights_bwd: workspace missing or too small"); return SEMICRF_EWORKSPACE; }
    launch_merge_weights_bwd(W, bias, dWm, dbm, D, size, dW, dbias, (float*)ws, (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("scorer_merge_weights_bwd");
    return SEMICRF_OK;
}

int interval_features_gather(const float* ctx, int C, int T, int D, int64_t ldc, const int32_t* pairs, int64_t K,
                             const int32_t* offsets, int nSym, float* out, int64_t* symIdx, int64_t* scatterIdx,
                             semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(C >= 1 && T >= 1 && D >= 1 && nSym >= 1, "C=%d T=%d D=%d nSym=%d must be >= 1", C, T, D, nSym);
    SEMICRF_CHECK_ARG(ctx && offsets, "ctx/offsets must be non-NULL");
    SEMICRF_CHECK_ARG(ldc >= D, "bad row stride");
    SEMICRF_CHECK_ARG(K >= 0 && K < (1ll << 31) && (K == 0 || (pairs && out)), "bad interval count / buffers");
    launch_interval_features(ctx, C, T, D, ldc, pairs, (int)K, offsets, nSym, out, (long long*)symIdx, (long long*)scatterIdx,
                             (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("interval_features_gather");
    return SEMICRF_OK;
}

int interval_features_gather_bwd(const float* gout, const float* ctx, int C, int T, int D, int64_t ldc, const int32_t* pairs,
                                 int64_t K, const int32_t* offsets, float* dctx, int64_t lddc, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(C >= 1 && T >= 1 && D >= 1, "C=%d T=%d D=%d must be >= 1", C, T, D);
    SEMICRF_CHECK_ARG(ctx && offsets && dctx, "ctx/offsets/dctx must be non-NULL");
    SEMICRF_CHECK_ARG(ldc >= D && lddc >= D, "bad row stride");
    SEMICRF_CHECK_ARG(K >= 0 && K < (1ll << 31) && (K == 0 || (pairs && gout)), "bad interval count / buffers");
    launch_interval_features_bwd(gout, ctx, C, T, D, ldc, pairs, (int)K, offsets, dctx, lddc, (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("interval_features_gather_bwd");
    return SEMICRF_OK;
}

int segment_onset_filter(const int32_t* pairs, const int32_t* offsets, int B, int bound, int32_t* pairs_out, int64_t cap,
                         int32_t* offsets_out, int32_t* counts_ws, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(B >= 1, "B=%d must be >= 1", B);
    SEMICRF_CHECK_ARG(offsets && offsets_out && counts_ws && cap >= 0 && (cap == 0 || (pairs && pairs_out)), "NULL buffer");
    launch_onset_filter(pairs, offsets, B, bound, pairs_out, (long long)cap, offsets_out, counts_ws, (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("segment_onset_filter");
    return SEMICRF_OK;
}

int segment_events(const int32_t* pairs, int64_t K, const int32_t* offsets, int B, int nSym, const float* ofValue,
                   const unsigned char* ofPresence, int lastFrameIdx, double frameDur, const double* beginTime, int stepFrames,
                   double* times, unsigned char* flags, int32_t* lastP, int32_t* nextStart, semicrf_stream_t stream)
{
    SEMICRF_CHECK_ARG(B >= 1 && nSym >= 1 && B % nSym == 0, "B=%d must be a positive multiple of nSym=%d", B, nSym);
    SEMICRF_CHECK_ARG(offsets && beginTime && lastP && nextStart, "offsets/beginTime/lastP/nextStart must be non-NULL");
    SEMICRF_CHECK_ARG(K >= 0 && (K == 0 || (pairs && ofValue && ofPresence && times && flags)), "bad interval count / buffers");
    launch_segment_events(pairs, offsets, B, nSym, ofValue, ofPresence, lastFrameIdx, frameDur, beginTime, stepFrames, times, flags,
                          lastP, nextStart, (hipStream_t)stream);
    SEMICRF_CHECK_LAUNCH("segment_events");
    return SEMICRF_OK;
}

}  // extern "C"
