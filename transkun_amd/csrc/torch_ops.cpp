// torch_ops.cpp -- LibTorch stable-ABI shim: registers the C ABI of include/semicrf_hip.h as torch ops (namespace
// `semicrf`, dispatch key CUDA = HIP tensors on ROCm), so that the Python mirror of the reference class calls
// torch.ops.semicrf.* -- dispatcher, stream and device handling by torch -- instead of ctypes.  Built into its own
// library (libsemicrf_torch.so) that links libsemicrf_hip.so: the C ABI itself stays free of torch.
//
// Only torch/csrc/stable/* and the aoti C shim are used (no ATen/c10 C++ ABI): the binary does not depend on the
// libtorch C++ ABI of the build.  Every op runs on the tensors' device (device guard) and enqueues on torch's current
// stream of that device; outputs and the workspace are allocated by the caller (the Python mirror) and passed in.
#define USE_ROCM 1
#include <torch/csrc/stable/accelerator.h>
#include <torch/csrc/stable/library.h>
#include <torch/csrc/stable/tensor.h>

#include "../../include/semicrf_hip.h"

using torch::stable::Tensor;

namespace {

struct Ctx {
    torch::stable::accelerator::DeviceGuard guard;
    void* stream = nullptr;
    explicit Ctx(const Tensor& t) : guard(t.get_device_index())
    {
        STD_TORCH_CHECK(t.is_cuda(), "semicrf: tensors must live on the GPU (there is no CPU path)");
        TORCH_ERROR_CODE_CHECK(aoti_torch_get_current_cuda_stream(t.get_device_index(), &stream));
    }
};

inline void same_device(const Tensor& a, const Tensor& b)
{
    STD_TORCH_CHECK(b.is_cuda() && a.get_device_index() == b.get_device_index(), "semicrf: all tensors of a call must share one device");
}
inline void check(int rc, const char* what)
{
    STD_TORCH_CHECK(rc == SEMICRF_OK, what, " failed (code ", rc, "): ", semicrf_last_error());
}
inline float* fp(const Tensor& t) { return t.defined() && t.numel() > 0 ? (float*)t.data_ptr() : nullptr; }
inline const float* cfp(const Tensor& t) { return t.defined() && t.numel() > 0 ? (const float*)t.data_ptr() : nullptr; }
inline int32_t* ip(const Tensor& t) { return t.defined() && t.numel() > 0 ? (int32_t*)t.data_ptr() : nullptr; }

// ---- semi-CRF --------------------------------------------------------------------------------------------------------
void logz_fwd(Tensor score, Tensor noise, Tensor logZ, Tensor v, bool want_v, Tensor ws)
{
    Ctx c(score); same_device(score, noise); same_device(score, logZ); same_device(score, ws);
    const int T = (int)score.size(0), B = (int)score.size(2);
    check(semicrf_logz_fwd(cfp(score), cfp(noise), T, B, fp(logZ), want_v ? fp(v) : nullptr, ws.data_ptr(), (size_t)ws.numel(), c.stream),
          "semicrf_logz_fwd");
}
void logz_bwd(Tensor score, Tensor noise, Tensor v, Tensor logZ, Tensor gout, Tensor dScore, Tensor dNoise, Tensor q, bool want_q,
              Tensor ws)
{
    Ctx c(score); same_device(score, noise); same_device(score, dScore); same_device(score, ws);
    const int T = (int)score.size(0), B = (int)score.size(2);
    check(semicrf_logz_bwd(cfp(score), cfp(noise), cfp(v), cfp(logZ), cfp(gout), T, B, fp(dScore), fp(dNoise), want_q ? fp(q) : nullptr,
                           ws.data_ptr(), (size_t)ws.numel(), c.stream),
          "semicrf_logz_bwd");
}
void beta(Tensor score, Tensor noise, Tensor out, Tensor ws)
{
    Ctx c(score); same_device(score, noise); same_device(score, out);
    const int T = (int)score.size(0), B = (int)score.size(2);
    check(semicrf_beta(cfp(score), cfp(noise), T, B, fp(out), ws.data_ptr(), (size_t)ws.numel(), c.stream), "semicrf_beta");
}
void viterbi(Tensor score, Tensor noise, Tensor start, bool has_start, bool forward, Tensor pairs, Tensor offsets, Tensor ws)
{
    Ctx c(score); same_device(score, noise); same_device(score, pairs); same_device(score, offsets);
    const int T = (int)score.size(0), B = (int)score.size(2);
    check(semicrf_viterbi(cfp(score), cfp(noise), T, B, has_start ? ip(start) : nullptr, forward ? 1 : 0, ip(pairs),
                          (int64_t)pairs.size(0), ip(offsets), ws.data_ptr(), (size_t)ws.numel(), c.stream),
          "semicrf_viterbi");
}
void eval_path(Tensor score, Tensor noise, Tensor pairs, int64_t K, Tensor offsets, Tensor out, Tensor ws)
{
    Ctx c(score); same_device(score, noise); same_device(score, pairs); same_device(score, out);
    const int T = (int)score.size(0), B = (int)score.size(2);
    check(semicrf_eval_path(cfp(score), cfp(noise), T, B, ip(pairs), K, ip(offsets), fp(out), ws.data_ptr(), (size_t)ws.numel(), c.stream),
          "semicrf_eval_path");
}
void eval_path_bwd(Tensor gout, int64_t T, int64_t B, Tensor pairs, int64_t K, Tensor offsets, Tensor dScore, bool has_ds, Tensor dNoise,
                   bool has_dn)
{
    Ctx c(gout); same_device(gout, pairs);
    check(semicrf_eval_path_bwd(cfp(gout), (int)T, (int)B, ip(pairs), K, ip(offsets), has_ds ? fp(dScore) : nullptr,
                                has_dn ? fp(dNoise) : nullptr, c.stream),
          "semicrf_eval_path_bwd");
}

// ---- interval scorer ---------------------------------------------------------------------------------------------------
void interval_score_fwd_op(Tensor q, Tensor k, Tensor diag, int64_t C, int64_t T, int64_t D, int64_t ldq, int64_t ldk, int64_t ldd,
                           double qscale, int64_t mode, int64_t full, Tensor S, Tensor noise)
{
    Ctx c(q); same_device(q, k); same_device(q, diag); same_device(q, S);
    check(interval_score_fwd(cfp(q), cfp(k), cfp(diag), (int)C, (int)T, (int)D, ldq, ldk, ldd, (float)qscale, (int)mode, (int)full, fp(S),
                             fp(noise), c.stream),
          "interval_score_fwd");
}
void interval_score_bwd_ws_op(Tensor dS, Tensor q, Tensor k, int64_t C, int64_t T, int64_t D, int64_t ldq, int64_t ldk, double qscale,
                              int64_t mode, Tensor dq, Tensor dk, Tensor ddiag, int64_t lddq, int64_t lddk, int64_t lddd, Tensor ws)
{
    Ctx c(dS); same_device(dS, q); same_device(dS, k); same_device(dS, dq);
    check(interval_score_bwd_ws(cfp(dS), cfp(q), cfp(k), (int)C, (int)T, (int)D, ldq, ldk, (float)qscale, (int)mode, fp(dq), fp(dk), fp(ddiag),
                                lddq, lddk, lddd, ws.numel() > 0 ? ws.data_ptr() : nullptr, (size_t)ws.numel(), c.stream),
          "interval_score_bwd_ws");
}
void interval_score_bwd_fused_ws_op(Tensor S, Tensor alpha, Tensor beta_, Tensor logZ, Tensor gout, Tensor q, Tensor k, int64_t C, int64_t T,
                                    int64_t D, int64_t ldq, int64_t ldk, double qscale, int64_t mode, Tensor dq, Tensor dk, Tensor ddiag,
                                    int64_t lddq, int64_t lddk, int64_t lddd, Tensor ws)
{
    Ctx c(S); same_device(S, q); same_device(S, k); same_device(S, dq);
    check(interval_score_bwd_fused_ws(cfp(S), cfp(alpha), cfp(beta_), cfp(logZ), cfp(gout), cfp(q), cfp(k), (int)C, (int)T, (int)D, ldq, ldk,
                                      (float)qscale, (int)mode, fp(dq), fp(dk), fp(ddiag), lddq, lddk, lddd,
                                      ws.numel() > 0 ? ws.data_ptr() : nullptr, (size_t)ws.numel(), c.stream),
          "interval_score_bwd_fused_ws");
}
void interval_score_path_bwd_op(Tensor gout, Tensor pairs, int64_t K, Tensor offsets, Tensor q, Tensor k, int64_t C, int64_t T, int64_t D,
                                int64_t ldq, int64_t ldk, double qscale, int64_t mode, Tensor dq, Tensor dk, Tensor ddiag, int64_t lddq,
                                int64_t lddk, int64_t lddd)
{
    Ctx c(gout); same_device(gout, q); same_device(gout, dq);
    check(interval_score_path_bwd(cfp(gout), ip(pairs), K, ip(offsets), cfp(q), cfp(k), (int)C, (int)T, (int)D, ldq, ldk, (float)qscale,
                                  (int)mode, fp(dq), fp(dk), fp(ddiag), lddq, lddk, lddd, c.stream),
          "interval_score_path_bwd");
}

// ---- attribute-head features ------------------------------------------------------------------------------------------
void interval_features_gather_op(Tensor ctx, int64_t C, int64_t T, int64_t D, int64_t ldc, Tensor pairs, int64_t K, Tensor offsets,
                                 int64_t nSym, Tensor out, Tensor symIdx, Tensor scatterIdx)
{
    Ctx c(ctx); same_device(ctx, pairs); same_device(ctx, out);
    check(interval_features_gather(cfp(ctx), (int)C, (int)T, (int)D, ldc, ip(pairs), K, ip(offsets), (int)nSym, fp(out),
                                   (int64_t*)symIdx.data_ptr(), (int64_t*)scatterIdx.data_ptr(), c.stream),
          "interval_features_gather");
}
void interval_features_gather_bwd_op(Tensor gout, Tensor ctx, int64_t C, int64_t T, int64_t D, int64_t ldc, Tensor pairs, int64_t K,
                                     Tensor offsets, Tensor dctx, int64_t lddc)
{
    Ctx c(gout); same_device(gout, ctx); same_device(gout, dctx);
    check(interval_features_gather_bwd(cfp(gout), cfp(ctx), (int)C, (int)T, (int)D, ldc, ip(pairs), K, ip(offsets), fp(dctx), lddc, c.stream),
          "interval_features_gather_bwd");
}

// ---- transcription segment loop ----------------------------------------------------------------------------------------
void segment_onset_filter_op(Tensor pairs, Tensor offsets, int64_t B, int64_t bound, Tensor pairs_out, Tensor offsets_out, Tensor counts_ws)
{
    Ctx c(offsets); same_device(offsets, pairs_out); same_device(offsets, offsets_out);
    check(segment_onset_filter(ip(pairs), ip(offsets), (int)B, (int)bound, ip(pairs_out), pairs_out.numel() / 2, ip(offsets_out),
                               ip(counts_ws), c.stream),
          "segment_onset_filter");
}
void segment_events_op(Tensor pairs, int64_t K, Tensor offsets, int64_t B, int64_t nSym, Tensor ofValue, Tensor ofPresence,
                       int64_t lastFrameIdx, double frameDur, Tensor beginTime, int64_t stepFrames, Tensor times, Tensor flags, Tensor lastP,
                       Tensor nextStart)
{
    Ctx c(offsets); same_device(offsets, beginTime); same_device(offsets, lastP); same_device(offsets, nextStart);
    check(segment_events(ip(pairs), K, ip(offsets), (int)B, (int)nSym, cfp(ofValue),
                         K > 0 ? (const unsigned char*)ofPresence.data_ptr() : nullptr, (int)lastFrameIdx, frameDur,
                         (const double*)beginTime.data_ptr(), (int)stepFrames, K > 0 ? (double*)times.data_ptr() : nullptr,
                         K > 0 ? (unsigned char*)flags.data_ptr() : nullptr, ip(lastP), ip(nextStart), c.stream),
          "segment_events");
}

}  // namespace

STABLE_TORCH_LIBRARY(semicrf, m)
{
    m.def("logz_fwd(Tensor score, Tensor noise, Tensor(a!) logZ, Tensor(b!) v, bool want_v, Tensor(c!) ws) -> ()");
    m.def("logz_bwd(Tensor score, Tensor noise, Tensor v, Tensor logZ, Tensor gout, Tensor(a!) dScore, Tensor(b!) dNoise, Tensor(c!) q, "
          "bool want_q, Tensor(d!) ws) -> ()");
    m.def("beta(Tensor score, Tensor noise, Tensor(a!) out, Tensor(b!) ws) -> ()");
    m.def("viterbi(Tensor score, Tensor noise, Tensor start, bool has_start, bool forward, Tensor(a!) pairs, Tensor(b!) offsets, "
          "Tensor(c!) ws) -> ()");
    m.def("eval_path(Tensor score, Tensor noise, Tensor pairs, int K, Tensor offsets, Tensor(a!) out, Tensor(b!) ws) -> ()");
    m.def("eval_path_bwd(Tensor gout, int T, int B, Tensor pairs, int K, Tensor offsets, Tensor(a!) dScore, bool has_ds, Tensor(b!) dNoise, "
          "bool has_dn) -> ()");
    m.def("interval_score_fwd(Tensor q, Tensor k, Tensor diag, int C, int T, int D, int ldq, int ldk, int ldd, float qscale, int mode, "
          "int full, Tensor(a!) S, Tensor(b!) noise) -> ()");
    m.def("interval_score_bwd_ws(Tensor dS, Tensor q, Tensor k, int C, int T, int D, int ldq, int ldk, float qscale, int mode, Tensor(a!) dq, "
          "Tensor(b!) dk, Tensor(c!) ddiag, int lddq, int lddk, int lddd, Tensor(d!) ws) -> ()");
    m.def("interval_score_bwd_fused_ws(Tensor S, Tensor alpha, Tensor beta, Tensor logZ, Tensor gout, Tensor q, Tensor k, int C, int T, int D, "
          "int ldq, int ldk, float qscale, int mode, Tensor(a!) dq, Tensor(b!) dk, Tensor(c!) ddiag, int lddq, int lddk, int lddd, "
          "Tensor(d!) ws) -> ()");
    m.def("interval_score_path_bwd(Tensor gout, Tensor pairs, int K, Tensor offsets, Tensor q, Tensor k, int C, int T, int D, int ldq, int ldk, "
          "float qscale, int mode, Tensor(a!) dq, Tensor(b!) dk, Tensor(c!) ddiag, int lddq, int lddk, int lddd) -> ()");
    m.def("interval_features_gather(Tensor ctx, int C, int T, int D, int ldc, Tensor pairs, int K, Tensor offsets, int nSym, Tensor(a!) out, "
          "Tensor(b!) symIdx, Tensor(c!) scatterIdx) -> ()");
    m.def("interval_features_gather_bwd(Tensor gout, Tensor ctx, int C, int T, int D, int ldc, Tensor pairs, int K, Tensor offsets, "
          "Tensor(a!) dctx, int lddc) -> ()");
    m.def("segment_onset_filter(Tensor pairs, Tensor offsets, int B, int bound, Tensor(a!) pairs_out, Tensor(b!) offsets_out, "
          "Tensor(c!) counts_ws) -> ()");
    m.def("segment_events(Tensor pairs, int K, Tensor offsets, int B, int nSym, Tensor ofValue, Tensor ofPresence, int lastFrameIdx, "
          "float frameDur, Tensor beginTime, int stepFrames, Tensor(a!) times, Tensor(b!) flags, Tensor(c!) lastP, Tensor(d!) nextStart) -> ()");
}

STABLE_TORCH_LIBRARY_IMPL(semicrf, CUDA, m)
{
    m.impl("logz_fwd", TORCH_BOX(&logz_fwd));
    m.impl("logz_bwd", TORCH_BOX(&logz_bwd));
    m.impl("beta", TORCH_BOX(&beta));
    m.impl("viterbi", TORCH_BOX(&viterbi));
    m.impl("eval_path", TORCH_BOX(&eval_path));
    m.impl("eval_path_bwd", TORCH_BOX(&eval_path_bwd));
    m.impl("interval_score_fwd", TORCH_BOX(&interval_score_fwd_op));
    m.impl("interval_score_bwd_ws", TORCH_BOX(&interval_score_bwd_ws_op));
    m.impl("interval_score_bwd_fused_ws", TORCH_BOX(&interval_score_bwd_fused_ws_op));
    m.impl("interval_score_path_bwd", TORCH_BOX(&interval_score_path_bwd_op));
    m.impl("interval_features_gather", TORCH_BOX(&interval_features_gather_op));
    m.impl("interval_features_gather_bwd", TORCH_BOX(&interval_features_gather_bwd_op));
    m.impl("segment_onset_filter", TORCH_BOX(&segment_onset_filter_op));
    m.impl("segment_events", TORCH_BOX(&segment_events_op));
}
