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
#include "cpu_ops.h"

#include <vector>

using torch::stable::Tensor;
using torch::headeronly::ScalarType;

namespace {

// Every op is dispatcher-visible: arguments are checked here (dtype, contiguity, element counts against T / B / K), not only
// in the Python mirror -- a wrong dtype or a short buffer is an error, never an out-of-bounds access.
inline void want(const Tensor& t, ScalarType st, int64_t min_numel, const char* name)
{
    STD_TORCH_CHECK(t.defined(), "semicrf: `", name, "` is undefined");
    STD_TORCH_CHECK(t.scalar_type() == st, "semicrf: `", name, "` has the wrong dtype");
    STD_TORCH_CHECK(t.is_contiguous(), "semicrf: `", name, "` must be contiguous");
    STD_TORCH_CHECK(t.numel() >= min_numel, "semicrf: `", name, "` holds ", t.numel(), " elements, the call needs ", min_numel);
}
inline const float* f32(const Tensor& t, int64_t n, const char* name) { want(t, ScalarType::Float, n, name); return n > 0 || t.numel() > 0 ? (const float*)t.data_ptr() : nullptr; }
inline float* f32w(const Tensor& t, int64_t n, const char* name) { return (float*)f32(t, n, name); }
inline int32_t* i32(const Tensor& t, int64_t n, const char* name) { want(t, ScalarType::Int, n, name); return n > 0 || t.numel() > 0 ? (int32_t*)t.data_ptr() : nullptr; }
inline void* bytes(const Tensor& t, const char* name) { want(t, ScalarType::Byte, 0, name); return t.numel() > 0 ? t.data_ptr() : nullptr; }

struct Dims { int T, B; };
inline Dims crf_dims(const Tensor& score, const Tensor& noise)
{
    STD_TORCH_CHECK(score.dim() == 3 && score.size(0) == score.size(1), "semicrf: score must be [T, T, B]");
    const int64_t T = score.size(0), B = score.size(2);
    STD_TORCH_CHECK(T >= 1 && B >= 1 && T < (1 << 29) && B < (1ll << 31), "semicrf: bad score shape");
    want(score, ScalarType::Float, T * T * B, "score");
    want(noise, ScalarType::Float, (T - 1) * B, "noise");
    return Dims{(int)T, (int)B};
}

struct Ctx {
    torch::stable::accelerator::DeviceGuard guard;
    void* stream = nullptr;
    static int32_t index_of(const Tensor& t)
    {
        STD_TORCH_CHECK(t.is_cuda(), "semicrf: this overload takes GPU tensors");
        return t.get_device_index();
    }
    explicit Ctx(const Tensor& t) : guard(index_of(t))          // is_cuda is checked BEFORE the guard is built
    {
        TORCH_ERROR_CODE_CHECK(aoti_torch_get_current_cuda_stream(t.get_device_index(), &stream));
    }
    // every tensor of a call lives on the device of the first one
    template <typename... Ts>
    void same(const Tensor& a, const Ts&... rest) const
    {
        const Tensor* ts[] = {&rest...};
        for (const Tensor* t : ts)
            STD_TORCH_CHECK(!t->defined() || t->numel() == 0 || (t->is_cuda() && t->get_device_index() == a.get_device_index()),
                            "semicrf: all tensors of a call must share one device");
    }
};
template <typename... Ts>
inline void all_cpu(const Ts&... ts)
{
    const Tensor* a[] = {&ts...};
    for (const Tensor* t : a)
        STD_TORCH_CHECK(!t->defined() || t->numel() == 0 || t->is_cpu(), "semicrf: all tensors of a call must share one device");
}

inline void check(int rc, const char* what)
{
    STD_TORCH_CHECK(rc == SEMICRF_OK, what, " failed (code ", rc, "): ", semicrf_last_error());
}
inline float* fp(const Tensor& t) { return t.defined() && t.numel() > 0 ? (float*)t.data_ptr() : nullptr; }
inline const float* cfp(const Tensor& t) { return t.defined() && t.numel() > 0 ? (const float*)t.data_ptr() : nullptr; }
inline int32_t* ip(const Tensor& t) { return t.defined() && t.numel() > 0 ? (int32_t*)t.data_ptr() : nullptr; }

// ---- semi-CRF, GPU (dispatch key CUDA = HIP tensors) ------------------------------------------------------------------
void logz_fwd(Tensor score, Tensor noise, Tensor logZ, Tensor v, bool want_v, Tensor ws)
{
    Ctx c(score); c.same(score, noise, logZ, v, ws);
    const Dims d = crf_dims(score, noise);
    const int64_t TB = (int64_t)d.T * d.B;
    check(semicrf_logz_fwd(cfp(score), cfp(noise), d.T, d.B, f32w(logZ, d.B, "logZ"), want_v ? f32w(v, TB, "v") : nullptr, bytes(ws, "ws"),
                           (size_t)ws.numel(), c.stream),
          "semicrf_logz_fwd");
}
void logz_bwd(Tensor score, Tensor noise, Tensor v, Tensor logZ, Tensor gout, Tensor dScore, Tensor dNoise, Tensor q, bool want_q,
              int64_t flags, Tensor ws)
{
    Ctx c(score); c.same(score, noise, v, logZ, gout, dScore, dNoise, q, ws);
    const Dims d = crf_dims(score, noise);
    const int64_t TB = (int64_t)d.T * d.B;
    check(semicrf_logz_bwd_f(cfp(score), cfp(noise), f32(v, TB, "v"), f32(logZ, d.B, "logZ"), f32(gout, d.B, "gout"), d.T, d.B,
                             f32w(dScore, TB * d.T, "dScore"), f32w(dNoise, TB - d.B, "dNoise"), want_q ? f32w(q, TB, "q") : nullptr,
                             (int)flags, bytes(ws, "ws"), (size_t)ws.numel(), c.stream),
          "semicrf_logz_bwd");
}
void beta(Tensor score, Tensor noise, Tensor out, Tensor ws)
{
    Ctx c(score); c.same(score, noise, out, ws);
    const Dims d = crf_dims(score, noise);
    check(semicrf_beta(cfp(score), cfp(noise), d.T, d.B, f32w(out, (int64_t)d.T * d.B, "beta"), bytes(ws, "ws"), (size_t)ws.numel(), c.stream),
          "semicrf_beta");
}
void viterbi(Tensor score, Tensor noise, Tensor start, bool has_start, bool forward, Tensor pairs, Tensor offsets, Tensor ws)
{
    Ctx c(score); c.same(score, noise, pairs, offsets, ws);
    if (has_start) c.same(score, start);
    const Dims d = crf_dims(score, noise);
    STD_TORCH_CHECK(pairs.dim() == 2 && pairs.size(1) == 2, "semicrf: pairs must be [cap, 2]");
    check(semicrf_viterbi(cfp(score), cfp(noise), d.T, d.B, has_start ? i32(start, d.B, "start") : nullptr, forward ? 1 : 0,
                          i32(pairs, 0, "pairs"), (int64_t)pairs.size(0), i32(offsets, d.B + 1, "offsets"), bytes(ws, "ws"),
                          (size_t)ws.numel(), c.stream),
          "semicrf_viterbi");
}
void eval_path(Tensor score, Tensor noise, Tensor pairs, int64_t K, Tensor offsets, Tensor out, Tensor ws)
{
    Ctx c(score); c.same(score, noise, pairs, offsets, out);
    const Dims d = crf_dims(score, noise);
    STD_TORCH_CHECK(K >= 0, "semicrf: negative interval count");
    check(semicrf_eval_path(cfp(score), cfp(noise), d.T, d.B, i32(pairs, 2 * K, "pairs"), K, i32(offsets, d.B + 1, "offsets"),
                            f32w(out, d.B, "out"), ws.numel() > 0 ? ws.data_ptr() : nullptr, (size_t)ws.numel(), c.stream),
          "semicrf_eval_path");
}
void eval_path_bwd(Tensor gout, int64_t T, int64_t B, Tensor pairs, int64_t K, Tensor offsets, Tensor dScore, bool has_ds, Tensor dNoise,
                   bool has_dn)
{
    Ctx c(gout); c.same(gout, pairs, offsets);
    if (has_ds) c.same(gout, dScore);
    if (has_dn) c.same(gout, dNoise);
    STD_TORCH_CHECK(T >= 1 && B >= 1 && K >= 0 && T < (1 << 29) && B < (1ll << 31), "semicrf: bad sizes");
    check(semicrf_eval_path_bwd(f32(gout, B, "gout"), (int)T, (int)B, i32(pairs, 2 * K, "pairs"), K, i32(offsets, B + 1, "offsets"),
                                has_ds ? f32w(dScore, T * T * B, "dScore") : nullptr, has_dn ? f32w(dNoise, (T - 1) * B, "dNoise") : nullptr,
                                c.stream),
          "semicrf_eval_path_bwd");
}

// logProb as one call each way (semicrf_logprob_fwd / _bwd): gout holds B values (gstride 1) or ONE (gstride 0)
void logprob_fwd(Tensor score, Tensor noise, Tensor pairs, int64_t K, Tensor offsets, Tensor logProb, Tensor logZ, Tensor v, bool want_v,
                 Tensor ws)
{
    Ctx c(score); c.same(score, noise, pairs, offsets, logProb, logZ, v, ws);
    const Dims d = crf_dims(score, noise);
    STD_TORCH_CHECK(K >= 0, "semicrf: negative interval count");
    check(semicrf_logprob_fwd(cfp(score), cfp(noise), d.T, d.B, i32(pairs, 2 * K, "pairs"), K, i32(offsets, d.B + 1, "offsets"),
                              f32w(logProb, d.B, "logProb"), f32w(logZ, d.B, "logZ"), want_v ? f32w(v, (int64_t)d.T * d.B, "v") : nullptr,
                              bytes(ws, "ws"), (size_t)ws.numel(), c.stream),
          "semicrf_logprob_fwd");
}
void logprob_bwd(Tensor score, Tensor noise, Tensor v, Tensor logZ, Tensor gout, int64_t gstride, Tensor pairs, int64_t K, Tensor offsets,
                 Tensor dScore, Tensor dNoise, int64_t flags, Tensor ws)
{
    Ctx c(score); c.same(score, noise, v, logZ, gout, pairs, offsets, dScore, dNoise, ws);
    const Dims d = crf_dims(score, noise);
    const int64_t TB = (int64_t)d.T * d.B;
    STD_TORCH_CHECK(K >= 0 && (gstride == 0 || gstride == 1), "semicrf: bad interval count / gout stride");
    check(semicrf_logprob_bwd_f(cfp(score), cfp(noise), f32(v, TB, "v"), f32(logZ, d.B, "logZ"), f32(gout, gstride ? d.B : 1, "gout"),
                                (int)gstride, d.T, d.B, i32(pairs, 2 * K, "pairs"), K, i32(offsets, d.B + 1, "offsets"),
                                f32w(dScore, TB * d.T, "dScore"), f32w(dNoise, TB - d.B, "dNoise"), (int)flags, bytes(ws, "ws"),
                                (size_t)ws.numel(), c.stream),
          "semicrf_logprob_bwd");
}

// ---- semi-CRF, CPU (dispatch key CPU): the product's own host kernels (cpu_ops.cpp) -- selected by the tensors' device, never
// a fallback for GPU tensors.  The workspace argument is ignored (pass an empty tensor).
void logz_fwd_cpu(Tensor score, Tensor noise, Tensor logZ, Tensor v, bool want_v, Tensor ws)
{
    all_cpu(score, noise, logZ, v);
    const Dims d = crf_dims(score, noise);
    const int64_t TB = (int64_t)d.T * d.B;
    std::vector<float> scratch;
    float* vv;
    if (want_v) vv = f32w(v, TB, "v");
    else { scratch.resize((size_t)TB); vv = scratch.data(); }
    semicrf_cpu::logz_fwd(cfp(score), cfp(noise), d.T, d.B, f32w(logZ, d.B, "logZ"), vv);
}
void logz_bwd_cpu(Tensor score, Tensor noise, Tensor v, Tensor logZ, Tensor gout, Tensor dScore, Tensor dNoise, Tensor q, bool want_q,
                  int64_t flags, Tensor ws)       // flags: a permission (SEMICRF_GRAD_UPPER_IS_ZERO); the host kernels write everything
{
    all_cpu(score, noise, v, logZ, gout, dScore, dNoise, q);
    const Dims d = crf_dims(score, noise);
    const int64_t TB = (int64_t)d.T * d.B;
    std::vector<float> scratch;
    float* qq;
    if (want_q) qq = f32w(q, TB, "q");
    else { scratch.resize((size_t)TB); qq = scratch.data(); }
    semicrf_cpu::logz_bwd(cfp(score), cfp(noise), f32(v, TB, "v"), f32(logZ, d.B, "logZ"), f32(gout, d.B, "gout"), d.T, d.B,
                          f32w(dScore, TB * d.T, "dScore"), f32w(dNoise, TB - d.B, "dNoise"), qq);
}
void beta_cpu(Tensor score, Tensor noise, Tensor out, Tensor ws)
{
    all_cpu(score, noise, out);
    const Dims d = crf_dims(score, noise);
    semicrf_cpu::logz_bwd(cfp(score), cfp(noise), nullptr, nullptr, nullptr, d.T, d.B, nullptr, nullptr, f32w(out, (int64_t)d.T * d.B, "beta"));
}
void viterbi_cpu(Tensor score, Tensor noise, Tensor start, bool has_start, bool forward, Tensor pairs, Tensor offsets, Tensor ws)
{
    all_cpu(score, noise, pairs, offsets);
    if (has_start) all_cpu(start);
    const Dims d = crf_dims(score, noise);
    STD_TORCH_CHECK(pairs.dim() == 2 && pairs.size(1) == 2, "semicrf: pairs must be [cap, 2]");
    const int32_t* st = has_start ? i32(start, d.B, "start") : nullptr;
    if (st)
        for (int c = 0; c < d.B; ++c) STD_TORCH_CHECK(st[c] >= 0 && st[c] < d.T, "semicrf: forcedStartPos out of range");
    semicrf_cpu::viterbi(cfp(score), cfp(noise), d.T, d.B, st, forward ? 1 : 0, i32(pairs, 0, "pairs"), (int64_t)pairs.size(0),
                         i32(offsets, d.B + 1, "offsets"));
}
inline void check_path(const int32_t* pairs, int64_t K, const int32_t* offsets, int T, int B)
{
    STD_TORCH_CHECK(offsets[0] == 0 && offsets[B] == K, "semicrf: offsets do not match the interval count");
    for (int c = 0; c < B; ++c) STD_TORCH_CHECK(offsets[c] <= offsets[c + 1], "semicrf: offsets must ascend");
    for (int64_t i = 0; i < K; ++i)
        STD_TORCH_CHECK(pairs[2 * i] >= 0 && pairs[2 * i] <= pairs[2 * i + 1] && pairs[2 * i + 1] < T, "semicrf: interval out of range");
}
void eval_path_cpu(Tensor score, Tensor noise, Tensor pairs, int64_t K, Tensor offsets, Tensor out, Tensor ws)
{
    all_cpu(score, noise, pairs, offsets, out);
    const Dims d = crf_dims(score, noise);
    STD_TORCH_CHECK(K >= 0, "semicrf: negative interval count");
    const int32_t* pp = i32(pairs, 2 * K, "pairs");
    const int32_t* oo = i32(offsets, d.B + 1, "offsets");
    check_path(pp, K, oo, d.T, d.B);
    semicrf_cpu::eval_path(cfp(score), cfp(noise), d.T, d.B, pp, oo, f32w(out, d.B, "out"));
}
void eval_path_bwd_cpu(Tensor gout, int64_t T, int64_t B, Tensor pairs, int64_t K, Tensor offsets, Tensor dScore, bool has_ds, Tensor dNoise,
                       bool has_dn)
{
    all_cpu(gout, pairs, offsets);
    STD_TORCH_CHECK(T >= 1 && B >= 1 && K >= 0 && T < (1 << 29) && B < (1ll << 31), "semicrf: bad sizes");
    const int32_t* pp = i32(pairs, 2 * K, "pairs");
    const int32_t* oo = i32(offsets, B + 1, "offsets");
    check_path(pp, K, oo, (int)T, (int)B);
    semicrf_cpu::eval_path_bwd(f32(gout, B, "gout"), (int)T, (int)B, pp, oo, has_ds ? f32w(dScore, T * T * B, "dScore") : nullptr,
                               has_dn ? f32w(dNoise, (T - 1) * B, "dNoise") : nullptr);
}

void logprob_fwd_cpu(Tensor score, Tensor noise, Tensor pairs, int64_t K, Tensor offsets, Tensor logProb, Tensor logZ, Tensor v, bool want_v,
                     Tensor ws)
{
    logz_fwd_cpu(score, noise, logZ, v, want_v, ws);
    eval_path_cpu(score, noise, pairs, K, offsets, logProb, ws);
    float* lp = (float*)logProb.data_ptr();
    const float* lz = (const float*)logZ.data_ptr();
    for (int64_t c = 0; c < score.size(2); ++c) lp[c] -= lz[c];
}
void logprob_bwd_cpu(Tensor score, Tensor noise, Tensor v, Tensor logZ, Tensor gout, int64_t gstride, Tensor pairs, int64_t K, Tensor offsets,
                     Tensor dScore, Tensor dNoise, int64_t flags, Tensor ws)
{
    all_cpu(score, noise, v, logZ, gout, pairs, offsets, dScore, dNoise);
    const Dims d = crf_dims(score, noise);
    const int64_t TB = (int64_t)d.T * d.B;
    STD_TORCH_CHECK(K >= 0 && (gstride == 0 || gstride == 1), "semicrf: bad interval count / gout stride");
    const float* g = f32(gout, gstride ? d.B : 1, "gout");
    std::vector<float> gp((size_t)d.B), gn((size_t)d.B), q((size_t)TB);
    for (int c = 0; c < d.B; ++c) { gp[(size_t)c] = g[(size_t)c * gstride]; gn[(size_t)c] = -gp[(size_t)c]; }
    const int32_t* pp = i32(pairs, 2 * K, "pairs");
    const int32_t* oo = i32(offsets, d.B + 1, "offsets");
    check_path(pp, K, oo, d.T, d.B);
    semicrf_cpu::logz_bwd(cfp(score), cfp(noise), f32(v, TB, "v"), f32(logZ, d.B, "logZ"), gn.data(), d.T, d.B, f32w(dScore, TB * d.T, "dScore"),
                          f32w(dNoise, TB - d.B, "dNoise"), q.data());
    semicrf_cpu::eval_path_bwd(gp.data(), d.T, d.B, pp, oo, fp(dScore), fp(dNoise));
}

// ---- interval scorer ---------------------------------------------------------------------------------------------------
// q / k / diag (and their gradients) are strided views ([C][T][D] rows of ld floats): dtype and the device are checked, the
// strides are the caller's statement; everything dense is checked for its element count.
inline const float* f32s(const Tensor& t, const char* name)
{
    STD_TORCH_CHECK(t.defined() && t.scalar_type() == ScalarType::Float, "semicrf: `", name, "` must be a float32 tensor");
    return t.numel() > 0 ? (const float*)t.data_ptr() : nullptr;
}
inline float* f32so(const Tensor& t, const char* name) { return t.defined() && t.numel() > 0 ? (float*)f32s(t, name) : nullptr; }
inline void score_dims(int64_t C, int64_t T, int64_t D)
{
    STD_TORCH_CHECK(C >= 1 && T >= 1 && D >= 1 && C < (1ll << 31) && T < (1 << 29) && D < (1 << 20), "semicrf: bad C / T / D");
}
inline int64_t slots_of(int64_t C, int64_t group, int64_t pitch)
{
    STD_TORCH_CHECK(group >= 1 && pitch >= group && C % group == 0 && group < (1ll << 31) && pitch < (1ll << 31), "semicrf: bad slot layout");
    return C / group * pitch;
}
// rowc / drowc (merged projection, *_pc entry points): strided [C][T] views, stride ldrc; ldrc == 0: none
void interval_score_fwd_op(Tensor q, Tensor k, Tensor diag, Tensor rowc, int64_t C, int64_t T, int64_t D, int64_t ldq, int64_t ldk,
                           int64_t ldd, int64_t ldrc, double qscale, int64_t mode, int64_t full, int64_t group, int64_t pitch, Tensor S,
                           Tensor noise)
{
    Ctx c(q); c.same(q, k, diag, rowc, S, noise);
    score_dims(C, T, D);
    const int64_t Cs = slots_of(C, group, pitch);
    check(interval_score_fwd_pc(f32s(q, "q"), f32s(k, "k"), f32s(diag, "diag"), ldrc > 0 ? f32s(rowc, "rowc") : nullptr, (int)C, (int)T, (int)D,
                                ldq, ldk, ldd, ldrc > 0 ? ldrc : 1, (float)qscale, (int)mode, (int)full, (int)group, (int)pitch,
                                f32w(S, T * T * Cs, "S"), noise.numel() > 0 ? f32w(noise, (T - 1) * Cs, "noise") : nullptr, c.stream),
          "interval_score_fwd");
}
void interval_score_bwd_ws_op(Tensor dS, Tensor q, Tensor k, int64_t C, int64_t T, int64_t D, int64_t ldq, int64_t ldk, double qscale,
                              int64_t mode, int64_t group, int64_t pitch, Tensor dq, Tensor dk, Tensor ddiag, Tensor drowc, int64_t lddq,
                              int64_t lddk, int64_t lddd, int64_t lddrc, Tensor ws)
{
    Ctx c(dS); c.same(dS, q, k, dq, dk, ddiag, drowc, ws);
    score_dims(C, T, D);
    const int64_t Cs = slots_of(C, group, pitch);
    check(interval_score_bwd_ws_pc(f32(dS, T * T * Cs, "dS"), f32s(q, "q"), f32s(k, "k"), (int)C, (int)T, (int)D, ldq, ldk, (float)qscale,
                                   (int)mode, (int)group, (int)pitch, f32so(dq, "dq"), f32so(dk, "dk"), f32so(ddiag, "ddiag"),
                                   lddrc > 0 ? f32so(drowc, "drowc") : nullptr, lddq, lddk, lddd, lddrc > 0 ? lddrc : 1, bytes(ws, "ws"),
                                   (size_t)ws.numel(), c.stream),
          "interval_score_bwd_ws");
}
void interval_score_bwd_fused_ws_op(Tensor S, Tensor alpha, Tensor beta_, Tensor logZ, Tensor gout, Tensor q, Tensor k, int64_t C, int64_t T,
                                    int64_t D, int64_t ldq, int64_t ldk, double qscale, int64_t mode, int64_t group, int64_t pitch, Tensor dq,
                                    Tensor dk, Tensor ddiag, Tensor drowc, int64_t lddq, int64_t lddk, int64_t lddd, int64_t lddrc, Tensor ws)
{
    Ctx c(S); c.same(S, alpha, beta_, logZ, gout, q, k, dq, dk, ddiag, drowc, ws);
    score_dims(C, T, D);
    const int64_t Cs = slots_of(C, group, pitch);
    check(interval_score_bwd_fused_ws_pc(f32(S, T * T * Cs, "S"), f32(alpha, T * Cs, "alpha"), f32(beta_, T * Cs, "beta"),
                                         f32(logZ, Cs, "logZ"), f32(gout, Cs, "gout"), f32s(q, "q"), f32s(k, "k"), (int)C, (int)T, (int)D, ldq,
                                         ldk, (float)qscale, (int)mode, (int)group, (int)pitch, f32so(dq, "dq"), f32so(dk, "dk"),
                                         f32so(ddiag, "ddiag"), lddrc > 0 ? f32so(drowc, "drowc") : nullptr, lddq, lddk, lddd,
                                         lddrc > 0 ? lddrc : 1, bytes(ws, "ws"), (size_t)ws.numel(), c.stream),
          "interval_score_bwd_fused_ws");
}
void interval_score_path_bwd_op(Tensor gout, Tensor pairs, int64_t K, Tensor offsets, Tensor q, Tensor k, int64_t C, int64_t T, int64_t D,
                                int64_t ldq, int64_t ldk, double qscale, int64_t mode, int64_t group, int64_t pitch, Tensor dq, Tensor dk,
                                Tensor ddiag, Tensor drowc, int64_t lddq, int64_t lddk, int64_t lddd, int64_t lddrc)
{
    Ctx c(gout); c.same(gout, pairs, offsets, q, k, dq, dk, ddiag, drowc);
    score_dims(C, T, D);
    STD_TORCH_CHECK(K >= 0, "semicrf: negative interval count");
    const int64_t Cs = slots_of(C, group, pitch);
    check(interval_score_path_bwd_pc(f32(gout, Cs, "gout"), i32(pairs, 2 * K, "pairs"), K, i32(offsets, Cs + 1, "offsets"), f32s(q, "q"),
                                     f32s(k, "k"), (int)C, (int)T, (int)D, ldq, ldk, (float)qscale, (int)mode, (int)group, (int)pitch,
                                     f32so(dq, "dq"), f32so(dk, "dk"), f32so(ddiag, "ddiag"), lddrc > 0 ? f32so(drowc, "drowc") : nullptr, lddq,
                                     lddk, lddd, lddrc > 0 ? lddrc : 1, c.stream),
          "interval_score_path_bwd");
}

// ---- attribute-head features ------------------------------------------------------------------------------------------
// the scorer's projection (csrc/proj_gemm.hip): strided row-major matrices, the strides are the caller's statement
void proj_nn_op(Tensor A, int64_t lda, int64_t M, int64_t K, Tensor B, int64_t ldb, int64_t N, Tensor out, int64_t ldout, Tensor bias,
                bool has_bias, Tensor w2, Tensor b2, bool has_w2, int64_t zero_cols, bool accumulate)
{
    Ctx c(A); c.same(A, B, out);
    if (has_bias) c.same(A, bias);
    if (has_w2) c.same(A, w2, b2);
    STD_TORCH_CHECK(M >= 1 && K >= 4 && N >= 1 && M < (1ll << 31) && K < (1 << 20) && N <= 256, "semicrf: bad M / K / N");
    STD_TORCH_CHECK(A.numel() >= (M - 1) * lda + K && out.numel() >= (M - 1) * ldout + N + (has_w2 ? 2 + zero_cols : 0), "semicrf: A / out too small");
    STD_TORCH_CHECK(B.numel() >= ((K + 31) / 32 * 32 - 1) * ldb + N, "semicrf: B must hold whole chunks of 32 rows (zero beyond K)");
    check(scorer_proj_nn(f32s(A, "A"), lda, M, (int)K, f32s(B, "B"), ldb, (int)N, f32so(out, "out"), ldout, has_bias ? f32(bias, N, "bias") : nullptr,
                         has_w2 ? f32(w2, 2 * K, "w2") : nullptr, has_w2 ? f32(b2, 2, "b2") : nullptr, (int)zero_cols, accumulate ? 1 : 0, c.stream),
          "scorer_proj_nn");
}
void proj_nn3_op(Tensor A, int64_t lda, int64_t M, int64_t K, Tensor B, int64_t ldb, int64_t N, Tensor out, int64_t ldout, Tensor bias,
                 bool has_bias, Tensor w2, Tensor b2, bool has_w2, int64_t zero_cols, bool accumulate, Tensor ws)
{
    Ctx c(A); c.same(A, B, out, ws);
    if (has_bias) c.same(A, bias);
    if (has_w2) c.same(A, w2, b2);
    STD_TORCH_CHECK(M >= 1 && K >= 4 && N >= 1 && M < (1ll << 31) && K < (1 << 20) && N <= 256, "semicrf: bad M / K / N");
    STD_TORCH_CHECK(A.numel() >= (M - 1) * lda + K && out.numel() >= (M - 1) * ldout + N + (has_w2 ? 2 + zero_cols : 0), "semicrf: A / out too small");
    STD_TORCH_CHECK(B.numel() >= ((K + 31) / 32 * 32 - 1) * ldb + N, "semicrf: B must hold whole chunks of 32 rows (zero beyond K)");
    check(scorer_proj_nn3(f32s(A, "A"), lda, M, (int)K, f32s(B, "B"), ldb, (int)N, f32so(out, "out"), ldout, has_bias ? f32(bias, N, "bias") : nullptr,
                          has_w2 ? f32(w2, 2 * K, "w2") : nullptr, has_w2 ? f32(b2, 2, "b2") : nullptr, (int)zero_cols, accumulate ? 1 : 0,
                          ws.numel() ? ws.data_ptr() : nullptr, (size_t)ws.numel() * ws.element_size(), c.stream),
          "scorer_proj_nn3");
}
void proj_tn_op(Tensor dy, int64_t lddy, int64_t M, int64_t R, int64_t extra_col0, int64_t total_rows, Tensor x, int64_t ldx, int64_t N, Tensor dW,
                int64_t lddw, Tensor db, Tensor ws)
{
    Ctx c(dy); c.same(dy, x, dW, db, ws);
    const int64_t x3flag = total_rows & SEMICRF_PROJ_TN_BF16X3;  // opt-in: the matrix part on the three-limb bf16 kernel
    total_rows &= ~(int64_t)SEMICRF_PROJ_TN_BF16X3;
    STD_TORCH_CHECK(M >= 1 && R >= 1 && N >= 1 && N <= 256 && M < (1ll << 31) && total_rows >= R && total_rows < (1 << 20), "semicrf: bad sizes");
    STD_TORCH_CHECK(dy.numel() >= (M - 1) * lddy + R && x.numel() >= (M - 1) * ldx + N && dW.numel() >= (total_rows - 1) * lddw + N,
                    "semicrf: dy / x / dW too small");
    check(scorer_proj_tn(f32s(dy, "dy"), lddy, M, (int)R, (int)extra_col0, (int)(total_rows | x3flag), f32s(x, "x"), ldx, (int)N, f32so(dW, "dW"), lddw,
                         f32w(db, total_rows, "db"), bytes(ws, "ws"), (size_t)ws.numel(), c.stream),
          "scorer_proj_tn");
}

void stage_linear_op(Tensor W, Tensor bias, int64_t D, int64_t size, int64_t rows_pad, Tensor BT, Tensor Wqd, Tensor w2, Tensor b2)
{
    Ctx c(W); c.same(W, bias, BT, Wqd); c.same(W, w2, b2);
    STD_TORCH_CHECK(D >= 1 && size >= 1 && rows_pad >= D + 1 && rows_pad < (1 << 20) && D < (1 << 20) && size < (1 << 20), "semicrf: bad D / size / rows_pad");
    check(scorer_stage_linear(f32(W, (2 * D + 1) * size, "W"), f32(bias, 2 * D + 1, "bias"), (int)D, (int)size, (int)rows_pad,
                              f32w(BT, size * 2 * D, "BT"), f32w(Wqd, rows_pad * size, "Wqd"), f32w(w2, 2 * size, "w2"), f32w(b2, 2, "b2"), c.stream),
          "scorer_stage_linear");
}
void merge_weights_fwd_op(Tensor W, Tensor bias, int64_t D, int64_t size, int64_t rows, Tensor Wm, Tensor bm, Tensor WmT, bool has_t)
{
    Ctx c(W); c.same(W, bias, Wm, bm);
    if (has_t) c.same(W, WmT);
    STD_TORCH_CHECK(D >= 1 && size >= 1 && size <= 256 && rows >= size + 2 && rows < (1 << 20), "semicrf: bad D / size / rows");
    check(scorer_merge_weights_fwd(f32(W, (2 * D + 1) * size, "W"), f32(bias, 2 * D + 1, "bias"), (int)D, (int)size, (int)rows,
                                   f32w(Wm, rows * size, "Wm"), f32w(bm, rows, "bm"), has_t ? f32w(WmT, size * size, "WmT") : nullptr, c.stream),
          "scorer_merge_weights_fwd");
}
void merge_weights_bwd_op(Tensor W, Tensor bias, Tensor dWm, Tensor dbm, int64_t D, int64_t size, int64_t rows, Tensor dW, Tensor dbias, Tensor ws)
{
    Ctx c(W); c.same(W, bias, dWm, dbm); c.same(W, dW, dbias, ws);
    STD_TORCH_CHECK(D >= 1 && size >= 1 && size <= 256 && rows >= size + 2 && rows < (1 << 20), "semicrf: bad D / size / rows");
    check(scorer_merge_weights_bwd(f32(W, (2 * D + 1) * size, "W"), f32(bias, 2 * D + 1, "bias"), f32(dWm, rows * size, "dWm"),
                                   f32(dbm, rows, "dbm"), (int)D, (int)size, (int)rows, f32w(dW, (2 * D + 1) * size, "dW"),
                                   f32w(dbias, 2 * D + 1, "dbias"), bytes(ws, "ws"), (size_t)ws.numel(), c.stream),
          "scorer_merge_weights_bwd");
}

void interval_features_gather_op(Tensor ctx, int64_t C, int64_t T, int64_t D, int64_t ldc, Tensor pairs, int64_t K, Tensor offsets,
                                 int64_t nSym, Tensor out, Tensor symIdx, Tensor scatterIdx)
{
    Ctx c(ctx); c.same(ctx, pairs, offsets, out, symIdx, scatterIdx);
    score_dims(C, T, D);
    STD_TORCH_CHECK(K >= 0, "semicrf: negative interval count");
    want(symIdx, ScalarType::Long, K, "symIdx"); want(scatterIdx, ScalarType::Long, K, "scatterIdx");
    check(interval_features_gather(f32s(ctx, "ctx"), (int)C, (int)T, (int)D, ldc, i32(pairs, 2 * K, "pairs"), K, i32(offsets, C + 1, "offsets"),
                                   (int)nSym, f32w(out, K * 3 * D, "out"), K > 0 ? (int64_t*)symIdx.data_ptr() : nullptr,
                                   K > 0 ? (int64_t*)scatterIdx.data_ptr() : nullptr, c.stream),
          "interval_features_gather");
}
void interval_features_gather_bwd_op(Tensor gout, Tensor ctx, int64_t C, int64_t T, int64_t D, int64_t ldc, Tensor pairs, int64_t K,
                                     Tensor offsets, Tensor dctx, int64_t lddc)
{
    Ctx c(gout); c.same(gout, ctx, pairs, offsets, dctx);
    score_dims(C, T, D);
    STD_TORCH_CHECK(K >= 0, "semicrf: negative interval count");
    check(interval_features_gather_bwd(f32(gout, K * 3 * D, "gout"), f32s(ctx, "ctx"), (int)C, (int)T, (int)D, ldc, i32(pairs, 2 * K, "pairs"),
                                       K, i32(offsets, C + 1, "offsets"), (float*)f32s(dctx, "dctx"), lddc, c.stream),
          "interval_features_gather_bwd");
}

// ---- transcription segment loop ----------------------------------------------------------------------------------------
void segment_onset_filter_op(Tensor pairs, Tensor offsets, int64_t B, int64_t bound, Tensor pairs_out, Tensor offsets_out, Tensor counts_ws)
{
    Ctx c(offsets); c.same(offsets, pairs, pairs_out, offsets_out, counts_ws);
    STD_TORCH_CHECK(B >= 1 && B < (1ll << 31), "semicrf: bad B");
    check(segment_onset_filter(i32(pairs, 0, "pairs"), i32(offsets, B + 1, "offsets"), (int)B, (int)bound, i32(pairs_out, 0, "pairs_out"),
                               pairs_out.numel() / 2, i32(offsets_out, B + 1, "offsets_out"), i32(counts_ws, B, "counts_ws"), c.stream),
          "segment_onset_filter");
}
void segment_events_op(Tensor pairs, int64_t K, Tensor offsets, int64_t B, int64_t nSym, Tensor ofValue, Tensor ofPresence,
                       int64_t lastFrameIdx, double frameDur, Tensor beginTime, int64_t stepFrames, Tensor times, Tensor flags, Tensor lastP,
                       Tensor nextStart)
{
    Ctx c(offsets); c.same(offsets, pairs, ofValue, ofPresence, beginTime, times, flags, lastP, nextStart);
    STD_TORCH_CHECK(B >= 1 && B < (1ll << 31) && nSym >= 1 && K >= 0, "semicrf: bad B / nSym / K");
    want(ofPresence, ScalarType::Byte, 2 * K, "ofPresence"); want(flags, ScalarType::Byte, 2 * K, "flags");
    want(beginTime, ScalarType::Double, B / nSym, "beginTime"); want(times, ScalarType::Double, 2 * K, "times");
    check(segment_events(i32(pairs, 2 * K, "pairs"), K, i32(offsets, B + 1, "offsets"), (int)B, (int)nSym, f32(ofValue, 2 * K, "ofValue"),
                         K > 0 ? (const unsigned char*)ofPresence.data_ptr() : nullptr, (int)lastFrameIdx, frameDur,
                         (const double*)beginTime.data_ptr(), (int)stepFrames, K > 0 ? (double*)times.data_ptr() : nullptr,
                         K > 0 ? (unsigned char*)flags.data_ptr() : nullptr, i32(lastP, B, "lastP"), i32(nextStart, B, "nextStart"), c.stream),
          "segment_events");
}

}  // namespace

STABLE_TORCH_LIBRARY(semicrf, m)
{
    m.def("logz_fwd(Tensor score, Tensor noise, Tensor(a!) logZ, Tensor(b!) v, bool want_v, Tensor(c!) ws) -> ()");
    m.def("logz_bwd(Tensor score, Tensor noise, Tensor v, Tensor logZ, Tensor gout, Tensor(a!) dScore, Tensor(b!) dNoise, Tensor(c!) q, "
          "bool want_q, int flags, Tensor(d!) ws) -> ()");
    m.def("beta(Tensor score, Tensor noise, Tensor(a!) out, Tensor(b!) ws) -> ()");
    m.def("viterbi(Tensor score, Tensor noise, Tensor start, bool has_start, bool forward, Tensor(a!) pairs, Tensor(b!) offsets, "
          "Tensor(c!) ws) -> ()");
    m.def("eval_path(Tensor score, Tensor noise, Tensor pairs, int K, Tensor offsets, Tensor(a!) out, Tensor(b!) ws) -> ()");
    m.def("eval_path_bwd(Tensor gout, int T, int B, Tensor pairs, int K, Tensor offsets, Tensor(a!) dScore, bool has_ds, Tensor(b!) dNoise, "
          "bool has_dn) -> ()");
    m.def("logprob_fwd(Tensor score, Tensor noise, Tensor pairs, int K, Tensor offsets, Tensor(a!) logProb, Tensor(b!) logZ, Tensor(c!) v, "
          "bool want_v, Tensor(d!) ws) -> ()");
    m.def("logprob_bwd(Tensor score, Tensor noise, Tensor v, Tensor logZ, Tensor gout, int gstride, Tensor pairs, int K, Tensor offsets, "
          "Tensor(a!) dScore, Tensor(b!) dNoise, int flags, Tensor(c!) ws) -> ()");
    // (group, pitch): the slot layout of the chain axis (include/semicrf_hip.h, *_p entry points); group == pitch: contiguous
    // rowc / drowc, ldrc / lddrc: the merged projection's per-(chain, end) constant (*_pc entry points); stride 0: none (pass any tensor)
    m.def("interval_score_fwd(Tensor q, Tensor k, Tensor diag, Tensor rowc, int C, int T, int D, int ldq, int ldk, int ldd, int ldrc, "
          "float qscale, int mode, int full, int group, int pitch, Tensor(a!) S, Tensor(b!) noise) -> ()");
    m.def("interval_score_bwd_ws(Tensor dS, Tensor q, Tensor k, int C, int T, int D, int ldq, int ldk, float qscale, int mode, int group, "
          "int pitch, Tensor(a!) dq, Tensor(b!) dk, Tensor(c!) ddiag, Tensor(d!) drowc, int lddq, int lddk, int lddd, int lddrc, "
          "Tensor(e!) ws) -> ()");
    m.def("interval_score_bwd_fused_ws(Tensor S, Tensor alpha, Tensor beta, Tensor logZ, Tensor gout, Tensor q, Tensor k, int C, int T, int D, "
          "int ldq, int ldk, float qscale, int mode, int group, int pitch, Tensor(a!) dq, Tensor(b!) dk, Tensor(c!) ddiag, Tensor(d!) drowc, "
          "int lddq, int lddk, int lddd, int lddrc, Tensor(e!) ws) -> ()");
    m.def("interval_score_path_bwd(Tensor gout, Tensor pairs, int K, Tensor offsets, Tensor q, Tensor k, int C, int T, int D, int ldq, int ldk, "
          "float qscale, int mode, int group, int pitch, Tensor(a!) dq, Tensor(b!) dk, Tensor(c!) ddiag, Tensor(d!) drowc, int lddq, int lddk, "
          "int lddd, int lddrc) -> ()");
    m.def("proj_nn(Tensor A, int lda, int M, int K, Tensor B, int ldb, int N, Tensor(a!) out, int ldout, Tensor bias, bool has_bias, Tensor w2, "
          "Tensor b2, bool has_w2, int zero_cols, bool accumulate) -> ()");
    m.def("proj_nn3(Tensor A, int lda, int M, int K, Tensor B, int ldb, int N, Tensor(a!) out, int ldout, Tensor bias, bool has_bias, Tensor w2, "
          "Tensor b2, bool has_w2, int zero_cols, bool accumulate, Tensor(b!) ws) -> ()");
    m.def("stage_linear(Tensor W, Tensor bias, int D, int size, int rows_pad, Tensor(a!) BT, Tensor(b!) Wqd, Tensor(c!) w2, Tensor(d!) b2) -> ()");
    m.def("merge_weights_fwd(Tensor W, Tensor bias, int D, int size, int rows, Tensor(a!) Wm, Tensor(b!) bm, Tensor(c!) WmT, bool has_t) -> ()");
    m.def("merge_weights_bwd(Tensor W, Tensor bias, Tensor dWm, Tensor dbm, int D, int size, int rows, Tensor(a!) dW, Tensor(b!) dbias, Tensor(c!) ws) -> ()");
    m.def("proj_tn(Tensor dy, int lddy, int M, int R, int extra_col0, int total_rows, Tensor x, int ldx, int N, Tensor(a!) dW, int lddw, "
          "Tensor(b!) db, Tensor(c!) ws) -> ()");
    m.def("interval_features_gather(Tensor ctx, int C, int T, int D, int ldc, Tensor pairs, int K, Tensor offsets, int nSym, Tensor(a!) out, "
          "Tensor(b!) symIdx, Tensor(c!) scatterIdx) -> ()");
    m.def("interval_features_gather_bwd(Tensor gout, Tensor ctx, int C, int T, int D, int ldc, Tensor pairs, int K, Tensor offsets, "
          "Tensor(a!) dctx, int lddc) -> ()");
    m.def("segment_onset_filter(Tensor pairs, Tensor offsets, int B, int bound, Tensor(a!) pairs_out, Tensor(b!) offsets_out, "
          "Tensor(c!) counts_ws) -> ()");
    m.def("segment_events(Tensor pairs, int K, Tensor offsets, int B, int nSym, Tensor ofValue, Tensor ofPresence, int lastFrameIdx, "
          "float frameDur, Tensor beginTime, int stepFrames, Tensor(a!) times, Tensor(b!) flags, Tensor(c!) lastP, Tensor(d!) nextStart) -> ()");
}

STABLE_TORCH_LIBRARY_IMPL(semicrf, CPU, m)
{
    m.impl("logz_fwd", TORCH_BOX(&logz_fwd_cpu));
    m.impl("logz_bwd", TORCH_BOX(&logz_bwd_cpu));
    m.impl("beta", TORCH_BOX(&beta_cpu));
    m.impl("viterbi", TORCH_BOX(&viterbi_cpu));
    m.impl("eval_path", TORCH_BOX(&eval_path_cpu));
    m.impl("eval_path_bwd", TORCH_BOX(&eval_path_bwd_cpu));
    m.impl("logprob_fwd", TORCH_BOX(&logprob_fwd_cpu));
    m.impl("logprob_bwd", TORCH_BOX(&logprob_bwd_cpu));
}

STABLE_TORCH_LIBRARY_IMPL(semicrf, CUDA, m)
{
    m.impl("logz_fwd", TORCH_BOX(&logz_fwd));
    m.impl("logz_bwd", TORCH_BOX(&logz_bwd));
    m.impl("beta", TORCH_BOX(&beta));
    m.impl("viterbi", TORCH_BOX(&viterbi));
    m.impl("eval_path", TORCH_BOX(&eval_path));
    m.impl("eval_path_bwd", TORCH_BOX(&eval_path_bwd));
    m.impl("logprob_fwd", TORCH_BOX(&logprob_fwd));
    m.impl("logprob_bwd", TORCH_BOX(&logprob_bwd));
    m.impl("interval_score_fwd", TORCH_BOX(&interval_score_fwd_op));
    m.impl("interval_score_bwd_ws", TORCH_BOX(&interval_score_bwd_ws_op));
    m.impl("interval_score_bwd_fused_ws", TORCH_BOX(&interval_score_bwd_fused_ws_op));
    m.impl("interval_score_path_bwd", TORCH_BOX(&interval_score_path_bwd_op));
    m.impl("proj_nn", TORCH_BOX(&proj_nn_op));
    m.impl("proj_nn3", TORCH_BOX(&proj_nn3_op));
    m.impl("proj_tn", TORCH_BOX(&proj_tn_op));
    m.impl("merge_weights_fwd", TORCH_BOX(&merge_weights_fwd_op));
    m.impl("stage_linear", TORCH_BOX(&stage_linear_op));
    m.impl("merge_weights_bwd", TORCH_BOX(&merge_weights_bwd_op));
    m.impl("interval_features_gather", TORCH_BOX(&interval_features_gather_op));
    m.impl("interval_features_gather_bwd", TORCH_BOX(&interval_features_gather_bwd_op));
    m.impl("segment_onset_filter", TORCH_BOX(&segment_onset_filter_op));
    m.impl("segment_events", TORCH_BOX(&segment_events_op));
}
