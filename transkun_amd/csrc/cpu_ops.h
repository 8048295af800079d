// cpu_ops.h -- host kernels behind the CPU dispatch key of torch.ops.semicrf.* (cpu_ops.cpp); argument meaning as the
// entry points of the same name in include/semicrf_hip.h, with host pointers and no stream / workspace.
#pragma once
#include <cstdint>

namespace semicrf_cpu {
void logz_fwd(const float* score, const float* noise, int T, int B, float* logZ, float* v /* [T][B], required */);
void logz_bwd(const float* score, const float* noise, const float* v, const float* logZ, const float* gout, int T, int B,
              float* dScore /* or null: beta only */, float* dNoise /* or null */, float* q /* [T][B], required */);
void viterbi(const float* score, const float* noise, int T, int B, const int32_t* start, int forward, int32_t* pairs, int64_t cap,
             int32_t* offsets);
void eval_path(const float* score, const float* noise, int T, int B, const int32_t* pairs, const int32_t* offsets, float* out);
void eval_path_bwd(const float* gout, int T, int B, const int32_t* pairs, const int32_t* offsets, float* dScore, float* dNoise);
}  // namespace semicrf_cpu
