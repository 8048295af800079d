// dpp_probe.hip -- stand-alone (hipcc --offload-arch=gfx950 tools/dpp_probe.hip -o /tmp/dpp_probe): latency of the two
// dependent chains of the spine's linear-domain diagonal solve, one lone wave per CU like a ring wave:
//   tropical pass   est = max(est, row_newbcast_j(est) + c_j)         (v_add_f32_dpp + v_max_f32)
//   linear solve    acc += row_newbcast_j(acc) * m_j                  (v_mov_b32_dpp + v_fmac_f32)
// against the log-domain step it replaces (ds_bpermute + logaddexp2: 2 transcendentals).  Prints cycles per 16-step block.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int J>
__device__ __forceinline__ float bcast(float x)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + J, 0xf, 0xf, false));
}
#define FOR15(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14)

__global__ void probe(const float* a, const float* m, float* o, unsigned long long* cyc, int iters)
{
    const int lane = threadIdx.x;
    float mm[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) mm[j] = m[j * 64 + lane];
    float acc = a[lane], est = a[lane], lg = a[lane];
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define STEP(J) est = fmaxf(est, bcast<J>(est) + mm[J]);
        FOR15(STEP)
#undef STEP
        est *= 0.5f;
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#define STEP(J) acc = fmaf(bcast<J>(acc), mm[J], acc);
        FOR15(STEP)
#undef STEP
        acc *= 0.001f;
    }
    unsigned long long t2 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 15; ++j) {
            const float u = __int_as_float(__builtin_amdgcn_ds_bpermute((j << 4) + ((lane & 3) << 2), __float_as_int(lg)));
            const float t = u + mm[j];
            lg = fmaxf(lg, t) + __builtin_amdgcn_logf(1.0f + __builtin_amdgcn_exp2f(-fabsf(lg - t)));
        }
        lg *= 0.5f;
    }
    unsigned long long t3 = __builtin_readcyclecounter();
    // 16 independent exp2 + 16 fma (the coefficient set-up), and 16 pushes of the shadow kind
    float s = 0.f, M = est;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) s += __builtin_amdgcn_exp2f(mm[j] - M);
        M += 1.0f;
    }
    unsigned long long t4 = __builtin_readcyclecounter();
    float aM = -1e30f, aS = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const float t = mm[j] + (float)it, d = t - aM;
            const float e = __builtin_amdgcn_exp2f(-fabsf(d));
            aS = d > 0.0f ? fmaf(aS, e, 1.0f) : aS + e;
            aM = fmaxf(aM, t);
        }
    }
    unsigned long long t5 = __builtin_readcyclecounter();
    o[blockIdx.x * 64 + lane] = acc + est + lg + s + aM + aS;
    if (lane == 0) {
        cyc[blockIdx.x * 5 + 0] = (t1 - t0) / iters; cyc[blockIdx.x * 5 + 1] = (t2 - t1) / iters; cyc[blockIdx.x * 5 + 2] = (t3 - t2) / iters;
        cyc[blockIdx.x * 5 + 3] = (t4 - t3) / iters; cyc[blockIdx.x * 5 + 4] = (t5 - t4) / iters;
    }
}

int main()
{
    float *a, *m, *o; unsigned long long* c;
    hipMalloc(&a, 64 * 4); hipMalloc(&m, 16 * 64 * 4); hipMalloc(&o, 256 * 64 * 4); hipMalloc(&c, 256 * 5 * 8);
    std::vector<float> ha(64, 1.0f), hm(16 * 64, 0.01f);
    hipMemcpy(a, ha.data(), 64 * 4, hipMemcpyHostToDevice); hipMemcpy(m, hm.data(), 16 * 64 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe, dim3(256), dim3(64), 0, 0, a, m, o, c, 2000);
    hipDeviceSynchronize();
    unsigned long long hc[5];
    hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
    printf("cycles per block of 15 steps (lone wave): tropical %llu, linear solve %llu, log-domain (bpermute + logaddexp2) %llu; "
           "16 independent exp2+add %llu; 16 acc_push1 %llu\n", hc[0], hc[1], hc[2], hc[3], hc[4]);
    return 0;
}
