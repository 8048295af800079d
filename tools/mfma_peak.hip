// mfma_peak.hip -- what the matrix pipe sustains on the box at hand: every wave issues independent v_mfma_f32_32x32x16_bf16 (or
// v_mfma_f32_32x32x2_f32) back to back, nothing else.  Prints TFLOP/s, the shader clock under that load (s_memtime against the
// 100 MHz s_memrealtime) and the pipe's busy share = instructions x passes x 4 cycles / cycles.  The numbers the scorer kernels'
// "share of the matrix pipe's time" should be read against: the clock under matrix load is not the 2.4 GHz of the data sheet.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak tools/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int NACC, int FILL, int RND = 0>
__global__ __launch_bounds__(1024) void peak_kernel(float* out, unsigned long long* clk, int iters, int duty)
{
    float fv[4] = {0.f, 0.f, 0.f, 0.f};
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // operands: small integers (RND = 0) or four sets of random bit patterns with normal-sized exponents (RND = 1: every
    // instruction sees other operand bits than the one before it, as in a real product -- the clock follows the power)
    bf16x8 av[4], bv[4];
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < 8; ++i) {
            if (RND) {
                h = h * 1664525u + 1013904223u; const unsigned short ua = (unsigned short)((h >> 16) & 0x807f) | 0x3f00;    // +-[0.5, 1)
                h = h * 1664525u + 1013904223u; const unsigned short ub = (unsigned short)((h >> 16) & 0x807f) | 0x3f00;
                av[s][i] = __builtin_bit_cast(__bf16, ua); bv[s][i] = __builtin_bit_cast(__bf16, ub);
            } else { av[s][i] = (__bf16)(float)(threadIdx.x + i); bv[s][i] = (__bf16)(float)(threadIdx.x * 3 + i); }
        }
    const float fa = (float)threadIdx.x, fb = 1.0f + threadIdx.x;
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int t = 0; t < NACC; ++t) {
                if (KIND == 0) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[(u + t) & 3], bv[u], acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
                _Pragma("unroll") for (int f = 0; f < FILL; ++f) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(fv[f & 3]) : "v"(fa), "v"(fb));     // (vector fillers)
            }
        for (int d = 0; d < duty; ++d) __builtin_amdgcn_s_sleep(1);          // (duty cycle: idle between the bursts)
    }
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = fv[0] + fv[1] + fv[2] + fv[3];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) { const int w = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64; clk[2 * w] = c1 - c0; clk[2 * w + 1] = r1 - r0; }
}

template <int KIND, int FILL = 0, int RND = 0>
static void run(const char* name, int threads, int blocks, int iters, int duty)
{
    float* out; unsigned long long* clk;
    hipMalloc(&out, (size_t)blocks * threads * 4); const int nw = blocks * threads / 64; hipMalloc(&clk, (size_t)nw * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        peak_kernel<KIND, 4, FILL, RND><<<blocks, threads>>>(out, clk, iters, duty);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * nw);
    hipMemcpy(h.data(), clk, (size_t)nw * 16, hipMemcpyDeviceToHost);
    double cyc = 0, rt = 0, rtmax = 0;                           // (a wave's own life: with two waves on a SIMD the older one goes first)
    for (int i = 0; i < nw; ++i) { cyc += h[2 * i]; rt += h[2 * i + 1]; if (h[2 * i + 1] > rtmax) rtmax = h[2 * i + 1]; }
    cyc /= nw; rt /= nw;
    const double ninst = (double)iters * 16, flop_inst = KIND == 0 ? 2.0 * 32 * 32 * 16 : 2.0 * 32 * 32 * 2;
    const double passes = KIND == 0 ? 8 : 16;
    const double waves = (double)blocks * threads / 64;
    const double clock = cyc / (rt * 10.0);                      // GHz: cycles per 10 ns tick / 10
    const double wps = (double)threads / 64 / 4;                 // waves per SIMD
    printf("%-5s %4d threads x %4d blocks, sleep %2d, fillers %d, %s: %8.3f ms  %7.1f TFLOP/s  clock %.2f GHz  pipe busy %.1f %% (kernel %.0f us by the waves' clocks)\n",
           name, threads, blocks, duty, FILL, RND ? "random operands" : "integer operands", ms, ninst * flop_inst * waves / (ms * 1e-3) / 1e12, clock,
           100.0 * ninst * wps * passes * 4 / (ms * 1e-3 * clock * 1e9), rtmax / 100.0);
    hipFree(out); hipFree(clk);
}

int main()
{
    run<0>("bf16", 256, 256, 20000, 0);
    run<0>("bf16", 512, 256, 10000, 0);
    run<0>("bf16", 1024, 256, 5000, 0);
    run<0>("bf16", 512, 256, 10000, 2);
    run<0>("bf16", 512, 256, 10000, 8);
    run<0, 1>("bf16", 512, 256, 10000, 0);
    run<0, 2>("bf16", 512, 256, 10000, 0);
    run<0, 3>("bf16", 512, 256, 10000, 0);
    run<0, 4>("bf16", 512, 256, 10000, 0);
    run<0, 6>("bf16", 512, 256, 10000, 0);
    run<0, 8>("bf16", 512, 256, 10000, 0);
    run<0, 3>("bf16", 256, 256, 20000, 0);
    run<0, 6>("bf16", 256, 256, 20000, 0);
    run<0, 0, 1>("bf16", 256, 256, 20000, 0);
    run<0, 0, 1>("bf16", 512, 256, 10000, 0);
    run<0, 3, 1>("bf16", 512, 256, 10000, 0);
    run<0, 6, 1>("bf16", 512, 256, 10000, 0);
    run<0, 0, 1>("bf16", 512, 256, 10000, 8);
    run<1>("fp32", 256, 256, 10000, 0);
    run<1>("fp32", 512, 256, 5000, 0);
    run<1, 4>("fp32", 512, 256, 5000, 0);
    return 0;
}
