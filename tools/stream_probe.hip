// stream_probe.hip -- what can ONE compute unit of an MI355X stream, and how?
//
// Measurement tool (not product code): workgroups of W waves, one per CU (LDS-sized so that two never share a CU),
// stream 8 KB tiles of a [T][T][B] fp32 tensor the way the panel waves of persist.hip do (eight 1 KB pieces, every
// piece = 8 lines of 128 bytes that are B*4 bytes apart) or as contiguous 8 KB, by
//   method 0: asynchronous global->LDS loads (buffer_load_dwordx4 ... lds), S stages per wave, nothing consumed
//   method 1: the same, each tile read back from LDS (8 x ds_read_b128) and summed
//   method 2: buffer_load_dwordx4 into registers, S tiles in flight per wave (32 VGPRs each), summed
// Output: one line per configuration with the aggregate and per-CU rate.
//
//   hipcc --offload-arch=gfx950 -O3 -o tools/_stream_probe tools/stream_probe.hip && tools/_stream_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int T = 1024, B = 352, NG = B / 32;

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

struct Args {
    const float* x;
    unsigned* sink;
    int ntiles;        // tiles in total; wave w of all takes w, w + nw, ...
    int pattern;       // 0 panel pattern, 1 contiguous
};

__device__ __forceinline__ size_t tile_base(int id, int pattern)
{
    if (pattern) return (size_t)id * 2048;                         // 8 KB contiguous
    const int g = id % NG, m = (id / NG) % (T / 16), r = id / (NG * (T / 16));
    return ((size_t)(4 * r) * T + (size_t)m * 16) * B + (size_t)g * 32;
}
__device__ __forceinline__ unsigned piece_soff(int e, int pattern)
{
    if (pattern) return (unsigned)e * 1024u;
    return (unsigned)((((size_t)(e >> 1) * T + 8 * (e & 1)) * B) * 4);
}

template <int METHOD, int S, int AUX, int WORK = 0, int BUBBLE = 0>
__global__ __launch_bounds__(1024) void probe(Args A)
{
    extern __shared__ __attribute__((aligned(1024))) char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    const int gw = blockIdx.x * nwv + wave, nw = gridDim.x * nwv;
    const unsigned voff = A.pattern ? (unsigned)lane * 16u : (unsigned)(((lane >> 3) * B + (lane & 7) * 4) * 4);
    unsigned acc = 0;
    if (METHOD <= 1) {
        char* st = lds + wave * (S * 8192);
        int issued = 0, done = 0;
        const int mine = (A.ntiles - gw + nw - 1) / nw;
        auto issue = [&](int i) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(A.x + tile_base(gw + i * nw, A.pattern)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(st + (i % S) * 8192 + e * 1024), 16, voff, piece_soff(e, A.pattern), 0, AUX);
        };
        for (; issued < S && issued < mine; ++issued) issue(issued);
        for (; done < mine; ++done) {
            const int younger = issued - done - 1;                  // tiles issued after tile `done`
            if (younger >= 3) wait_vm<24>(); else if (younger == 2) wait_vm<16>(); else if (younger == 1) wait_vm<8>(); else wait_vm<0>();
            if (METHOD == 1) {
                const unsigned a = (unsigned)(size_t)(st + (done % S) * 8192) + lane * 16;
                v4u o[8];
                asm volatile("ds_read_b128 %0, %8\n\tds_read_b128 %1, %8 offset:1024\n\tds_read_b128 %2, %8 offset:2048\n\t"
                             "ds_read_b128 %3, %8 offset:3072\n\tds_read_b128 %4, %8 offset:4096\n\tds_read_b128 %5, %8 offset:5120\n\t"
                             "ds_read_b128 %6, %8 offset:6144\n\tds_read_b128 %7, %8 offset:7168\n\ts_waitcnt lgkmcnt(0)"
                             : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                             : "v"(a));
#pragma unroll
                for (int e = 0; e < 8; ++e) acc += o[e].x ^ o[e].y ^ o[e].z ^ o[e].w;
                if (WORK) {
                    // the panels' math, roughly: per tile 32 exponentials and ~80 plain instructions on dependent-ish data
                    float f = __uint_as_float((acc & 0x007fffffu) | 0x3f800000u);
#pragma unroll
                    for (int w = 0; w < WORK; ++w) {
                        f = __builtin_amdgcn_exp2f(f - 1.5f);
                        f = __builtin_fmaf(f, 0.75f, 0.5f); f = __builtin_fmaf(f, 0.75f, 0.25f); f = __builtin_fmaf(f, 0.5f, 0.5f);
                    }
                    acc ^= __float_as_uint(f);
                }
            }
            if (BUBBLE && (done % 12) == 11) {
                // a task boundary: nothing of the next task is requested before the current one is through, plus the
                // reduction / queue draw (~2 us)
                wait_vm<0>();
                for (int w = 0; w < BUBBLE; ++w) __builtin_amdgcn_s_sleep(72);       // 72 x 64 cycles ~ 2 us
            }
            if (issued < mine && !(BUBBLE && issued > done + 1 && ((done + 1) / 12) != (issued / 12))) { issue(issued); ++issued; }
        }
    } else {
        const int mine = (A.ntiles - gw + nw - 1) / nw;
        v4u r[S][8];
        auto issue = [&](int i, int slot) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(A.x + tile_base(gw + i * nw, A.pattern)), 0, 0x7fffffff, 0x00020000);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[slot][e] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, piece_soff(e, A.pattern), AUX);
        };
        // S tiles in flight: the loop body is unrolled S times so that the slots are compile-time registers
#pragma unroll
        for (int s = 0; s < S; ++s)
            if (s < mine) issue(s, s);
        for (int i = 0; i < mine; i += S) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                if (i + s < mine) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc += r[s][e].x ^ r[s][e].y ^ r[s][e].z ^ r[s][e].w;
                    if (i + s + S < mine) issue(i + s + S, s);
                }
            }
        }
    }
    if (acc == 0x12345u) A.sink[0] = acc;
}

template <int METHOD, int S, int AUX, int WORK = 0, int BUBBLE = 0>
static void run(const char* name, const float* x, unsigned* sink, int cus, int waves, int pattern, hipStream_t st)
{
    Args A{x, sink, 80000, pattern};
    const size_t lds = METHOD <= 1 ? (size_t)waves * S * 8192 : 0;
    const size_t want = lds > 88 * 1024 ? lds : 88 * 1024;          // more than half a CU's LDS: one workgroup per CU
    CK(hipFuncSetAttribute((const void*)probe<METHOD, S, AUX, WORK, BUBBLE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)want));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((probe<METHOD, S, AUX, WORK, BUBBLE>), dim3(cus), dim3(waves * 64), want, st, A);
    CK(hipEventRecord(e0, st));
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((probe<METHOD, S, AUX, WORK, BUBBLE>), dim3(cus), dim3(waves * 64), want, st, A);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1e3 / reps, bytes = 80000.0 * 8192.0;
    printf("%-34s pattern=%s cus=%3d waves=%d  %7.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", name, pattern ? "contig" : "panel ", cus, waves, us,
           bytes / us * 1e-6, bytes / us * 1e-3 / cus);
    fflush(stdout);
}

int main()
{
    float* x; unsigned* sink;
    const size_t bytes = (size_t)T * T * B * 4;
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&sink, 64));
    CK(hipMemset(x, 0, bytes));
    hipStream_t st; CK(hipStreamCreate(&st));
    // second part (round 2): what do the panels' math and their task boundaries cost a compute unit?
    if (getenv("STREAM_PROBE_PART2")) {
        for (int cus : {168, 256}) {
            run<1, 3, 2>("lds-dma 3 stages, read only", x, sink, cus, 4, 0, st);
            run<1, 3, 2, 32>("  + math (32 exp + 96 fma per tile)", x, sink, cus, 4, 0, st);
            run<1, 3, 2, 0, 1>("  + task boundary (drain + 2 us) every 12 tiles", x, sink, cus, 4, 0, st);
            run<1, 3, 2, 32, 1>("  + math + task boundaries: 4 waves x 3 stages", x, sink, cus, 4, 0, st);
            run<1, 3, 2, 32, 1>("  ... 5 waves x 3 stages", x, sink, cus, 5, 0, st);
            run<1, 2, 2, 32, 1>("  ... 6 waves x 2 stages", x, sink, cus, 6, 0, st);
            run<1, 2, 2, 32, 1>("  ... 8 waves x 2 stages", x, sink, cus, 8, 0, st);
            run<1, 4, 2, 32, 1>("  ... 4 waves x 4 stages", x, sink, cus, 4, 0, st);
            run<1, 2, 2, 32, 0>("  math, no boundaries: 8 waves x 2 stages", x, sink, cus, 8, 0, st);
            run<1, 3, 2, 32, 0>("  math, no boundaries: 5 waves x 3 stages", x, sink, cus, 5, 0, st);
        }
        return 0;
    }
    const int cus_list[] = {88, 168, 256};
    for (int pattern = 0; pattern < 2; ++pattern)
        for (int cus : cus_list) {
            run<0, 3, 2>("lds-dma 3 stages nt, no read", x, sink, cus, 4, pattern, st);
            run<1, 3, 2>("lds-dma 3 stages nt + ds_read", x, sink, cus, 4, pattern, st);
            run<1, 3, 0>("lds-dma 3 stages plain + ds_read", x, sink, cus, 4, pattern, st);
            run<1, 2, 2>("lds-dma 2 stages nt + ds_read", x, sink, cus, 8, pattern, st);
            run<1, 4, 2>("lds-dma 4 stages nt + ds_read", x, sink, cus, 4, pattern, st);
            run<2, 2, 2>("regs 2 tiles nt", x, sink, cus, 4, pattern, st);
            run<2, 3, 2>("regs 3 tiles nt", x, sink, cus, 4, pattern, st);
            run<2, 2, 2>("regs 2 tiles nt", x, sink, cus, 8, pattern, st);
            run<2, 3, 2>("regs 3 tiles nt", x, sink, cus, 8, pattern, st);
            run<2, 3, 0>("regs 3 tiles plain", x, sink, cus, 8, pattern, st);
            run<2, 2, 2>("regs 2 tiles nt", x, sink, cus, 16, pattern, st);
        }
    return 0;
}
