// proj_gemm3.hip -- the projection's two NN forms (forward y = x W^T + b with the two extra columns; input gradient dx (+)= dy W) on the
// bf16 matrix instructions with three exact bf16 limbs per operand (bf16x3.h): the opt-in counterpart of proj_gemm.hip's exact-fp32
// kernel (scorer_proj_nn3; LayersTransformer.py:388-397, :406-410 and their autograd).  fp32-grade, not bit-identical.
//
// Same skeleton as score_bwd_gemm3_kernel (scorer_bwd_gemm.hip): a persistent workgroup of 8 waves owns a 128 x 256 output tile, two
// wave groups in opposite phases (ONE barrier per chunk: group 0 multiplies and then splits, group 1 splits and then multiplies), two
// LDS stages of 72 KB (limbs in the matrix instruction's layout), requests and their waits written out.  What makes this product the better customer of the three-limb contraction: the second operand is the SAME small
// matrix for every tile (W: 256-288 rows x 256 columns), so it is split ONCE per call by a kernel of its own into the LDS image of
// its chunks (proj_split_b_kernel: [chunk][limb][column][4 pieces of 8 rows], 48 KB per chunk, 0.4 MB in all: L2-resident) and a
// chunk's share arrives as plain 16-byte copies; only the A rows (one unit of eight values per lane and chunk) are split in the
// loop: 0.9 vector instructions per matrix instruction of the workgroup where the scorer's backward has 2.75.
#include "common.h"
#include "bf16x3.h"

#include <type_traits>

namespace semicrf {

int proj_ncu();

namespace pj3 {

template <int I, int N_, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N_) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N_>(f);
    }
}

#ifndef SEMICRF_P3_MMA_ORDER
#define SEMICRF_P3_MMA_ORDER 0     // 1: a slab's limb products round-robin over the four accumulators (needs all four B operand sets)
#endif
#ifndef SEMICRF_MMA_PRIO
#define SEMICRF_MMA_PRIO 0       // 1: raised wave priority while a wave issues its chunk's matrix instructions
#endif
#ifndef SEMICRF_P3_PHASED
#define SEMICRF_P3_PHASED 1        // 0: all waves in the same order (multiply, then split): within 3 % of the opposite phases
#endif
#ifndef SEMICRF_P3_PROBE
#define SEMICRF_P3_PROBE 0         // 1: per-wave cycle accounting in place of the first output rows (tools/bwd3_probe.py --proj-probe)
#endif
constexpr int GM = 128;            // rows of an output tile
constexpr int GK = 32;             // contraction values per chunk
constexpr int N = 256;             // output columns through the matrix cores (the only width this kernel takes)
constexpr int NW = 4;              // 32-column blocks per wave
constexpr int APL = GM * 64;       // bytes of one limb plane of the A part: 128 rows x 32 values x 2 bytes
constexpr int BPL = N * 64;
constexpr int BOFF = 3 * APL;
constexpr int STAGE = 3 * (APL + BPL);
constexpr int CHUNK_B = 3 * BPL;   // bytes of a chunk's B limbs in the workspace (= their LDS image)
constexpr int NLD_A = 2;           // loads per A register set: two 16-byte pieces of a row (requested two chunks ahead; the six pieces of
                                   // the B image one chunk ahead, IN FRONT of them: what is in flight behind a set and its B are the other set's two)

// B [Brows][ldb] (row = contraction index, whole chunks of rows, zero beyond K: scorer_proj_nn's promise) -> limb image
__global__ __launch_bounds__(256) void proj_split_b_kernel(const float* __restrict__ B, long long ldb, int Brows, char* __restrict__ ws, int nchunks)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;          // (chunk, column, piece)
    const int p = idx & 3, n = (idx >> 2) % N, c = idx / (4 * N);
    if (c >= nchunks) return;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int row = c * GK + 8 * p + i;
        v[i] = row < Brows ? B[(size_t)row * ldb + n] : 0.0f;
    }
    const Limbs3 L = split8((v4f){v[0], v[1], v[2], v[3]}, (v4f){v[4], v[5], v[6], v[7]});
    char* dst = ws + (size_t)c * CHUNK_B + n * 64 + ((p ^ ((n >> 2) & 3)) * 16);
    *(bf16x8*)dst = L.h;
    *(bf16x8*)(dst + BPL) = L.m;
    *(bf16x8*)(dst + 2 * BPL) = L.l;
}

struct Args {
    const float* A; long long lda; int M, K;             // [M][K] (rows = output rows), K % 4 == 0
    const char* Bl; int nchunks;                         // the limb image of B: nchunks x CHUNK_B bytes
    float* out; long long ldout;
    int accumulate;                                      // out += (the part through a second operand is there already)
    const float* bias;                                   // [N] added to the result (not with accumulate), or NULL
    const float* w2; const float* b2;                    // EX: two extra output columns N, N+1 = <A row, w2[j]> + b2[j] ([2][K], [2])
    int zero_cols;                                       // EX: columns N+2 .. N+1+zero_cols are set to zero
};

template <bool EX>
__global__ __launch_bounds__(512, 2) void proj_gemm3_kernel(Args P_)
{
    // (the arguments as locals: referenced through the struct inside the lambdas below, the whole struct lived in scratch memory)
    const float* const a_A = P_.A; const long long a_lda = P_.lda; const int a_M = P_.M, a_K = P_.K; const char* const a_Bl = P_.Bl;
    const int a_nchunks = P_.nchunks; float* const a_out = P_.out; const long long a_ldout = P_.ldout; const int a_accumulate = P_.accumulate;
    const float* const a_bias = P_.bias; const float* const a_w2 = P_.w2; const float* const a_b2 = P_.b2; const int a_zero_cols = P_.zero_cols;
    extern __shared__ __attribute__((aligned(16))) char glds[];    // [2][STAGE] (+ EX: w2 [2][Kpad] floats)
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;           // this wave: rows 32*wm.., columns 128*wn.. of the tile
    const int grp = SEMICRF_P3_PHASED ? wave >> 2 : 0; // (waves w and w + 4 share a SIMD)
    const int nm = (a_M + GM - 1) / GM;
    const int nk = a_nchunks;
    const int Kpad = nk * GK;
    float* const w2l = (float*)(glds + 2 * STAGE);

    // ---- reading lanes: lane = (row, half); the instruction of slab sl takes piece 2*half + sl of the row ----------------
    unsigned rdA[2], rdB[2];
    {
        const int ra = 32 * wm + l31, rb = 32 * NW * wn + l31;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            rdA[sl] = (unsigned)(ra * 64 + (((2 * half + sl) ^ ((ra >> 2) & 3)) * 16));
            rdB[sl] = (unsigned)(BOFF + rb * 64 + (((2 * half + sl) ^ ((rb >> 2) & 3)) * 16));
        }
    }
    // ---- staging lanes: A unit = (row tid / 4, piece tid % 4) along the row; B: pieces tid + 512 i of the chunk's image ------
    const int aRow = tid >> 2, aPiece = tid & 3;
    const unsigned wA = (unsigned)(aRow * 64 + ((aPiece ^ ((aRow >> 2) & 3)) * 16));
    const unsigned bVoff = (unsigned)(tid * 16);

    typedef int v4i __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const void* p, size_t bytes) -> v4i {       // (readfirstlane: an "s" operand must BE in scalar registers)
        const unsigned long long a = (unsigned long long)(uintptr_t)p;
        return (v4i){__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)),
                     __builtin_amdgcn_readfirstlane((int)(unsigned)bytes), 0x00020000};
    };
    // (A: the buffer ends with the matrix -- a last row's contraction values past K that lie past the end read as zero; the others
    // are masked in the split, below)
    const v4i ra = make_rsrc(a_A, ((size_t)(a_M - 1) * a_lda + a_K) * 4);
    const v4i rb = make_rsrc(a_Bl, (size_t)nk * CHUNK_B);

    // ---- request side (identical in all waves): item = row tile, all of them nk chunks long --------------------------------
    int nx_round = 0, nx_j = 0;
    bool nx_valid = (int)blockIdx.x < nm;
    if (!nx_valid) return;                             // uniform
    int nx_mi = (int)blockIdx.x;
    unsigned aVoff = 0;
    auto set_item_offsets = [&]() __attribute__((always_inline)) {
        const int row = nx_mi * GM + aRow;
        aVoff = (unsigned)(((size_t)(row < a_M ? row : a_M - 1) * a_lda + aPiece * 8) * 4);     // (a clamped row is never stored)
    };
    set_item_offsets();

    struct Regs { v4f alo, ahi; };
    v4f bq[6];                                          // the B image's pieces of the NEXT chunk to be split (one set: the registers do not hold two)
    int nb_j = 0;                                       // its chunk (the image does not depend on the item)
    struct Meta { bool valid, last; int mi, k0; };
    auto fetch = [&](Regs& g, Meta& m, bool withB = true) __attribute__((always_inline)) {
        m.valid = nx_valid;
        const unsigned k0 = (unsigned)nx_j * GK;
        {
            // (A: the whole offset in the per-lane part, which is what the buffer's range check looks at)
            const unsigned va = aVoff + k0 * 4, zero = 0u;
            if (withB) {
                const unsigned s0 = __builtin_amdgcn_readfirstlane((unsigned)nb_j * CHUNK_B);
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bq[0]) : "v"(bVoff), "s"(rb), "s"(s0));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bq[1]) : "v"(bVoff), "s"(rb), "s"(s0 + 8192u));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bq[2]) : "v"(bVoff), "s"(rb), "s"(s0 + 16384u));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bq[3]) : "v"(bVoff), "s"(rb), "s"(s0 + 24576u));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bq[4]) : "v"(bVoff), "s"(rb), "s"(s0 + 32768u));
                asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(bq[5]) : "v"(bVoff), "s"(rb), "s"(s0 + 40960u));
                nb_j = nb_j + 1 == nk ? 0 : nb_j + 1;
            }
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g.alo) : "v"(va), "s"(ra), "s"(zero));
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(g.ahi) : "v"(va), "s"(ra), "s"(zero));
        }
        m.last = nx_j + 1 == nk;
        m.mi = nx_mi;
        m.k0 = (int)k0;
        if (nx_valid && ++nx_j == nk) {
            const int mi2 = (int)blockIdx.x + (nx_round + 1) * (int)gridDim.x;
            if (mi2 < nm) {
                ++nx_round;
                nx_mi = mi2; nx_j = 0;
                set_item_offsets();
            } else {
                nx_valid = false;
                nx_j = nk - 1;
            }
        }
    };
    // the set's loads and the B pieces requested behind it have landed (the other set's two younger loads may be in flight)
    auto landed = [&](Regs& g) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD_A));
        asm volatile("" : "+v"(g.alo), "+v"(g.ahi));
#pragma unroll
        for (int i = 0; i < 6; ++i) asm volatile("" : "+v"(bq[i]));
    };
    float e0 = 0.0f, e1 = 0.0f;                        // EX: this lane's part of <row, w2[j]>
    float b2v[2] = {0.f, 0.f};
    if (EX) { b2v[0] = a_b2[0]; b2v[1] = a_b2[1]; }
    auto convert = [&](Regs& g, const Meta& m, int stage) __attribute__((always_inline)) {
        char* base = glds + stage * STAGE;
        v4f alo = g.alo, ahi = g.ahi;
        if (m.k0 + GK > a_K) {                          // the last chunk of a contraction that is no whole number of chunks (uniform):
            const int kk = m.k0 + 8 * aPiece;           // values past K are the next row's (or padding): they must not meet anything
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (kk + i >= a_K) alo[i] = 0.0f;
                if (kk + 4 + i >= a_K) ahi[i] = 0.0f;
            }
        }
        {
            const Limbs3 L = split8(alo, ahi);
            *(bf16x8*)(base + wA) = L.h;
            *(bf16x8*)(base + APL + wA) = L.m;
            *(bf16x8*)(base + 2 * APL + wA) = L.l;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) *(v4f*)(base + BOFF + i * 8192 + tid * 16) = bq[i];
        if (EX && m.valid) {
            const float* w0 = w2l + m.k0 + 8 * aPiece;
            const v4f x0 = *(const v4f*)w0, x1 = *(const v4f*)(w0 + 4), y0 = *(const v4f*)(w0 + Kpad), y1 = *(const v4f*)(w0 + Kpad + 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) { e0 = fmac1(alo[i], x0[i], e0); e1 = fmac1(alo[i], y0[i], e1); }
#pragma unroll
            for (int i = 0; i < 4; ++i) { e0 = fmac1(ahi[i], x1[i], e0); e1 = fmac1(ahi[i], y1[i], e1); }
            if (m.last) {
                float t0 = e0 + __shfl_xor(e0, 1), t1 = e1 + __shfl_xor(e1, 1);
                t0 += __shfl_xor(t0, 2);
                t1 += __shfl_xor(t1, 2);
                const int mrow = m.mi * GM + aRow;
                if (aPiece == 0 && mrow < a_M) {
                    float* o = a_out + (size_t)mrow * a_ldout + N;
                    o[0] = t0 + b2v[0];
                    o[1] = t1 + b2v[1];
                    for (int z = 0; z < a_zero_cols; ++z) o[2 + z] = 0.0f;
                }
                e0 = e1 = 0.0f;
            }
        }
    };

    f32x16 acc[NW];
    // 2 NW groups of six matrix instructions per chunk: group i = (slab i / NW, column block i % NW); the reads of group i + 1 are
    // issued before the instructions of group i (two operand sets, alternating: the register budget has no room for a whole slab's)
    auto multiply = [&](int stage) __attribute__((always_inline)) {
        const char* base = glds + stage * STAGE;
#if SEMICRF_P3_MMA_ORDER == 0
        Limbs3 A, B[2];
        auto ldA = [&](Limbs3& L, int sl) __attribute__((always_inline)) {
            L.h = *(const bf16x8*)(base + rdA[sl]);
            L.m = *(const bf16x8*)(base + APL + rdA[sl]);
            L.l = *(const bf16x8*)(base + 2 * APL + rdA[sl]);
        };
        auto ldB = [&](Limbs3& L, int sl, int t) __attribute__((always_inline)) {
            L.h = *(const bf16x8*)(base + rdB[sl] + t * 2048);
            L.m = *(const bf16x8*)(base + BPL + rdB[sl] + t * 2048);
            L.l = *(const bf16x8*)(base + 2 * BPL + rdB[sl] + t * 2048);
        };
        ldA(A, 0);
        ldB(B[0], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 2 * NW>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int sl = i / NW, t = i % NW;
            if constexpr (i + 1 < 2 * NW) {
                constexpr int sl2 = (i + 1) / NW, t2 = (i + 1) % NW;
                if constexpr (t2 != 0) ldB(B[(i + 1) & 1], sl2, t2);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = mma6(A, B[i & 1], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i + 1 < 2 * NW && (i + 1) % NW == 0) {            // the next slab: its A limbs go where this slab's were
                ldA(A, (i + 1) / NW);
                ldB(B[(i + 1) & 1], (i + 1) / NW, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
#else
        // a slab at a time: all its operands, then the six limb products round-robin over the accumulators (independent neighbours)
        Limbs3 A, B[NW];
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            A.h = *(const bf16x8*)(base + rdA[sl]);
            A.m = *(const bf16x8*)(base + APL + rdA[sl]);
            A.l = *(const bf16x8*)(base + 2 * APL + rdA[sl]);
#pragma unroll
            for (int t = 0; t < NW; ++t) {
                B[t].h = *(const bf16x8*)(base + rdB[sl] + t * 2048);
                B[t].m = *(const bf16x8*)(base + BPL + rdB[sl] + t * 2048);
                B[t].l = *(const bf16x8*)(base + 2 * BPL + rdB[sl] + t * 2048);
            }
#pragma unroll
            for (int t = 0; t < NW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B[t].l, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l, B[t].h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B[t].m, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B[t].m, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B[t].h, acc[t], 0, 0, 0);
#pragma unroll
            for (int t = 0; t < NW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B[t].h, acc[t], 0, 0, 0);
        }
#endif
    };

    // the two extra weight rows (forward), zero-padded to whole chunks; the bias values of this lane's columns
    if (EX) {
        for (int i = tid; i < 2 * Kpad; i += 512) {
            const int j = i / Kpad, k = i % Kpad;
            w2l[i] = k < a_K ? a_w2[(size_t)j * a_K + k] : 0.0f;
        }
    }
    float bvv[NW] = {0.f, 0.f, 0.f, 0.f};
    if (a_bias && !a_accumulate)
#pragma unroll
        for (int t = 0; t < NW; ++t) bvv[t] = a_bias[32 * NW * wn + 32 * t + l31];
#pragma unroll
    for (int t = 0; t < NW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = bvv[t];         // (the bias as the accumulators' start value: added at the end it cost 64 registers of splats)
    __syncthreads();

    Regs gX, gY;
    Meta mX = {false, false, 0, 0}, mY = {false, false, 0, 0};
    auto sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my limb stores are in the LDS ...
        __builtin_amdgcn_s_barrier();                            // ... and so are everybody's; everybody is done reading the other stage
        asm volatile("" ::: "memory");
    };
    // Schedule: ONE barrier per chunk.  Behind it the limbs of chunk n are complete in stage P and nobody reads the other stage any
    // more; group 0 multiplies chunk n and then splits its share of chunk n+1 into the other stage, group 1 does the same in the
    // opposite order -- so on every SIMD one wave's matrix instructions run next to the other wave's split, and whichever is
    // shorter does not wait for a barrier in the middle of the chunk (score_bwd_gemm3_kernel has one there: its second group runs the
    // loop a slot later and splits into the stage it has just read).  X is the register set split in the steps with P = 0.
    fetch(gY, mY);
    fetch(gX, mX, false);
    landed(gY);
    convert(gY, mY, 0);
    fetch(gY, mY);

    int cur_round = 0, mi = (int)blockIdx.x, j = 0;
    // ---- the item's 128 x 256 block (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); rows >= M are dropped
    //      by the buffer's range check; then ONE wait for everything in flight (see score_bwd_gemm3_kernel) ----
    auto finish = [&]() __attribute__((always_inline)) -> bool {
        if (++j < nk) return true;
        const auto ro = __builtin_amdgcn_make_buffer_rsrc((void*)a_out, 0, (int)(((size_t)(a_M - 1) * a_ldout + N) * 4), 0x00020000);
        const unsigned v0 = (unsigned)(((size_t)(mi * GM + 32 * wm + 4 * half) * a_ldout + 32 * NW * wn + l31) * 4);
        if (a_accumulate) {                                      // (the loads first, all of them: one wait)
#pragma unroll
            for (int t = 0; t < NW; ++t) {
                float old[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned vr = v0 + (unsigned)((size_t)((r & 3) + 8 * (r >> 2)) * a_ldout * 4);
                    old[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ro, vr, t * 128, 0));       // rows >= M: zero
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] += old[r];
                __builtin_amdgcn_sched_barrier(0);               // (a column block at a time: all 64 old values at once do not fit the registers)
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned vr = v0 + (unsigned)((size_t)((r & 3) + 8 * (r >> 2)) * a_ldout * 4);
#pragma unroll
            for (int t = 0; t < NW; ++t) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][r]), ro, vr, t * 128, 0);
                acc[t][r] = bvv[t];                              // the next item starts from the bias (zero with accumulate)
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
        j = 0;
        ++cur_round;
        mi = (int)blockIdx.x + cur_round * (int)gridDim.x;
        return mi < nm;
    };
#if SEMICRF_P3_PROBE
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_readcyclecounter();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();
#define P3_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); pc[i] += now_ - pt; pt = now_; } while (0)
#else
#define P3_STAMP(i) do { } while (0)
#endif
    auto step = [&](auto PC, Regs& gn, Meta& mn) __attribute__((always_inline)) -> bool {
        constexpr int Pst = decltype(PC)::value;
        bool more = true;
        sync();
        P3_STAMP(2);
        if (grp == 0) {
            if (SEMICRF_MMA_PRIO) __builtin_amdgcn_s_setprio(1);
            multiply(Pst);
            if (SEMICRF_MMA_PRIO) __builtin_amdgcn_s_setprio(0);
            P3_STAMP(0);
            more = finish();
            P3_STAMP(1);
        }
        landed(gn);
        P3_STAMP(3);
        convert(gn, mn, Pst ^ 1);
        P3_STAMP(4);
        fetch(gn, mn);
        P3_STAMP(5);
        if (grp != 0) {
            if (SEMICRF_MMA_PRIO) __builtin_amdgcn_s_setprio(1);
            multiply(Pst);
            if (SEMICRF_MMA_PRIO) __builtin_amdgcn_s_setprio(0);
            P3_STAMP(0);
            more = finish();
            P3_STAMP(1);
        }
        return more;
    };
    while (true) {
        if (!step(std::integral_constant<int, 0>{}, gX, mX)) break;
        if (!step(std::integral_constant<int, 1>{}, gY, mY)) break;
    }
#if SEMICRF_P3_PROBE
    pc[7] = __builtin_amdgcn_s_memrealtime() - rt0;              // (cycle accounting over the first rows of `out`: probe builds only)
    __syncthreads();
    if (lane == 0)
        for (int i = 0; i < 8; ++i) a_out[(size_t)(blockIdx.x * 8 + wave) * 8 + i] = (float)pc[i];
#endif
#undef P3_STAMP
}

// ---------------------------------------------------------------------------------------------
// The weight gradient's matrix part on the same contraction: part[sl][r][n] = sum over the slice's rows m of dy[m][r] x[m][n]
// (launch_proj_tn's split-K partial slabs; its fixed-order reduction, the two extra rows and the bias gradient's extras stay as they
// are).  Both operands have the contraction index as their ROW index in memory: column units for dy (one column, eight rows: 4-byte
// loads, the row in the scalar offset) and column-pair units for x (8-byte loads), both split in the loop -- 2.75 vector
// instructions per matrix instruction like the scorer's backward, whose skeleton this is; the bias gradient (column sums of dy) falls
// out of the staged dy values, combined over a column's four pieces through the LDS.
// ---------------------------------------------------------------------------------------------
struct ArgsT {
    const float* A; long long lda; int M, lda_cols;      // dy [M][lda] (lda_cols: columns a lane may touch, <= lda)
    const float* X; long long ldx;                        // x [M][256]
    float* part; int Mp;                                  // [S][Mp][256]
    float* pbias; int bias_pitch;                         // [S][bias_pitch]
    int nslices, kslice;                                  // row slices of the contraction (chunks per slice)
};

__global__ __launch_bounds__(512, 2) void proj_tn3_kernel(ArgsT P_)
{
    const float* const a_A = P_.A; const long long a_lda = P_.lda; const int a_M = P_.M, a_cols = P_.lda_cols;
    const float* const a_X = P_.X; const long long a_ldx = P_.ldx; float* const a_part = P_.part; const int a_Mp = P_.Mp;
    float* const a_pbias = P_.pbias; const int a_bpitch = P_.bias_pitch; const int a_S = P_.nslices, a_ks = P_.kslice;
    extern __shared__ __attribute__((aligned(16))) char glds[];    // [2][STAGE] | column sums [2][4][128] floats
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int grp = wave >> 2;
    const int nm = a_Mp / GM;
    const int nkt = (a_M + GK - 1) / GK;
    const int nitems = nm * a_S;
    float* const csl = (float*)(glds + 2 * STAGE);

    unsigned rdA[2], rdB[2];
    {
        const int ra = 32 * wm + l31, rb = 32 * NW * wn + l31;
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            rdA[sl] = (unsigned)(ra * 64 + (((2 * half + sl) ^ ((ra >> 2) & 3)) * 16));
            rdB[sl] = (unsigned)(BOFF + rb * 64 + (((2 * half + sl) ^ ((rb >> 2) & 3)) * 16));
        }
    }
    // staging lanes: dy -- unit = (column tid % 128 of the tile, piece tid / 128) across rows; x -- units (column 2 cp + j, piece wave / 2)
    const int aRow = tid & 127, aPiece = wave >> 1;
    const unsigned wA = (unsigned)(aRow * 64 + ((aPiece ^ ((aRow >> 2) & 3)) * 16));
    const int bCp = (wave & 1) * 64 + lane, bPiece = wave >> 1;
    unsigned wB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int d = 2 * bCp + j;
        wB[j] = (unsigned)(BOFF + d * 64 + ((bPiece ^ ((d >> 2) & 3)) * 16));
    }
    const unsigned bVoff = (unsigned)(bCp * 8);

    typedef int v4i __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const void* p, size_t bytes) -> v4i {
        const unsigned long long a = (unsigned long long)(uintptr_t)p;
        return (v4i){__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)((a >> 32) & 0xffffu)),
                     __builtin_amdgcn_readfirstlane((int)(unsigned)bytes), 0x00020000};
    };
    const v4i ra = make_rsrc(a_A, ((size_t)(a_M - 1) * a_lda + a_cols) * 4);
    const v4i rb = make_rsrc(a_X, ((size_t)(a_M - 1) * a_ldx + N) * 4);
    const unsigned lda4 = (unsigned)(a_lda * 4), ldx4 = (unsigned)(a_ldx * 4);
    const unsigned amax = (unsigned)(a_M - 1) * lda4, xmax = (unsigned)(a_M - 1) * ldx4;

    auto item_of = [&](int n, int& mi, int& sl, int& kbeg, int& nk) __attribute__((always_inline)) -> bool {
        if (n >= nitems) return false;
        mi = __builtin_amdgcn_readfirstlane(n % nm);
        sl = __builtin_amdgcn_readfirstlane(n / nm);
        kbeg = sl * a_ks;
        nk = nkt - kbeg < a_ks ? nkt - kbeg : a_ks;
        return nk > 0;
    };
    int nx_n = (int)blockIdx.x, nx_mi = 0, nx_sl = 0, nx_kbeg = 0, nx_nk = 0, nx_j = 0;
    bool nx_valid = item_of(nx_n, nx_mi, nx_sl, nx_kbeg, nx_nk);
    if (!nx_valid) return;                             // uniform (slices are never empty: S = ceil(chunks / kslice))
    unsigned aVoff = 0;
    auto set_item_offsets = [&]() __attribute__((always_inline)) {
        const int col = nx_mi * GM + aRow;
        aVoff = (unsigned)((col < a_cols ? col : a_cols - 1) * 4);            // (a clamped column's output row is beyond R: never reduced)
    };
    set_item_offsets();

    constexpr int NLD = 8 + 8;
    struct Regs { float a[8]; f32x2 b[8]; };
    struct Meta { bool valid, last; int mi, sl, m0; };
    auto fetch = [&](Regs& g, Meta& m) __attribute__((always_inline)) {
        m.valid = nx_valid;
        const unsigned m0 = (unsigned)(nx_kbeg + nx_j) * GK;
        {
            unsigned sr = (m0 + 8 * aPiece) * lda4;             // rows past M: the last row's values, zeroed in the split
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned so = sr < amax ? sr : amax;
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(g.a[i]) : "v"(aVoff), "s"(ra), "s"(so));
                sr += lda4;
            }
        }
        {
            unsigned sr = (m0 + 8 * bPiece) * ldx4;             // rows past M meet dy == 0: any finite value will do
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned so = sr < xmax ? sr : xmax;
                asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(g.b[i]) : "v"(bVoff), "s"(rb), "s"(so));
                sr += ldx4;
            }
        }
        m.last = nx_j + 1 == nx_nk;
        m.mi = nx_mi;
        m.sl = nx_sl;
        m.m0 = (int)m0;
        if (nx_valid && ++nx_j == nx_nk) {
            int mi2, sl2, kbeg2, nk2;
            if (item_of(nx_n + (int)gridDim.x, mi2, sl2, kbeg2, nk2)) {
                nx_n += (int)gridDim.x;
                nx_mi = mi2; nx_sl = sl2; nx_kbeg = kbeg2; nx_nk = nk2; nx_j = 0;
                set_item_offsets();
            } else {
                nx_valid = false;
                nx_j = nx_nk - 1;
            }
        }
    };
    auto landed = [&](Regs& g) __attribute__((always_inline)) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(g.a[i]));
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(g.b[i]));
    };
    float cs = 0.0f;                                    // this lane's part of its column's sum over the slice (bias gradient)
    int cpar = 0;                                       // which of the two column-sum buffers the item being split uses
    auto convert = [&](Regs& g, const Meta& m, int stage) __attribute__((always_inline)) {
        char* base = glds + stage * STAGE;
        if (m.m0 + GK > a_M) {                          // the contraction's last chunk (uniform): rows past M must not meet anything
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (m.m0 + 8 * aPiece + i >= a_M) g.a[i] = 0.0f;
        }
        {
            const Limbs3 L = split8((v4f){g.a[0], g.a[1], g.a[2], g.a[3]}, (v4f){g.a[4], g.a[5], g.a[6], g.a[7]});
            *(bf16x8*)(base + wA) = L.h;
            *(bf16x8*)(base + APL + wA) = L.m;
            *(bf16x8*)(base + 2 * APL + wA) = L.l;
        }
        if (m.valid) {
            cs = add1(cs, add1(add1(add1(g.a[0], g.a[1]), add1(g.a[2], g.a[3])), add1(add1(g.a[4], g.a[5]), add1(g.a[6], g.a[7]))));
            if (m.last) {                               // parked for whoever stores the item's bias sums (behind the next barrier)
                csl[cpar * 512 + aPiece * 128 + aRow] = cs;
                cs = 0.0f;
                cpar ^= 1;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const Limbs3 L = split8((v4f){g.b[0][j], g.b[1][j], g.b[2][j], g.b[3][j]}, (v4f){g.b[4][j], g.b[5][j], g.b[6][j], g.b[7][j]});
            *(bf16x8*)(base + wB[j]) = L.h;
            *(bf16x8*)(base + BPL + wB[j]) = L.m;
            *(bf16x8*)(base + 2 * BPL + wB[j]) = L.l;
        }
    };

    f32x16 acc[NW];
#pragma unroll
    for (int t = 0; t < NW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    auto multiply = [&](int stage) __attribute__((always_inline)) {
        const char* base = glds + stage * STAGE;
        Limbs3 A, B[2];
        auto ldA = [&](Limbs3& L, int sl) __attribute__((always_inline)) {
            L.h = *(const bf16x8*)(base + rdA[sl]);
            L.m = *(const bf16x8*)(base + APL + rdA[sl]);
            L.l = *(const bf16x8*)(base + 2 * APL + rdA[sl]);
        };
        auto ldB = [&](Limbs3& L, int sl, int t) __attribute__((always_inline)) {
            L.h = *(const bf16x8*)(base + rdB[sl] + t * 2048);
            L.m = *(const bf16x8*)(base + BPL + rdB[sl] + t * 2048);
            L.l = *(const bf16x8*)(base + 2 * BPL + rdB[sl] + t * 2048);
        };
        ldA(A, 0);
        ldB(B[0], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 2 * NW>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int t = i % NW;
            if constexpr (i + 1 < 2 * NW) {
                constexpr int sl2 = (i + 1) / NW, t2 = (i + 1) % NW;
                if constexpr (t2 != 0) ldB(B[(i + 1) & 1], sl2, t2);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[t] = mma6(A, B[i & 1], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (i + 1 < 2 * NW && (i + 1) % NW == 0) {
                ldA(A, (i + 1) / NW);
                ldB(B[(i + 1) & 1], (i + 1) / NW, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
    };

    Regs gX, gY;
    Meta mX = {false, false, 0, 0, 0}, mY = {false, false, 0, 0, 0};
    auto sync = [&]() __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    fetch(gY, mY);
    fetch(gX, mX);
    landed(gY);
    convert(gY, mY, 0);
    fetch(gY, mY);

    int cur_n = (int)blockIdx.x, mi = 0, sl = 0, kbeg = 0, nk = 0, j = 0;
    (void)item_of(cur_n, mi, sl, kbeg, nk);
    int bpar = 0;                                       // the column-sum buffer of the item being multiplied
    auto finish = [&]() __attribute__((always_inline)) -> bool {
        if (++j < nk) return true;
        float* ob = a_part + ((size_t)sl * a_Mp + (size_t)mi * GM) * N;
        const auto ro = __builtin_amdgcn_make_buffer_rsrc((void*)ob, 0, (int)((size_t)GM * N * 4), 0x00020000);
        const unsigned v0 = (unsigned)(((32 * wm + 4 * half) * N + 32 * NW * wn + l31) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned vr = v0 + (unsigned)(((r & 3) + 8 * (r >> 2)) * N * 4);
#pragma unroll
            for (int t = 0; t < NW; ++t) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][r]), ro, vr, t * 128, 0);
                acc[t][r] = 0.0f;
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
        j = 0;
        cur_n += (int)gridDim.x;
        return item_of(cur_n, mi, sl, kbeg, nk);
    };
    auto step = [&](auto PC, Regs& gn, Meta& mn) __attribute__((always_inline)) -> bool {
        constexpr int Pst = decltype(PC)::value;
        bool more = true;
        sync();
        // chunk (kbeg + j) of item (mi, sl) is complete in stage Pst; if it is the item's last one, its column sums are in the LDS as well
        if (j + 1 == nk) {
            if (tid < 128) {
                const float* c4 = csl + bpar * 512 + tid;
                a_pbias[(size_t)sl * a_bpitch + mi * GM + tid] = (c4[0] + c4[128]) + (c4[256] + c4[384]);
            }
            bpar ^= 1;
        }
        if (grp == 0) {
            multiply(Pst);
            more = finish();
        }
        landed(gn);
        convert(gn, mn, Pst ^ 1);
        fetch(gn, mn);
        if (grp != 0) {
            multiply(Pst);
            more = finish();
        }
        return more;
    };
    while (true) {
        if (!step(std::integral_constant<int, 0>{}, gX, mX)) break;
        if (!step(std::integral_constant<int, 1>{}, gY, mY)) break;
    }
}

}  // namespace pj3

bool proj_gemm_supported(long long M, int K, int N, const void* A, long long lda, const void* B, long long ldb, const void* out,
                         long long ldout);

size_t proj_nn3_workspace_bytes(int K, int N)
{
    if (N != pj3::N || K < 4) return 0;
    return (size_t)((K + pj3::GK - 1) / pj3::GK) * pj3::CHUNK_B + 256;
}

// 0: ran; 1: the shape is not this kernel's (the caller runs the exact kernel)
int launch_proj_nn3(const float* A, long long lda, long long M, int K, const float* B, long long ldb, int N, float* out, long long ldout,
                    const float* bias, const float* w2, const float* b2, int zero_cols, int accumulate, void* ws, size_t ws_bytes,
                    hipStream_t stream)
{
    if (N != pj3::N || !proj_gemm_supported(M, K, N, A, lda, B, ldb, out, ldout)) return 1;
    if (!ws || ((uintptr_t)ws & 15) || ws_bytes < proj_nn3_workspace_bytes(K, N)) return 1;
    if ((long long)M * ldout * 4 >= (1ll << 31)) return 1;
    const int nchunks = (K + pj3::GK - 1) / pj3::GK;
    const size_t lds = (size_t)2 * pj3::STAGE + (w2 ? (size_t)2 * nchunks * pj3::GK * 4 : 0);
    if (lds > 160 * 1024 - 512) return 1;                    // (before anything is enqueued: the caller runs the exact kernel instead)
    hipLaunchKernelGGL(pj3::proj_split_b_kernel, dim3((nchunks * 4 * pj3::N + 255) / 256), dim3(256), 0, stream, B, ldb, nchunks * pj3::GK,
                       (char*)ws, nchunks);
    pj3::Args P{};
    P.A = A; P.lda = lda; P.M = (int)M; P.K = K; P.Bl = (const char*)ws; P.nchunks = nchunks; P.out = out; P.ldout = ldout;
    P.accumulate = accumulate; P.bias = bias; P.w2 = w2; P.b2 = b2; P.zero_cols = zero_cols;
    static PerDeviceOnce attr_once[2];
    const long long nitems = (M + pj3::GM - 1) / pj3::GM;
    const int ncu = proj_ncu();
    const int grid = nitems < ncu ? (int)nitems : ncu;
    if (w2) {
        if (attr_once[0].first())
            (void)hipFuncSetAttribute((const void*)pj3::proj_gemm3_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        hipLaunchKernelGGL((pj3::proj_gemm3_kernel<true>), dim3(grid), dim3(512), lds, stream, P);
    } else {
        if (attr_once[1].first())
            (void)hipFuncSetAttribute((const void*)pj3::proj_gemm3_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
        hipLaunchKernelGGL((pj3::proj_gemm3_kernel<false>), dim3(grid), dim3(512), lds, stream, P);
    }
    return 0;
}

// the matrix part of launch_proj_tn (proj_gemm.hip) on the three-limb kernels: 0 = ran
int launch_proj_tn3_core(const float* A, long long lda, long long M, int R, const float* X, long long ldx, int N, float* part, int Mp,
                         float* pbias, int bias_pitch, int S, int kslice, hipStream_t stream)
{
    if (N != pj3::N || Mp % pj3::GM != 0 || M * lda * 4 >= (1ll << 31) || M * ldx * 4 >= (1ll << 31)) return 1;
    pj3::ArgsT P{};
    P.A = A; P.lda = lda; P.M = (int)M; P.lda_cols = (int)(lda < Mp ? lda : Mp);
    P.X = X; P.ldx = ldx; P.part = part; P.Mp = Mp; P.pbias = pbias; P.bias_pitch = bias_pitch; P.nslices = S; P.kslice = kslice;
    (void)R;
    const size_t lds = (size_t)2 * pj3::STAGE + 4096;
    static PerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)pj3::proj_tn3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    const long long nitems = (long long)(Mp / pj3::GM) * S;
    const int ncu = proj_ncu();
    const int grid = nitems < ncu ? (int)nitems : ncu;
    hipLaunchKernelGGL(pj3::proj_tn3_kernel, dim3(grid), dim3(512), lds, stream, P);
    return 0;
}

}  // namespace semicrf
