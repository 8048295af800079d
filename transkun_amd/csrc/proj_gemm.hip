// proj_gemm.hip -- the interval scorer's projection (the nn.Linear in front of the contraction, LayersTransformer.py:388-397,
// :406-410) and its autograd as exact-fp32 matrix-core GEMMs of this library: no BLAS call is left on the scorer's path.
//
//   forward    Y[M][N + 4] = X[M][K] W^T + b          N = 64 NW columns through the matrix cores, the two extra columns of the
//                                                     packed outputs ([q | diag | 0 0 0], [z | c | diag | 0 0]) as dot products of
//                                                     the A values the kernel reads anyway, the last ones zero
//   input grad dX[M][N] (+)= dY[M][K'] W[K'][N]       the same kernel (row-major B), K' = the packed width (260)
//   weight grad dW[K'][N] = dY^T X, db[K'] = 1^T dY   contraction over the M rows: split over row slices, partial sums in a
//                                                     workspace, reduced in a fixed order; the bias gradient falls out of the A
//                                                     values, the two extra rows out of the B values
//
// One kernel template, the machinery of scorer_bwd_gemm.hip: a persistent workgroup of 8 waves owns a 128 x N output tile (wave =
// 32 rows x N/2 columns, v_mfma_f32_32x32x2_f32: an exact fp32 fmaf chain), operand chunks of 32 contraction values arrive by
// `buffer_load ... lds` in 1 KB pieces, three stages deep, one s_barrier per chunk; A rows are XOR-swizzled by the loading lanes so
// that the row-per-lane ds_read_b128 has no bank conflicts; in the transposed form (AT) the stage rows are the contraction index
// and the same data is walked with ds_read_b32.  Everything a lane asks for outside its matrix is masked by the buffer's range
// check (the row part of every offset travels in the VGPR), so that no value from outside ever meets a matrix instruction.
#include "common.h"

#include <type_traits>

namespace semicrf {

namespace pj {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int GM = 128;            // rows of an output tile
constexpr int GK = 32;             // contraction values per chunk
constexpr int GNS = 3;             // LDS stages
constexpr int GA_BYTES = GM * GK * 4;      // 16 KB: the A part of a stage
constexpr unsigned OOB = 0xfffffff0u;      // a byte offset no buffer holds: the load returns zeros

__device__ __forceinline__ unsigned lds_addr(const void* p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

struct Args {
    const float* A; long long lda; int Arows, Acols;     // !AT: [M][K] (rows = output rows); AT: [M][>= 128 tiles] (rows = contraction)
    const float* B; long long ldb; int Brows;             // [K][64 NW] row-major, Brows = K (rows beyond read as zero)
    float* out; long long ldout; int Mout;                // !AT: [M][...]; AT: partial slabs [S][Mout_pad][64 NW]
    int K;                                                // contraction length
    int accumulate;                                       // !AT: out += (the part through a second operand is there already)
    const float* bias;                                    // !AT: [64 NW] added to the result, or NULL
    const float* w2; const float* b2;                     // !AT: two extra output columns N, N+1 = <A row, w2[j]> + b2[j] ([2][K], [2]); NULL: none
    int zero_cols;                                        // !AT with w2: columns N+2 .. N+1+zero_cols are set to zero
    int ktail;                                            // !AT: 1 = the last chunk holds at most 8 contraction values (a packed width of 256 + 8): it runs as ONE
                                                          // group of matrix instructions (k = 4 half + component) instead of four
    int nslices, kslice;                                  // AT: row slices of the contraction (chunks per slice)
    int extra_col0;                                       // AT: A columns extra_col0, +1 give two extra output rows (or -1)
    float* part_bias;                                     // AT: [S][bias_pitch] column sums of A over the slice
    int bias_pitch;
    float* part_extra;                                    // AT: [S][2][64 NW]
};

template <bool AT, int NW, bool EX>
__global__ __launch_bounds__(512, 2) void proj_gemm_kernel(Args P)
{
    constexpr int D = 64 * NW;
    constexpr int GB_BYTES = GK * D * 4;               // the B part of a stage: 32 rows
    constexpr int GSTAGE = GA_BYTES + GB_BYTES;
    constexpr int RPP = 4 / NW;                        // B rows per 1 KB piece
    constexpr int NLOAD = 2 + NW;                      // pieces per wave and chunk
    extern __shared__ __attribute__((aligned(16))) char glds[];    // [GNS][GSTAGE] (+ !AT: w2 [2][Kpad])
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;           // this wave: rows 32*wm.., columns 32*NW*wn.. of the tile
    const unsigned lds0 = lds_addr(glds);
    const int nkt = (P.K + GK - 1) / GK;               // chunks of the whole contraction axis
    const int nm = AT ? (P.Mout + GM - 1) / GM : (P.Arows + GM - 1) / GM;
    // (item indices are 32-bit on purpose: with 64-bit ones this compiler kept the item's row index in an s_cselect on a stale SCC
    // -- the 64-bit compare had moved to the VALU -- and every item after a workgroup's first was stored over its first)
    const int nitems = AT ? nm * P.nslices : nm;
    // the two extra columns are a COMPILE-TIME variant: with a run-time flag the reads' registers become conditional values, the
    // compiler copies them ahead of the wait they are tied to and the copies hold whatever the registers held before
    constexpr bool extras = !AT && EX;                 // (AT: the two extra rows are proj_extra_tn_kernel's)
    const int Kpad = nkt * GK;
    float* const w2l = (float*)(glds + GNS * GSTAGE);  // !AT: [2][Kpad]

    auto item_of = [&](int n, int& mi, int& sl, int& kbeg, int& nk) -> bool {
        mi = 0; sl = 0; kbeg = 0; nk = 0;
        if (n >= nitems) return false;
        if (AT) {
            mi = n % nm; sl = n / nm;
            kbeg = sl * P.kslice;
            nk = nkt - kbeg < P.kslice ? nkt - kbeg : P.kslice;
            if (nk < 0) nk = 0;
        } else {
            mi = n; sl = 0; kbeg = 0; nk = nkt;
        }
        return true;
    };

    // ---- LDS read addresses (bytes within a stage): contraction order inside a chunk as in scorer_bwd_gemm.hip -----------------
    unsigned rdA[4];
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
        const int row = 32 * wm + l31;
        rdA[mm] = (unsigned)(row * 128 + (((4 * half + mm) ^ ((row >> 1) & 7)) * 16));
    }
    const unsigned rdAT = (unsigned)(half * 16 * (GM * 4) + (32 * wm + l31) * 4);
    const unsigned rdAtail = (unsigned)((32 * wm + l31) * 128 + ((half ^ (((32 * wm + l31) >> 1) & 7)) * 16));
    const unsigned rdBtail = (unsigned)(GA_BYTES + half * 4 * (D * 4) + (32 * NW * wn + l31) * 4);
    const unsigned rdB = (unsigned)(GA_BYTES + half * 16 * (D * 4) + (32 * NW * wn + l31) * 4);

    // ---- request side (identical in all waves) -------------------------------------------------------------------------------------
    const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)P.A, 0, (int)((size_t)P.Arows * P.lda * 4 < 0x7fffffffu ? (size_t)P.Arows * P.lda * 4 : 0x7fffffffu), 0x00020000);
    const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)P.B, 0, (int)((size_t)P.Brows * P.ldb * 4), 0x00020000);
    int nx_n = (int)blockIdx.x;
    int nx_mi = 0, nx_sl = 0, nx_kbeg = 0, nx_nk = 0, nx_j = 0, nx_stage = 0;
    bool nx_valid = item_of(nx_n, nx_mi, nx_sl, nx_kbeg, nx_nk);
    while (nx_valid && nx_nk == 0) { nx_n += (int)gridDim.x; nx_valid = item_of(nx_n, nx_mi, nx_sl, nx_kbeg, nx_nk); }
    if (!nx_valid) return;                             // uniform
    auto issue_chunk = [&]() {
        const int k0 = (nx_kbeg + nx_j) * GK;
        char* da = glds + nx_stage * GSTAGE + (2 * wave) * 1024;
        char* db = glds + nx_stage * GSTAGE + GA_BYTES + (NW * wave) * 1024;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = 2 * wave + j;                // A piece 0..15
            unsigned vo;
            if (!AT) {
                // rows past the end and contraction values past K are CLAMPED to the matrix (an out-of-range global -> LDS load
                // leaves the stage's old contents): a clamped row is never stored, a clamped contraction value meets a B row that
                // is zero IN MEMORY (B is padded to whole chunks: the caller's promise, checked on the host)
                int row = nx_mi * GM + 8 * p + (lane >> 3);                // output row; 128 contiguous bytes along k
                row = row < P.Arows ? row : P.Arows - 1;
                const int seg = (lane & 7) ^ (((8 * p + (lane >> 3)) >> 1) & 7);
                int kk = k0 + seg * 4;
                kk = kk < P.K ? kk : P.K - 4;                              // (K % 4 == 0)
                vo = (unsigned)(((size_t)row * P.lda + kk) * 4);
            } else {
                const int row = k0 + 2 * p + (lane >> 5);                  // contraction row; 512 contiguous bytes along the output rows
                const int col = nx_mi * GM + (lane & 31) * 4;
                const bool ok = row < P.Arows && col < P.Acols;
                vo = ok ? (unsigned)(((size_t)row * P.lda + col) * 4) : OOB;
            }
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)(da + j * 1024), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int p = NW * wave + j;               // B piece 0 .. 8 NW - 1
            const int row = k0 + p * RPP + lane / (16 * NW);               // (!AT: Brows covers whole chunks; AT: the tail is zeroed below)
            const unsigned vo = row < P.Brows ? (unsigned)(((size_t)row * P.ldb + (lane % (16 * NW)) * 4) * 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)(db + j * 1024), 16, vo, 0, 0, 0);
        }
        nx_stage = nx_stage + 1 == GNS ? 0 : nx_stage + 1;
        if (++nx_j == nx_nk) {
            nx_j = 0;
            do {
                nx_n += (int)gridDim.x;
                nx_valid = item_of(nx_n, nx_mi, nx_sl, nx_kbeg, nx_nk);
            } while (nx_valid && nx_nk == 0);
        }
    };

    // the two extra weight rows (forward), zero-padded to whole chunks
    if (!AT && extras) {
        for (int i = threadIdx.x; i < 2 * Kpad; i += 512) {
            const int j = i / Kpad, k = i % Kpad;
            w2l[i] = k < P.K ? P.w2[(size_t)j * P.K + k] : 0.0f;
        }
    }
    __syncthreads();

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < NW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float e0 = 0.0f, e1 = 0.0f;                          // !AT extras: this lane's part of <row, w2[j]>
    float cs = 0.0f;                                     // AT: this lane's part of its column's sum (bias gradient)

    // the bias values of this lane's columns: loaded ONCE, ahead of the first global -> LDS load (an ordinary load's use behind one
    // makes the compiler drain the whole queue)
    float bvv[4] = {0.f, 0.f, 0.f, 0.f}, b2v[2] = {0.f, 0.f};
    if (!AT) {
        if (P.bias && !P.accumulate)
#pragma unroll
            for (int t = 0; t < NW; ++t) bvv[t] = P.bias[32 * NW * wn + 32 * t + l31];
        if (extras) { b2v[0] = P.b2[0]; b2v[1] = P.b2[1]; }
    }

    int inflight = 0;
    int st_old = 0, st_new = 0;                          // stores issued behind the oldest / second oldest chunk in flight
#pragma unroll
    for (int i = 0; i < GNS - 1; ++i)
        if (nx_valid) { issue_chunk(); ++inflight; }
    int rd_stage = 0;

    int cur_n = (int)blockIdx.x;
    while (true) {
        int mi, sl, kbeg, nk;
        if (!item_of(cur_n, mi, sl, kbeg, nk)) break;
        if (nk == 0) { cur_n += (int)gridDim.x; continue; }
        for (int j = 0; j < nk; ++j) {
            // this wave's pieces of the chunk have landed when at most `allow` younger vector-memory operations are outstanding (they
            // complete in issue order, stores included): the next chunk's loads and the previous item's stores, which were issued
            // BEHIND this chunk's loads -- waiting for those stores here would expose a whole write round trip per item
            {
                const int allow = (inflight >= 2 ? NLOAD : 0) + st_old;
                if (allow >= 63) asm volatile("s_waitcnt vmcnt(63)" ::: "memory");
                else if (allow >= 32) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
                else if (allow >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
                else if (inflight >= 2) {
                    if (NLOAD == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    else if (NLOAD == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else if (NLOAD == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                    else if (NLOAD == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                st_old = st_new; st_new = 0;             // the chunk behind it becomes the oldest; the one issued below has no stores behind it yet
            }
            const int kc0 = (kbeg + j) * GK;
            if (AT && kc0 + GK > P.Arows) {
                // the contraction's last chunk: rows past the end were not loaded (out of range: the stage keeps what it held) --
                // every wave clears the invalid parts of ITS pieces before anybody reads the stage
                char* st = glds + rd_stage * GSTAGE;
                const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    if (kc0 + 2 * (2 * wave + jj) + (lane >> 5) >= P.Arows) *(float4*)(st + (2 * wave + jj) * 1024 + lane * 16) = z;
#pragma unroll
                for (int jj = 0; jj < NW; ++jj)
                    if (kc0 + (NW * wave + jj) * RPP + lane / (16 * NW) >= P.Arows) *(float4*)(st + GA_BYTES + (NW * wave + jj) * 1024 + lane * 16) = z;
            }
            __builtin_amdgcn_s_barrier();
            --inflight;
            if (nx_valid) { issue_chunk(); ++inflight; }
            const unsigned sb = lds0 + (unsigned)(rd_stage * GSTAGE);
            rd_stage = rd_stage + 1 == GNS ? 0 : rd_stage + 1;
            const int kc = (kbeg + j) * GK;              // first contraction value of this chunk
            v4f a4[2];
            float a1[2][4];
            float bq[2][4][4];
            v4f xw[2][2];                                // !AT extras: the weights of the group's four k (two vectors)
            auto read_group = [&](auto mmc, v4f& av4, float (&av)[4], float (&bv)[4][4], v4f (&xv)[2]) {
                constexpr int mm = decltype(mmc)::value;
                if (!AT) {
                    const unsigned addr = sb + rdA[mm];
                    asm volatile("ds_read_b128 %0, %1" : "=v"(av4) : "v"(addr));
                    if (extras) {
                        const unsigned wa = lds0 + (unsigned)(GNS * GSTAGE + (kc + 16 * half + 4 * mm) * 4);
                        asm volatile("ds_read_b128 %0, %1" : "=v"(xv[0]) : "v"(wa));
                        asm volatile("ds_read_b128 %0, %1" : "=v"(xv[1]) : "v"(wa + (unsigned)(Kpad * 4)));
                    }
                } else {
                    static_for<0, 4>([&](auto cc) {
                        constexpr int comp = decltype(cc)::value;
                        float& dst = av[comp];
                        const unsigned addr = sb + rdAT;
                        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"((4 * mm + comp) * (GM * 4)));
                    });
                }
                static_for<0, 4>([&](auto cc) {
                    constexpr int comp = decltype(cc)::value;
                    static_for<0, NW>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        float& dst = bv[comp][t];
                        const unsigned addr = sb + rdB;
                        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"((4 * mm + comp) * (D * 4) + t * 128));
                    });
                });
            };
            auto wait_group = [&](v4f& av4, float (&av)[4], float (&bv)[4][4], v4f (&xv)[2]) {
                // (every register an asm read returned is an operand of THE wait: a separate statement behind it lets the compiler
                // copy the register before the data has landed)
                if (AT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]));
                else if (extras) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av4), "+v"(xv[0]), "+v"(xv[1]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av4));
#pragma unroll
                for (int comp = 0; comp < 4; ++comp)
#pragma unroll
                    for (int t = 0; t < NW; ++t) asm volatile("" : "+v"(bv[comp][t]));
            };
            auto mul_group = [&](const v4f& av4, const float (&av)[4], const float (&bv)[4][4], const v4f (&xv)[2]) {
#pragma unroll
                for (int comp = 0; comp < 4; ++comp)
#pragma unroll
                    for (int t = 0; t < NW; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AT ? av[comp] : av4[comp], bv[comp][t], acc[t], 0, 0, 0);
                if (!AT) {
                    if (extras) {
#pragma unroll
                        for (int comp = 0; comp < 4; ++comp) { e0 = fmaf(av4[comp], xv[0][comp], e0); e1 = fmaf(av4[comp], xv[1][comp], e1); }
                        asm volatile("" : "+v"(e0), "+v"(e1));       // here, not sunk to the end of the chunk on copies of the operands
                    }
                } else {
                    cs += (av[0] + av[1]) + (av[2] + av[3]);
                    asm volatile("" : "+v"(cs));
                }
            };
            if (!AT && !EX && P.ktail && kbeg + j == nkt - 1) {
                // the short last chunk: its (at most) 8 values as one group, k = 4 half + component
                {
                    const unsigned addr = sb + rdAtail;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(a4[0]) : "v"(addr));
                }
                static_for<0, 4>([&](auto cc) {
                    constexpr int comp = decltype(cc)::value;
                    static_for<0, NW>([&](auto tc) {
                        constexpr int t = decltype(tc)::value;
                        float& dst = bq[0][comp][t];
                        const unsigned addr = sb + rdBtail;
                        asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(comp * (D * 4) + t * 128));
                    });
                });
                wait_group(a4[0], a1[0], bq[0], xw[0]);
                __builtin_amdgcn_sched_barrier(0);
                mul_group(a4[0], a1[0], bq[0], xw[0]);
                __builtin_amdgcn_sched_barrier(0);
                continue;
            }
            read_group(std::integral_constant<int, 0>{}, a4[0], a1[0], bq[0], xw[0]);
            wait_group(a4[0], a1[0], bq[0], xw[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_group(std::integral_constant<int, 1>{}, a4[1], a1[1], bq[1], xw[1]);
            __builtin_amdgcn_sched_barrier(0);
            mul_group(a4[0], a1[0], bq[0], xw[0]);
            __builtin_amdgcn_sched_barrier(0);
            wait_group(a4[1], a1[1], bq[1], xw[1]);
            __builtin_amdgcn_sched_barrier(0);
            read_group(std::integral_constant<int, 2>{}, a4[0], a1[0], bq[0], xw[0]);
            __builtin_amdgcn_sched_barrier(0);
            mul_group(a4[1], a1[1], bq[1], xw[1]);
            __builtin_amdgcn_sched_barrier(0);
            wait_group(a4[0], a1[0], bq[0], xw[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_group(std::integral_constant<int, 3>{}, a4[1], a1[1], bq[1], xw[1]);
            __builtin_amdgcn_sched_barrier(0);
            mul_group(a4[0], a1[0], bq[0], xw[0]);
            __builtin_amdgcn_sched_barrier(0);
            wait_group(a4[1], a1[1], bq[1], xw[1]);
            __builtin_amdgcn_sched_barrier(0);
            mul_group(a4[1], a1[1], bq[1], xw[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- the item's 128 x D block (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ------------------
        if (!AT) {
            const bool full = mi * GM + GM <= P.Arows;           // a whole tile: unconditional stores (counted for the waits above)
#pragma unroll
            for (int t = 0; t < NW; ++t) {
                const int col = 32 * NW * wn + 32 * t + l31;
                if (P.accumulate) {                              // (the loads first, all of them: one wait, not one per element)
                    float old[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mi * GM + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
                        old[r] = m < P.Arows ? P.out[(size_t)m * P.ldout + col] : 0.0f;
                    }
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += old[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] += bvv[t];
                }
            }
            if (full) {
#pragma unroll
                for (int t = 0; t < NW; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mi * GM + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
                        P.out[(size_t)m * P.ldout + 32 * NW * wn + 32 * t + l31] = acc[t][r];
                        acc[t][r] = 0.0f;
                    }
                st_old += 16 * NW; st_new += 16 * NW;
            } else {
#pragma unroll
                for (int t = 0; t < NW; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = mi * GM + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (m < P.Arows) P.out[(size_t)m * P.ldout + 32 * NW * wn + 32 * t + l31] = acc[t][r];
                        acc[t][r] = 0.0f;
                    }
            }
            if (extras) {
                const float t0 = e0 + __shfl_xor(e0, 32), t1 = e1 + __shfl_xor(e1, 32);
                const int m = mi * GM + 32 * wm + l31;
                if (wn == 0 && half == 0 && m < P.Arows) {
                    float* o = P.out + (size_t)m * P.ldout + D;
                    o[0] = t0 + b2v[0];
                    o[1] = t1 + b2v[1];
                    for (int z = 0; z < P.zero_cols; ++z) o[2 + z] = 0.0f;
                }
                e0 = e1 = 0.0f;
            }
        } else {
            const int Mp = nm * GM;                              // rows of a partial slab
            float* ob = P.out + (size_t)sl * Mp * D;
#pragma unroll
            for (int t = 0; t < NW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mi * GM + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
                    ob[(size_t)m * D + 32 * NW * wn + 32 * t + l31] = acc[t][r];
                    acc[t][r] = 0.0f;
                }
            st_old += 16 * NW; st_new += 16 * NW;
            if (P.part_bias) {
                const float tot = cs + __shfl_xor(cs, 32);
                if (wn == 0 && half == 0) P.part_bias[(size_t)sl * P.bias_pitch + mi * GM + 32 * wm + l31] = tot;
            }
            cs = 0.0f;
        }
        cur_n += (int)gridDim.x;
    }
}

// The two extra rows of the weight gradient and the two extra entries of the bias gradient (the packed outputs' columns c / diag):
// out[j][n] = sum_m dy[m][col0 + j] x[m][n] over this block's rows -- a dot product per column on the side of the matrix-core part
// (memory-bound: x is read once more).  Partial sums per block, reduced with the rest in a fixed order.
constexpr int XROWS_MIN = 128;              // rows per block: at least this many, and about 512 blocks (two per compute unit; the reduction
                                            // walks one partial result per block)
__host__ inline int xrows_of(long long M) { long long x = (M + 511) / 512; x = (x + 31) / 32 * 32; return (int)(x < XROWS_MIN ? XROWS_MIN : x); }
// (round 6: a wave per row and 16 bytes of x per lane -- four rows of a block in flight per wave, the two dy values of a row through the
// scalar path -- instead of a thread per column with 4-byte loads: 75 -> ~50 us at 248 760 rows x 256; the four waves' sums are added
// in wave order, the result does not depend on timing)
__global__ __launch_bounds__(256) void proj_extra_tn_kernel(const float* __restrict__ dy, long long lddy, long long M, int col0,
                                                            const float* __restrict__ x, long long ldx, int N,
                                                            float* __restrict__ part_extra, float* __restrict__ part_xbias, int xrows)
{
    __shared__ float sh[4][2][256];
    __shared__ float shb[4][2];
    const long long m0 = (long long)blockIdx.x * xrows;
    const long long m1 = m0 + xrows < M ? m0 + xrows : M;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int n4 = lane * 4;
    const bool act = n4 < N;                                    // (N is 64 / 128 / 256: whole quads)
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    float b0 = 0.f, b1 = 0.f;
    long long m = m0 + wave;
    for (; m + 12 < m1; m += 16) {                              // rows m, m + 4, m + 8, m + 12 of this wave
        float4 xv[4];
        float a0[4], a1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long long r = m + 4 * i;
            xv[i] = act ? *(const float4*)(x + r * ldx + n4) : make_float4(0.f, 0.f, 0.f, 0.f);
            a0[i] = dy[r * lddy + col0];
            a1[i] = dy[r * lddy + col0 + 1];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            s0.x = fmaf(a0[i], xv[i].x, s0.x); s0.y = fmaf(a0[i], xv[i].y, s0.y); s0.z = fmaf(a0[i], xv[i].z, s0.z); s0.w = fmaf(a0[i], xv[i].w, s0.w);
            s1.x = fmaf(a1[i], xv[i].x, s1.x); s1.y = fmaf(a1[i], xv[i].y, s1.y); s1.z = fmaf(a1[i], xv[i].z, s1.z); s1.w = fmaf(a1[i], xv[i].w, s1.w);
            b0 += a0[i]; b1 += a1[i];
        }
    }
    for (; m < m1; m += 4) {
        const float4 xv = act ? *(const float4*)(x + m * ldx + n4) : make_float4(0.f, 0.f, 0.f, 0.f);
        const float a0 = dy[m * lddy + col0], a1 = dy[m * lddy + col0 + 1];
        s0.x = fmaf(a0, xv.x, s0.x); s0.y = fmaf(a0, xv.y, s0.y); s0.z = fmaf(a0, xv.z, s0.z); s0.w = fmaf(a0, xv.w, s0.w);
        s1.x = fmaf(a1, xv.x, s1.x); s1.y = fmaf(a1, xv.y, s1.y); s1.z = fmaf(a1, xv.z, s1.z); s1.w = fmaf(a1, xv.w, s1.w);
        b0 += a0; b1 += a1;
    }
    if (act) {
        *(float4*)&sh[wave][0][n4] = s0;
        *(float4*)&sh[wave][1][n4] = s1;
    }
    if (lane == 0) { shb[wave][0] = b0; shb[wave][1] = b1; }
    __syncthreads();
    const int n = threadIdx.x;
    if (n < N) {
        part_extra[((size_t)blockIdx.x * 2 + 0) * N + n] = ((sh[0][0][n] + sh[1][0][n]) + sh[2][0][n]) + sh[3][0][n];
        part_extra[((size_t)blockIdx.x * 2 + 1) * N + n] = ((sh[0][1][n] + sh[1][1][n]) + sh[2][1][n]) + sh[3][1][n];
    }
    if (n < 2) part_xbias[(size_t)blockIdx.x * 2 + n] = ((shb[0][n] + shb[1][n]) + shb[2][n]) + shb[3][n];
}

// dW[r][n] = sum_s part[s][r][n] (r < rows: the matrix part), the two extra rows from part_extra, db from part_bias.  A block per
// output row and 64 columns (and a block row for the bias gradient); 16 groups of lanes take every 16th partial result each (independent loads, four columns
// per lane) and are combined in a fixed order: the result does not depend on timing.
__global__ __launch_bounds__(256) void proj_reduce_kernel(const float* __restrict__ part, int S, int Mp, int D, int rows,
                                                          const float* __restrict__ part_extra, const float* __restrict__ part_xbias,
                                                          int SX, int extra_row0,
                                                          const float* __restrict__ part_bias, int bias_pitch, int nbias,
                                                          float* __restrict__ dW, long long lddw, int total_rows,
                                                          float* __restrict__ db)
{
    // a block = one output row x 64 columns: 16 quads of columns x 16 groups that take every 16th partial result each
    constexpr int RG = 16, RQ = 16;
    __shared__ float sh[RG][4 * RQ];
    const int g = threadIdx.x / RQ, q = threadIdx.x % RQ;
    const int n0 = blockIdx.x * (4 * RQ) + 4 * q;
    const int r = blockIdx.y;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const float* src = nullptr;          // partial result i, column n: src[i * pitch + n]
    size_t pitch = 0;
    int count = 0, ncols = 0;
    if (r < total_rows) {
        ncols = D;
        if (r < rows) { src = part + (size_t)r * D; pitch = (size_t)Mp * D; count = S; }
        else if (part_extra && r >= extra_row0 && r < extra_row0 + 2) { src = part_extra + (size_t)(r - extra_row0) * D; pitch = (size_t)2 * D; count = SX; }
    } else if (db) {
        ncols = nbias;
    }
    if (r < total_rows) {
        if (src && n0 < ncols) {         // (D is a multiple of 64: whole quads)
            int i = g;
            for (; i + 3 * RG < count; i += 4 * RG) {
                const float4 a = *(const float4*)(src + (size_t)i * pitch + n0), b = *(const float4*)(src + (size_t)(i + RG) * pitch + n0);
                const float4 c = *(const float4*)(src + (size_t)(i + 2 * RG) * pitch + n0), d = *(const float4*)(src + (size_t)(i + 3 * RG) * pitch + n0);
                s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
                s[0] += b.x; s[1] += b.y; s[2] += b.z; s[3] += b.w;
                s[0] += c.x; s[1] += c.y; s[2] += c.z; s[3] += c.w;
                s[0] += d.x; s[1] += d.y; s[2] += d.z; s[3] += d.w;
            }
            for (; i < count; i += RG) {
                const float4 a = *(const float4*)(src + (size_t)i * pitch + n0);
                s[0] += a.x; s[1] += a.y; s[2] += a.z; s[3] += a.w;
            }
        }
    } else if (db) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = n0 + c;
            if (n >= nbias) continue;
            if (n < rows) {
                for (int i = g; i < S; i += RG) s[c] += part_bias[(size_t)i * bias_pitch + n];
            } else if (part_xbias && n >= extra_row0 && n < extra_row0 + 2) {
                for (int i = g; i < SX; i += RG) s[c] += part_xbias[(size_t)i * 2 + (n - extra_row0)];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) sh[g][4 * q + c] = s[c];
    __syncthreads();
    if (g == 0) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int n = n0 + c;
            float v = sh[0][4 * q + c];
#pragma unroll
            for (int k = 1; k < RG; ++k) v += sh[k][4 * q + c];       // fixed order
            if (r < total_rows) { if (n < D) dW[(size_t)r * lddw + n] = v; }
            else if (db && n < nbias) db[n] = v;
        }
    }
}

}  // namespace pj

int proj_ncu()
{
    int ncu = 256, dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ncu = v;
    return ncu;
}

template <bool AT, int NW, bool EX>
static void proj_launch(const pj::Args& P, long long nitems, int Kpad, hipStream_t stream)
{
    const size_t lds = (size_t)pj::GNS * (pj::GA_BYTES + pj::GK * 64 * NW * 4) + (AT ? 0 : (size_t)2 * Kpad * 4);
    static PerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)pj::proj_gemm_kernel<AT, NW, EX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512);
    const int ncu = proj_ncu();
    const int grid = nitems < ncu ? (int)nitems : ncu;
    hipLaunchKernelGGL((pj::proj_gemm_kernel<AT, NW, EX>), dim3(grid), dim3(512), lds, stream, P);
}

bool proj_gemm_supported(long long M, int K, int N, const void* A, long long lda, const void* B, long long ldb, const void* out,
                         long long ldout)
{
    if (!(N == 64 || N == 128 || N == 256) || K < 4 || K % 4 != 0 || K > 4096 || M < 1) return false;
    if (((uintptr_t)A & 15) || ((uintptr_t)B & 15) || ((uintptr_t)out & 3) || lda % 4 || ldb % 4) return false;
    if (M * lda * 4 >= (1ll << 31) || (long long)K * ldb * 4 >= (1ll << 31)) return false;          // 32-bit buffer offsets
    return true;
}

// Y[M][N (+2+zero_cols)] (+)= A[M][K] B[K][N] (+ bias); extra columns from w2 [2][K], b2 [2]
int launch_proj_nn(const float* A, long long lda, long long M, int K, const float* B, long long ldb, int N, float* out, long long ldout,
                   const float* bias, const float* w2, const float* b2, int zero_cols, int accumulate, hipStream_t stream)
{
    if (!proj_gemm_supported(M, K, N, A, lda, B, ldb, out, ldout)) return 1;
    pj::Args P{};
    P.A = A; P.lda = lda; P.Arows = (int)M; P.Acols = K;
    P.B = B; P.ldb = ldb; P.Brows = (K + pj::GK - 1) / pj::GK * pj::GK;     // B holds whole chunks of rows, zero beyond K (the caller's promise)
    P.out = out; P.ldout = ldout; P.Mout = (int)M; P.K = K; P.accumulate = accumulate; P.bias = bias;
    P.w2 = w2; P.b2 = b2; P.zero_cols = zero_cols; P.nslices = 1; P.kslice = 0; P.extra_col0 = -1;
    P.ktail = (!w2 && K > pj::GK && K % pj::GK != 0 && K % pj::GK <= 8) ? 1 : 0;
    const int Kpad = (K + pj::GK - 1) / pj::GK * pj::GK;
    const long long nitems = (M + pj::GM - 1) / pj::GM;
    switch (N) {
    case 64: w2 ? proj_launch<false, 1, true>(P, nitems, Kpad, stream) : proj_launch<false, 1, false>(P, nitems, Kpad, stream); break;
    case 128: w2 ? proj_launch<false, 2, true>(P, nitems, Kpad, stream) : proj_launch<false, 2, false>(P, nitems, Kpad, stream); break;
    default: w2 ? proj_launch<false, 4, true>(P, nitems, Kpad, stream) : proj_launch<false, 4, false>(P, nitems, Kpad, stream); break;
    }
    return 0;
}

// workspace of launch_proj_tn: partial slabs + partial bias sums + partial extra rows
static void proj_tn_geometry(long long M, int R, int* S, int* kslice, int* Mp)
{
    const int nkt = (int)((M + pj::GK - 1) / pj::GK);
    const int nm = (R + pj::GM - 1) / pj::GM;
    int s = proj_ncu() / (nm > 0 ? nm : 1);
    if (s < 1) s = 1;
    if (s > nkt) s = nkt;
    *kslice = (nkt + s - 1) / s;
    *S = (nkt + *kslice - 1) / *kslice;
    *Mp = nm * pj::GM;
}
size_t proj_tn_workspace_bytes(long long M, int R, int N)
{
    int S, ks, Mp;
    proj_tn_geometry(M, R, &S, &ks, &Mp);
    const size_t SX = (size_t)((M + pj::xrows_of(M) - 1) / pj::xrows_of(M));
    return ((size_t)S * Mp * N + (size_t)S * (Mp + 8) + SX * 2 * N + SX * 2) * sizeof(float) + 256;
}

// dW[R (+2)][N] = A[:, :R]^T X, db = column sums of A (first R columns, + the two extra columns extra_col0, +1 when >= 0); rows
// of dW beyond R + 2 up to total_rows are set to zero
int launch_proj_tn3_core(const float* A, long long lda, long long M, int R, const float* X, long long ldx, int N, float* part, int Mp,
                         float* pbias, int bias_pitch, int S, int kslice, hipStream_t stream);      // proj_gemm3.hip

int launch_proj_tn(const float* A, long long lda, long long M, int R, int extra_col0, int total_rows, const float* X, long long ldx, int N,
                   float* dW, long long lddw, float* db, void* ws, size_t ws_bytes, hipStream_t stream)
{
    const int prec = (total_rows & SEMICRF_PROJ_TN_BF16X3) ? 1 : 0;        // opt-in: the matrix part on the three-limb bf16 kernel (N == 256)
    total_rows &= ~SEMICRF_PROJ_TN_BF16X3;
    if (!(N == 64 || N == 128 || N == 256) || R < 1 || M < 1) return 1;
    if (((uintptr_t)A & 15) || ((uintptr_t)X & 15) || lda % 4 || ldx % 4 || M * lda * 4 >= (1ll << 31) || M * ldx * 4 >= (1ll << 31)) return 1;
    if (!ws || ws_bytes < proj_tn_workspace_bytes(M, R, N) || ((uintptr_t)ws & 15)) return 2;
    int S, ks, Mp;
    proj_tn_geometry(M, R, &S, &ks, &Mp);
    float* part = (float*)ws;
    float* pbias = part + (size_t)S * Mp * N;
    float* pextra = pbias + (size_t)S * (Mp + 8);
    const int xrows = pj::xrows_of(M);
    const int SX = (int)((M + xrows - 1) / xrows);
    float* pxbias = pextra + (size_t)SX * 2 * N;
    pj::Args P{};
    P.A = A; P.lda = lda; P.Arows = (int)M; P.Acols = R;
    P.B = X; P.ldb = ldx; P.Brows = (int)M;
    P.out = part; P.ldout = N; P.Mout = R; P.K = (int)M; P.accumulate = 0; P.bias = nullptr; P.w2 = nullptr; P.b2 = nullptr;
    P.zero_cols = 0; P.nslices = S; P.kslice = ks; P.extra_col0 = -1; P.part_bias = pbias; P.bias_pitch = Mp + 8; P.part_extra = nullptr;
    const long long nitems = (long long)(Mp / pj::GM) * S;
    if (prec && launch_proj_tn3_core(A, lda, M, R, X, ldx, N, part, Mp, pbias, Mp + 8, S, ks, stream) == 0) {
    } else
    switch (N) {
    case 64: proj_launch<true, 1, false>(P, nitems, 0, stream); break;
    case 128: proj_launch<true, 2, false>(P, nitems, 0, stream); break;
    default: proj_launch<true, 4, false>(P, nitems, 0, stream); break;
    }
    if (extra_col0 >= 0)
        hipLaunchKernelGGL(pj::proj_extra_tn_kernel, dim3((unsigned)SX), dim3(256), 0, stream, A, lda, M, extra_col0, X, ldx, N, pextra, pxbias, xrows);
    const int nb = total_rows;
    const dim3 grid((unsigned)(((N > nb ? N : nb) + 63) / 64), (unsigned)(total_rows + 1));
    hipLaunchKernelGGL(pj::proj_reduce_kernel, grid, dim3(256), 0, stream, part, S, Mp, N, R, extra_col0 >= 0 ? pextra : nullptr,
                       extra_col0 >= 0 ? pxbias : nullptr, SX, extra_col0, pbias, Mp + 8, nb, dW, lddw, total_rows, db);
    return 0;
}

}  // namespace semicrf
