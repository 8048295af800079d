// scorer_mfma.hip -- scaled-inner-product interval scores on the CDNA4 matrix cores
// (LayersTransformer.py:406-441 after the Linear map).
//
//   S[e,b,c] = qscale * <q[c,e,:], k[c,b,:]> * len(|e-b|)  (+ diag[c,e] when e == b),   layout [T][T][C]
//
// The contraction is exact fp32: v_mfma_f32_32x32x2_f32 (f32 in / f32 accumulate, a k-ordered fmaf chain; no
// xf32/TF32 exists on gfx950), so scores match the reference's fp32 einsum to round-off and the CRF decode
// that consumes them stays comparable.  One workgroup = one 32x32 (end, begin) tile for 16 chains:
// each of the 4 waves runs the 32x32xD MFMA chain for 4 chains in turn (lane = one row of q / one row of k,
// the two half-waves take the two halves of the d axis -- any d permutation applied to both operands is
// legal), results are scaled, transposed through a padded LDS tile and written chain-contiguous (64-byte
// segments), lower triangle only unless the caller asks for the full square.
#include <atomic>
#include "common.h"
#include "scorer_tiles.h"
#include "bf16x3.h"

#include <cstdlib>
#include <map>
#include <mutex>
#include <tuple>
#include <type_traits>
#include <vector>

namespace semicrf {

typedef float f32x16 __attribute__((ext_vector_type(16)));

#if defined(SEMICRF_DEBUG_BUILD) && SEMICRF_DEBUG_BUILD
#define SEMICRF_HAVE_TILE_REF 1
#else
#define SEMICRF_HAVE_TILE_REF 0
#endif

constexpr int ST = 32;        // tile edge (positions)
constexpr int SC = 16;        // chains per workgroup
constexpr int SPAD = SC + 1;  // LDS row padding: conflict-free column writes

__device__ __forceinline__ float len_scale_mfma(int len, int mode)
{
    if (mode == SEMICRF_LEN_LINEAR) return (float)len;
    if (mode == SEMICRF_LEN_SQRT) return sqrtf((float)len);
    return 1.0f;
}

// Work list (XCD-aware): workgroups are dispatched round-robin over the 8 XCDs, each with its own 4 MB L2, and the
// operands are heavy (a 32-row tile of q or k for 16 chains is 512 KB at D=256) while HBM delivers only ~10 B/clk/CU --
// a third of what the matrix pipe consumes with 32x32 tiles.  So the host orders the tiles such that the workgroups
// resident on ONE XCD at a time work on a band of SBAND tile rows of one chain group, column by column: the band's q
// tiles stay in that L2 and each k tile is fetched once for the SBAND workgroups that use it back to back.
// work[blockIdx.x] = {et | bt << 16, chain group} (et < 0: padding).
// NCH = chunks of 32 contraction values per half-wave (D / 64) when known at compile time (<= 4), else 0: with a
// compile-time trip count the two-stage load/multiply pipeline unrolls without branches and every wait is exact
template <bool ALIGNED, int NCH>
__global__ __launch_bounds__(256) void interval_score_mfma_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ diag, int C, int T, int D,
    long long ldq, long long ldk, long long ldd, float qscale, int mode, int full, float* __restrict__ S,
    const int2* __restrict__ work)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [ST*ST][SPAD]
    const int2 wk = work[blockIdx.x];
    if (wk.x < 0) return;
    const int et = wk.x & 0xffff, bt = wk.x >> 16;
    const int e0 = et * ST, b0 = bt * ST;
    const int cg = wk.y * SC;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int row = lane & 31, half = lane >> 5;
    const int Dh = D >> 1;                                  // d range of this half-wave: [half*Dh, half*Dh + Dh)
    const int er = e0 + row < T ? e0 + row : T - 1;         // clamped rows (masked at the write)
    const int br = b0 + row < T ? b0 + row : T - 1;

    for (int round = 0; round < SC / 4; ++round) {
        const int ci = wave + 4 * round;
        const int c = cg + ci < C ? cg + ci : C - 1;
        const float* qp = q + ((size_t)c * T + er) * ldq + (size_t)half * Dh;
        const float* kp = k + ((size_t)c * T + br) * ldk + (size_t)half * Dh;
        f32x16 acc;
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
        // chunks of 32 d-values per half-wave: 8 float4 of q and of k per lane; the next chunk is requested
        // before the 32 MFMAs of the current one so that the loads hide under the matrix pipe
        auto load_chunk = [&](float4 (&qa)[8], float4 (&ka)[8], int d0) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const float* qq = qp + d0 + 4 * m;
                const float* kk = kp + d0 + 4 * m;
                if (ALIGNED) {
                    qa[m] = *(const float4*)qq;
                    ka[m] = *(const float4*)kk;
                } else {                      // rows only 4-byte aligned (packed Linear output, ld = 2D+1)
                    qa[m] = make_float4(qq[0], qq[1], qq[2], qq[3]);
                    ka[m] = make_float4(kk[0], kk[1], kk[2], kk[3]);
                }
            }
        };
        auto mma_chunk = [&](const float4 (&qa)[8], const float4 (&ka)[8]) {
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].x, ka[m].x, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].y, ka[m].y, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].z, ka[m].z, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].w, ka[m].w, acc, 0, 0, 0);
            }
        };
        float4 qa0[8], ka0[8], qa1[8], ka1[8];
        load_chunk(qa0, ka0, 0);
        if (NCH > 0) {
#pragma unroll
            for (int ch = 0; ch < NCH; ++ch) {
                if (ch & 1) {
                    if (ch + 1 < NCH) load_chunk(qa0, ka0, (ch + 1) * 32);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_chunk(qa1, ka1);
                } else {
                    if (ch + 1 < NCH) load_chunk(qa1, ka1, (ch + 1) * 32);
                    __builtin_amdgcn_sched_barrier(0);
                    mma_chunk(qa0, ka0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            for (int d0 = 0; d0 < Dh; d0 += 64) {
                if (d0 + 32 < Dh) load_chunk(qa1, ka1, d0 + 32);
                mma_chunk(qa0, ka0);
                if (d0 + 32 >= Dh) break;
                if (d0 + 64 < Dh) load_chunk(qa0, ka0, d0 + 64);
                mma_chunk(qa1, ka1);
            }
        }
        // C/D layout of 32x32: col j = lane & 31, row i = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
        const int bj = lane & 31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ei = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int e = e0 + ei, b = b0 + bj;
            float v = acc[r] * qscale;
            const int len = e > b ? e - b : b - e;
            v *= len_scale_mfma(len, mode);
            if (e == b && e < T) v += diag[((size_t)c * T + e) * ldd];
            tile[(ei * ST + bj) * SPAD + ci] = v;
        }
    }
    __syncthreads();
    // write out: 4 threads per cell (4 chains each), chain axis contiguous
    const int nq = SC / 4;
    for (int idx = threadIdx.x; idx < ST * ST * nq; idx += 256) {
        const int cell = idx / nq, qd = idx % nq;
        const int ei = cell / ST, bj = cell % ST;
        const int e = e0 + ei, b = b0 + bj;
        if (e >= T || b >= T || (!full && b > e)) continue;
        float* dst = S + ((size_t)e * T + b) * C + cg + qd * 4;
        const float* src = tile + cell * SPAD + qd * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (cg + qd * 4 + i < C) dst[i] = src[i];
    }
}

bool interval_score_mfma_supported(int C, int T, int D) { return D % 64 == 0 && T >= 1 && C >= 1; }

constexpr int NXCD = 8;

// builds (once per device and shape) the device-resident work list described above.  The cache holds the SCORE_WL_MAX most
// recently used lists (least recently used one freed first), the upload is enqueued on the caller's stream (from a pageable
// staging vector kept alive in the cache entry), and a failed upload frees the allocation.  The first call for a shape
// allocates (hipMalloc: not stream-capturable); later calls only look the list up.
constexpr size_t SCORE_WL_MAX = 16;
struct ScoreWorkList { int2* dev = nullptr; int n = 0; unsigned long long stamp = 0; std::vector<int2> host; };

static const int2* score_work_list(int nt, int ngroups, int full, int band, int* grid_out, hipStream_t stream)
{
    static std::mutex mu;
    static std::map<std::tuple<int, int, int, int, int>, ScoreWorkList> cache;
    static unsigned long long clock = 0;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_tuple(dev, nt, ngroups, full, band);
    auto it = cache.find(key);
    if (it != cache.end()) { it->second.stamp = ++clock; *grid_out = it->second.n; return it->second.dev; }
    // units: (chain group, pair of bands p and nb-1-p) -- equal work per unit in the triangle
    const int nb = (nt + band - 1) / band;
    std::vector<std::vector<int2>> per(NXCD);
    int u = 0;
    auto add_band = [&](std::vector<int2>& out, int g, int j) {
        const int r0 = j * band, r1 = (r0 + band < nt) ? r0 + band : nt;
        const int ncol = full ? nt : r1;                 // columns 0 .. r1-1 (bt <= et)
        for (int bt = 0; bt < ncol; ++bt)
            for (int et = r0; et < r1; ++et)
                if (full || bt <= et) out.push_back(make_int2(et | (bt << 16), g));
    };
    for (int g = 0; g < ngroups; ++g)
        for (int p = 0; p < (nb + 1) / 2; ++p, ++u) {
            std::vector<int2>& out = per[u % NXCD];
            add_band(out, g, p);
            if (nb - 1 - p != p) add_band(out, g, nb - 1 - p);
        }
    size_t longest = 0;
    for (auto& v : per) longest = v.size() > longest ? v.size() : longest;
    ScoreWorkList e;
    e.host.assign(longest * NXCD, make_int2(-1, 0));
    for (int x = 0; x < NXCD; ++x)
        for (size_t i = 0; i < per[x].size(); ++i) e.host[i * NXCD + x] = per[x][i];
    if (cache.size() >= SCORE_WL_MAX) {                  // evict the least recently used list of this process
        auto lru = cache.begin();
        for (auto c = cache.begin(); c != cache.end(); ++c)
            if (c->second.stamp < lru->second.stamp) lru = c;
        // kernels that still read the list were enqueued before this call: free after they have drained
        (void)hipStreamSynchronize(stream);
        (void)hipFree(lru->second.dev);
        cache.erase(lru);
    }
    if (hipMalloc((void**)&e.dev, e.host.size() * sizeof(int2)) != hipSuccess) return nullptr;
    e.n = (int)e.host.size();
    e.stamp = ++clock;
    auto ins = cache.emplace(key, std::move(e)).first;          // the staging vector lives on in the cache entry
    if (hipMemcpyAsync(ins->second.dev, ins->second.host.data(), ins->second.host.size() * sizeof(int2), hipMemcpyHostToDevice,
                       stream) != hipSuccess) {
        (void)hipFree(ins->second.dev);
        cache.erase(ins);
        return nullptr;
    }
    *grid_out = ins->second.n;
    return ins->second.dev;
}

// ---------------------------------------------------------------------------------------------
// LDS-staged operands (16-byte aligned rows, D % 64 == 0)
// ---------------------------------------------------------------------------------------------
// With one matrix row per lane a register load touches 64 different 128-byte lines and the CU's L1 looks them up one
// per clock: the operand loads, not the matrix pipe, set the pace of the kernel above (28 % of the fp32 MFMA rate).
// The kernels below copy operands with `buffer_load ... lds`: one instruction moves 8 rows x 128 contiguous bytes
// (8 full lines), a chunk is 32 contraction values of every row.  The 16-byte segments of a row are XOR-swizzled by
// the LOADING lanes ((row >> 1) & 7) so that the row-per-lane ds_read_b128 of the MFMA operands is bank-conflict free
// without padding.  LDS reads in the main loops are asm: the compiler would order every DS operation it sees after ALL
// outstanding LDS-DMA (s_waitcnt vmcnt(0)).
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int ZSTAGE = 8192;              // bytes per stage: q chunk (4 KB) + k chunk (4 KB)
constexpr int ZCH = 32;                   // contraction values per chunk

__device__ __forceinline__ unsigned z_lds_addr(const void* p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

// ---------------------------------------------------------------------------------------------
// Streaming variant: persistent, barrier-free, no output tile
// ---------------------------------------------------------------------------------------------
// A first LDS-staged version kept the output tile of the kernel above (one workgroup per 32x32 tile and 8 chains, 1.85 ms
// at T=1024, C=352, D=256): 19 % of the time no workgroup was resident (133 KB of LDS: the next one starts only when the
// previous one has drained its stores) and a quarter of a workgroup's cycles were pipeline fill, barrier and write-out.
// Here
//   * a wave owns FOUR ADJACENT chains of one 32x32 tile: it multiplies them one after the other, keeps the finished
//     blocks in registers and writes each cell's four chains as one 16-byte piece straight from registers -- the
//     eight waves of a workgroup cover 32 adjacent chains of the same tile, i.e. whole 128-byte lines, which L2 merges;
//     no output tile in LDS, no barrier, waves never wait for each other;
//   * workgroups are persistent (one per CU, 8 waves = 2 per SIMD) and walk their XCD's part of the work list; the
//     operand stream (two 8 KB LDS stages per wave + the register double buffer) runs on across chains and tiles;
//   * consecutive matrix instructions alternate between two accumulators (summed at the end of a chain).
constexpr int YW = 8;                      // waves per workgroup
constexpr int YNS = 2;                     // LDS stages per wave
constexpr int YG = 4 * YW;                 // chains per workgroup item

__global__ __launch_bounds__(64 * YW) void interval_score_stream_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ diag, int C, int T, int D,
    long long ldq, long long ldk, long long ldd, float qscale, int mode, int full, float* __restrict__ S,
    const int2* __restrict__ work, int nlist)
{
    extern __shared__ __attribute__((aligned(16))) char ylds[];    // [YW waves][YNS][ZSTAGE]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = lane & 31, half = lane >> 5;
    char* const stage0 = ylds + wave * (YNS * ZSTAGE);
    const unsigned stage0_addr = z_lds_addr(stage0);
    const int nchunk = D / ZCH;                                     // even (D % 64 == 0)
    const int xcd = blockIdx.x % NXCD, slot0 = blockIdx.x / NXCD, nslots = gridDim.x / NXCD;
    const int nper = nlist / NXCD;                                  // entries per XCD (the tail may be padding)

    // reading lanes: lane = (row, half); segment 4*half + m of the row, m = 0..3
    unsigned rd[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) rd[m] = (unsigned)(row * 128 + (((4 * half + m) ^ ((row >> 1) & 7)) * 16));

    // ---- request side: (item, chain of the quad, chunk) walked with counters ---------------------------------
    int nx_item = slot0;                       // index into this XCD's list
    int nx_j = 0, nx_ch = 0, nx_stage = 0;
    bool nx_valid = false;
    unsigned voq[4], vok[4];
    const float* nx_q = q;
    const float* nx_k = k;
    int nx_c4 = 0;
    auto set_chain = [&]() {
        const int c = nx_c4 + nx_j < C ? nx_c4 + nx_j : C - 1;
        nx_q = q + (size_t)c * T * ldq;
        nx_k = k + (size_t)c * T * ldk;
    };
    auto set_item = [&]() {
        nx_valid = false;
        if (nx_item < nper) {
            const int2 wk = work[(size_t)nx_item * NXCD + xcd];
            if (wk.x >= 0) {
                nx_valid = true;
                const int e0 = (wk.x & 0xffff) * ST, b0 = (wk.x >> 16) * ST;
                nx_c4 = wk.y * YG + wave * 4;
                // loading lanes: piece j (0..3) covers rows 8j .. 8j+7, lane = (row within the piece, 16-byte position)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int lr = 8 * j + (lane >> 3);
                    const int seg = (lane & 7) ^ ((lr >> 1) & 7);
                    const int er = e0 + lr < T ? e0 + lr : T - 1;          // clamped rows (masked at the write)
                    const int br = b0 + lr < T ? b0 + lr : T - 1;
                    voq[j] = (unsigned)(((size_t)er * ldq + seg * 4) * 4);
                    vok[j] = (unsigned)(((size_t)br * ldk + seg * 4) * 4);
                }
                nx_j = 0;
                nx_ch = 0;
                set_chain();
            }
        }
    };
    // one 1 KB piece of the next chunk: p < 4 -> q rows 8p.., else k rows 8(p-4)..
    auto issue_piece = [&](int p) {
        char* dst = stage0 + nx_stage * ZSTAGE + p * 1024;
        if (p < 4) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)nx_q, 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)dst, 16, voq[p & 3], nx_ch * (ZCH * 4), 0, 0);
        } else {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)nx_k, 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)dst, 16, vok[p & 3], nx_ch * (ZCH * 4), 0, 0);
        }
    };
    auto advance_next = [&]() {
        nx_stage ^= 1;
        if (++nx_ch == nchunk) {
            nx_ch = 0;
            if (++nx_j == 4) {
                nx_item += nslots;
                set_item();
            } else {
                set_chain();
            }
        }
    };
    auto read_chunk = [&](int stage, v4f (&qa)[4], v4f (&ka)[4]) {
        const unsigned sb = stage0_addr + (unsigned)(stage * ZSTAGE);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const unsigned a = sb + rd[m];
            asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:4096" : "=&v"(qa[m]), "=&v"(ka[m]) : "v"(a));
        }
    };
    auto wait_reads = [&](v4f (&qa)[4], v4f (&ka)[4]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]), "+v"(ka[0]), "+v"(ka[1]), "+v"(ka[2]), "+v"(ka[3]));
    };

    set_item();
    if (!nx_valid) return;
    // the compute side's view of the current item
    int cur_item = nx_item;
    int2 cur_wk = work[(size_t)cur_item * NXCD + xcd];

    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.0f; accB[r] = 0.0f; }
    float hold[3][16];
    v4f qa0[4], ka0[4], qa1[4], ka1[4];

    // prologue: chunk 0 requested and read into registers, chunk 1 requested
#pragma unroll
    for (int p = 0; p < 8; ++p) issue_piece(p);
    advance_next();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    read_chunk(0, qa0, ka0);
    wait_reads(qa0, ka0);
    if (nx_valid) {
#pragma unroll
        for (int p = 0; p < 8; ++p) issue_piece(p);
        advance_next();
    }
    bool have_next = true;                     // a chunk after the current one exists (requested into stage rd_stage)
    int rd_stage = 1;

    // step: the current chunk is in registers (qc, kc); the next one (in flight since the previous step) is awaited and
    // read into the other set; the one after it is requested between the matrix instructions
    auto step = [&](v4f (&qc)[4], v4f (&kc)[4], v4f (&qn)[4], v4f (&kn)[4], bool last_of_stream) {
        if (!last_of_stream) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            // the chunk after the next goes into the stage of the current one (which is in registers); all of it is
            // requested NOW so that every piece has the whole step to arrive
            if (nx_valid) {
#pragma unroll
                for (int p = 0; p < 8; ++p) issue_piece(p);
                advance_next();
            }
            read_chunk(rd_stage, qn, kn);
            rd_stage ^= 1;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[m].x, kc[m].x, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[m].y, kc[m].y, accB, 0, 0, 0);
            accA = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[m].z, kc[m].z, accA, 0, 0, 0);
            accB = __builtin_amdgcn_mfma_f32_32x32x2f32(qc[m].w, kc[m].w, accB, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (!last_of_stream) wait_reads(qn, kn);
    };

    while (true) {
        // is there an item after this one (for this workgroup)?  the request side knows: it is at most two chunks ahead
        const int e0 = (cur_wk.x & 0xffff) * ST, b0 = (cur_wk.x >> 16) * ST;
        const int c4 = cur_wk.y * YG + wave * 4;
        int nxt_item = cur_item + nslots;
        bool nxt_ok = false;
        int2 nxt_wk = make_int2(-1, 0);
        if (nxt_item < nper) {
            nxt_wk = work[(size_t)nxt_item * NXCD + xcd];
            nxt_ok = nxt_wk.x >= 0;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            for (int ch = 0; ch < nchunk; ch += 2) {
                step(qa0, ka0, qa1, ka1, false);
                step(qa1, ka1, qa0, ka0, !nxt_ok && j == 3 && ch + 2 == nchunk);
            }
            if (j < 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { hold[j][r] = accA[r] + accB[r]; accA[r] = 0.0f; accB[r] = 0.0f; }
            }
        }
        // ---- write the item's four chains: one 16-byte piece per cell (C/D layout: col = lane & 31,
        //      row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
        {
            const int bj = row, b = b0 + bj;
            const bool vec = (C & 3) == 0 && c4 + 3 < C;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ei = (r & 3) + 8 * (r >> 2) + 4 * half;
                const int e = e0 + ei;
                const int len = e > b ? e - b : b - e;
                const float sc = qscale * len_scale_mfma(len, mode);
                float v[4] = {hold[0][r] * sc, hold[1][r] * sc, hold[2][r] * sc, (accA[r] + accB[r]) * sc};
                accA[r] = 0.0f; accB[r] = 0.0f;
                if (e < T && b < T && (full || b <= e) && c4 < C) {
                    if (e == b) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (c4 + i < C) v[i] += diag[((size_t)(c4 + i) * T + e) * ldd];
                    }
                    float* dst = S + ((size_t)e * T + b) * C + c4;
                    if (vec) {
                        *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (c4 + i < C) dst[i] = v[i];
                    }
                }
            }
        }
        if (!nxt_ok) break;
        cur_item = nxt_item;
        cur_wk = nxt_wk;
    }
}

static int launch_score_stream(const float* q, const float* k, const float* diag, int C, int T, int D, long long ldq,
                               long long ldk, long long ldd, float qscale, int mode, int full, float* S, int band,
                               hipStream_t stream)
{
    const int nt = (T + ST - 1) / ST;
    const size_t lds = (size_t)YW * YNS * ZSTAGE;
    int nlist = 0;
    const int2* work = score_work_list(nt, (C + YG - 1) / YG, full ? 1 : 0, band, &nlist, stream);
    if (!work) return 1;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)interval_score_stream_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int ncu = 256, dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        ncu = v;
    int grid = ncu / NXCD * NXCD;                       // one persistent workgroup per CU, a whole number per XCD
    if (grid < NXCD) grid = NXCD;
    if (grid > nlist) grid = nlist;                     // nlist is a multiple of NXCD
    hipLaunchKernelGGL(interval_score_stream_kernel, dim3(grid), dim3(64 * YW), lds, stream, q, k, diag, C, T, D, ldq, ldk, ldd,
                       qscale, mode, full, S, work, nlist);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// 128x128 variant: operands shared by the whole workgroup through LDS
// ---------------------------------------------------------------------------------------------
// Counters of the streaming kernel (1.8 ms): the L1 waits on L2 80 % of the time and delivers 15 B/clk/CU -- its miss
// queue holds about 8 KB and an L2 round trip is ~600 cycles, so that IS what a CU can pull, whatever the access
// pattern.  32x32 tiles need 32 B/clk/CU to keep the fp32 matrix pipe busy (8 flop per operand byte).  Hence the
// classic answer: the eight waves of a workgroup multiply ONE chain's 128x128 tile together (wave = 32 rows x 64
// columns, two accumulators), the q and k chunks (128 rows x 32 contraction values each, 32 KB per stage, swizzled as
// above) are fetched once per workgroup: 32 flop per byte, 8 B/clk/CU at full rate.  A workgroup item is one tile for
// FOUR adjacent chains: finished blocks stay in registers and every cell leaves as one 16-byte piece.  The eight
// workgroups that cover the 32 chains of a 128-byte output line run side by side on the same XCD (persistent
// workgroups, static schedule: slot % 8 = chain quad), so its L2 assembles whole lines.  One s_barrier per chunk (no
// fence: the prefetch stays in flight); the matrix instructions of a wave alternate between its two accumulators.
constexpr int XTB = 128;                   // tile columns (begin positions)
#ifndef SEMICRF_SCORE_PRIO
#define SEMICRF_SCORE_PRIO 0      // 1: raised priority during a chunk's matrix instructions (measured: profiles/r02_scorer.md)
#endif
#ifndef SEMICRF_SCORE_XNS
#define SEMICRF_SCORE_XNS 3
#endif
constexpr int XNS = SEMICRF_SCORE_XNS;     // LDS stages (4: measured no faster, profiles/r02_scorer.md)

// Opt-in contraction on the bf16 matrix instructions (PREC = 1; interval_score_fwd, full_square bit 2).  Every fp32 operand
// is split into three bf16 limbs, x = hi + mid + lo EXACTLY (round-to-nearest splits: 8 + 8 + 8 significant bits and the
// signs of the remainders), and six of the nine limb products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16, smallest
// first: hi*lo, lo*hi, mid*mid, hi*mid, mid*hi, hi*hi.  Every limb product is exact (8 x 8 bits); what is dropped (mid*lo,
// lo*mid, lo*lo) is below 2^-23 |q_d k_d| per term -- the size of the rounding of one fp32 fmaf of the default path.  Six
// instructions of 8 passes replace eight fp32 instructions of 16 per 16 contraction values: 2.7x less matrix time; the
// split costs ~5 vector instructions per operand value and is done by every wave for its own operands.
// (Limbs3, split8, mma6: bf16x3.h)

// XTE = tile rows (end positions): 128 -> 8 waves, one workgroup per CU; 64 -> 4 waves, two (independent) workgroups
// per CU whose barriers and operand reads fall into each other's matrix phases (24 instead of 32 flop per byte).
// C = real chains; Cs = slots (the chain pitch of S); see SlotGeom.
// The 64- / 128-row shared-operand tile kernels: superseded as the default by scorer_tiled.hip (bit-identical results); since
// round 4 they are compiled into the DEBUG library only (libsemicrf_hip_debug.so, -DSEMICRF_DEBUG_BUILD=1), where the parity tests
// load them explicitly as the reference of test_scorer_tiled_bits.  The release library keeps the tiled kernel (default), the
// streaming kernel (T < 256), the register-load kernel (any alignment / contraction size) and the opt-in three-limb kernel.
#if SEMICRF_HAVE_TILE_REF
template <int XTE>
__global__ __launch_bounds__(512, 2) void interval_score_tile_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ diag, int C, int T, int D,
    long long ldq, long long ldk, long long ldd, float qscale, int mode, int full, float* __restrict__ S,
    int ntiles, int nquadp, int Cs, SlotGeom G, const float* __restrict__ rowc, long long ldrc)
{
    constexpr int XW = XTE / 16;                   // waves: (row block of 32, column half of 64)
    constexpr int KP = 16 / XW;                    // k pieces (8 rows x 128 bytes) per wave and chunk; q pieces: 2
    constexpr int XSTAGE = (XTE + XTB) * 128;      // bytes per stage: q chunk | k chunk
    extern __shared__ __attribute__((aligned(16))) char xlds[];    // [XNS][XSTAGE]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = lane & 31, half = lane >> 5;
    const int wer = wave >> 1, wh = wave & 1;                       // this wave: rows 32*wer.., columns 64*wh.. of the tile
    const unsigned lds0 = z_lds_addr(xlds);
    // timing ablations (SEMICRF_SCORE_DEBUG, debug builds only; results are wrong when set): 1 no matrix instructions, 2 no
    // operand requests, 4 no stores, 8 no LDS reads; a constant 0 in release builds
#ifdef SEMICRF_DEBUG_BUILD
    const int dbg = mode >> 8;
#else
    constexpr int dbg = 0;
#endif
    mode &= 0xff;
    const int nchunk = D / ZCH;
    const int nbt = (T + XTB - 1) / XTB;                            // column tiles
    const int xcd = blockIdx.x % NXCD, slot0 = blockIdx.x / NXCD, nslots = gridDim.x / NXCD;
    const long long nitems = (long long)ntiles * nquadp;            // nquadp: chain quads, padded to a multiple of 8

    // entry u of this XCD's list -> global item n: 8 consecutive entries = the 8 quads of one 128-byte line group
    auto item_of = [&](int u, int& et, int& bt, QuadInfo& qi) -> bool {
        const long long n = (long long)(u >> 3) * (8 * NXCD) + xcd * 8 + (u & 7);
        if (n >= nitems) return false;
        const int t = (int)(n / nquadp);
        qi = quad_info(G, (int)(n % nquadp));
        if (full) {
            et = t / nbt; bt = t % nbt;
        } else if (XTE == XTB) {
            // row et holds et+1 tiles
            et = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (et * (et + 1) / 2 > t) --et;
            while ((et + 1) * (et + 2) / 2 <= t) ++et;
            bt = t - et * (et + 1) / 2;
        } else {
            // rows 2p and 2p+1 hold p+1 tiles each; p(p+1) tiles lie before the pair
            int pp = (int)((sqrtf(4.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (pp * (pp + 1) > t) --pp;
            while ((pp + 1) * (pp + 2) <= t) ++pp;
            const int r = t - pp * (pp + 1);
            et = r < pp + 1 ? 2 * pp : 2 * pp + 1;
            bt = r < pp + 1 ? r : r - (pp + 1);
        }
        return true;
    };

    // reading lanes: lane = (row, half); segment 4*half + m of the row
    unsigned rdq[4], rdk[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const unsigned sw = (unsigned)((((4 * half + m) ^ ((row >> 1) & 7)) * 16));
        rdq[m] = (unsigned)((32 * wer + row) * 128) + sw;
        rdk[m] = (unsigned)(XTE * 128 + (64 * wh + row) * 128) + sw;       // second column block: + 32 rows = + 4096 bytes
    }

    // ---- request side (identical in all waves): (entry, chain of the quad, chunk, stage) ----------------------
    int nx_u = slot0, nx_j = 0, nx_ch = 0, nx_stage = 0;
    QuadInfo nx_q4 = {0, 0, 0, 0};
    bool nx_valid = false;
    unsigned voq[2], vok[4];      // (a [KP] array captured by the lambdas below trips the host compiler)
    const float* nx_q = q;
    const float* nx_k = k;
    // row constants (rowc != NULL, the merged projection): the XTE constants of chain j of a quad ride along with the chain's first
    // chunk -- wave j asks for them (one dword per lane, straight into LDS) right after its operand pieces, so every vmcnt wait
    // below stays at least as strict as without them -- into one of two buffers [chain][row] behind the stages, alternating by
    // REAL item (padding items request nothing).  The epilogue of item n reads buffer n & 1 while the requests of item n + 1 are
    // in flight; those of item n + 2 are issued only after a barrier of item n + 1 (>= 2 chunks per chain: D % 64 == 0), which
    // every wave reaches after its epilogue of item n.
    static_assert(XNS <= 3, "the two row-constant buffers assume requests run at most two chunks ahead");
    constexpr int RC_BYTES = 4 * XTE * 4;
    unsigned vorc[2] = {0u, 0u};
    const float* nx_rc = rowc;
    int nx_cnt = 0;
    auto set_chain = [&]() {
        const int c = nx_q4.ck + nx_j;                                       // nx_j < nx_q4.nr: a real chain
        nx_q = q + (size_t)c * T * ldq;
        nx_k = k + (size_t)c * T * ldk;
        if (rowc) nx_rc = rowc + (size_t)c * T * ldrc;
    };
    auto set_item = [&]() {
        int et = 0, bt = 0;
        // padding items (no real chain) request nothing: the consuming side skips them the same way
        while ((nx_valid = item_of(nx_u, et, bt, nx_q4)) && nx_q4.nr == 0) nx_u += nslots;
        if (nx_valid && rowc) {
#pragma unroll
            for (int h = 0; h < XTE / 64; ++h) {
                const int er = et * XTE + 64 * h + lane < T ? et * XTE + 64 * h + lane : T - 1;
                vorc[h] = (unsigned)((size_t)er * ldrc * 4);
            }
        }
        if (nx_valid) {
            // loading lanes: a piece is 8 rows x 128 bytes; this wave's q pieces 2*wave.. and k pieces KP*wave..
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int lr = 8 * (2 * wave + j) + (lane >> 3);
                const int seg = (lane & 7) ^ ((lr >> 1) & 7);
                const int er = et * XTE + lr < T ? et * XTE + lr : T - 1;     // clamped rows (masked at the write)
                voq[j] = (unsigned)(((size_t)er * ldq + seg * 4) * 4);
            }
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                const int lr = 8 * (KP * wave + j) + (lane >> 3);
                const int seg = (lane & 7) ^ ((lr >> 1) & 7);
                const int br = bt * XTB + lr < T ? bt * XTB + lr : T - 1;
                vok[j] = (unsigned)(((size_t)br * ldk + seg * 4) * 4);
            }
            nx_j = 0;
            nx_ch = 0;
            set_chain();
        }
    };
    auto issue_chunk = [&]() {
        if (!(dbg & 2)) {
        char* dq = xlds + nx_stage * XSTAGE + (2 * wave) * 1024;
        char* dk = xlds + nx_stage * XSTAGE + XTE * 128 + (KP * wave) * 1024;
        const auto rq = __builtin_amdgcn_make_buffer_rsrc((void*)nx_q, 0, 0x7fffffff, 0x00020000);
        const auto rk = __builtin_amdgcn_make_buffer_rsrc((void*)nx_k, 0, 0x7fffffff, 0x00020000);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void_t*)dq, 16, voq[0], nx_ch * (ZCH * 4), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void_t*)(dq + 1024), 16, voq[1], nx_ch * (ZCH * 4), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void_t*)dk, 16, vok[0], nx_ch * (ZCH * 4), 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void_t*)(dk + 1024), 16, vok[1], nx_ch * (ZCH * 4), 0, 0);
        if (KP == 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void_t*)(dk + 2048), 16, vok[KP - 2], nx_ch * (ZCH * 4), 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void_t*)(dk + 3072), 16, vok[KP - 1], nx_ch * (ZCH * 4), 0, 0);
        }
        if (rowc && nx_ch == 0 && wave == nx_j) {
            const auto rr = __builtin_amdgcn_make_buffer_rsrc((void*)nx_rc, 0, 0x7fffffff, 0x00020000);
            char* dr = xlds + XNS * XSTAGE + (nx_cnt & 1) * RC_BYTES + nx_j * (XTE * 4);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_void_t*)dr, 4, vorc[0], 0, 0, 0);
            if (XTE == 128) __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_void_t*)(dr + 256), 4, vorc[XTE / 64 - 1], 0, 0, 0);
        }
        }
        nx_stage = nx_stage + 1 == XNS ? 0 : nx_stage + 1;
        if (++nx_ch == nchunk) {
            nx_ch = 0;
            if (++nx_j == nx_q4.nr) {
                nx_u += nslots;
                ++nx_cnt;
                set_item();
            } else {
                set_chain();
            }
        }
    };

    set_item();
    int cur_u = slot0;
    {
        int e0_, b0_; QuadInfo q0_;
        if (!item_of(cur_u, e0_, b0_, q0_)) return;                 // uniform over the workgroup
    }

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    float hold[3][2][16];

    // prologue: XNS-1 chunks requested
    int inflight = 0;                           // chunks requested and not yet consumed
#pragma unroll
    for (int i = 0; i < XNS - 1; ++i)
        if (nx_valid) { issue_chunk(); ++inflight; }
    int rd_stage = 0;
    int cur_cnt = 0;                            // real items consumed: which row-constant buffer

    while (true) {
        int et, bt;
        QuadInfo qi;
        (void)item_of(cur_u, et, bt, qi);
        const int c4 = qi.c4;
        // 32x32 blocks of this wave that lie entirely above the diagonal are not multiplied (nor written)
        const int erow = et * (XTE / 32) + wer, bcol = bt * (XTB / 32) + 2 * wh;
        const bool on0 = full || bcol <= erow, on1 = full || bcol + 1 <= erow;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            for (int ch = 0; ch < (j < qi.nr ? nchunk : 0); ++ch) {
                // this wave's pieces of the current chunk have landed (a younger request may stay in flight) ...
                if (XNS >= 4 && inflight >= 3) {
                    if (KP == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
                } else if (inflight >= 2) {
                    if (KP == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                // ... and so have everybody's; everybody is also done reading the previous chunk (no fence: the
                // prefetch stays in flight)
                __builtin_amdgcn_s_barrier();
                --inflight;
                if (nx_valid) { issue_chunk(); ++inflight; }        // into the stage of the previous chunk
                v4f qa[4], ka0[4], ka1[4];
                const unsigned sb = lds0 + (unsigned)(rd_stage * XSTAGE);
                rd_stage = rd_stage + 1 == XNS ? 0 : rd_stage + 1;
                if (dbg & 8) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) { qa[m] = (v4f)(1.0f); ka0[m] = (v4f)(1.0f); ka1[m] = (v4f)(1.0f); }
                } else
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    asm volatile("ds_read_b128 %0, %3\n\tds_read_b128 %1, %4\n\tds_read_b128 %2, %4 offset:4096"
                                 : "=&v"(qa[m]), "=&v"(ka0[m]), "=&v"(ka1[m])
                                 : "v"(sb + rdq[m]), "v"(sb + rdk[m]));
                }
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]), "+v"(ka0[0]), "+v"(ka0[1]), "+v"(ka0[2]),
                               "+v"(ka0[3]), "+v"(ka1[0]), "+v"(ka1[1]), "+v"(ka1[2]), "+v"(ka1[3]));
                __builtin_amdgcn_sched_barrier(0);
#if SEMICRF_SCORE_PRIO
                __builtin_amdgcn_s_setprio(2);
#endif
                if (dbg & 1) {
                    acc0[0] += qa[0].x + qa[1].y + qa[2].z + qa[3].w + ka0[0].x + ka0[3].w;
                    acc1[0] += ka1[0].x + ka1[3].w;
                } else if (on0 && on1) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].x, ka0[m].x, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].x, ka1[m].x, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].y, ka0[m].y, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].y, ka1[m].y, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].z, ka0[m].z, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].z, ka1[m].z, acc1, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].w, ka0[m].w, acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].w, ka1[m].w, acc1, 0, 0, 0);
                    }
                } else if (on0) {
#pragma unroll
                    for (int m = 0; m < 4; ++m) {
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].x, ka0[m].x, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].y, ka0[m].y, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].z, ka0[m].z, acc0, 0, 0, 0);
                        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(qa[m].w, ka0[m].w, acc0, 0, 0, 0);
                    }
                }
#if SEMICRF_SCORE_PRIO
                __builtin_amdgcn_s_setprio(0);
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
            if (j < 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hold[j][0][r] = acc0[r]; hold[j][1][r] = acc1[r];
                    acc0[r] = 0.0f; acc1[r] = 0.0f;
                }
            }
        }
        // ---- write the item's four chains: one 16-byte piece per cell (C/D layout: col = lane & 31,
        //      row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
        {
            const bool vec = (Cs & 3) == 0 && c4 + 3 < Cs;
            // merged projection (interval_score_fwd_pc): the row constants of this wave's rows, four consecutive rows (one row group
            // r >> 2) of the four chains per pass
            const unsigned rb = lds0 + (unsigned)(XNS * XSTAGE + (cur_cnt & 1) * RC_BYTES + (32 * wer + 4 * half) * 4);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v4f rcv[4];
                if (rowc) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        asm volatile("ds_read_b128 %0, %1" : "=v"(rcv[i]) : "v"(rb + (unsigned)((i * XTE + 8 * g) * 4)));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rcv[0]), "+v"(rcv[1]), "+v"(rcv[2]), "+v"(rcv[3]));
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int b = bt * XTB + 64 * wh + 32 * t + row;
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        const int r = 4 * g + rr;
                        const int e = et * XTE + 32 * wer + rr + 8 * g + 4 * half;
                        const int len = e > b ? e - b : b - e;
                        const float sc = qscale * len_scale_mfma(len, mode);
                        const float last = t == 0 ? acc0[r] : acc1[r];
                        float v[4] = {hold[0][t][r], hold[1][t][r], hold[2][t][r], last};
                        if (rowc) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < qi.nr) v[i] += rcv[i][rr];
                        }
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] *= sc;
                        if (e < T && b < T && (full || b <= e) && qi.nr > 0 && !(dbg & 4)) {
                            if (e == b) {
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (i < qi.nr) v[i] += diag[((size_t)(qi.ck + i) * T + e) * ldd];
                            }
                            float* dst = S + ((size_t)e * T + b) * Cs + c4;
                            if (vec) {
                                // (plain stores: L2 merges the eight 16-byte pieces of a line; nontemporal ones do not -- 2.4 ms)
                                *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);      // ghost slots of the quad: exact zeros (hold = 0)
                                for (int z = 4; z <= qi.tz; z += 4) *(float4*)(dst + z) = make_float4(0.f, 0.f, 0.f, 0.f);   // the group's ghost tail
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    if (i < qi.nr) dst[i] = v[i];
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        }
        if (qi.nr > 0) ++cur_cnt;
        cur_u += nslots;
        int e2, b2;
        QuadInfo q2;
        if (!item_of(cur_u, e2, b2, q2)) break;
    }
}

template <int XTE>
static int launch_score_tile(const float* q, const float* k, const float* diag, int C, int T, int D, long long ldq,
                             long long ldk, long long ldd, float qscale, int mode, int full, float* S, hipStream_t stream,
                             int group, int pitch, const float* rowc, long long ldrc)
{
    const SlotGeom G = slot_geom(C, group, pitch);
    const int Cs = (C / group) * pitch;
    const int net = (T + XTE - 1) / XTE, nbt = (T + XTB - 1) / XTB;
    int ntiles = 0;
    if (full) ntiles = net * nbt;
    else
        for (int et = 0; et < net; ++et) ntiles += (et * XTE + XTE - 1) / XTB + 1 < nbt ? (et * XTE + XTE - 1) / XTB + 1 : nbt;
    const int nquadp = (G.nrq + 7) / 8 * 8;
    const size_t lds = (size_t)XNS * (XTE + XTB) * 128 + 2 * (4 * XTE * 4);     // stages + the two row-constant buffers
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)interval_score_tile_kernel<XTE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int ncu = 256, dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        ncu = v;
    // persistent workgroups (128 / XTE per CU); per XCD a multiple of 8 slots
    int grid = ncu * (128 / XTE) / (8 * NXCD) * (8 * NXCD);
    if (grid < 8 * NXCD) grid = 8 * NXCD;
    const long long nitems = (long long)ntiles * nquadp;
    const long long need = (nitems + 8 * NXCD - 1) / (8 * NXCD) * (8 * NXCD);
    if (grid > need) grid = (int)need;
    int dbg = 0;
#ifdef SEMICRF_DEBUG_BUILD
    if (const char* e = getenv("SEMICRF_SCORE_DEBUG")) dbg = atoi(e) & 0xff;      // timing ablations (wrong results): debug builds only
#endif
    hipLaunchKernelGGL(interval_score_tile_kernel<XTE>, dim3(grid), dim3(XTE * 4), lds, stream, q, k, diag, C, T, D, ldq, ldk, ldd,
                       qscale, mode | (dbg << 8), full, S, ntiles, nquadp, Cs, G, rowc, ldrc);
    return 0;
}
#endif  // SEMICRF_HAVE_TILE_REF

// ---------------------------------------------------------------------------------------------
// 128x128 tile kernel with the three-limb bf16 contraction (opt-in: interval_score_fwd, full_square | SEMICRF_SCORE_BF16X3)
// ---------------------------------------------------------------------------------------------
// Same items, schedule and epilogue as interval_score_tile_kernel<128>.  What differs is the operand path: the vector
// work of the split (~5 instructions per value) would cost as much as the matrix instructions save if every wave split
// its own operands (each value is used by 2 or 4 waves: measured 1.17 vs 1.23 ms).  So every value is split ONCE: the
// 512 lanes fetch a chunk (256 rows x 32 contraction values) from global memory into registers, two chunks ahead, split
// it and store the limbs to LDS ([stage][limb][row][4 pieces of 8 values], pieces swizzled by row so that writes and
// reads are conflict-free); after one s_barrier per chunk the waves read bf16x8 operands and issue 24 instructions of
// 8 passes (the fp32 kernel: 32 of 16).
constexpr int W3_NS = 2;
#ifndef SEMICRF_SCORE_SCHED3
#define SEMICRF_SCORE_SCHED3 1
#endif

// XTE = 128: eight waves, one workgroup per CU (the one that is launched).  XTE = 64 -- four waves and 72 KB, two independent
// workgroups per CU that could fall into each other's phases -- measured slower everywhere (1.28 vs 0.96 ms at T=1024, C=352;
// 634 vs 552 us at T=691, C=360) and spills with the pinned instruction order.
template <int XTE>
__global__ __launch_bounds__(512, 2) void interval_score_tile3_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ diag, int C, int T, int D,
    long long ldq, long long ldk, long long ldd, float qscale, int mode, int full, float* __restrict__ S,
    int ntiles, int nquadp, int Cs, SlotGeom G, const float* __restrict__ rowc, long long ldrc)
{
    constexpr int NTH = XTE * 4;                 // threads
    constexpr int NR = XTE + XTB;                // rows per stage: q rows | k rows
    constexpr int LIMB = NR * 64;                // bytes per limb plane: 32 values x 2 bytes per row
    constexpr int STAGE = 3 * LIMB;
    constexpr int NU = NR / XTE;                 // units (one row, 8 values) per lane and chunk: unit 0 is a q row, the others k rows
    extern __shared__ __attribute__((aligned(16))) char xlds[];    // [W3_NS][STAGE] | row constants [XTE][4]
    float* const rcl = (float*)(xlds + W3_NS * 3 * (XTE + XTB) * 64);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = lane & 31, half = lane >> 5;
    const int wer = wave >> 1, wh = wave & 1;                       // this wave: rows 32*wer.., columns 64*wh.. of the tile
    // timing ablations (SEMICRF_SCORE_DEBUG, debug builds only; results are wrong when set): 1 no matrix instructions, 2 no
    // operand fetch, 4 no stores.  A constant 0 in release builds: a run-time test in front of every matrix instruction costs
    // two scalar instructions and ends its basic block
#ifdef SEMICRF_DEBUG_BUILD
    const int dbg = mode >> 8;
#else
    constexpr int dbg = 0;
#endif
    mode &= 0xff;
    const int nchunk = D / ZCH;                                     // even (D % 64 == 0)
    const int nbt = (T + XTB - 1) / XTB;
    const int xcd = blockIdx.x % NXCD, slot0 = blockIdx.x / NXCD, nslots = gridDim.x / NXCD;
    const long long nitems = (long long)ntiles * nquadp;

    auto item_of = [&](int u, int& et, int& bt, QuadInfo& qi) -> bool {
        const long long n = (long long)(u >> 3) * (8 * NXCD) + xcd * 8 + (u & 7);
        if (n >= nitems) return false;
        // line group (32 chains = 8 quads) major: all tiles of a group before the next group, so that the group's q and k
        // (64 MB at T=1024) come from HBM once and from the memory-side cache afterwards.  In tile-major order over all
        // chains (the fp32 kernel's, which is bound by its matrix instructions) every operand tile is fetched again:
        // 3.3 GB at T=1024, C=352 -- 1.09 ms instead of 0.86 here.
        const long long v = n >> 3;
        const int t = (int)(v % ntiles);
        qi = quad_info(G, (int)(v / ntiles) * 8 + (int)(n & 7));
        if (full) {
            et = t / nbt; bt = t % nbt;
        } else if (XTE == XTB) {
            et = (int)((sqrtf(8.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (et * (et + 1) / 2 > t) --et;
            while ((et + 1) * (et + 2) / 2 <= t) ++et;
            bt = t - et * (et + 1) / 2;
        } else {
            // rows 2p and 2p+1 hold p+1 tiles each; p(p+1) tiles lie before the pair
            int pp = (int)((sqrtf(4.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (pp * (pp + 1) > t) --pp;
            while ((pp + 1) * (pp + 2) <= t) ++pp;
            const int r = t - pp * (pp + 1);
            et = r < pp + 1 ? 2 * pp : 2 * pp + 1;
            bt = r < pp + 1 ? r : r - (pp + 1);
        }
        return true;
    };

    // reading lanes: lane = (row, half); the instruction of slab sl takes piece 2*half + sl of the row
    unsigned rdA[2], rdB[2];
    {
        const int ra = 32 * wer + row, rb = XTE + 64 * wh + row;   // second column block: + 32 rows = + 2048 bytes, same swizzle
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            rdA[sl] = (unsigned)(ra * 64 + (((2 * half + sl) ^ ((ra >> 2) & 3)) * 16));
            rdB[sl] = (unsigned)(rb * 64 + (((2 * half + sl) ^ ((rb >> 2) & 3)) * 16));
        }
    }
    // loading lanes: unit j = (row j * XTE + tid / 4 of the stage, piece tid % 4)
    const int oct = threadIdx.x & 3;
    unsigned wof[NU];
#pragma unroll
    for (int j = 0; j < NU; ++j) {
        const int R = j * XTE + (int)(threadIdx.x >> 2);
        wof[j] = (unsigned)(R * 64 + ((oct ^ ((R >> 2) & 3)) * 16));
    }

    // ---- request side (identical in all waves): (entry, chain of the quad, chunk) --------------------------------
    int nx_u = slot0, nx_j = 0, nx_ch = 0;
    QuadInfo nx_q4 = {0, 0, 0, 0};
    bool nx_valid = false;
    size_t nx_off[NU];
    const float* nx_q = q;
    const float* nx_k = k;
    auto set_chain = [&]() {
        const int c = nx_q4.ck + nx_j;                                       // nx_j < nx_q4.nr: a real chain
        nx_q = q + (size_t)c * T * ldq;
        nx_k = k + (size_t)c * T * ldk;
    };
    auto set_item = [&]() {
        int et = 0, bt = 0;
        // padding items (no real chain) request nothing: the consuming side skips them the same way
        while ((nx_valid = item_of(nx_u, et, bt, nx_q4)) && nx_q4.nr == 0) nx_u += nslots;
        if (nx_valid) {
#pragma unroll
            for (int j = 0; j < NU; ++j) {
                const int gr = (j == 0 ? et * XTE : bt * XTB + (j - 1) * XTE) + (int)(threadIdx.x >> 2);
                const int cr = gr < T ? gr : T - 1;                          // clamped rows (masked at the write)
                nx_off[j] = (size_t)cr * (size_t)(j == 0 ? ldq : ldk) + (size_t)(oct * 8);
            }
            nx_j = 0;
            nx_ch = 0;
            set_chain();
        }
    };
    auto fetch = [&](v4f (&g)[NU][2]) {
        if (!nx_valid) return;
        if (!(dbg & 2))
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const float* src = (j == 0 ? nx_q : nx_k) + nx_off[j] + (size_t)nx_ch * ZCH;
            g[j][0] = *(const v4f*)src;
            g[j][1] = *(const v4f*)(src + 4);
        }
        if (++nx_ch == nchunk) {
            nx_ch = 0;
            if (++nx_j == nx_q4.nr) {
                nx_u += nslots;
                set_item();
            } else {
                set_chain();
            }
        }
    };
    auto convert = [&](const v4f (&g)[NU][2], int stage) {
        char* base = xlds + stage * STAGE;
#pragma unroll
        for (int j = 0; j < NU; ++j) {
            const Limbs3 L = split8_pairs(g[j][0], g[j][1]);
            *(bf16x8*)(base + wof[j]) = L.h;
            *(bf16x8*)(base + LIMB + wof[j]) = L.m;
            *(bf16x8*)(base + 2 * LIMB + wof[j]) = L.l;
        }
    };

    set_item();
    int cur_u = slot0;
    {
        int e0_, b0_; QuadInfo q0_;
        if (!item_of(cur_u, e0_, b0_, q0_)) return;                 // uniform over the workgroup
    }
#ifdef SEMICRF_SCORE_PROBE
    // cycle accounting of one wave (probe build; tools/score_probe.py): [0] waiting at the barrier, [1] operand reads + matrix
    // instructions + split + limb stores, [2] the fetch of the chunk after next, [3] the epilogue; written over the (unused)
    // cells S[0, 1.., :] of a lower-triangle-only launch
    unsigned long long pc[4] = {0, 0, 0, 0}, pt = __builtin_readcyclecounter();
#define SCORE_PROBE(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); pc[i] += now_ - pt; pt = now_; } while (0)
#else
#define SCORE_PROBE(i) do { } while (0)
#endif

    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
    float hold[3][2][16];

    v4f g0[NU][2], g1[NU][2];
#pragma unroll
    for (int j = 0; j < NU; ++j) { g0[j][0] = (v4f)(0.0f); g0[j][1] = (v4f)(0.0f); g1[j][0] = (v4f)(0.0f); g1[j][1] = (v4f)(0.0f); }
    fetch(g0);
    fetch(g1);
    convert(g0, 0);
    fetch(g0);

    while (true) {
        int et, bt;
        QuadInfo qi;
        (void)item_of(cur_u, et, bt, qi);
        const int c4 = qi.c4;
        // merged projection (rowc): the item's XTE row constants of its four chains, one per thread, requested here, parked in the LDS
        // behind the first chain's loop and read as one 16-byte piece per row in the epilogue (as 128 scattered loads INSIDE the
        // epilogue they cost this kernel 0.26 ms at the training shape: 843 against the exact kernel's 710 us).  Zero for ghost
        // chains and rows past T.
        float rcv = 0.0f;
        if (rowc) {
            const int ri = (int)threadIdx.x / XTE, re = et * XTE + (int)threadIdx.x % XTE;
            if (ri < qi.nr && re < T) rcv = rowc[((size_t)(qi.ck + ri) * T + re) * ldrc];
        }
        const int erow = et * (XTE / 32) + wer, bcol = bt * (XTB / 32) + 2 * wh;
        const bool on0 = full || bcol <= erow, on1 = full || bcol + 1 <= erow;
        // (the blocks-above-the-diagonal cases are separate instantiations: a branch inside the half iteration would end the
        // basic block the instruction order below is imposed on)
        auto multiply = [&](int stage, auto ON0, auto ON1) {
            const char* base = xlds + stage * STAGE;
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                Limbs3 A, B;
                A.h = *(const bf16x8*)(base + rdA[sl]);
                A.m = *(const bf16x8*)(base + LIMB + rdA[sl]);
                A.l = *(const bf16x8*)(base + 2 * LIMB + rdA[sl]);
                if constexpr (decltype(ON0)::value) {
                    B.h = *(const bf16x8*)(base + rdB[sl]);
                    B.m = *(const bf16x8*)(base + LIMB + rdB[sl]);
                    B.l = *(const bf16x8*)(base + 2 * LIMB + rdB[sl]);
                    acc0 = mma6(A, B, acc0);
                }
                if constexpr (decltype(ON1)::value) {
                    B.h = *(const bf16x8*)(base + rdB[sl] + 2048);
                    B.m = *(const bf16x8*)(base + LIMB + rdB[sl] + 2048);
                    B.l = *(const bf16x8*)(base + 2 * LIMB + rdB[sl] + 2048);
                    acc1 = mma6(A, B, acc1);
                }
            }
        };
        // Instruction order of a half iteration.  Inside a workgroup the barrier keeps every wave in the same phase, so whatever
        // overlaps has to overlap inside a wave: the matrix pipe takes an instruction every 32 cycles, and the operand reads
        // of the next slab, the split of the chunk after next (~90 vector instructions) and its limb stores go into those
        // gaps, one piece per matrix instruction, the order pinned by scheduling barriers (in source order the compiler puts
        // them behind the last matrix instruction: 1.09 ms, no better than fp32; sched_group_barrier hints were not followed).
        auto half_full = [&](int srd, const v4f (&g)[NU][2], int swr) {
            const char* rb = xlds + srd * STAGE;
            char* wb = xlds + swr * STAGE;
            Limbs3 A0, A1, B0, B1;
            auto ld = [&](Limbs3& L, unsigned off) {
                L.h = *(const bf16x8*)(rb + off);
                L.m = *(const bf16x8*)(rb + LIMB + off);
                L.l = *(const bf16x8*)(rb + 2 * LIMB + off);
            };
            unsigned ph[4], pm[4], pl[4];
            auto piece = [&](int j, int pr) {
                const v4f x = g[j][pr >> 1];
                if (pr & 1) split_pair(x.z, x.w, ph[pr], pm[pr], pl[pr]);
                else split_pair(x.x, x.y, ph[pr], pm[pr], pl[pr]);
            };
            auto store = [&](int j) {
                *(u32x4*)(wb + wof[j]) = (u32x4){ph[0], ph[1], ph[2], ph[3]};
                *(u32x4*)(wb + LIMB + wof[j]) = (u32x4){pm[0], pm[1], pm[2], pm[3]};
                *(u32x4*)(wb + 2 * LIMB + wof[j]) = (u32x4){pl[0], pl[1], pl[2], pl[3]};
            };
            auto mma1 = [&](const Limbs3& A, const Limbs3& B, f32x16& acc, int p) {
                switch (p) {
                case 0: acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.l, acc, 0, 0, 0); break;
                case 1: acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l, B.h, acc, 0, 0, 0); break;
                case 2: acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.m, acc, 0, 0, 0); break;
                case 3: acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.m, acc, 0, 0, 0); break;
                case 4: acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.h, acc, 0, 0, 0); break;
                default: acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.h, acc, 0, 0, 0); break;
                }
            };
            ld(A0, rdA[0]);
            ld(B0, rdB[0]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                // acc0 slab 0 (A0, B0) | acc1 slab 0 (A0, B1) | acc0 slab 1 (A1, B0) | acc1 slab 1 (A1, B1)
                if (dbg & 1) {
                    if (i == 0) acc0[0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, A0.h).x);
                } else if (i < 6) mma1(A0, B0, acc0, i);
                else if (i < 12) mma1(A0, B1, acc1, i - 6);
                else if (i < 18) mma1(A1, B0, acc0, i - 12);
                else mma1(A1, B1, acc1, i - 18);
                __builtin_amdgcn_sched_barrier(0);
                if (i == 0) ld(B1, rdB[0] + 2048);
                if (i >= 1 && i <= 4) piece(0, i - 1);
                if (i == 5) store(0);
                if (i == 6) ld(A1, rdA[1]);
                if (i == 7) ld(B0, rdB[1]);
                if (i >= 8 && i <= 11) piece(1, i - 8);
                if (i == 12) store(1);
                if (i == 13) ld(B1, rdB[1] + 2048);
                if (NU > 2 && i >= 14 && i <= 17) piece(NU - 1, i - 14);
                if (NU > 2 && i == 18) store(NU - 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        auto chain_loop = [&](auto ON0, auto ON1) {
            constexpr bool both = decltype(ON0)::value && decltype(ON1)::value;
            for (int ch = 0; ch < nchunk; ch += 2) {
                // the limbs of chunk ch are stored (mine: lgkmcnt, everybody's: the barrier) and everybody has read the
                // other stage (no fence: the global prefetch stays in flight)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                SCORE_PROBE(0);
                if constexpr (both && SEMICRF_SCORE_SCHED3) {
                    half_full(0, g1, 1);
                } else {
                    multiply(0, ON0, ON1);
                    convert(g1, 1);
                }
                SCORE_PROBE(1);
                fetch(g1);
                SCORE_PROBE(2);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                SCORE_PROBE(0);
                if constexpr (both && SEMICRF_SCORE_SCHED3) {
                    half_full(1, g0, 0);
                } else {
                    multiply(1, ON0, ON1);
                    convert(g0, 0);
                }
                SCORE_PROBE(1);
                fetch(g0);
                SCORE_PROBE(2);
            }
        };
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (j >= qi.nr) { }                                       // a ghost slot of the quad: nothing was requested for it
            else if (on0 && on1) chain_loop(std::true_type{}, std::true_type{});
            else if (on0) chain_loop(std::true_type{}, std::false_type{});
            else chain_loop(std::false_type{}, std::false_type{});
            // (the previous item's epilogue is behind everybody: chain 0's loop has barriers.  A PADDING item -- a quad without a real chain,
            // e.g. slots 92..95 of a 96-slot segment -- runs no loop, i.e. passes no barrier: it must not touch the buffer, or its zeros
            // land under the waves that are still reading the previous item's constants.  Found by the round-6 parity test of the
            // "bf16x3-all" route against the reference's segment goldens: quads 7 and 15 of a 90-symbol segment wrong by up to 176.)
            if (j == 0 && rowc && qi.nr > 0) rcl[((int)threadIdx.x % XTE) * 4 + (int)threadIdx.x / XTE] = rcv;
            if (j < 3) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    hold[j][0][r] = acc0[r]; hold[j][1][r] = acc1[r];
                    acc0[r] = 0.0f; acc1[r] = 0.0f;
                }
            }
        }
        // ---- write the item's four chains: one 16-byte piece per cell (C/D layout: col = lane & 31,
        //      row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
        {
            const bool vec = (Cs & 3) == 0 && c4 + 3 < Cs;
            if (rowc) {                                                  // everybody's row constants are in the LDS
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int b = bt * XTB + 64 * wh + 32 * t + row;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int e = et * XTE + 32 * wer + (r & 3) + 8 * (r >> 2) + 4 * half;
                    const int len = e > b ? e - b : b - e;
                    const float sc = qscale * len_scale_mfma(len, mode);
                    const float last = t == 0 ? acc0[r] : acc1[r];
                    float v[4] = {hold[0][t][r], hold[1][t][r], hold[2][t][r], last};
                    if (rowc) {
                        // merged projection (interval_score_fwd_p, rowc): a per-(chain, end) constant joins the contraction
                        const v4f rc4 = *(const v4f*)(rcl + (32 * wer + (r & 3) + 8 * (r >> 2) + 4 * half) * 4);
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += rc4[i];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] *= sc;
                    if (e < T && b < T && (full || b <= e) && qi.nr > 0 && !(dbg & 4)) {
                        if (e == b) {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < qi.nr) v[i] += diag[((size_t)(qi.ck + i) * T + e) * ldd];
                        }
                        float* dst = S + ((size_t)e * T + b) * Cs + c4;
                        if (vec) {
                            *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);      // ghost slots of the quad: exact zeros (hold = 0)
                            for (int z = 4; z <= qi.tz; z += 4) *(float4*)(dst + z) = make_float4(0.f, 0.f, 0.f, 0.f);   // the group's ghost tail
                        } else {
#pragma unroll
                            for (int i = 0; i < 4; ++i)
                                if (i < qi.nr) dst[i] = v[i];
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        }
        SCORE_PROBE(3);
        cur_u += nslots;
        int e2, b2;
        QuadInfo q2;
        if (!item_of(cur_u, e2, b2, q2)) break;
    }
#ifdef SEMICRF_SCORE_PROBE
    if (lane == 0 && !full)
        for (int i = 0; i < 4; ++i) S[(size_t)Cs + (size_t)((blockIdx.x * (XTE / 16) + wave) * 4 + i)] = (float)pc[i];
#endif
#undef SCORE_PROBE
}

template <int XTE>
static int launch_score_tile3(const float* q, const float* k, const float* diag, int C, int T, int D, long long ldq,
                              long long ldk, long long ldd, float qscale, int mode, int full, float* S, hipStream_t stream,
                              int group, int pitch, const float* rowc, long long ldrc)
{
    const SlotGeom G = slot_geom(C, group, pitch);
    const int Cs = (C / group) * pitch;
    const int net = (T + XTE - 1) / XTE, nbt = (T + XTB - 1) / XTB;
    int ntiles = 0;
    if (full) ntiles = net * nbt;
    else
        for (int et = 0; et < net; ++et) ntiles += (et * XTE + XTE - 1) / XTB + 1 < nbt ? (et * XTE + XTE - 1) / XTB + 1 : nbt;
    const int nquadp = (G.nrq + 7) / 8 * 8;
    const size_t lds = (size_t)W3_NS * 3 * (XTE + XTB) * 64 + (size_t)XTE * 16;
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)interval_score_tile3_kernel<XTE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int ncu = 256, dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        ncu = v;
    int grid = ncu * (128 / XTE) / (8 * NXCD) * (8 * NXCD);       // persistent workgroups (128 / XTE per CU); per XCD a multiple of 8 slots
    if (grid < 8 * NXCD) grid = 8 * NXCD;
    const long long nitems = (long long)ntiles * nquadp;
    const long long need = (nitems + 8 * NXCD - 1) / (8 * NXCD) * (8 * NXCD);
    if (grid > need) grid = (int)need;
    int dbg = 0;
#ifdef SEMICRF_DEBUG_BUILD
    if (const char* e = getenv("SEMICRF_SCORE_DEBUG")) dbg = atoi(e) & 0xff;      // timing ablations (wrong results): debug builds only
#endif
    hipLaunchKernelGGL(interval_score_tile3_kernel<XTE>, dim3(grid), dim3(XTE * 4), lds, stream, q, k, diag, C, T, D, ldq, ldk, ldd, qscale,
                       mode | (dbg << 8), full, S, ntiles, nquadp, Cs, G, rowc, ldrc);
    return 0;
}

void launch_interval_score_tiled(const float* q, const float* k, const float* diag, const float* rowc, int C, int T, int D,
                                 long long ldq, long long ldk, long long ldd, long long ldrc, float qscale, int mode, int full,
                                 float* S, hipStream_t stream, int group, int pitch);      // scorer_tiled.hip

// test hook: force one of the forward kernels (0 register loads, 32 streaming, 64 / 128 shared-operand tiles, 2 the tiles with
// the epilogue inside the contraction loop (scorer_tiled.hip); -1 = auto)
static std::atomic<int> g_score_variant{-1};
void set_score_variant(int v) { g_score_variant.store(v, std::memory_order_relaxed); }

// group / pitch: the slot layout of S's chain axis (SlotGeom); group == pitch == C is the contiguous layout.  Returns 2 when a
// slot layout is asked for and the shared-operand tile kernels do not apply (the caller reports SEMICRF_EINVAL).
bool interval_score_slots_supported(int C, int T, int D, const float* q, const float* k, long long ldq, long long ldk)
{
    const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ldq % 4 == 0 && ldk % 4 == 0;
    return aligned && D % 64 == 0 && T >= 128 && (long long)T * ldq * 4 < (1ll << 31) && (long long)T * ldk * 4 < (1ll << 31);
}

int launch_interval_score_mfma(const float* q, const float* k, const float* diag, int C, int T, int D,
                                long long ldq, long long ldk, long long ldd, float qscale, int mode, int full,
                                float* S, hipStream_t stream, int prec, int group, int pitch, const float* rowc, long long ldrc)
{
    // rowc (merged projection) lives in the tile kernels like the slot layout
    const bool slots = !(group == C && pitch == C) || rowc != nullptr;
    if (slots && !interval_score_slots_supported(C, T, D, q, k, ldq, ldk)) return 2;
    const int nt = (T + ST - 1) / ST;
    const size_t lds = (size_t)ST * ST * SPAD * sizeof(float);
    const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ldq % 4 == 0 && ldk % 4 == 0;
    const int nch = (D / 64 <= 4) ? D / 64 : 0;
    const int band = 4;
    // 16-byte aligned rows: the LDS-staged kernel (T*ld*4 < 2^31: 32-bit buffer offsets)
    if (aligned && D % 64 == 0 && (long long)T * ldq * 4 < (1ll << 31) && (long long)T * ldk * 4 < (1ll << 31)) {
        // 64-row tiles when they waste less of the last tile row (e.g. T=691: 704 vs 768 rows) -- measured 779 vs 814 us
        int variant = T < 256 ? 32 : ((T + 63) / 64 * 64 < (T + 127) / 128 * 128 ? 64 : 128);
        const int forced = g_score_variant.load(std::memory_order_relaxed);     // test hook (semicrf_debug_score_variant), -1 = auto
        // the tiles with the epilogue inside the contraction loop (scorer_tiled.hip): a tie with the kernels below at T=1024 x 352
        // (1.115 vs 1.107 ms, a third less written to HBM), faster at the model's shapes (691 x 360: 0.653 vs 0.707 ms, 1024 x 88:
        // 0.312 vs 0.365, 691 x 90: 0.183 vs 0.189)
        const int Cs = (C / group) * pitch;
        const bool tiled_ok = prec == 0 && T >= 128 && D <= 256 && (long long)32 * T * Cs * 4 < (1ll << 31);
        if (tiled_ok && (T >= 256 || slots)) variant = 2;
        if (forced >= 0) variant = forced;
        if (variant == 2 && tiled_ok) {
            launch_interval_score_tiled(q, k, diag, rowc, C, T, D, ldq, ldk, ldd, ldrc, qscale, mode, full, S, stream, group, pitch);
            return 0;
        }
        if (variant == 2) variant = 128;
        if (slots && variant != 64) variant = 128;                 // the slot layout lives in the tile kernels
        if (prec == 1 && T >= 128)
            return launch_score_tile3<128>(q, k, diag, C, T, D, ldq, ldk, ldd, qscale, mode, full, S, stream, group, pitch, rowc, ldrc);
#if SEMICRF_HAVE_TILE_REF
        if (variant == 128)
            return launch_score_tile<128>(q, k, diag, C, T, D, ldq, ldk, ldd, qscale, mode, full, S, stream, group, pitch, rowc, ldrc);
        if (variant == 64)
            return launch_score_tile<64>(q, k, diag, C, T, D, ldq, ldk, ldd, qscale, mode, full, S, stream, group, pitch, rowc, ldrc);
#else
        // release library: what the tiled kernel does not take (D > 256, 32 T Cs floats beyond 32-bit offsets) runs on the
        // register-load kernel below; a slot layout / row constant there is refused by the caller's check
        if (slots) return 2;
#endif
        if (variant == 32)
            return launch_score_stream(q, k, diag, C, T, D, ldq, ldk, ldd, qscale, mode, full, S, band, stream);
    }
    int ngrid = 0;
    const int2* work = score_work_list(nt, (C + SC - 1) / SC, full ? 1 : 0, band, &ngrid, stream);
    if (!work) return 1;
    const dim3 grid(ngrid), block(256);
#define SEMICRF_FWD_LAUNCH(A, N)                                                                                        \
    do {                                                                                                                \
        static PerDeviceOnce attr_once;                                                                                   \
        if (attr_once.first()) {                                                                                                \
            (void)hipFuncSetAttribute((const void*)interval_score_mfma_kernel<A, N>,                                    \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                           \
        }                                                                                                               \
        hipLaunchKernelGGL((interval_score_mfma_kernel<A, N>), grid, block, lds, stream, q, k, diag, C, T, D, ldq, ldk, \
                           ldd, qscale, mode, full, S, work);                                                           \
    } while (0)
#define SEMICRF_FWD_DISPATCH(A)                                                                                         \
    switch (nch) {                                                                                                      \
    case 1: SEMICRF_FWD_LAUNCH(A, 1); break;                                                                            \
    case 2: SEMICRF_FWD_LAUNCH(A, 2); break;                                                                            \
    case 3: SEMICRF_FWD_LAUNCH(A, 3); break;                                                                            \
    case 4: SEMICRF_FWD_LAUNCH(A, 4); break;                                                                            \
    default: SEMICRF_FWD_LAUNCH(A, 0); break;                                                                           \
    }
    if (aligned) { SEMICRF_FWD_DISPATCH(true) } else { SEMICRF_FWD_DISPATCH(false) }
#undef SEMICRF_FWD_DISPATCH
#undef SEMICRF_FWD_LAUNCH
    return 0;
}

}  // namespace semicrf
