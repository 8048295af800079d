// scorer_tiled.hip -- interval scores on the matrix cores, exact fp32, with the epilogue INSIDE the contraction loop
// (LayersTransformer.py:406-441; same definition and results as interval_score_tile_kernel in scorer_mfma.hip).
//
// What interval_score_tile_kernel loses (profiles/r03_derived.json: the matrix pipe is busy 57 % of the kernel, the backward
// GEMMs with the same barrier structure 74 %) is its epilogue: S is chain-minor, so a workgroup owns 16 bytes (four chains) of
// every 128-byte line of its tile and a wave's 32 stores per item touch 64 lines each -- ~950 cycles of store issue apiece
// during which no wave of the workgroup multiplies.  Here the finished blocks of item n stay in registers WHILE ITEM n+1 IS
// MULTIPLIED and leave one store per chunk, in the shadow of that chunk's matrix instructions.  That needs the results twice
// (being stored + being accumulated), so a wave owns a 32 x 32 block of FOUR chains (64 + 64 registers) instead of 32 x 64:
//   workgroup = 8 waves = a 64 x 128 tile (2 row blocks x 4 column blocks), one workgroup per CU, 2 waves per SIMD;
//   chunk     = 64 contraction values (two 128-byte halves per row: 16 KB of q rows + 32 KB of k rows), three stages,
//               `buffer_load ... lds` in 1 KB pieces of 8 full lines, XOR-swizzled as in scorer_mfma.hip;
//               32 matrix instructions per wave and chunk on ONE accumulator (v_mfma_f32_32x32x2_f32: issue interval =
//               dependent latency = 64 cycles; the same fmaf chain as the other kernels: bit-identical scores), operands of
//               group g+1 read while group g multiplies;
//   chunk     = [wait for the wave's pieces of the chunk (the next one stays in flight), s_barrier, request chunk + 2, the
//               previous item's row(s) of this chunk, 32 matrix instructions].
//   ping-pong (SEMICRF_TILED_PINGPONG=1, off): waves 0-3 and 4-7 (one of each on every SIMD) one phase apart -- one group in
//               its memory phase (store + requests) while the other multiplies; three stages still suffice because group A
//               loads the first halves of all rows and requests one chunk ahead, group B the second halves, two ahead.
//               Correct (bit-identical) and SLOWER: 1.18 vs 1.11 ms at T=1024 x 352.  Cycle stamps (tools/tiled_probe.py) say
//               why: the memory phase is not instruction-bound but bound by the CU's ONE vector-memory path -- every LDS-DMA
//               piece (1 KB) and every scattered store costs its issuing wave ~100-400 cycles there, 56 of them per chunk --
//               so the memory phase of four waves (~3000 cycles) is longer than the other four's contraction (2048) and the
//               two phases add up to more than the in-step chunk.  64 x 128 tiles move 12 bytes of operands per matrix cycle
//               and CU where the L1 delivers ~15; the 128 x 128 tiles of scorer_mfma.hip need 8 but
//               cannot hold their results twice.
//   item      = one tile x four adjacent chains: chain by chain, 4 D / 64 chunks; the 16 result rows of the PREVIOUS item are
//               written during these chunks (16 / (4 D / 64) rows per chunk);
//   order     = the 32 workgroups of an XCD work side by side on the 8 chain quads of one 128-byte line group (its L2
//               assembles whole lines, as before) x 4 tiles that form a 2 x 2 block (rows 2a, 2a+1 x two column tiles): the
//               q rows and k rows each tile needs are fetched by two workgroups at the same time; line groups are the
//               outermost index so that a group's operands (64 MB at T=1024) stay in the memory-side cache across its tiles.
// The row constants of the merged projection (include/semicrf_hip.h: interval_score_fwd_pc) and the diagonal terms ride along
// with the operand requests into a ring of small LDS buffers, so the loop contains no register load at all and the vmcnt
// waits count exactly the LDS-DMA pieces (+ the stores, which have a whole chunk to retire).
#include "common.h"
#include "scorer_tiles.h"

#include <type_traits>

namespace semicrf {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int YTE = 64;                      // tile rows (end positions)
constexpr int YTB = 128;                     // tile columns (begin positions)
constexpr int YNS = 3;                       // LDS stages
constexpr int YQH = YTE * 128;               // bytes of one half (32 contraction values) of the q rows: 8 KB
constexpr int YKH = YTB * 128;               // ... of the k rows: 16 KB
constexpr int YSTAGE = 2 * YQH + 2 * YKH;    // [q half 0][q half 1][k half 0][k half 1] = 48 KB
constexpr int YRING = 4;                     // row-constant buffers: item n's is read while n+1 multiplies and n+2 is requested
constexpr int YRC = 2 * 4 * YTE * 4;         // [rowc | diag][chain][row] floats = 2 KB
constexpr int YLDS = YNS * YSTAGE + YRING * YRC;     // 152 KB
constexpr int YXCD = 8;
#ifndef SEMICRF_TILED_SPREAD
#define SEMICRF_TILED_SPREAD 0       // 1: a chunk's store and requests between its matrix instructions (in-step schedule), 0: in a block behind the barrier.
                                     // Measured (round 5): 1: 1.131-1.134 ms, 0: 1.115-1.119 at T=1024 x 352 (0.669 / 0.661 at T=691 x 360) -- what gains 1 % in
                                     // the backward GEMMs (SEMICRF_GEMM_SPREAD) loses 1.3 % here: the store's row constants come through the LDS
                                     // and its wait sits in the middle of the operand reads
#endif
#ifndef SEMICRF_TILED_PINGPONG
#define SEMICRF_TILED_PINGPONG 0     // 1: the two halves of the workgroup run one phase apart (measured slower, see the header)
#endif

__device__ __forceinline__ float y_len_scale(int len, int mode)
{
    if (mode == SEMICRF_LEN_LINEAR) return (float)len;
    if (mode == SEMICRF_LEN_SQRT) return sqrtf((float)len);
    return 1.0f;
}

__device__ __forceinline__ unsigned y_lds_addr(const void* p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}

template <int I, int N, class F>
__device__ __forceinline__ void y_static_for(F&& f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        y_static_for<I + 1, N>(f);
    }
}

struct TileGeom {
    int net, nbt, ntiles, ntg, nlg;          // row tiles (64), column tiles (128), tiles, tile groups of 4, line groups (8 quads)
};

}  // namespace

// NCH = D / 64 chunks per chain.  C = real chains, Cs = slots (the chain pitch of S), G: scorer_tiles.h.
template <int NCH>
__global__ __launch_bounds__(512, 2) void interval_score_tiled_kernel(
    const float* __restrict__ q, const float* __restrict__ k, const float* __restrict__ diag, const float* __restrict__ rowc, int C,
    int T, long long ldq, long long ldk, long long ldd, long long ldrc, float qscale, int mode, int full, float* __restrict__ S,
    int Cs, SlotGeom G, TileGeom TG)
{
    constexpr int NCI = 4 * NCH;                 // chunks of a full item
    extern __shared__ __attribute__((aligned(16))) char ylds[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int row = lane & 31, hl = lane >> 5;
    const int wer = wave >> 2, wc = wave & 3;        // this wave: rows 32 wer.., columns 32 wc.. of the tile
    const unsigned lds0 = y_lds_addr(ylds);
    const int xcd = blockIdx.x % YXCD, slot0 = blockIdx.x / YXCD, nslots = gridDim.x / YXCD;

    // entry u of this XCD's list: quad u & 7 of a line group, tile (u >> 3) & 3 of a group of four, super item xcd + 8 (u >> 5)
    auto item_of = [&](int u, int& et, int& bt, QuadInfo& qi) __attribute__((always_inline)) -> bool {
        const int s = xcd + YXCD * (u >> 5);
        if (s >= TG.nlg * TG.ntg) return false;
        const int lg = s / TG.ntg, tg = s - lg * TG.ntg;
        const int t = 4 * tg + ((u >> 3) & 3);
        qi = quad_info(G, 8 * lg + (u & 7));
        if (t >= TG.ntiles) { qi.nr = 0; qi.tz = 0; et = 0; bt = 0; return true; }     // padding
        if (full) {
            // pairs of rows: 2 nbt tiles each, column-major inside the pair
            const int a = t / (2 * TG.nbt), p = t - a * (2 * TG.nbt);
            if (2 * a + 1 >= TG.net) { et = 2 * a; bt = p; } else { et = 2 * a + (p & 1); bt = p >> 1; }
        } else {
            // rows 2a and 2a+1 hold a+1 column tiles each; a (a+1) tiles lie before the pair; column-major inside the pair, so
            // four consecutive tiles are a 2 x 2 block
            int a = (int)((sqrtf(4.0f * (float)t + 1.0f) - 1.0f) * 0.5f);
            while (a * (a + 1) > t) --a;
            while ((a + 1) * (a + 2) <= t) ++a;
            const int p = t - a * (a + 1);
            if (2 * a + 1 >= TG.net) { et = 2 * a; bt = p; } else { et = 2 * a + (p & 1); bt = p >> 1; }
        }
        return true;
    };

    // ---- LDS read offsets inside a stage: lane = (row, hl); segment 4 hl + m of the row's 128-byte half -----------------
    unsigned rdq[4], rdk[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const unsigned sw = (unsigned)((((4 * hl + m) ^ ((row >> 1) & 7)) * 16));
        rdq[m] = (unsigned)((32 * wer + row) * 128) + sw;
        rdk[m] = (unsigned)(2 * YQH + (32 * wc + row) * 128) + sw;
    }

    // ---- request side (identical in all waves): (entry, chain of the quad, chunk, stage) ------------------------------------
    int nx_u = slot0, nx_j = 0, nx_ch = 0, nx_stage = 0, nx_cnt = 0;
    QuadInfo nx_q4 = {0, 0, 0, 0};
    bool nx_valid = false;
    unsigned voq[2], vok[4], vorc = 0u, vodg = 0u;
    const float* nx_q = q;
    const float* nx_k = k;
    const float* nx_rc = rowc;
    const float* nx_dg = diag;
    auto set_chain = [&]() __attribute__((always_inline)) {
        const int c = nx_q4.ck + nx_j;                                       // nx_j < nx_q4.nr: a real chain
        nx_q = q + (size_t)c * T * ldq;
        nx_k = k + (size_t)c * T * ldk;
        nx_dg = diag + (size_t)c * T * ldd;
        if (rowc) nx_rc = rowc + (size_t)c * T * ldrc;
    };
    auto set_item = [&]() __attribute__((always_inline)) {
        int et = 0, bt = 0;
        // padding items (no real chain) request nothing: the consuming side skips them the same way
        while ((nx_valid = item_of(nx_u, et, bt, nx_q4)) && nx_q4.nr == 0) nx_u += nslots;
        if (nx_valid) {
            // loading lanes: a piece is 8 rows x 128 bytes.  q pieces 2 wave + j of 16 (half = piece >> 3), k pieces 4 wave + j
            // of 32 (half = piece >> 4); clamped rows are masked at the write
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int p = 2 * wave + j;
                const int lr = 8 * (p & 7) + (lane >> 3);
                const int seg = (lane & 7) ^ ((lr >> 1) & 7);
                const int er = et * YTE + lr < T ? et * YTE + lr : T - 1;
                voq[j] = (unsigned)(((size_t)er * ldq + (p >> 3) * 32 + seg * 4) * 4);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int p = 4 * wave + j;
                const int lr = 8 * (p & 15) + (lane >> 3);
                const int seg = (lane & 7) ^ ((lr >> 1) & 7);
                const int br = bt * YTB + lr < T ? bt * YTB + lr : T - 1;
                vok[j] = (unsigned)(((size_t)br * ldk + (p >> 4) * 32 + seg * 4) * 4);
            }
            const int er = et * YTE + lane < T ? et * YTE + lane : T - 1;
            vorc = (unsigned)((size_t)er * ldrc * 4);
            vodg = (unsigned)((size_t)er * ldd * 4);
            nx_j = 0;
            nx_ch = 0;
            set_chain();
        }
    };
    // A chunk's requests: req(0 .. 5) (the wave's two q pieces and four k pieces, from the CURRENT request state), req_tail() (the row
    // constants with a chain's first chunk; then the request state moves on).  issue_chunk() = all of them in a block;
    // SEMICRF_TILED_SPREAD issues them one at a time between the chunk's matrix instructions instead (see the loop).
    auto req = [&](int i) __attribute__((always_inline)) {
        char* st = ylds + nx_stage * YSTAGE;
        const int so = nx_ch * 256;                                          // 64 floats per chunk
        if (i < 2) {
            const auto rq = __builtin_amdgcn_make_buffer_rsrc((void*)nx_q, 0, 0x7fffffff, 0x00020000);
            const int p0 = 2 * wave;                                         // both pieces lie in the same half
            char* dq = st + (p0 >> 3) * YQH + (p0 & 7) * 1024 + i * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (lds_void_t*)dq, 16, voq[i], so, 0, 0);
        } else {
            const auto rk = __builtin_amdgcn_make_buffer_rsrc((void*)nx_k, 0, 0x7fffffff, 0x00020000);
            const int p1 = 4 * wave;
            char* dk = st + 2 * YQH + (p1 >> 4) * YKH + (p1 & 15) * 1024 + (i - 2) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rk, (lds_void_t*)dk, 16, vok[i - 2], so, 0, 0);
        }
    };
    auto req_tail = [&]() __attribute__((always_inline)) {
        // the chain's 64 row constants and diagonal terms, one dword per lane, AFTER the wave's operand pieces (every vmcnt wait
        // below stays at least as strict as without them): wave j for chain j, with the chain's first chunk
        if (nx_ch == 0 && wave == nx_j) {
            char* dr = ylds + YNS * YSTAGE + (nx_cnt & (YRING - 1)) * YRC + nx_j * (YTE * 4);
            const auto rd = __builtin_amdgcn_make_buffer_rsrc((void*)nx_dg, 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rd, (lds_void_t*)(dr + 4 * YTE * 4), 4, vodg, 0, 0, 0);
            if (rowc) {
                const auto rr = __builtin_amdgcn_make_buffer_rsrc((void*)nx_rc, 0, 0x7fffffff, 0x00020000);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rr, (lds_void_t*)dr, 4, vorc, 0, 0, 0);
            }
        }
        nx_stage = nx_stage + 1 == YNS ? 0 : nx_stage + 1;
        if (++nx_ch == NCH) {
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
    auto issue_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 6; ++i) req(i);
        req_tail();
    };

    set_item();
    int cur_u = slot0;
    {
        int e0_, b0_; QuadInfo q0_;
        if (!item_of(cur_u, e0_, b0_, q0_)) return;                         // uniform over the workgroup
    }

    f32x16 acc, prv[4];
    float cur[3][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc[r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 3; ++j) cur[j][r] = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) prv[j][r] = 0.0f;
    }
    // the item whose blocks wait in prv: tile, quad, whether this wave's block lies on / below the diagonal, its constants
    bool has_prev = false, p_on = false, p_dg = false, p_fast = false;
    int p_et = 0, p_bt = 0, p_cnt = 0, p_dlt = 0;
    float* p_ptr = S;
    // per lane, for the fast store: byte offset of the lane's cell in row 0 of a block, and its (row - column) inside the block
    const unsigned vo_lane = (unsigned)(((size_t)(4 * hl) * T + row) * Cs * 4);
    const int d0 = 4 * hl - row;
    const unsigned rowstride = (unsigned)((size_t)T * Cs * 4);
    QuadInfo p_qi = {0, 0, 0, 0};

    // row r (wave-uniform, a run-time index into the register blocks) of the previous item's block: scale, add the constants, one
    // 16-byte piece per cell (C/D layout of the 32 x 32 block: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5))
    auto store_row = [&](int r) __attribute__((always_inline)) {
        if (p_fast) {
            // the block lies strictly below the diagonal and inside T x T, whole 16-byte pieces, no ghost tail: no per-lane test,
            // one buffer store off the block's base (row offset in the scalar part)
            const int rr = (r & 3) + 8 * (r >> 2);
            const int len = p_dlt + rr + d0;                                 // e - b > 0
            const float sc = qscale * y_len_scale(len, mode);
            float v[4] = {prv[0][r], prv[1][r], prv[2][r], prv[3][r]};
            if (rowc) {
                const unsigned rcb = lds0 + (unsigned)(YNS * YSTAGE + (p_cnt & (YRING - 1)) * YRC) + (unsigned)((32 * wer + 4 * hl + rr) * 4);
                float cst[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned a = rcb + (unsigned)(i * YTE * 4);
                    asm volatile("ds_read_b32 %0, %1" : "=v"(cst[i]) : "v"(a));
                }
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cst[0]), "+v"(cst[1]), "+v"(cst[2]), "+v"(cst[3]));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = i < p_qi.nr ? (v[i] + cst[i]) * sc : 0.0f;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] *= sc;                      // (a chain the quad does not have: 0 * sc)
            }
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)p_ptr, 0, 0x7fffffff, 0x00020000);
            const u32x4 dv = {__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]), __builtin_bit_cast(unsigned, v[2]),
                              __builtin_bit_cast(unsigned, v[3])};
            __builtin_amdgcn_raw_buffer_store_b128(dv, rs, (int)vo_lane, (int)((unsigned)rr * rowstride), 0);
            return;
        }
        const bool vec = (Cs & 3) == 0 && p_qi.c4 + 3 < Cs;
        const int b = p_bt * YTB + 32 * wc + row;
        const unsigned rcb = lds0 + (unsigned)(YNS * YSTAGE + (p_cnt & (YRING - 1)) * YRC);
        const int lr = 32 * wer + (r & 3) + 8 * (r >> 2) + 4 * hl;           // row inside the tile
        const int e = p_et * YTE + lr;
        float cst[4] = {0.f, 0.f, 0.f, 0.f}, dgv[4] = {0.f, 0.f, 0.f, 0.f};
        if (rowc) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned a = rcb + (unsigned)((i * YTE + lr) * 4);
                asm volatile("ds_read_b32 %0, %1" : "=v"(cst[i]) : "v"(a));
            }
        }
        if (p_dg) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned a = rcb + (unsigned)(((4 + i) * YTE + lr) * 4);
                asm volatile("ds_read_b32 %0, %1" : "=v"(dgv[i]) : "v"(a));
            }
        }
        if (rowc || p_dg)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(cst[0]), "+v"(cst[1]), "+v"(cst[2]), "+v"(cst[3]), "+v"(dgv[0]), "+v"(dgv[1]),
                         "+v"(dgv[2]), "+v"(dgv[3]));
        const int len = e > b ? e - b : b - e;
        const float sc = qscale * y_len_scale(len, mode);
        float v[4] = {prv[0][r], prv[1][r], prv[2][r], prv[3][r]};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < p_qi.nr) {                                               // ghost slots of the quad keep their exact zeros
                v[i] = (v[i] + cst[i]) * sc;
                if (e == b) v[i] += dgv[i];
            }
        }
        if (e < T && b < T && (full || b <= e)) {
            float* dst = S + ((size_t)e * T + b) * Cs + p_qi.c4;
            if (vec) {
                *(float4*)dst = make_float4(v[0], v[1], v[2], v[3]);
                for (int z = 4; z <= p_qi.tz; z += 4) *(float4*)(dst + z) = make_float4(0.f, 0.f, 0.f, 0.f);   // the group's ghost tail
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < p_qi.nr) dst[i] = v[i];
            }
        }
    };
    // chunk ci of an item writes rows ci 16 / NCI .. (ci + 1) 16 / NCI - 1 of the previous one
    auto store_chunk = [&](int ci) __attribute__((always_inline)) {
        if (!has_prev || !p_on) return;
        for (int r = ci * 16 / NCI; r < (ci + 1) * 16 / NCI; ++r) store_row(r);
    };

    // prologue: group A requests one chunk ahead, group B two (see the header); B's first phase is only the wait for its pieces
    // of chunk 0
    const bool grpB = SEMICRF_TILED_PINGPONG ? wave >= 4 : true;
    int inflight = 0;                           // group B: chunks requested and not yet consumed
    if (nx_valid) { issue_chunk(); ++inflight; }
    if (grpB && nx_valid) { issue_chunk(); ++inflight; }
    int rd_stage = 0, cur_cnt = 0;
    if (SEMICRF_TILED_PINGPONG && grpB) {
        if (inflight >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }
#ifdef SEMICRF_TILED_PROBE
    // cycle accounting of one wave (tools/tiled_probe.py): 0 vmcnt wait, 1 stores, 2 requests, 3 barrier after the requests,
    // 4 contraction, 5 barrier after the contraction, 7 the rest
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long pt = __builtin_readcyclecounter();
#define TILED_PROBE(i) do { const unsigned long long n_ = __builtin_readcyclecounter(); pc[i] += n_ - pt; pt = n_; } while (0)
#else
#define TILED_PROBE(i) do { } while (0)
#endif

    while (true) {
        int et, bt;
        QuadInfo qi;
        (void)item_of(cur_u, et, bt, qi);
        // a 32 x 32 block that lies entirely above the diagonal is neither multiplied nor written
        const int erow = et * (YTE / 32) + wer, bcol = bt * (YTB / 32) + wc;
        const bool on = full || bcol <= erow;

        for (int j = 0; j < qi.nr; ++j) {
            for (int ch = 0; ch < NCH; ++ch) {
                const int ci = j * NCH + ch;
                // ---- phase S: this wave's memory instructions, while the SIMD's other wave multiplies ---------------------
                TILED_PROBE(7);
                // group A: its pieces of THIS chunk (requested one chunk ago; nothing younger is in flight) have landed
                if (!grpB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (!SEMICRF_TILED_PINGPONG) {
                    // in step: everybody waits for its pieces of this chunk (a younger request may stay in flight), one barrier
                    --inflight;
                    if (inflight >= 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __builtin_amdgcn_s_barrier();
                }
                TILED_PROBE(0);
                // SEMICRF_TILED_SPREAD (in-step schedule only): the store and the six requests do not go out in a block here -- where
                // all eight waves queue for the CU's one vector-memory path at the same time and nobody multiplies -- but one behind
                // each group of four matrix instructions, the two waves of a SIMD one group apart
                const bool spread = SEMICRF_TILED_SPREAD && !SEMICRF_TILED_PINGPONG && on;
                const bool doreq = nx_valid;
                if (!spread) store_chunk(ci);                               // the previous item's rows of this chunk
                TILED_PROBE(1);
                if (doreq) {
                    if (!spread) issue_chunk();                             // A: chunk + 1, B: chunk + 2
                    ++inflight;
                }
                TILED_PROBE(2);
                if (SEMICRF_TILED_PINGPONG) __builtin_amdgcn_s_barrier();
                TILED_PROBE(3);
                // ---- phase M: the chunk's 32 matrix instructions; everybody's pieces of it were waited for before the barrier
                // above (B's before the one in front of it) --------------------------------------------------------------
                const unsigned sb = lds0 + (unsigned)(rd_stage * YSTAGE);
                rd_stage = rd_stage + 1 == YNS ? 0 : rd_stage + 1;
                // eight groups of four contraction pairs (two halves x four segments); the operands of group g+1 are read
                // while group g multiplies (the registers an asm read returns must not be touched before the wait tied to them)
                v4f qa[2], ka[2];
                auto read_g = [&](auto gc, v4f& q_, v4f& k_) __attribute__((always_inline)) {
                    constexpr int g = decltype(gc)::value;
                    constexpr int h = g >> 2, m = g & 3;
                    const unsigned aq = sb + rdq[m], ak = sb + rdk[m];            // (asm operands alone do not capture)
                    asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%5"
                                 : "=&v"(q_), "=&v"(k_)
                                 : "v"(aq), "v"(ak), "n"(h * YQH), "n"(h * YKH));
                };
                auto wait_g = [&](v4f& q_, v4f& k_) __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(q_), "+v"(k_)); };
                auto mul_g = [&](const v4f& q_, const v4f& k_) __attribute__((always_inline)) {
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q_.x, k_.x, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q_.y, k_.y, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q_.z, k_.z, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(q_.w, k_.w, acc, 0, 0, 0);
                };
                if (on) {
                    read_g(std::integral_constant<int, 0>{}, qa[0], ka[0]);
                    wait_g(qa[0], ka[0]);
                    __builtin_amdgcn_sched_barrier(0);
                    y_static_for<0, 8>([&](auto gc) __attribute__((always_inline)) {
                        constexpr int g = decltype(gc)::value;
                        if constexpr (g < 7) {
                            read_g(std::integral_constant<int, g + 1>{}, qa[(g + 1) & 1], ka[(g + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        mul_g(qa[g & 1], ka[g & 1]);
                        __builtin_amdgcn_sched_barrier(0);
                        if (spread) {
                            // slot = g (waves 0-3) or g - 1 (waves 4-7): slot 0 the store, slots 1 .. 6 the requests
                            if (wer == 0) {
                                if constexpr (g == 0) store_chunk(ci);
                                if constexpr (g >= 1 && g <= 6) { if (doreq) req(g - 1); }
                            } else {
                                if constexpr (g == 1) store_chunk(ci);
                                if constexpr (g >= 2) { if (doreq) req(g - 2); }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        if constexpr (g < 7) {
                            wait_g(qa[(g + 1) & 1], ka[(g + 1) & 1]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    });
                    asm volatile("" : "+v"(acc));                            // (the matrix instructions stay in front of the barrier)
                    if (spread && doreq) req_tail();
                }
                TILED_PROBE(4);
                // group B: its pieces of the NEXT chunk (requested one chunk ago; the requests of this chunk's phase S may stay
                // in flight, its store has had the whole contraction to retire) have landed before group A reads them
                if (SEMICRF_TILED_PINGPONG) {
                    if (grpB) {
                        --inflight;
                        if (inflight >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_s_barrier();
                }
                TILED_PROBE(5);
            }
            // the chain's block waits in registers for the quad's other chains (the last one stays in acc)
            if (j == qi.nr - 1) {
            } else if (j == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { cur[0][r] = acc[r]; acc[r] = 0.0f; }
            } else if (j == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { cur[1][r] = acc[r]; acc[r] = 0.0f; }
            } else if (j == 2) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { cur[2][r] = acc[r]; acc[r] = 0.0f; }
            }
        }
        // rows of the previous item whose chunks did not run (a quad of fewer than four chains, a padding item)
        for (int ci = qi.nr * NCH; ci < NCI; ++ci) store_chunk(ci);
        // this item's blocks wait for the next item's chunks
        // (a quad of nr < 4 chains: chain nr-1's block is still in acc, the blocks behind it are zero)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float last = acc[r];
            prv[0][r] = qi.nr > 1 ? cur[0][r] : (qi.nr == 1 ? last : 0.0f);
            prv[1][r] = qi.nr > 2 ? cur[1][r] : (qi.nr == 2 ? last : 0.0f);
            prv[2][r] = qi.nr > 3 ? cur[2][r] : (qi.nr == 3 ? last : 0.0f);
            prv[3][r] = qi.nr == 4 ? last : 0.0f;
            acc[r] = 0.0f;
        }
        has_prev = qi.nr > 0;
        p_on = on; p_dg = erow == bcol; p_et = et; p_bt = bt; p_qi = qi; p_cnt = cur_cnt;
        {
            const int E0 = et * YTE + 32 * wer, B0 = bt * YTB + 32 * wc;
            p_dlt = E0 - B0;
            p_ptr = S + ((size_t)E0 * T + B0) * Cs + qi.c4;
            p_fast = on && (Cs & 3) == 0 && qi.c4 + 3 < Cs && qi.tz == 0 && E0 >= B0 + 32 && E0 + 32 <= T && B0 + 32 <= T;
        }
        if (qi.nr > 0) ++cur_cnt;
        cur_u += nslots;
        int e2, b2;
        QuadInfo q2;
        if (!item_of(cur_u, e2, b2, q2)) break;
    }
    if (SEMICRF_TILED_PINGPONG && !grpB) __builtin_amdgcn_s_barrier();     // (group B's last contraction phase)
    for (int ci = 0; ci < NCI; ++ci) store_chunk(ci);
#ifdef SEMICRF_TILED_PROBE
    if (lane == 0 && !full)
        for (int i = 0; i < 8; ++i) S[(size_t)Cs + (size_t)((blockIdx.x * 8 + wave) * 8 + i)] = (float)pc[i];
#endif
#undef TILED_PROBE
}

bool interval_score_tiled_supported(int C, int T, int D, const float* q, const float* k, long long ldq, long long ldk)
{
    const bool aligned = ((uintptr_t)q % 16 == 0) && ((uintptr_t)k % 16 == 0) && ldq % 4 == 0 && ldk % 4 == 0;
    return aligned && D % 64 == 0 && D <= 256 && T >= 128 && C >= 1 && (long long)T * ldq * 4 < (1ll << 31) &&
           (long long)T * ldk * 4 < (1ll << 31);      // (+ interval_score_tiled_fits for the slot count: 32-bit offsets inside a block of S)
}

template <int NCH>
static void launch_tiled(const float* q, const float* k, const float* diag, const float* rowc, int C, int T, long long ldq,
                         long long ldk, long long ldd, long long ldrc, float qscale, int mode, int full, float* S, int Cs,
                         const SlotGeom& G, const TileGeom& TG, int grid, hipStream_t stream)
{
    static PerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)interval_score_tiled_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, YLDS);
    hipLaunchKernelGGL(interval_score_tiled_kernel<NCH>, dim3(grid), dim3(512), YLDS, stream, q, k, diag, rowc, C, T, ldq, ldk, ldd, ldrc,
                       qscale, mode, full, S, Cs, G, TG);
}

// rowc may be NULL; group / pitch: the slot layout (scorer_tiles.h).  The caller checked interval_score_tiled_supported.
void launch_interval_score_tiled(const float* q, const float* k, const float* diag, const float* rowc, int C, int T, int D,
                                 long long ldq, long long ldk, long long ldd, long long ldrc, float qscale, int mode, int full,
                                 float* S, hipStream_t stream, int group, int pitch)
{
    const SlotGeom G = slot_geom(C, group, pitch);
    const int Cs = (C / group) * pitch;
    TileGeom TG;
    TG.net = (T + YTE - 1) / YTE;
    TG.nbt = (T + YTB - 1) / YTB;
    if (full) {
        TG.ntiles = TG.net * TG.nbt;
    } else {
        TG.ntiles = 0;
        for (int et = 0; et < TG.net; ++et) TG.ntiles += et / 2 + 1;           // row et: ceil(min((et+1) 64, T) / 128) column tiles
    }
    TG.ntg = (TG.ntiles + 3) / 4;
    TG.nlg = (G.nrq + 7) / 8;
    int ncu = 256, dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        ncu = v;
    // persistent workgroups, one per CU; per XCD a multiple of 32 slots (one super item at a time)
    int grid = ncu / (32 * YXCD) * (32 * YXCD);
    if (grid < 32 * YXCD) grid = 32 * YXCD;
    const long long supers = (long long)TG.nlg * TG.ntg;
    const long long need = (supers + YXCD - 1) / YXCD * 32 * YXCD;
    if (grid > need) grid = (int)need;
    switch (D / 64) {
    case 1: launch_tiled<1>(q, k, diag, rowc, C, T, ldq, ldk, ldd, ldrc, qscale, mode, full, S, Cs, G, TG, grid, stream); break;
    case 2: launch_tiled<2>(q, k, diag, rowc, C, T, ldq, ldk, ldd, ldrc, qscale, mode, full, S, Cs, G, TG, grid, stream); break;
    case 3: launch_tiled<3>(q, k, diag, rowc, C, T, ldq, ldk, ldd, ldrc, qscale, mode, full, S, Cs, G, TG, grid, stream); break;
    default: launch_tiled<4>(q, k, diag, rowc, C, T, ldq, ldk, ldd, ldrc, qscale, mode, full, S, Cs, G, TG, grid, stream); break;
    }
}

}  // namespace semicrf
