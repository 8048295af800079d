// scorer_bwd_gemm.hip -- backward of the interval scores as two batched triangular GEMMs on a repacked cotangent
// (the autograd of LayersTransformer.py:406-441 after the Linear map; same definition as scorer_bwd.hip).
//
//   G[e,b,c] = dS[e,b,c] * qscale * len(e-b) for e >= b, 0 otherwise
//   dq[c,e,:] = sum_b G[e,b,c] k[c,b,:]           dk[c,b,:] = sum_e G[e,b,c] q[c,e,:]
//
// scorer_bwd.hip multiplies straight out of the CRF's chain-minor gradient layout: a 32-row tile of 8 chains per
// workgroup, the k/q rows of every chain re-read by every row tile -- 16 B/clk/CU of operands at full matrix rate,
// which is what a CU's L1 can deliver at best (see scorer_mfma.hip), and the kernel ends up at a quarter of the fp32
// MFMA rate.  Here the cotangent is repacked ONCE into per-chain block-triangular matrices Gt[c][e][b] (scaled, zero above
// the diagonal inside the 128-aligned diagonal blocks; whole 128-byte lines in, 128-byte rows out), and both products become ordinary tiled GEMMs of one chain
// at a time: a persistent workgroup of 8 waves owns a 128 x D output tile (wave = 32 rows x D/2 columns), the
// operand chunks (32 contraction values: 16 KB of Gt + 32 rows of k or q) arrive by `buffer_load ... lds`, three
// stages deep, one s_barrier per chunk: 6 B/clk/CU.  dq reads Gt rows along the contraction (ds_read_b128, swizzled as
// in scorer_mfma.hip); dk needs the transpose, which is just a different walk of the same LDS stage (ds_read_b32 with
// consecutive lanes on consecutive b): no second copy.  The triangular structure shows up as the contraction range
// of an output tile (dq: b < 128 (i+1); dk: e >= 128 i); items are dealt longest first.
#include "common.h"
#include "bf16x3.h"

#include <type_traits>

namespace semicrf {

typedef __attribute__((address_space(3))) void lds_void_t;

constexpr int GM = 128;            // rows of an output tile
constexpr int GK = 32;             // contraction values per chunk
constexpr int GNS = 3;             // LDS stages
constexpr int GA_BYTES = GM * GK * 4;      // 16 KB: the Gt part of a stage

__device__ __forceinline__ float len_scale_pack(int len, int mode)
{
    if (mode == SEMICRF_LEN_LINEAR) return (float)len;
    if (mode == SEMICRF_LEN_SQRT) return sqrtf((float)len);
    return 1.0f;
}

__device__ __forceinline__ unsigned g_lds_addr(const void* p)
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

// ---------------------------------------------------------------------------------------------
// Gt layout: per chain a block-triangular matrix.  Rows come in blocks of 128 (GM); the rows of block i hold columns
// 0 .. RL(i)-1 with RL(i) = min(128 (i+1), Tp) -- everything an output tile ever reads (the diagonal block included,
// zero above the diagonal) and nothing else: 56 % of the square at T=1024.
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ int gt_row_len(int blk, int Tp) { return (blk + 1) * GM < Tp ? (blk + 1) * GM : Tp; }
// floats before block blk (blocks before it are full: 128 rows of 128 (i+1) columns)
__host__ __device__ __forceinline__ size_t gt_block_off(int blk) { return (size_t)GM * GM * ((size_t)blk * (blk + 1) / 2); }
__host__ __device__ __forceinline__ size_t gt_chain_floats(int Tp)
{
    const int nb = (Tp + GM - 1) / GM;
    return gt_block_off(nb - 1) + (size_t)(Tp - (nb - 1) * GM) * Tp;
}

// ---------------------------------------------------------------------------------------------
// repack: dS [T][T][C] -> Gt [C][block-triangular] (Tp = T rounded up to 32), scaled; zero for b > e and for e >= T
// ---------------------------------------------------------------------------------------------
// One block = 8 end frames x 32 begin frames x 32 chains (whole 128-byte lines in; 128-byte rows out).  Only the
// b tiles up to the end of the row's 128-aligned diagonal block are written: the GEMMs read nothing beyond.
constexpr int PK_CH = 32, PK_E = 8, PK_B = 32;
#ifndef SEMICRF_PACK_ORDER
#define SEMICRF_PACK_ORDER 1
#endif
constexpr int PK_ROW = PK_B + 4;                   // LDS row pitch (floats): 16-byte aligned rows
constexpr int PK_CHS = PK_E * PK_ROW + 4;          // chain pitch

// FUSED: the cotangent is built on the fly from the CRF's own quantities (the dense [T][T][C]
// gradient of ComputeLogZFasterGrad.backward is never written):  dS[e,b,c] = gout[c] * marginal[e,b,c],
//   marginal = exp(alpha[b,c] + beta[e,c] + S[e,b,c] - logZ[c])                         (e > b)
//            = exp(alpha[t,c] + beta[t,c] + S[t,t,c] - 2 softplus(S[t,t,c]) - logZ[c])  (e == b == t)
// (NeuralSemiCRFInterval.py:424-447); `dS` then points at S itself.
struct PackFused {
    const float* alpha;   // [T][C] by frame
    const float* beta;    // [T][C] by frame
    const float* logZ;    // [C]
    const float* gout;    // [C]
};

// C here is the number of SLOTS (the chain pitch of dS, alpha, beta, logZ, gout); a slot's matrix goes to its chain's slab,
// ghost slots are skipped (SL: common.h).
template <bool FUSED>
__global__ __launch_bounds__(256) void score_bwd_pack_kernel(const float* __restrict__ dS, float* __restrict__ Gt, int C,
                                                             int T, int Tp, float qscale, int mode, PackFused F, ChainSlots SL)
{
    __shared__ __attribute__((aligned(16))) float L[PK_CH * PK_CHS];
    // (chain groups fastest: the blocks that run side by side read neighbouring 128-byte pieces of the SAME cells -- whole runs of the
    // chain axis -- instead of one piece each of cells 1.4 KB apart)
#if SEMICRF_PACK_ORDER == 1
    const int cg = blockIdx.x * PK_CH, b0 = blockIdx.y * PK_B, e0 = blockIdx.z * PK_E;
#else
    const int b0 = blockIdx.x * PK_B, e0 = blockIdx.y * PK_E, cg = blockIdx.z * PK_CH;
#endif
    if (b0 >= (e0 / GM + 1) * GM) return;
    const int tid = threadIdx.x;
    const bool vec = (C % 4 == 0) && (((uintptr_t)dS & 15) == 0) &&
                     (!FUSED || ((((uintptr_t)F.alpha | (uintptr_t)F.beta) & 15) == 0));
    // this thread's four chains are the same in every iteration (the quad depends on tid only)
    float lz[4] = {0.f, 0.f, 0.f, 0.f}, gz[4] = {0.f, 0.f, 0.f, 0.f};
    if (FUSED) {
        const int cq = cg + (tid & 7) * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (cq + i < C) { lz[i] = F.logZ[cq + i]; gz[i] = F.gout[cq + i]; }
    }
#pragma unroll
    for (int it = 0; it < (PK_E * PK_B * PK_CH / 4) / 256; ++it) {
        const int idx = tid + it * 256;
        const int quad = idx & 7, cell = idx >> 3;
        const int bl = cell & 31, el = cell >> 5;
        const int e = e0 + el, b = b0 + bl, c4 = cg + quad * 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < T && b <= e && c4 < C) {
            const float* src = dS + ((size_t)e * T + b) * C + c4;
            if (vec) {
                v = *(const float4*)src;
            } else {
                v.x = src[0];
                if (c4 + 1 < C) v.y = src[1];
                if (c4 + 2 < C) v.z = src[2];
                if (c4 + 3 < C) v.w = src[3];
            }
            if (FUSED) {
                float al[4] = {0.f, 0.f, 0.f, 0.f}, be[4] = {0.f, 0.f, 0.f, 0.f};
                const float* ap = F.alpha + (size_t)b * C + c4;
                const float* bp = F.beta + (size_t)e * C + c4;
                if (vec) {
                    const float4 a4 = *(const float4*)ap, b4 = *(const float4*)bp;
                    al[0] = a4.x; al[1] = a4.y; al[2] = a4.z; al[3] = a4.w;
                    be[0] = b4.x; be[1] = b4.y; be[2] = b4.z; be[3] = b4.w;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (c4 + i < C) { al[i] = ap[i]; be[i] = bp[i]; }
                }
                float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float x = al[i] + be[i] + vv[i] - lz[i];
                    if (e == b) x -= 2.0f * softplus_f(vv[i]);
                    vv[i] = gz[i] * __expf(x);
                }
                v = make_float4(vv[0], vv[1], vv[2], vv[3]);
            }
            const float sc = qscale * len_scale_pack(e - b, mode);
            v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc;
        }
        float* dst = L + (quad * 4) * PK_CHS + el * PK_ROW + bl;
        dst[0] = v.x; dst[PK_CHS] = v.y; dst[2 * PK_CHS] = v.z; dst[3 * PK_CHS] = v.w;
    }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < (PK_E * PK_B * PK_CH / 4) / 256; ++it) {
        const int idx = tid + it * 256;
        const int j4 = idx & 7, rowid = idx >> 3;
        const int el = rowid & 7, ch = rowid >> 3;
        const int c = cg + ch < C ? chain_of_slot(SL, cg + ch) : -1;
        if (c >= 0) {
            const float4 v = *(const float4*)(L + ch * PK_CHS + el * PK_ROW + 4 * j4);
            const int blk = e0 / GM;                          // the 8 rows of a block lie in one 128-row block
            float* row = Gt + (size_t)c * gt_chain_floats(Tp) + gt_block_off(blk) + (size_t)(e0 + el - blk * GM) * gt_row_len(blk, Tp);
            *(float4*)(row + b0 + 4 * j4) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// GEMM: out[c][m][:] = sum_k A[m][k] other[c][k][:]   with A = Gt (dq: m = e, k = b) or Gt^T (dk: m = b, k = e)
// ---------------------------------------------------------------------------------------------
// AT: A is read transposed (dk).  NW: 32-column blocks per wave (D = 64 NW; NW in {1, 2, 4}).
// rsum (!AT only, may be NULL): rsum[(c T + m) ldrs] = sum_k A[m][k] -- the gradient of the merged projection's row constant
// (include/semicrf_hip.h: interval_score_bwd_ws_pc) falls out of the A values this kernel reads anyway: every lane adds up the
// contraction values of its row that pass through it (a fixed order: chunk, group, component), the two half-waves are combined at
// the end of the item.
#ifndef SEMICRF_GEMM_SPREAD
#define SEMICRF_GEMM_SPREAD 0     // 1: a chunk's LDS-DMA requests between its matrix instructions, the two waves of a SIMD at different places; 0: in a block
                                  // behind the barrier.  Measured (round 5, T=1024 x 352, plateau): with run-time tests around the requests 2.117-2.121 ms
                                  // against 2.143-2.147 -- but the compiler then lays the groups' blocks out of order (LDS reads textually behind their
                                  // waits: tools/check_asm_waits.py cannot vouch for that); as straight-line bodies per wave group 2.19 against 2.125: off
#endif
template <bool AT, int NW>
__global__ __launch_bounds__(512, 2) void score_bwd_gemm_kernel(const float* __restrict__ Gt, int Tp,
                                                                const float* __restrict__ other, long long ldo,
                                                                float* __restrict__ out, long long ldout, int C, int T,
                                                                float* __restrict__ rsum, long long ldrs)
{
    constexpr int D = 64 * NW;
    constexpr int GB_BYTES = GK * D * 4;               // the k/q part of a stage: 32 rows
    constexpr int GSTAGE = GA_BYTES + GB_BYTES;
    constexpr int RPP = 4 / NW;                        // k/q rows per 1 KB piece
    constexpr int NLOAD = 2 + NW;                      // pieces per wave and chunk
    extern __shared__ __attribute__((aligned(16))) char glds[];    // [GNS][GSTAGE]
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;           // this wave: rows 32*wm.., columns 32*NW*wn.. of the tile
    const unsigned lds0 = g_lds_addr(glds);
    const int nm = (T + GM - 1) / GM;                  // output tiles per chain
    const long long nitems = (long long)nm * C;
    const int nkt = Tp / GK;                           // chunks of the whole contraction axis

    // item n: tiles dealt longest first (dq: the last row tile has the longest contraction range, dk: the first)
    auto item_of = [&](long long n, int& c, int& mi, int& kbeg, int& nk) -> bool {
        if (n >= nitems) return false;
        const int r = (int)(n / C);
        c = (int)(n % C);
        mi = AT ? r : nm - 1 - r;
        if (AT) {
            kbeg = mi * (GM / GK);
            nk = nkt - kbeg;
        } else {
            kbeg = 0;
            nk = (mi + 1) * (GM / GK) < nkt ? (mi + 1) * (GM / GK) : nkt;
        }
        return true;
    };

    // ---- LDS read addresses (bytes within a stage) ----------------------------------------------------------
    // contraction order inside a chunk: matrix instruction (mm, comp) takes k = 4 mm + comp from lanes 0-31 and
    // k = 16 + 4 mm + comp from lanes 32-63 -- the same permutation for both operands
    unsigned rdA[4];                                   // !AT: one 16-byte segment per mm (row = m, swizzled)
#pragma unroll
    for (int mm = 0; mm < 4; ++mm) {
        const int row = 32 * wm + l31;
        rdA[mm] = (unsigned)(row * 128 + (((4 * half + mm) ^ ((row >> 1) & 7)) * 16));
    }
    const unsigned rdAT = (unsigned)(half * 16 * (GM * 4) + (32 * wm + l31) * 4);        // AT: stage rows = k, 512 bytes each
    const unsigned rdB = (unsigned)(GA_BYTES + half * 16 * (D * 4) + (32 * NW * wn + l31) * 4);

    // ---- request side (identical in all waves) --------------------------------------------------------------
    long long nx_n = blockIdx.x;
    int nx_c = 0, nx_mi = 0, nx_kbeg = 0, nx_nk = 0, nx_j = 0, nx_stage = 0;
    bool nx_valid = item_of(nx_n, nx_c, nx_mi, nx_kbeg, nx_nk);
    if (!nx_valid) return;                             // uniform
    // per-lane byte offsets of this wave's pieces (relative to the chain slab, without the chunk)
    unsigned voA[2], voB[4];                           // (dependent array bounds captured by lambdas trip the host compiler)
    auto set_offsets = [&]() {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = 2 * wave + j;                // Gt piece 0..15
            if (!AT) {
                const int row = 8 * p + (lane >> 3);   // m within the tile; 128 contiguous bytes along k
                const int seg = (lane & 7) ^ ((row >> 1) & 7);
                voA[j] = (unsigned)((gt_block_off(nx_mi) + (size_t)row * gt_row_len(nx_mi, Tp) + seg * 4) * 4);
            } else {
                const int row = 2 * p + (lane >> 5);   // k within the chunk; 512 contiguous bytes along m
                voA[j] = (unsigned)row;                  // the row length depends on the chunk: finished in issue_chunk
            }
        }
#pragma unroll
        for (int j = 0; j < NW; ++j) {
            const int p = NW * wave + j;               // k/q piece 0 .. 8 NW - 1
            const int row = p * RPP + lane / (16 * NW);
            voB[j] = (unsigned)(((size_t)row * ldo + (lane % (16 * NW)) * 4) * 4);
        }
    };
    set_offsets();
    // A chunk's requests = prep (scalars; the walk over the items moves on) + req(0 .. NLOAD-1).  SEMICRF_GEMM_SPREAD: the requests
    // are not issued in a block behind the barrier -- where all eight waves queue for the CU's one vector-memory path at the same
    // time and no wave multiplies -- but one at a time between the chunk's matrix instructions, the two waves of a SIMD at different places.
    // (the pending request as plain locals, the resource words made at the request: through a struct handed to the lambdas the HOST
    // pass drops the kernel's instantiation without a diagnostic and the library fails to load with an undefined kernel symbol)
    const float* rq_pa = Gt; const float* rq_pb = other; int rq_na = 0, rq_nb = 0; unsigned rq_da = 0, rq_db = 0, rq_sa = 0, rq_va0 = 0, rq_va1 = 0, rq_kb = 0;
    auto prep = [&]() __attribute__((always_inline)) {
        const int k0 = (nx_kbeg + nx_j) * GK;
        // bounds-checked buffers over the chain's slab: rows past the end read as zero
        const size_t slab = gt_chain_floats(Tp);
        rq_pa = Gt + (size_t)nx_c * slab; rq_na = (int)(slab * 4);
        rq_pb = other + (size_t)nx_c * T * ldo; rq_nb = (int)((size_t)T * ldo * 4);
        rq_da = (unsigned)(nx_stage * GSTAGE + (2 * wave) * 1024);
        rq_db = (unsigned)(nx_stage * GSTAGE + GA_BYTES + (NW * wave) * 1024);
        // AT: the chunk's 32 rows (k = e) lie in one 128-row block kblk; its row length enters the per-lane offsets
        const int kblk = k0 / GM, krl = gt_row_len(kblk, Tp);
        rq_sa = AT ? (unsigned)((gt_block_off(kblk) + (size_t)(k0 - kblk * GM) * krl) * 4) : (unsigned)(k0 * 4);   // inside the slab
        rq_va0 = AT ? (unsigned)((voA[0] * krl + nx_mi * GM + (lane & 31) * 4) * 4) : voA[0];
        rq_va1 = AT ? (unsigned)((voA[1] * krl + nx_mi * GM + (lane & 31) * 4) * 4) : voA[1];
        // the k/q rows of the last chunk may lie past T: their offset goes into the per-lane part, which is what the
        // buffer's range check looks at (the scalar offset is not checked); such rows keep the stage's old contents
        // and meet Gt == 0 (the stages are cleared once at the start so that they never hold a NaN pattern)
        rq_kb = (unsigned)((size_t)k0 * ldo * 4);
        nx_stage = nx_stage + 1 == GNS ? 0 : nx_stage + 1;
        if (++nx_j == nx_nk) {
            nx_j = 0;
            nx_n += gridDim.x;
            nx_valid = item_of(nx_n, nx_c, nx_mi, nx_kbeg, nx_nk);
            if (nx_valid) set_offsets();
        }
    };
    auto req = [&](int i) __attribute__((always_inline)) {
        // (the destination through a named char*: with the cast applied to `glds + ...` directly the HOST pass drops the kernel's
        // instantiation without a diagnostic and the library fails to load with an undefined kernel symbol)
        if (i < 2) {
            const auto ra = __builtin_amdgcn_make_buffer_rsrc((void*)rq_pa, 0, rq_na, 0x00020000);
            char* dst = glds + rq_da + i * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lds_void_t*)dst, 16, i == 0 ? rq_va0 : rq_va1, rq_sa, 0, 0);
        } else {
            const auto rb = __builtin_amdgcn_make_buffer_rsrc((void*)rq_pb, 0, rq_nb, 0x00020000);
            char* dst = glds + rq_db + (i - 2) * 1024;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lds_void_t*)dst, 16, voB[i - 2] + rq_kb, 0, 0, 0);
        }
    };
    auto issue_chunk = [&]() __attribute__((always_inline)) {
        prep();
#pragma unroll
        for (int i = 0; i < NLOAD; ++i) req(i);
    };

    // clear the stages once (see issue_chunk)
    for (int i = threadIdx.x; i < GNS * GSTAGE / 16; i += 512) ((float4*)glds)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    __syncthreads();

    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < NW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    int inflight = 0;
#pragma unroll
    for (int i = 0; i < GNS - 1; ++i)
        if (nx_valid) { issue_chunk(); ++inflight; }
    int rd_stage = 0;
    float rs = 0.0f;                                    // !AT && rsum: this lane's part of its row's sum
    const bool want_rs = !AT && rsum != nullptr;

    long long cur_n = blockIdx.x;
    while (true) {
        int c, mi, kbeg, nk;
        if (!item_of(cur_n, c, mi, kbeg, nk)) break;
        for (int j = 0; j < nk; ++j) {
            // this wave's pieces of the current chunk have landed (a younger request may stay in flight) ...
            if (inflight >= 2) {
                if (NLOAD == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else if (NLOAD == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            // ... and so have everybody's; everybody is also done reading the previous chunk
            __builtin_amdgcn_s_barrier();
            --inflight;
            const bool doreq = nx_valid;
            if (doreq) {
                if (SEMICRF_GEMM_SPREAD) prep(); else issue_chunk();
                ++inflight;
            }
            const unsigned sb = lds0 + (unsigned)(rd_stage * GSTAGE);
            rd_stage = rd_stage + 1 == GNS ? 0 : rd_stage + 1;
            // four groups of four contraction pairs; the operands of group g+1 are read while group g multiplies
            // (the registers an asm read returns must not be touched before the wait that is tied to them: no copies).
            // One straight-line body per (wave group, requests or not): with run-time tests around the requests the compiler lays the
            // groups' blocks out of order (reads textually behind their waits: tools/check_asm_waits.py cannot follow that).
            auto body = [&](auto WGC, auto RQC) __attribute__((always_inline)) {
            constexpr int WG = decltype(WGC)::value;
            constexpr bool RQ = decltype(RQC)::value;
            v4f a4[2];                         // !AT: four consecutive contraction values of the row
            float a1[2][4];                    // AT: one value per read
            float bq[2][4][4];
            auto read_group = [&](auto mmc, v4f& av4, float (&av)[4], float (&bv)[4][4]) {
                constexpr int mm = decltype(mmc)::value;
                if (!AT) {
                    const unsigned addr = sb + rdA[mm];
                    asm volatile("ds_read_b128 %0, %1" : "=v"(av4) : "v"(addr));
                } else {
                    static_for<0, 4>([&](auto cc) {
                        constexpr int comp = decltype(cc)::value;
                        float& dst = av[comp];                       // (asm operands alone do not capture)
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
            auto wait_group = [&](v4f& av4, float (&av)[4], float (&bv)[4][4]) {
                if (!AT) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av4));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]));
#pragma unroll
                for (int comp = 0; comp < 4; ++comp)
#pragma unroll
                    for (int t = 0; t < NW; ++t) asm volatile("" : "+v"(bv[comp][t]));
            };
            // (grp: the chunk's group of 16 NW / 4 instructions; with SEMICRF_GEMM_SPREAD two of the wave's requests go behind the
            // instructions of groups 0 .. NLOAD/2 - 1: waves 0-3 behind the 1st and 3rd quarter, waves 4-7 behind the 2nd and 4th)
            auto mul_group = [&](auto gc, const v4f& av4, const float (&av)[4], const float (&bv)[4][4]) __attribute__((always_inline)) {
                constexpr int grp = decltype(gc)::value;
                static_for<0, 4>([&](auto cc) __attribute__((always_inline)) {
                    constexpr int comp = decltype(cc)::value;
#pragma unroll
                    for (int t = 0; t < NW; ++t)
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(AT ? av[comp] : av4[comp], bv[comp][t], acc[t], 0, 0, 0);
                    if constexpr (SEMICRF_GEMM_SPREAD != 0) {
                        constexpr int r0 = 2 * grp + (comp >> 1);            // request behind quarter comp: 2 grp (comp 0, 1) or 2 grp + 1 (comp 2, 3)
                        if constexpr (r0 < NLOAD) {
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (RQ && WG == (comp & 1)) req(r0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                });
                if (!AT && want_rs) rs += (av4[0] + av4[1]) + (av4[2] + av4[3]);
            };
            read_group(std::integral_constant<int, 0>{}, a4[0], a1[0], bq[0]);
            wait_group(a4[0], a1[0], bq[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_group(std::integral_constant<int, 1>{}, a4[1], a1[1], bq[1]);
            __builtin_amdgcn_sched_barrier(0);
            mul_group(std::integral_constant<int, 0>{}, a4[0], a1[0], bq[0]);
            __builtin_amdgcn_sched_barrier(0);
            wait_group(a4[1], a1[1], bq[1]);
            __builtin_amdgcn_sched_barrier(0);
            read_group(std::integral_constant<int, 2>{}, a4[0], a1[0], bq[0]);
            __builtin_amdgcn_sched_barrier(0);
            mul_group(std::integral_constant<int, 1>{}, a4[1], a1[1], bq[1]);
            __builtin_amdgcn_sched_barrier(0);
            wait_group(a4[0], a1[0], bq[0]);
            __builtin_amdgcn_sched_barrier(0);
            read_group(std::integral_constant<int, 3>{}, a4[1], a1[1], bq[1]);
            __builtin_amdgcn_sched_barrier(0);
            mul_group(std::integral_constant<int, 2>{}, a4[0], a1[0], bq[0]);
            __builtin_amdgcn_sched_barrier(0);
            wait_group(a4[1], a1[1], bq[1]);
            __builtin_amdgcn_sched_barrier(0);
            mul_group(std::integral_constant<int, 3>{}, a4[1], a1[1], bq[1]);
            __builtin_amdgcn_sched_barrier(0);
            };
            if (!SEMICRF_GEMM_SPREAD || !doreq) body(std::integral_constant<int, 0>{}, std::false_type{});
            else if ((wave >> 2) == 0) body(std::integral_constant<int, 0>{}, std::true_type{});
            else body(std::integral_constant<int, 1>{}, std::true_type{});
        }
        // ---- the item's 128 x D block (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)) ----
        {
            float* ob = out + (size_t)c * T * ldout;
#pragma unroll
            for (int t = 0; t < NW; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mi * GM + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < T) ob[(size_t)m * ldout + 32 * NW * wn + 32 * t + l31] = acc[t][r];
                    acc[t][r] = 0.0f;
                }
            if (!AT && want_rs) {
                const float tot = rs + __shfl_xor(rs, 32);
                const int m = mi * GM + 32 * wm + l31;
                if (wn == 0 && half == 0 && m < T) rsum[((size_t)c * T + m) * ldrs] = tot;
                rs = 0.0f;
            }
        }
        cur_n += gridDim.x;
    }
}

// ---------------------------------------------------------------------------------------------
// The same two products on the bf16 matrix instructions, every operand as three exact bf16 limbs (bf16x3.h; opt-in:
// length_scaling | SEMICRF_LEN_BF16X3).  Six instructions of 8 passes replace eight fp32 instructions of 16 per 16
// contraction values: 2.67x less matrix time, fp32-grade results (not bit-identical to the exact-fp32 kernel above).
// ---------------------------------------------------------------------------------------------
// Same items, same output tile (128 x D, wave = 32 rows x D/2 columns), same epilogue.  What differs is the operand path.
// v_mfma_f32_32x32x16_bf16 wants EIGHT consecutive contraction values of one row (A) / one column (B) per lane, and one of the
// two operands of either product has the contraction index as its ROW index in memory (k[b][:] for dq; Gt[e][:] and q[e][:] for
// dk).  So nothing goes from memory to LDS directly: every lane fetches "units" of eight contraction values into registers two
// chunks ahead -- a unit along a memory row is two 16-byte loads, a unit across rows is eight 4-byte loads of one column
// (consecutive lanes on consecutive columns: 256-byte runs; the row is wave-uniform and sits in the scalar offset of a buffer
// load, so a unit costs no address arithmetic in the vector unit) -- splits each value once (~5.5 vector instructions) and stores
// the three limbs as 16-byte pieces in exactly the layout the matrix instruction reads: [limb][row][4 pieces of 8 values],
// pieces swizzled by row as in the forward's three-limb kernel (reads conflict-free; the column units' writes two-way).
// Two LDS stages of 3 x (8 KB + D x 64 bytes) (144 KB at D = 256).
//
// Schedule.  Timing ablations of the first version (every wave: barrier, multiply, split the next chunk, request; T=1024 x 352,
// both products): matrix instructions alone 0.51 ms, everything 1.32 -- the parts ADD, because a barrier per chunk keeps the two
// waves of a SIMD in the same phase.  So the waves are two GROUPS (0-3 and 4-7: one of each per SIMD) half a chunk apart: while
// one group multiplies chunk n, the other splits its share of chunk n+1 and requests chunk n+3, then they swap; two barriers
// per chunk.  A multiplying wave has the SIMD's matrix pipe to itself, so its operand reads are issued one group of six
// instructions ahead (two sets of operand registers), the order pinned by scheduling barriers.
// That form is still here (SEMICRF_G3_INTERLEAVE = 0).  The DEFAULT is the interleaved form below (multiply_fill): every wave
// multiplies the chunk in one stage and, between its own matrix instructions, splits the next chunk into the other stage (a third
// of a piece behind every instruction) and requests the one after it; one barrier per chunk.  1.53 -> 1.45 ms at T=1024 x 352 at
// the clock's plateau (same box: two groups 1.528, interleaved with the requests behind the chunk 1.549, whole pieces 1.493, this 1.453).
// What the kernel is bound by, measured: DESIGN.md section 3 ("The last schedule, and what bounds it") and profiles/r05_mfma_peak.txt.
#ifndef SEMICRF_G3_PROBE
#define SEMICRF_G3_PROBE 0        // 1: cycle accounting of every wave instead of results (tools/bwd3_probe.py --probe): [0] multiply, [1] epilogue,
#endif                            // [2] first barrier, [3] wait for the set's loads, [4] split + limb stores, [5] requests, [6] second barrier
constexpr int G3_NS = 2;
constexpr int NXCD_G3 = 8;        // workgroup b runs on XCD b % 8
#ifndef SEMICRF_G3_DBG
#define SEMICRF_G3_DBG 0          // timing ablations (variant builds only; results are wrong): 1 no matrix instructions, 2 no requests,
#endif                            // 4 no split / limb stores, 8 no operand reads
#ifndef SEMICRF_G3_MMA_ORDER
#define SEMICRF_G3_MMA_ORDER 1    // 1: the limb products of a slab round-robin over the accumulators (independent neighbours); 0: six in a row per accumulator, operand reads one group ahead (within 2 % of each other, like the phase shift)
#endif
#ifndef SEMICRF_MMA_PRIO
#define SEMICRF_MMA_PRIO 0       // 1: raised wave priority while a wave issues its chunk's matrix instructions
#endif
#ifndef SEMICRF_G3_DRAIN
#define SEMICRF_G3_DRAIN 1        // 1: vmcnt(0) behind an item's stores.  0: none -- the hand-written vmcnt(NLD) waits stay SAFE with stores in flight (loads return
#endif                            // in order among themselves: at most NLD operations outstanding means at most NLD loads outstanding, i.e. only the younger set's)
                                  // -- measured the same (1.434-1.439 / 1.439-1.443 ms): the next wait for a register set drains the stores anyway
#ifndef SEMICRF_G3_FINE
#define SEMICRF_G3_FINE 1         // (interleaved form, NW = 4) a THIRD of a piece behind every matrix instruction instead of a piece behind every third
#endif
#ifndef SEMICRF_G3_LEAD
#define SEMICRF_G3_LEAD 0         // (with SEMICRF_G3_FINE) this many thirds in FRONT of a chunk's first matrix instruction, in the shadow of its first LDS
                                  // reads (measured: 3 the same, 6 one per cent slower)
#endif
#ifndef SEMICRF_G3_ABL
#define SEMICRF_G3_ABL 0          // timing ablations of the interleaved form (variant builds only; results are wrong): 1 no split pieces,
#endif                            // 2 the operands of a chunk's first instruction group for all of it (no LDS reads inside), 4 no requests
#ifndef SEMICRF_G3_PRIO_FLIP
#define SEMICRF_G3_PRIO_FLIP 0     // (interleaved form) the younger wave of a SIMD goes first for this many matrix instructions of a chunk (0: never).
                                   // 24 evens out the two waves' arrival at the barrier (cycle stamps) and measures 0.8 % SLOWER at the clock's plateau
#endif
#ifndef SEMICRF_G3_FILL_LOADS
#define SEMICRF_G3_FILL_LOADS 1    // (interleaved form) the requests for the chunk after next between the matrix instructions, too
#endif
#ifndef SEMICRF_G3_INTERLEAVE
#define SEMICRF_G3_INTERLEAVE 1   // 1: every wave splits the next chunk between its own matrix instructions, one barrier per chunk; 0: two wave groups
#endif
#ifndef SEMICRF_G3_PHASED
#define SEMICRF_G3_PHASED 1       // 0: both groups in the same phase (the first version's schedule)
#endif

template <bool AT, int NW>
__global__ __launch_bounds__(512, 2) void score_bwd_gemm3_kernel(const float* __restrict__ Gt, int Tp,
                                                                 const float* __restrict__ other, long long ldo,
                                                                 float* __restrict__ out, long long ldout, int C, int T,
                                                                 float* __restrict__ rsum, long long ldrs)
{
    constexpr int D = 64 * NW;
    constexpr int APL = GM * 64;                       // bytes of one limb plane of the Gt part: 128 rows x 32 values x 2 bytes
    constexpr int BPL = D * 64;                        // ... of the k/q part: D rows (columns of k/q)
    constexpr int BOFF = 3 * APL;
    constexpr int STAGE = 3 * (APL + BPL);
    constexpr int WPP = D / 128 > 0 ? D / 128 : 1;     // waves per piece (8 rows) of the k/q part: 64 column pairs per wave
    extern __shared__ __attribute__((aligned(16))) char glds[];    // [G3_NS][STAGE]: Gt limbs h | m | l | k/q limbs h | m | l
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;           // this wave: rows 32*wm.., columns 32*NW*wn.. of the tile
    const int grp = SEMICRF_G3_PHASED ? wave >> 2 : 0; // (waves w and w + 4 share a SIMD)
    const int nm = (T + GM - 1) / GM;
    const long long nitems = (long long)nm * C;
    const int nkt = Tp / GK;
    const size_t slab = gt_chain_floats(Tp);

    // Item order.  An output tile re-reads its chain's k (q) rows, 1 MB per chain at T=1024, and with tiles dealt chain-fastest (the
    // fp32 kernel's order: 256 workgroups on 256 chains) nearly all of that comes from memory: 2.4 GB per product, which the fp32
    // kernel's 1.0 ms does not notice and this kernel's matrix time (0.26 ms) does.  Here a chain's tiles run side by side on ONE
    // XCD (its L2 serves the other seven readers of a row): workgroup b = (XCD b % 8, slot b / 8); XCD x owns the chains c = x mod 8
    // and walks the list [chain][position] with its slots; the tile behind a position rotates from round to round, mirrored in
    // every other round (tiles a and nm-1-a together are one chain's average), so that the slots stay within a round of each other
    // although their tiles differ 8 : 1 in length.  All tiles of a chain start at the same end of the contraction (dk walks it
    // backwards).  Few chains (or a grid that is no multiple of 8): the plain order.
    // (with few items per slot the rotation cannot even out the tile lengths: T=691 x 90 chains 0.30 -> 0.38 ms in this order)
    const bool xcd_order = (gridDim.x % NXCD_G3) == 0 && C >= 8 * NXCD_G3 && (long long)C * nm >= 4 * (long long)gridDim.x;
    const int nslots = gridDim.x / NXCD_G3, xcd = blockIdx.x % NXCD_G3;
    const int cpr = nslots / nm > 0 ? nslots / nm : 1;               // chains of an XCD per round
    // the workgroup's item number `round` (32-bit arithmetic throughout: nm * C items)
    auto item_of = [&](int round, int& c, int& mi, int& kbeg, int& nk) -> bool {
        unsigned ti;
        if (xcd_order) {
            const unsigned nch = (unsigned)(C - xcd + NXCD_G3 - 1) / NXCD_G3;       // chains of this XCD
            const unsigned u = blockIdx.x / NXCD_G3 + (unsigned)nslots * (unsigned)round;
            if (u >= nch * (unsigned)nm) return false;
            const unsigned lc = u / (unsigned)nm, pos = u % (unsigned)nm, rr = lc / (unsigned)cpr;
            const unsigned base = (pos + (rr >> 1)) % (unsigned)nm;
            ti = (rr & 1) ? nm - 1 - base : base;                    // 0 = the longest contraction range
            c = __builtin_amdgcn_readfirstlane((int)(lc * NXCD_G3 + xcd));  // (divisions run in the vector unit: back to scalar registers)
        } else {
            const unsigned n = blockIdx.x + gridDim.x * (unsigned)round;
            if (n >= (unsigned)nitems) return false;
            ti = n / (unsigned)C;                                    // longest first
            c = __builtin_amdgcn_readfirstlane((int)(n % (unsigned)C));
        }
        mi = __builtin_amdgcn_readfirstlane(AT ? (int)ti : nm - 1 - (int)ti);
        if (AT) {
            kbeg = mi * (GM / GK);                                   // chunks nkt-1 down to kbeg
            nk = nkt - kbeg;
        } else {
            kbeg = 0;
            nk = (mi + 1) * (GM / GK) < nkt ? (mi + 1) * (GM / GK) : nkt;
        }
        return true;
    };

    // ---- reading lanes: lane = (row, half); the instruction of slab sl takes piece 2*half + sl of the row ----------------
    unsigned rdA[2], rdB[2];
    {
        const int ra = 32 * wm + l31, rb = 32 * NW * wn + l31;          // (column block t: + 32 rows = + 2048 bytes, same swizzle)
#pragma unroll
        for (int sl = 0; sl < 2; ++sl) {
            rdA[sl] = (unsigned)(ra * 64 + (((2 * half + sl) ^ ((ra >> 2) & 3)) * 16));
            rdB[sl] = (unsigned)(BOFF + rb * 64 + (((2 * half + sl) ^ ((rb >> 2) & 3)) * 16));
        }
    }
    // ---- staging lanes --------------------------------------------------------------------------------------------------
    // Gt: dq -- unit = (row tid / 4, piece tid % 4) along the row; dk -- unit = (row = b = tid % 128, piece tid / 128) across rows e
    const int aRow = AT ? (tid & 127) : (tid >> 2);
    const int aPiece = AT ? (wave >> 1) : (tid & 3);
    const unsigned wA = (unsigned)(aRow * 64 + ((aPiece ^ ((aRow >> 2) & 3)) * 16));
    // k/q: a lane takes TWO adjacent columns (8-byte loads: half the vector-memory instructions of 4-byte ones, and the issue of
    // those -- 24 per wave and chunk -- was 17 % of a wave's time) of the 8 rows of one piece: units (column 2 cp + j, piece), j = 0, 1;
    // the piece is wave-uniform.  D = 256: all 512 lanes; D = 128: waves 0-3; D = 64: their lanes 0-31 (the others fetch a valid
    // address again and store nothing)
    const int bPieceRaw = wave / WPP, bCpRaw = (wave % WPP) * 64 + lane;
    const bool bOn = D >= 256 || (bPieceRaw < 4 && bCpRaw < D / 2);
    const int bPiece = bPieceRaw < 4 ? bPieceRaw : 3, bCp = bCpRaw < D / 2 ? bCpRaw : D / 2 - 1;
    unsigned wB[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int d = 2 * bCp + j;
        wB[j] = (unsigned)(BOFF + d * 64 + ((bPiece ^ ((d >> 2) & 3)) * 16));
    }
    const unsigned bVoff = (unsigned)(bCp * 8);

    // ---- request side (identical in all waves) ---------------------------------------------------------------------------
    // (per item: the buffer descriptors of the chain's slabs and the block offset; per chunk the scalar unit computes 32-bit
    // offsets by increments -- the first version's 64-bit address arithmetic was ~500 scalar instructions per wave and chunk,
    // 4000 issue cycles of the CU's one scalar unit against 3072 matrix cycles)
    typedef int v4i __attribute__((ext_vector_type(4)));
    auto make_rsrc = [](const void* p, size_t bytes) -> v4i {
        const unsigned long long a = (unsigned long long)(uintptr_t)p;
        return (v4i){(int)(unsigned)a, (int)((a >> 32) & 0xffffu), (int)(unsigned)bytes, 0x00020000};
    };
    const unsigned ldo4 = (unsigned)(ldo * 4), somax = (unsigned)(T - 1) * ldo4;
    int nx_round = 0;
    int nx_c = 0, nx_mi = 0, nx_kbeg = 0, nx_nk = 0, nx_j = 0;
    bool nx_valid = item_of(nx_round, nx_c, nx_mi, nx_kbeg, nx_nk);
    if (!nx_valid) return;                             // uniform
    unsigned aVoff = AT ? (unsigned)((tid & 127) * 4) : 0u;
    v4i nx_ra, nx_rb;
    unsigned nx_aoff = 0;                              // dq: byte offset of the item's 128-row block of Gt
    auto set_item_offsets = [&]() {
        nx_ra = make_rsrc(Gt + (size_t)nx_c * slab, slab * 4);
        nx_rb = make_rsrc(other + (size_t)nx_c * T * ldo, ((size_t)(T - 1) * ldo + D) * 4);
        if (!AT) {
            const int rows = Tp - nx_mi * GM < GM ? Tp - nx_mi * GM : GM;      // rows of the block that exist (the others are not stored)
            const int R = aRow < rows ? aRow : rows - 1;
            aVoff = (unsigned)((R * gt_row_len(nx_mi, Tp) + aPiece * 8) * 4);
            nx_aoff = (unsigned)(gt_block_off(nx_mi) * 4);
        }
    };
    set_item_offsets();

    // Requests and their waits are written out (asm): the compiler's own bookkeeping of loads in flight turned the wait for
    // the OLDER register set into a wait for everything in every other step (vmcnt 17 -> 0 where 35 -> 18 was meant, with and
    // without the epilogue's stores in the loop), i.e. a stall of one memory latency per chunk -- timing ablations: without the
    // requests 1.35 ms, without the split 1.34, without both 1.33, complete 1.99.  Here a set's NLD loads are issued back to
    // back, nothing else is in flight except the other set's NLD (the epilogue drains its stores), and the wait in front of
    // a set's split is vmcnt(NLD).  The loaded registers are tied to that wait ("+v"): no use can move above it; the loads are
    // UNCONDITIONAL (past the last chunk the last one is fetched again and never used) so that the count is always the same.
    constexpr int NLD = (AT ? 8 : 2) + 8;
    struct Regs { v4f alo, ahi; float a[8]; f32x2 b[8]; };
    struct Meta { bool valid, last; int c, mi; };
    // A request for one chunk = prepare (the chunk's scalar offsets, and the walk over the items moves on) + issue (the loads; the
    // interleaved schedule issues them between its matrix instructions, the B loads and the A loads at different places).
    struct Pend { v4i ra, rb; unsigned av, sa, krl4, sr; };
    auto prepare = [&](Meta& m, Pend& q) __attribute__((always_inline)) {
        m.valid = nx_valid;
        const unsigned k0 = (unsigned)(AT ? nkt - 1 - nx_j : nx_kbeg + nx_j) * GK;
        q.ra = nx_ra; q.rb = nx_rb; q.av = aVoff; q.krl4 = 0;
        if (!AT) {
            q.sa = nx_aoff + k0 * 4;
        } else {
            // the chunk's 32 rows (e) lie in one 128-row block; columns past the block's row length (a last block that is not
            // 128 wide) read the next row or the workspace's slack: they only reach output rows >= T, which are not written
            const unsigned kblk = k0 / GM;
            q.krl4 = (unsigned)gt_row_len((int)kblk, Tp) * 4;
            q.sa = (unsigned)GM * GM * 4 * (kblk * (kblk + 1) / 2) + (k0 - kblk * GM + 8 * aPiece) * q.krl4 + (unsigned)nx_mi * (GM * 4);
        }
        q.sr = (k0 + 8 * bPiece) * ldo4;                         // rows past T meet Gt == 0: any finite value will do (the last row's)
        m.last = nx_j + 1 == nx_nk;
        m.c = nx_c;
        m.mi = nx_mi;
        if (nx_valid && ++nx_j == nx_nk) {
            int c2, mi2, kbeg2, nk2;
            if (item_of(nx_round + 1, c2, mi2, kbeg2, nk2)) {
                ++nx_round;
                nx_c = c2; nx_mi = mi2; nx_kbeg = kbeg2; nx_nk = nk2; nx_j = 0;
                set_item_offsets();
            } else {
                nx_valid = false;
                nx_j = nx_nk - 1;
            }
        }
    };
    auto issue_a = [&](Regs& g, const Pend& q, int i0, int n) __attribute__((always_inline)) {     // AT: loads i0 .. i0 + n - 1 of 8; else both
        if (SEMICRF_G3_DBG & 2) return;
        if (!AT) {
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen" : "=v"(g.alo) : "v"(q.av), "s"(q.ra), "s"(q.sa));
            asm volatile("buffer_load_dwordx4 %0, %1, %2, %3 offen offset:16" : "=v"(g.ahi) : "v"(q.av), "s"(q.ra), "s"(q.sa));
        } else {
#pragma unroll
            for (int i = i0; i < i0 + n; ++i) {
                const unsigned so = q.sa + (unsigned)i * q.krl4;
                asm volatile("buffer_load_dword %0, %1, %2, %3 offen" : "=v"(g.a[i]) : "v"(q.av), "s"(q.ra), "s"(so));
            }
        }
    };
    auto issue_b = [&](Regs& g, const Pend& q, int i0, int n) __attribute__((always_inline)) {
        if (SEMICRF_G3_DBG & 2) return;
#pragma unroll
        for (int i = i0; i < i0 + n; ++i) {
            const unsigned sr = q.sr + (unsigned)i * ldo4;
            const unsigned so = sr < somax ? sr : somax;
            asm volatile("buffer_load_dwordx2 %0, %1, %2, %3 offen" : "=v"(g.b[i]) : "v"(bVoff), "s"(q.rb), "s"(so));
        }
    };
    auto fetch = [&](Regs& g, Meta& m) __attribute__((always_inline)) {
        Pend q;
        prepare(m, q);
        issue_a(g, q, 0, 8);
        issue_b(g, q, 0, 8);
    };
    // the set's loads have landed (the other set's NLD younger ones may be in flight)
    auto landed = [&](Regs& g) {
        if (SEMICRF_G3_DBG & 2) return;
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NLD));
        if (!AT) {
            asm volatile("" : "+v"(g.alo), "+v"(g.ahi));
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(g.a[i]));
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(g.b[i]));
    };
    float rs = 0.0f;                                    // !AT && rsum: this lane's part of its row's sum
    const bool want_rs = !AT && rsum != nullptr;
    auto convert = [&](Regs& g, const Meta& m, int stage) {
        if (SEMICRF_G3_DBG & 4) return;
        char* base = glds + stage * STAGE;
        v4f alo, ahi;
        if (AT) { alo = (v4f){g.a[0], g.a[1], g.a[2], g.a[3]}; ahi = (v4f){g.a[4], g.a[5], g.a[6], g.a[7]}; }
        else { alo = g.alo; ahi = g.ahi; }
        {
            const Limbs3 L = split8(alo, ahi);
            *(bf16x8*)(base + wA) = L.h;
            *(bf16x8*)(base + APL + wA) = L.m;
            *(bf16x8*)(base + 2 * APL + wA) = L.l;
        }
        if (!AT && want_rs && m.valid) {
            rs = add1(rs, add1(add1(add1(alo.x, alo.y), add1(alo.z, alo.w)), add1(add1(ahi.x, ahi.y), add1(ahi.z, ahi.w))));
            if (m.last) {
                float tot = rs + __shfl_xor(rs, 1);
                tot += __shfl_xor(tot, 2);
                const int mrow = m.mi * GM + aRow;
                if ((tid & 3) == 0 && mrow < T) rsum[((size_t)m.c * T + mrow) * ldrs] = tot;
                rs = 0.0f;
            }
        }
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const Limbs3 L = split8((v4f){g.b[0][j], g.b[1][j], g.b[2][j], g.b[3][j]}, (v4f){g.b[4][j], g.b[5][j], g.b[6][j], g.b[7][j]});
            if (D >= 256 || bOn) {
                *(bf16x8*)(base + wB[j]) = L.h;
                *(bf16x8*)(base + BPL + wB[j]) = L.m;
                *(bf16x8*)(base + 2 * BPL + wB[j]) = L.l;
            }
        }
    };

    f32x16 acc[NW];
#pragma unroll
    for (int t = 0; t < NW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // 2 NW groups of six matrix instructions per chunk: group i = (slab i / NW, column block i % NW).  The reads of group i + 1
    // are issued before the instructions of group i (two operand sets, alternating); the slab's A limbs are read with the
    // slab's first group.
    auto multiply = [&](int stage) {
        const char* base = glds + stage * STAGE;
        if (SEMICRF_G3_DBG & 8) {
            if (!(SEMICRF_G3_DBG & 1)) {
                Limbs3 A, B;
                A.h = A.m = A.l = B.h = B.m = B.l = __builtin_bit_cast(bf16x8, (u32x4){(unsigned)stage, 1u, 2u, 3u});
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                    for (int t = 0; t < NW; ++t) acc[t] = mma6(A, B, acc[t]);
            }
            return;
        }
#if SEMICRF_G3_MMA_ORDER == 0
        Limbs3 A[2], B[2];
        auto ldA = [&](Limbs3& L, int sl) {
            L.h = *(const bf16x8*)(base + rdA[sl]);
            L.m = *(const bf16x8*)(base + APL + rdA[sl]);
            L.l = *(const bf16x8*)(base + 2 * APL + rdA[sl]);
        };
        auto ldB = [&](Limbs3& L, int sl, int t) {
            L.h = *(const bf16x8*)(base + rdB[sl] + t * 2048);
            L.m = *(const bf16x8*)(base + BPL + rdB[sl] + t * 2048);
            L.l = *(const bf16x8*)(base + 2 * BPL + rdB[sl] + t * 2048);
        };
        ldA(A[0], 0);
        ldB(B[0], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 2 * NW>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            constexpr int sl = i / NW, t = i % NW;
            if constexpr (i + 1 < 2 * NW) {
                constexpr int sl2 = (i + 1) / NW, t2 = (i + 1) % NW;
                if constexpr (t2 == 0) ldA(A[sl2], sl2);
                ldB(B[(i + 1) & 1], sl2, t2);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (SEMICRF_G3_DBG & 1)
                acc[t][0] += __builtin_bit_cast(float, __builtin_bit_cast(u32x4, A[sl].h).x ^ __builtin_bit_cast(u32x4, B[i & 1].l).y ^
                                                           __builtin_bit_cast(u32x4, A[sl].m).z ^ __builtin_bit_cast(u32x4, B[i & 1].h).w ^
                                                           __builtin_bit_cast(u32x4, A[sl].l).x ^ __builtin_bit_cast(u32x4, B[i & 1].m).x);
            else
                acc[t] = mma6(A[sl], B[i & 1], acc[t]);
            __builtin_amdgcn_sched_barrier(0);
        });
#else
        // a slab at a time: all its operands (A and the NW column blocks' B: 12 (1 + NW) registers), then the six limb products
        // round-robin over the NW accumulators -- consecutive matrix instructions are independent
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

    // The interleaved form (SEMICRF_G3_INTERLEAVE): the SAME wave splits the next chunk between its own matrix instructions -- one
    // piece (a pair of values through the three-limb split, or a unit's three limb stores) behind every third instruction, the order
    // pinned by scheduling barriers.  Cycle stamps and priority experiments of the two-group schedule said why: a SIMD issues to its
    // oldest wave first, so whichever group is older runs at full speed and the other one starves (multiply 1960 cycles per chunk for
    // the waves 0-3, 2950 for 4-7; with raised priority for 4-7 it is the other way round) -- the split next to ANOTHER wave's matrix
    // instructions does not overlap, the split between a wave's OWN matrix instructions does (a filler per instruction is nearly free).
    auto mma1 = [&](const Limbs3& A, const Limbs3& B, f32x16& ac, int pz) __attribute__((always_inline)) {
        switch (pz) {
        case 0: ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.l, ac, 0, 0, 0); break;
        case 1: ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.l, B.h, ac, 0, 0, 0); break;
        case 2: ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.m, ac, 0, 0, 0); break;
        case 3: ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.m, ac, 0, 0, 0); break;
        case 4: ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.m, B.h, ac, 0, 0, 0); break;
        default: ac = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A.h, B.h, ac, 0, 0, 0); break;
        }
    };
    // SEMICRF_G3_FILL_LOADS: the chunk after next is requested from inside, too -- the k/q columns first through the split (pieces
    // 0-9), their loads two at a time behind the instructions 30, 33, 36, 39, the Gt unit last (pieces 10-14) and its loads behind
    // 45 and 46: the address path works while the matrix pipe does, instead of ~550 cycles per wave behind the chunk.
    constexpr bool FL = SEMICRF_G3_FILL_LOADS && NW == 4;
    auto multiply_fill = [&](int stage, Regs& g, Meta& mref, int wstage) __attribute__((always_inline)) {
        const char* base = glds + stage * STAGE;
        char* wbase = glds + wstage * STAGE;
        const Meta m = mref;
        Pend q;
        if (FL) prepare(mref, q);
        // the values to split: unit 0 = the Gt unit, units 1, 2 = the two k/q columns of the lane's pair
        v4f alo, ahi;
        if (AT) { alo = (v4f){g.a[0], g.a[1], g.a[2], g.a[3]}; ahi = (v4f){g.a[4], g.a[5], g.a[6], g.a[7]}; }
        else { alo = g.alo; ahi = g.ahi; }
        unsigned ph[4], pm[4], pl[4];
        auto pairv = [&](int u, int pr, float& x, float& y) __attribute__((always_inline)) {
            if (u == 0) {
                const v4f v = pr < 2 ? alo : ahi;
                x = (pr & 1) ? v.z : v.x; y = (pr & 1) ? v.w : v.y;
            } else {
                x = g.b[2 * pr][u - 1]; y = g.b[2 * pr + 1][u - 1];
            }
        };
        auto piece = [&](int k) __attribute__((always_inline)) {             // k = 0 .. 14: (unit, step k % 5)
            const int u = FL ? (k < 10 ? 1 + k / 5 : 0) : k / 5, st = k % 5;
            if (st < 4) {
                float x, y;
                pairv(u, st, x, y);
                split_pair(x, y, ph[st], pm[st], pl[st]);
            } else {
                const unsigned wo = u == 0 ? wA : wB[u - 1];
                const int pitch = u == 0 ? APL : BPL;
                if (u == 0 || D >= 256 || bOn) {
                    *(u32x4*)(wbase + wo) = (u32x4){ph[0], ph[1], ph[2], ph[3]};
                    *(u32x4*)(wbase + pitch + wo) = (u32x4){pm[0], pm[1], pm[2], pm[3]};
                    *(u32x4*)(wbase + 2 * pitch + wo) = (u32x4){pl[0], pl[1], pl[2], pl[3]};
                }
                if (u == 0 && !AT && want_rs && m.valid) {
                    rs = add1(rs, add1(add1(add1(alo.x, alo.y), add1(alo.z, alo.w)), add1(add1(ahi.x, ahi.y), add1(ahi.z, ahi.w))));
                    if (m.last) {
                        float tot = rs + __shfl_xor(rs, 1);
                        tot += __shfl_xor(tot, 2);
                        const int mrow = m.mi * GM + aRow;
                        if ((tid & 3) == 0 && mrow < T) rsum[((size_t)m.c * T + mrow) * ldrs] = tot;
                        rs = 0.0f;
                    }
                }
            }
        };
        // the same work in thirds (SEMICRF_G3_FINE): a split level (convert, widen, subtract: 5 instructions) or one limb plane's store
        // behind EVERY matrix instruction -- a whole piece is 13 dependent-ish vector instructions, longer than the instruction in
        // front of it runs, and the wave's next matrix instruction waits behind them
        f32x2 srem = {0.0f, 0.0f};
        float rsa = 0.0f;
        auto subpiece = [&](int k3) __attribute__((always_inline)) {         // k3 = 0 .. 44: (piece k3 / 3, part k3 % 3)
            const int k = k3 / 3, part = k3 % 3;
            const int u = FL ? (k < 10 ? 1 + k / 5 : 0) : k / 5, st = k % 5;
            if (st < 4) {
                if (part == 0) {
                    float x, y;
                    pairv(u, st, x, y);
                    const unsigned hu = cvt_pk_bf16(x, y);
                    const f32x2 xv = {x, y}, hf = {__builtin_bit_cast(float, hu << 16), __builtin_bit_cast(float, hu & 0xffff0000u)};
                    ph[st] = hu;
                    srem = sub2(xv, hf);
                } else if (part == 1) {
                    const unsigned mu = cvt_pk_bf16(srem.x, srem.y);
                    const f32x2 mf = {__builtin_bit_cast(float, mu << 16), __builtin_bit_cast(float, mu & 0xffff0000u)};
                    pm[st] = mu;
                    srem = sub2(srem, mf);
                } else {
                    pl[st] = cvt_pk_bf16(srem.x, srem.y);
                }
            } else {
                const unsigned wo = u == 0 ? wA : wB[u - 1];
                const int pitch = u == 0 ? APL : BPL;
                if (u == 0 || D >= 256 || bOn) {
                    if (part == 0) *(u32x4*)(wbase + wo) = (u32x4){ph[0], ph[1], ph[2], ph[3]};
                    else if (part == 1) *(u32x4*)(wbase + pitch + wo) = (u32x4){pm[0], pm[1], pm[2], pm[3]};
                    else *(u32x4*)(wbase + 2 * pitch + wo) = (u32x4){pl[0], pl[1], pl[2], pl[3]};
                }
                if (u == 0 && !AT && want_rs && m.valid) {
                    if (part == 0) rsa = add1(add1(alo.x, alo.y), add1(alo.z, alo.w));              // (the same order as piece())
                    else if (part == 1) rs = add1(rs, add1(rsa, add1(add1(ahi.x, ahi.y), add1(ahi.z, ahi.w))));
                    else if (m.last) {
                        float tot = rs + __shfl_xor(rs, 1);
                        tot += __shfl_xor(tot, 2);
                        const int mrow = m.mi * GM + aRow;
                        if ((tid & 3) == 0 && mrow < T) rsum[((size_t)m.c * T + mrow) * ldrs] = tot;
                        rs = 0.0f;
                    }
                }
            }
        };
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
        // A SIMD issues to its OLDER wave first: of the two waves that share it the older one would finish its chunk ~1000 cycles ahead
        // and wait at the barrier while the younger one goes on alone (a single wave does not fill the pipe).  So the younger one
        // goes first in the first half of a chunk and the older one in the second: they reach the barrier together.
        if (SEMICRF_G3_PRIO_FLIP > 0 && (wave >> 2)) __builtin_amdgcn_s_setprio(1);
        ldA(A, 0);
        ldB(B[0], 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        constexpr int LEAD = (SEMICRF_G3_FINE && NW == 4) ? SEMICRF_G3_LEAD : 0;
        static_for<0, LEAD>([&](auto kc) __attribute__((always_inline)) {
            if (!(SEMICRF_G3_ABL & 1)) subpiece(decltype(kc)::value);
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 2 * NW>([&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int t = i % NW;
            if constexpr (i + 1 < 2 * NW && (i + 1) % NW != 0 && !(SEMICRF_G3_ABL & 2)) ldB(B[(i + 1) & 1], (i + 1) / NW, (i + 1) % NW);
            __builtin_amdgcn_sched_barrier(0);
            static_for<0, 6>([&](auto pc) __attribute__((always_inline)) {
                constexpr int pz = decltype(pc)::value;
                constexpr int n = 6 * i + pz;                              // 0 .. 12 NW - 1
                if constexpr (SEMICRF_G3_PRIO_FLIP > 0 && n == SEMICRF_G3_PRIO_FLIP * NW / 4)
                    if (wave >> 2) __builtin_amdgcn_s_setprio(0);
                mma1(A, B[(SEMICRF_G3_ABL & 2) ? 0 : (i & 1)], acc[t], pz);
                __builtin_amdgcn_sched_barrier(0);
                constexpr int every = (12 * NW) / 15 > 0 ? (12 * NW) / 15 : 1;       // NW = 4: a piece behind every third instruction
                constexpr bool FINE = SEMICRF_G3_FINE && NW == 4;
                if constexpr (FINE && n + LEAD < 45 && !(SEMICRF_G3_ABL & 1)) {
                    subpiece(n + LEAD);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (!FINE && n % every == every - 1 && n / every < 15 && !(SEMICRF_G3_ABL & 1)) {
                    piece(n / every);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (FL && n >= 30 && n <= 39 && n % 3 == 0 && !(SEMICRF_G3_ABL & 4)) {
                    issue_b(g, q, 2 * ((n - 30) / 3), 2);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (FL && (n == 45 || (AT && n == 46)) && !(SEMICRF_G3_ABL & 4)) {
                    issue_a(g, q, 4 * (n - 45), 4);
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            if constexpr (i + 1 < 2 * NW && (i + 1) % NW == 0 && !(SEMICRF_G3_ABL & 2)) {
                ldA(A, (i + 1) / NW);
                ldB(B[(i + 1) & 1], (i + 1) / NW, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        // (NW < 4: fewer instructions than pieces -- the rest of the split behind the last one)
        static_for<0, 15>([&](auto kc) __attribute__((always_inline)) {
            constexpr int k = decltype(kc)::value;
            constexpr int every = (12 * NW) / 15 > 0 ? (12 * NW) / 15 : 1;
            if constexpr (k >= (12 * NW) / every) piece(k);
        });
    };

    Regs gX, gY;
    Meta mX = {false, false, 0, 0}, mY = {false, false, 0, 0};
    auto sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // my limb stores are in the LDS ...
        __builtin_amdgcn_s_barrier();                            // ... and so are everybody's; everybody is done reading the other stage
        asm volatile("" ::: "memory");
    };
    // Slots (one barrier each): group 0 multiplies chunk n in slot 2n and splits chunk n+1 (into the OTHER stage) in slot 2n+1;
    // group 1 splits chunk n+1 in slot 2n and multiplies chunk n in slot 2n+1 -- i.e. it runs the same loop one slot later and two
    // chunks ahead with its requests: in the loop it splits chunk n+2 into the stage it has just multiplied.  Chunk 0 is split by
    // everybody in front of the first barrier.  X is the register set split in the steps with P = 0, Y with P = 1.
#if SEMICRF_G3_INTERLEAVE
    fetch(gY, mY);
    fetch(gX, mX);
    landed(gY);
    convert(gY, mY, 0);
    fetch(gY, mY);
#else
    if (grp == 0) {
        fetch(gY, mY);
        fetch(gX, mX);
        landed(gY);
        convert(gY, mY, 0);
        fetch(gY, mY);
        sync();
    } else {
        fetch(gX, mX);
        fetch(gY, mY);
        landed(gX);
        convert(gX, mX, 0);
        fetch(gX, mX);
        sync();
        landed(gY);
        convert(gY, mY, 1);
        fetch(gY, mY);
        sync();
    }

#endif

    int cur_round = 0;
    int c = 0, mi = 0, kbeg = 0, nk = 0, j = 0;
    (void)item_of(cur_round, c, mi, kbeg, nk);
    // ---- the item's 128 x D block (C/D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)); rows >= T are
    //      dropped by the buffer's range check (the row sits in the per-lane offset): no branches.  Then ONE wait for everything
    //      in flight: stores and loads complete out of order with respect to each other, so with a store pending the compiler
    //      turns every later wait for a load into vmcnt(0) -- on every path through that wait, i.e. in every chunk (the first
    //      version: 17 -> 0 where 35 -> 18 was meant).  Draining here, once per item, keeps the chunks' waits exact. ----
    auto finish = [&]() -> bool {
        if (++j < nk) return true;
        const auto ro = __builtin_amdgcn_make_buffer_rsrc((void*)(out + (size_t)c * T * ldout), 0, (int)((size_t)T * ldout * 4), 0x00020000);
        const unsigned v0 = (unsigned)(((size_t)(mi * GM + 32 * wm + 4 * half) * ldout + 32 * NW * wn + l31) * 4);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const unsigned vr = v0 + (unsigned)((size_t)((r & 3) + 8 * (r >> 2)) * ldout * 4);
#pragma unroll
            for (int t = 0; t < NW; ++t) {
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(acc[t][r]), ro, vr, t * 128, 0);     // (not __builtin_bit_cast of a vector element: bf16x3.h)
                acc[t][r] = 0.0f;
            }
        }
#if SEMICRF_G3_DRAIN
        __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0)
#endif
        j = 0;
        return item_of(++cur_round, c, mi, kbeg, nk);
    };
#if SEMICRF_G3_PROBE
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pt = __builtin_readcyclecounter();
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();          // 100 MHz
#if SEMICRF_G3_PROBE == 2         // (2: the kernel's own clock -- one stamp at each end of a wave's life, nothing in the loop)
#define G3_STAMP(i) do { } while (0)
#else
#define G3_STAMP(i) do { const unsigned long long now_ = __builtin_readcyclecounter(); pc[i] += now_ - pt; pt = now_; } while (0)
#endif
#else
#define G3_STAMP(i) do { } while (0)
#endif
#if SEMICRF_G3_INTERLEAVE
    // one chunk (ONE barrier): its limbs are in stage P; every wave multiplies it and splits the next chunk into the other stage between
    // its matrix instructions
    auto step = [&](auto PC, Regs& gn, Meta& mn) __attribute__((always_inline)) -> bool {
        constexpr int P = decltype(PC)::value;
        sync();
        G3_STAMP(2);
        landed(gn);
        G3_STAMP(3);
        multiply_fill(P, gn, mn, P ^ 1);
        G3_STAMP(0);
        const bool more = finish();
        G3_STAMP(1);
        if (!FL) fetch(gn, mn);
        G3_STAMP(5);
        return more;
    };
#else
    // one chunk: its limbs are in stage P
    auto step = [&](auto PC, Regs& gn, Meta& mn) -> bool {
        constexpr int P = decltype(PC)::value;
        if (SEMICRF_MMA_PRIO) __builtin_amdgcn_s_setprio(1);
        multiply(P);
        if (SEMICRF_MMA_PRIO) __builtin_amdgcn_s_setprio(0);
        G3_STAMP(0);
        const bool more = finish();
        G3_STAMP(1);
        sync();
        G3_STAMP(2);
        landed(gn);
        G3_STAMP(3);
        convert(gn, mn, P ^ 1 ^ grp);
        G3_STAMP(4);
        fetch(gn, mn);
        G3_STAMP(5);
        sync();
        G3_STAMP(6);
        return more;
    };
#endif
    while (true) {
        if (!step(std::integral_constant<int, 0>{}, gX, mX)) break;
        if (!step(std::integral_constant<int, 1>{}, gY, mY)) break;
    }
    if (!SEMICRF_G3_INTERLEAVE && SEMICRF_G3_PHASED && grp == 0) sync();                   // (group 1's last slot)
#if SEMICRF_G3_PROBE
    if (SEMICRF_G3_PROBE == 2) pc[0] = __builtin_readcyclecounter() - pt;
    pc[7] = __builtin_amdgcn_s_memrealtime() - rt0;
    __syncthreads();
    if (lane == 0 && rsum)        // (behind the C x T row sums: the probe script allocates 16 K floats more)
        for (int i = 0; i < 8; ++i) rsum[(size_t)C * T * ldrs + (size_t)(blockIdx.x * 8 + wave) * 8 + i] = (float)pc[i];
#endif
#undef G3_STAMP
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int round_up32(int T) { return (T + 31) / 32 * 32; }

// 0: the packed path does not apply (the caller uses the direct kernels)
size_t interval_score_bwd_ws_bytes(int C, int T, int D)
{
    if (!(D == 64 || D == 128 || D == 256) || T < 64 || C < 1) return 0;
    const size_t Tp = (size_t)round_up32(T);
    const size_t slab = gt_chain_floats((int)Tp);
    if (slab * 4 >= (1ull << 31)) return 0;                    // 32-bit buffer offsets inside a chain's slab
    return (size_t)C * slab * sizeof(float) + 4096;            // + slack for the transposed walk of the last tile
}

template <bool AT, int NW>
static void launch_gemm(const float* Gt, int Tp, const float* other, long long ldo, float* out, long long ldout, int C,
                        int T, hipStream_t stream, float* rsum = nullptr, long long ldrs = 1)
{
    const size_t lds = (size_t)GNS * (GA_BYTES + GK * 64 * NW * 4);
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)score_bwd_gemm_kernel<AT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int ncu = 256, dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        ncu = v;
    const long long nitems = (long long)((T + GM - 1) / GM) * C;
    const int grid = nitems < ncu ? (int)nitems : ncu;
    hipLaunchKernelGGL((score_bwd_gemm_kernel<AT, NW>), dim3(grid), dim3(512), lds, stream, Gt, Tp, other, ldo, out, ldout, C, T,
                       AT ? nullptr : rsum, ldrs);
}

template <bool AT, int NW>
static void launch_gemm3(const float* Gt, int Tp, const float* other, long long ldo, float* out, long long ldout, int C,
                         int T, hipStream_t stream, float* rsum = nullptr, long long ldrs = 1)
{
    const size_t lds = (size_t)G3_NS * 3 * (GM * 64 + 64 * NW * 64);
    static PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)score_bwd_gemm3_kernel<AT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    int ncu = 256, dev = 0, v = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
        ncu = v;
    const long long nitems = (long long)((T + GM - 1) / GM) * C;
    const int grid = nitems < ncu ? (int)nitems : ncu;
    hipLaunchKernelGGL((score_bwd_gemm3_kernel<AT, NW>), dim3(grid), dim3(512), lds, stream, Gt, Tp, other, ldo, out, ldout, C, T,
                       AT ? nullptr : rsum, ldrs);
}

// true when the packed path ran (q/k rows must be 16-byte aligned for the LDS loads)
// prec: 0 exact fp32 (v_mfma_f32_32x32x2_f32), 1 the three-limb bf16 contraction (score_bwd_gemm3_kernel)
// fused != nullptr: {alpha, beta, logZ, gout} and dS is the score tensor itself
bool launch_interval_score_bwd_packed(const float* dS, const float* q, const float* k, int C, int T, int D, long long ldq,
                                      long long ldk, float qscale, int mode, float* dq, float* dk, long long lddq,
                                      long long lddk, void* ws, size_t ws_bytes, hipStream_t stream,
                                      const float* const* fused, int group, int pitch, float* drowc, long long lddrc, int prec)
{
    // drowc (may be NULL): row sums of the packed cotangent, out of the dq GEMM (which then must run: dq != NULL)
    if (drowc && !dq) return false;
    const ChainSlots SL{group, pitch};
    const int Cs = (C / group) * pitch;                       // slots: the chain pitch of dS and of the CRF-side vectors
    const size_t need = interval_score_bwd_ws_bytes(C, T, D);
    if (need == 0 || !ws || ws_bytes < need) return false;
    if (((uintptr_t)q & 15) || ((uintptr_t)k & 15) || ldq % 4 || ldk % 4 || ((uintptr_t)ws & 15)) return false;
    if ((long long)T * ldq * 4 >= (1ll << 31) || (long long)T * ldk * 4 >= (1ll << 31)) return false;
    const int Tp = round_up32(T);
    float* Gt = (float*)ws;
#if SEMICRF_PACK_ORDER == 1
    const dim3 pgrid((Cs + PK_CH - 1) / PK_CH, Tp / PK_B, Tp / PK_E);
#else
    const dim3 pgrid(Tp / PK_B, Tp / PK_E, (Cs + PK_CH - 1) / PK_CH);
#endif
    if (fused) {
        const PackFused F{fused[0], fused[1], fused[2], fused[3]};
        hipLaunchKernelGGL(score_bwd_pack_kernel<true>, pgrid, dim3(256), 0, stream, dS, Gt, Cs, T, Tp, qscale, mode, F, SL);
    } else {
        const PackFused F{nullptr, nullptr, nullptr, nullptr};
        hipLaunchKernelGGL(score_bwd_pack_kernel<false>, pgrid, dim3(256), 0, stream, dS, Gt, Cs, T, Tp, qscale, mode, F, SL);
    }
#define SEMICRF_GEMM_DISPATCH(AT_, OTHER, LDO, OUT, LDOUT)                                                              \
    switch (D) {                                                                                                        \
    case 64: launch_gemm<AT_, 1>(Gt, Tp, OTHER, LDO, OUT, LDOUT, C, T, stream, drowc, lddrc); break;                    \
    case 128: launch_gemm<AT_, 2>(Gt, Tp, OTHER, LDO, OUT, LDOUT, C, T, stream, drowc, lddrc); break;                   \
    default: launch_gemm<AT_, 4>(Gt, Tp, OTHER, LDO, OUT, LDOUT, C, T, stream, drowc, lddrc); break;                    \
    }
#define SEMICRF_GEMM3_DISPATCH(AT_, OTHER, LDO, OUT, LDOUT)                                                             \
    switch (D) {                                                                                                        \
    case 64: launch_gemm3<AT_, 1>(Gt, Tp, OTHER, LDO, OUT, LDOUT, C, T, stream, drowc, lddrc); break;                   \
    case 128: launch_gemm3<AT_, 2>(Gt, Tp, OTHER, LDO, OUT, LDOUT, C, T, stream, drowc, lddrc); break;                  \
    default: launch_gemm3<AT_, 4>(Gt, Tp, OTHER, LDO, OUT, LDOUT, C, T, stream, drowc, lddrc); break;                   \
    }
    // (the three-limb kernels address their outputs through buffers: 32-bit offsets inside a chain's [T][ld] block)
    if (prec == 1 && ((long long)T * lddq * 4 >= (1ll << 31) || (long long)T * lddk * 4 >= (1ll << 31))) prec = 0;
    if (prec == 1) {
        if (dq) { SEMICRF_GEMM3_DISPATCH(false, k, ldk, dq, lddq) }
        if (dk) { SEMICRF_GEMM3_DISPATCH(true, q, ldq, dk, lddk) }
    } else {
        if (dq) { SEMICRF_GEMM_DISPATCH(false, k, ldk, dq, lddq) }
        if (dk) { SEMICRF_GEMM_DISPATCH(true, q, ldq, dk, lddk) }
    }
#undef SEMICRF_GEMM3_DISPATCH
#undef SEMICRF_GEMM_DISPATCH
    return true;
}

}  // namespace semicrf
