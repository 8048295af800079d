// persist.hip -- persistent blocked sweep for the semi-CRF recurrences on gfx950 (impl 0 / auto).
//
// One launch computes, for every chain, the T-long dependent recurrence
//     u[p] = finalize( (+)_{j<p} ( u[j] (x) cell(p,j) ), skip(p) )        p = 0..T-1 (position order)
// in the (logsumexp,+) semiring (alpha/beta sweeps, NeuralSemiCRFInterval.py:402-410) or the
// (max,+) semiring with argmax (viterbi / viterbiBackward, :27-51, :122-144).  Positions run over
// frames ascending (DIR 0) or descending (DIR 1); cell(p,j) is score[end][begin] of the two frames.
//
// Work decomposition (positions in blocks of 16, workgroups of 8 waves, ONE workgroup per compute unit):
//   * SPINE workgroup, one per 4 chains: a ring of four waves, wave w owns position blocks k = w, w+4, ...;
//     a lane is (row r of the block, ONE chain).  Every wave applies each newly finished u[j] to its own
//     block's rows (band = the current block and the next three); the owner of the current block finalises
//     one position per step and publishes it through LDS to its ring mates; once per block it publishes 16
//     positions to HBM for the panels.  The T-step dependent chain never leaves one CU.  A lone wave issues
//     one plain instruction per ~4.5 cycles and one transcendental per ~16, so the step is kept at ~24
//     instructions with 4 transcendentals.  The ring waves never wait for memory: a LOADER wave stages the
//     band cells in LDS with asynchronous global->LDS loads several blocks ahead, and a FAR wave collects the
//     panels' partial results into LDS (see the SPINE section).
//   * PANEL waves (four per panel workgroup; every wave pulls its own tasks, no workgroup-level
//     synchronisation): a task is (position block k >= FAR0, column part of <= 16 tiles, 32 chains, 4 of the 16
//     rows).  The wave streams the far field -- cells (p in its rows, j < 16(k-FAR0+1)) -- tile by tile through
//     three LDS stages filled by asynchronous global->LDS loads (it counts its own outstanding loads), keeps
//     the partial accumulators in registers and hands ONE number per (position, chain, part) to the spine.
//   * Hand-offs carry their own flag -- u: a float that reads U_EMPTY until published; far-field partials: 8-byte
//     {tag, value} granules -- written with relaxed agent-scope atomic stores and polled with tagged loads
//     (no fences, placement independent).
//     Roles go by workgroup index (wg_ticket: the eight rings of a 32-chain panel group share one XCD); every workgroup of
//     a launch is resident (one per CU), every spin is bounded and raises the error word instead of hanging.
//
// HBM traffic: every lower-triangle cell is read exactly once (128-byte lines, non-temporal, in the panels;
// 16-byte segments in the band).  Algorithmic bytes per sweep: 4*B*(T(T+1)/2 + T-1).
#include <atomic>
#include <stdlib.h>
#include "common.h"
#include "pathscore.h"

#ifndef SEMICRF_PANEL_PROBES
#define SEMICRF_PANEL_PROBES 0      // 1: keep the panel timing probes (debug flags 4 and 32) in the hot loop
#endif

#ifndef SEMICRF_PROBE_TASKS
#define SEMICRF_PROBE_TASKS 0      // 1 (with SEMICRF_PANEL_PROBES): the task timeline of tools/task_trace.py (its own build: registers)
#endif
#ifndef SEMICRF_PROBE_HIST
#define SEMICRF_PROBE_HIST 0       // 1 (with SEMICRF_PANEL_PROBES): the per-tile activity histogram of tools/activity_hist.py (spills: its own build)
#endif

#ifndef SEMICRF_CT_DBG
#define SEMICRF_CT_DBG -1          // >= 0 (with SEMICRF_PANEL_PROBES=1): the debug flags as a compile-time constant -- a timing ablation
#endif                              // without the probe build's register pressure (the flags' dead branches fold away)

namespace semicrf {

constexpr int PB = 16;             // positions per block
#ifndef SEMICRF_RING
#define SEMICRF_RING 4
#endif
constexpr int RING = SEMICRF_RING; // waves per ring; band = RING-1 off-diagonal blocks + the diagonal block
#ifndef SEMICRF_NNEAR
#define SEMICRF_NNEAR 0
#endif
// NEAR tiles (round 4): the NNEAR column blocks just beyond the band -- tile (k, k-RING-n), n < NNEAR -- are applied INSIDE the spine
// workgroup, by its far wave, from cells the loader stages in LDS and from the u values in the LDS ring: no trip through memory.
// The panels' far field of block k ends NNEAR blocks earlier (newest tile k - FAR0), so the hand-off chain (publish u -> a panel
// notices, finishes its newest tile, stores its partial -> the far wave notices -> the owner) has RING + NNEAR blocks of slack
// instead of RING.  Measured chain: 7.5 us (88 chains) to 10 us (352) against 4 x 1.4 us of ring time per band; every extra tile
// costs the loader 4 more 64-line loads per row block (ring alone: 1.34 -> 1.64 us per block at 88 chains, 1.41 -> 1.88 at 352).
constexpr int NNEAR = SEMICRF_NNEAR;
#ifndef SEMICRF_BANDX
#define SEMICRF_BANDX 1            // 1: launches with few chains copy the band into spine-major order while they run (copy_role) and
#endif                              //    their loaders read whole 128-byte lines; 0: the loader always reads the score tensor itself
#ifndef SEMICRF_BANDX_MAXB
#define SEMICRF_BANDX_MAXB 192     // ... for at most this many chains per launch (the copy moves the band twice more: at 352 chains
#endif                              //    that is 1.5 TB/s each way in the head of the sweep, where the hand-offs want a quiet fabric)
#ifndef SEMICRF_BAND_LOAD_AUX
#define SEMICRF_BAND_LOAD_AUX 16
#endif
#ifndef SEMICRF_BANDX_WGSTRIDE
#define SEMICRF_BANDX_WGSTRIDE 3   // ... in every n-th panel workgroup (every one: the copy's burst delays the first hand-offs)
#endif
#ifndef SEMICRF_BANDX_WAVES
#define SEMICRF_BANDX_WAVES 1      // ... copy waves per panel workgroup (1 / 2 / 4: 127 / 130 / 139 us at T=1024 x 88)
#endif
#ifndef SEMICRF_BANDX_K0
#define SEMICRF_BANDX_K0 8         // ... row blocks below this one are loaded from the score tensor (the copy has not started yet)
#endif
constexpr int NBT = RING + NNEAR;  // band tiles per row block and spine (the ring's RING, then the near tiles)
constexpr int FAR0 = RING + NNEAR; // first block with a far field of its own; the newest far tile of block k is k - FAR0
constexpr int RS = 4;              // chains per ring (one per lane)
constexpr int GS = RS;              // chains per spine workgroup
constexpr int GP = 32;             // chains per panel task (4 per lane)
#ifndef SEMICRF_TPT
#define SEMICRF_TPT 12            // 12 tiles per task and a last part of at least 4: 184-186 us vs 188-190 with 16 / 1 (T=1024, NBatch=352)
#endif
constexpr int TPT = SEMICRF_TPT;   // tiles (column blocks) per panel task
#ifndef SEMICRF_LEADT
#define SEMICRF_LEADT 4
#endif
// Parts of a block with far tiles 0..q: FULL parts of TPT tiles at fixed columns, then the LAST part, which ends with the
// newest tile and is at least LEADT tiles long (when the block has that many): every full part lies LEADT tiles or more
// behind the newest one, i.e. all of its columns are published LEADT + RING blocks before its result is needed.
constexpr int LEADT = SEMICRF_LEADT;
__host__ __device__ inline int nfull_of(int q) { return q + 1 - LEADT >= 0 ? (q + 1 - LEADT) / TPT : 0; }
__host__ __device__ inline int nparts_of(int q) { return nfull_of(q) + 1; }
__host__ __device__ inline void part_tiles(int q, int part, int& m0, int& m1)
{
    m0 = part * TPT;
    m1 = part < nfull_of(q) ? m0 + TPT : q + 1;
}
#ifndef SEMICRF_NT
#define SEMICRF_NT 640
#endif
constexpr int NT = SEMICRF_NT;     // threads per workgroup (10 waves: a spine workgroup = 4 ring + loader + far + 4 streaming waves)
constexpr int MAX_CHUNKS = 16;     // chain chunks (launches) per call
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int SPIN_LIMIT = 1 << 20;       // global-memory polls (with s_sleep): ~0.3 s
constexpr int SPIN_LIMIT_LDS = 1 << 24;   // LDS polls (s_sleep 1): ~0.5 s
// Panel accumulators (reference M, sum S of exp2(t - M)).  u grows by ~45 log2 units per tile of 16 positions on N(0,1)
// scores, so the old rule -- "move M when a term exceeds M + 64, to exactly that term" -- moved every accumulator every
// 1.4 tiles, each at its own time; the test is a wave-wide __any over 32 chains x 8 columns, so the slow path ran on EVERY
// tile and cost as much as the accumulation itself (a tile of real values took 1.5 us, one of NaNs 0.9).  Now: M is an
// integer placed RESC_LIFT ABOVE the largest term seen when it is (re)set -- fp32 still holds every term within 2^-24 of
// that one -- the slow path is entered when a term exceeds M + RESC_HI, and it then moves every accumulator of the wave
// that is within RESC_EARLY of that limit, so that the chains stay in step: one visit every ~4 tiles.  A move is exact: S
// is scaled by a power of two (ldexp), no exponential.  Bounds: S <= 32 terms x 2^RESC_HI, 8 such in the final reduction.
constexpr float RESC_LIFT = 96.0f, RESC_HI = 108.0f, RESC_EARLY = 56.0f;
constexpr unsigned CTRL_INIT = 0xffffffffu;  // initial value of every workspace word (one 0xff fill per launch)
constexpr unsigned U_EMPTY = 0xffffffffu;  // not-yet-published u (a NaN pattern the spine never stores): the value is its own flag

// control words of a launch (all start at 0xffffffff).  Separate 128-byte lines: counters that take atomics must
// not share a line with words that are polled (hundreds of idle waves reading a line that others update atomically
// slow every dequeue down to tens of microseconds).
constexpr size_t CTRL_WORDS = 256;   // control words per chain chunk
constexpr int CTRL_EXIT = 64;         // [64]: workgroups that have left (leased workspaces: the last one resets the control words)
constexpr int CTRL_COPYQ = 192;       // [192]: band copy task queue head (a line of its own)
constexpr int CTRL_PATHQ = 160;       // [160]: path task queue head (a line of its own)
constexpr int CTRL_GEN = 96;          // [96]: leased workspaces: the tag of the launch that last cleaned up here (CTRL_INIT after a fill);
                                      // every workgroup of the next launch compares it with what the host expects (error 13)

typedef unsigned long long u64;
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v4u_a4 __attribute__((ext_vector_type(4), aligned(4)));      // rows of odd multiples of 8 bytes
template <int V>
struct IC { static constexpr int value = V; };

#define DBGF(P) (SEMICRF_CT_DBG >= 0 ? (unsigned)SEMICRF_CT_DBG : (P).dbg)
struct SweepParams {
    const float* score;
    const float* noise;
    int T, B, K;
    int c0, c1;            // chains [c0, c1) of the batch are handled by this launch
    int nSpine, nPanelGroups;
    int nTasks;            // panel tasks (k ascending, then column part, then chain group, then row quarter)
    int taskBase;          // tasks [0, taskBase) are the first tasks of the panel workgroups' waves (no draw); the queue hands out the rest
    int panelWaves;        // waves per non-spine workgroup that work as panels (the rest exit at once)
    int hybridPanelWaves;  // panel waves of a spine workgroup (0..2)
    int hybridStart;       // ... which start once the ring has reached this row block
    int zeroWaves;         // GRAD: waves per panel workgroup that write the zero upper triangle of dScore (0: separate kernel)
    unsigned tag;          // nonzero launch epoch
    int selfclean;         // leased workspace (semicrf_workspace_register): the launch leaves u and its control words as the fill would
    unsigned expect_gen;   // ... and finds ctrl[CTRL_GEN] == expect_gen: the previous launch into this workspace was the one the host
                           // counted, and it finished its clean-up (otherwise: error 13, outputs poisoned)
    unsigned dbg;          // SEMICRF_DEBUG_FLAGS (timing experiments only; results are wrong when set):
                           // 1 spine ignores far partials, 2 panels exit at once, 4 panels do not wait for u,
                           // 8 spine exits at once, 16 spine 0 records per-block timestamps,
                           // 32 panels only stream their cells (no granules, no math)
    unsigned* ctrl;        // [1] error, [2] panel task queue head, [3] zero-fill row queue head, [64] workgroups that have left (leases)
    u64* ts;               // [2T] debug timestamps of spine 0, ring 0 (dbg & 16)
    unsigned* ug;          // [T][B] u as float bits (position-major: index p*B + c); U_EMPTY until the spine publishes it
    unsigned* ug_other;    // leased workspaces: the u buffer of the PREVIOUS launch into this workspace (the two alternate); this launch
                           // puts it back to U_EMPTY, at its start and off every critical path (nullptr: nothing to clean)
    int gradLazyShort;     // GRAD: panel waves poll for a tile that is not their task's newest every ~0.5 us instead of every ~4
    int bandWaves;         // GRAD: waves per panel workgroup that write the band's marginals (band_role); 0: the ring waves do
    u64* farg;             // [parts][T][B] granules of far-field partials (part = column range of TPT tiles)
    float* u_out;          // [T][B] by FRAME (natural-log units for LSE) or nullptr
    float* last_out;       // [B] value at the last position (logZ for DIR 0) or nullptr
    int* code;             // MAX: [B][T] backtrack codes by frame
    // GRAD (LSE, DIR 1 only): marginals are a by-product of the beta sweep (NeuralSemiCRFInterval.py:424-447, :469-472)
    const float* vfwd;     // [T][B] alpha values by frame (natural log)
    const float* logZ;     // [B]
    const float* gout;     // upstream gradient of chain c: gscale * gout[c * gstride] (gstride 0: one value for every chain)
    int gstride;
    float gscale;
    float* dScore;         // [T][T][B]: lower triangle + diagonal written here (the upper triangle by zero_upper_kernel)
    float* dNoise;         // [T-1][B]
    // logProb as ONE launch (round 5; LSE, DIR 0 only): a spare wave of the panel workgroups computes the path score of every chain
    // (path_role, pathscore.h) while the sweep runs and leaves it as a {tag, value} granule; the ring wave that finalises the last
    // position takes it and writes logProb = path - logZ (NeuralSemiCRFInterval.py:587-588)
    const int* pathPairs;  // [K][2] (begin, end) or nullptr: no path role
    const int* pathOffsets;// [B+1]
    int pathK;
    int pathSpine;         // the spine workgroups' spare wave takes path tasks too (launches without panel workgroups)
    float* pathOut;        // [B] logProb
    u64* pathg;            // [B] granules (workspace)
    float noiseAdd;        // GRAD: dNoise gets noiseAdd * gout[c] on top of the marginal (d cum[T-1] / d noise of the path score), 0: nothing
    float* band;           // SEMICRF_BANDX: [K][bandSpines][NBT][16 columns][16 rows][4 chains]: the band, spine-major (workspace), or nullptr
    unsigned* bandFlags;   // ... [32-chain groups][K + 2][8]: {launch tag, row block} once tile t of (row block, group) has been copied
    int bandSpines;        // ... spines of the whole batch (the copy is indexed by the batch's spine number)
    int bandGroups;        // ... 32-chain groups of the whole batch
    int bandK0;            // ... row blocks < bandK0 are loaded from the score tensor itself
    int copyWaves;         // ... waves per panel workgroup that copy (copy_role)
    int cellNT;            // 1: the panels stream their cells non-temporal (large problems); 0: ordinary loads (see cell_policy_nt)
    int gradNT;            // GRAD: likewise for the marginals' stores
    int nfarw;             // far waves of a spine that take turns on the blocks (1 .. NFARW; near tiles: always NFARW)
};

// ---------------------------------------------------------------------------------------------
// small helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float fexp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float flog2(float x) { return __builtin_amdgcn_logf(x); }

__device__ __forceinline__ u64 make_granule(unsigned tag, float v)
{
    return ((u64)tag << 32) | (u64)__float_as_uint(v);
}
__device__ __forceinline__ u64 make_granule_key(unsigned tag, float v, int key)
{
    return ((u64)(((tag & 0xffffu) << 16) | ((unsigned)key & 0xffffu)) << 32) | (u64)__float_as_uint(v);
}
__device__ __forceinline__ void store_granule(u64* p, u64 g)
{
    __hip_atomic_store(p, g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 load_granule(const u64* p)
{
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int DIR>
__device__ __forceinline__ int frame_of(int p, int T) { return DIR == 0 ? p : T - 1 - p; }

// element offset (without the chain) of cell(pi, pj), pj < pi, in score [T][T][B]
template <int DIR>
__device__ __forceinline__ size_t cell_index(int pi, int pj, int T)
{
    return DIR == 0 ? (size_t)pi * T + pj : (size_t)(T - 1 - pj) * T + (T - 1 - pi);
}
// noise row between positions p-1 and p
template <int DIR>
__device__ __forceinline__ int gap_of(int p, int T) { return DIR == 0 ? p - 1 : T - 1 - p; }

// log2-domain accumulator, exact running max, one exp per push: value = M + log2(S); empty = (-inf, 0)
__device__ __forceinline__ void acc_push1(float& M, float& S, float t)
{
    const float d = t - M;                       // M = -inf -> +inf
    const float e = fexp2(-fabsf(d));
    S = d > 0.0f ? fmaf(S, e, 1.0f) : S + e;
    M = fmaxf(M, t);
}
__device__ __forceinline__ void acc_merge(float& M, float& S, float M2, float S2)
{
    if (S2 == 0.0f) return;
    if (M2 > M) {
        S = S * fexp2(M - M2) + S2;   // M = -inf -> S*0 (S = 0)
        M = M2;
    } else {
        S += S2 * fexp2(M2 - M);
    }
}
// (max, key) push for candidates that arrive in position order within one accumulator: frames ascend with the
// positions in DIR 0 (the first maximum stays) and descend in DIR 1 (a later equal candidate has the smaller
// frame and wins) -- one compare instead of the general three (the reference keeps the smallest frame index)
template <int DIR>
__device__ __forceinline__ void max_push_seq(float& best, int& key, float t, int k)
{
    const bool take = DIR == 0 ? t > best : t >= best;
    best = take ? t : best;
    key = take ? k : key;
}
// softplus in log2 units: log2(1 + 2^(x*log2e)), linear above the reference's threshold (20)
__device__ __forceinline__ float softplus2(float x)
{
    const float x2 = x * LOG2E;
    return x > 20.0f ? x2 : flog2(1.0f + fexp2(x2));
}

// sticky device-side status word (0 = fine); read and cleared by semicrf_debug_device_status()
__device__ unsigned g_dev_status = 0;
// set (LDS) by any wave of this workgroup that gave up on a bounded wait or saw another one give up: a spine workgroup
// whose flag is up poisons its final outputs with NaN (every wrong result is preceded by a timed-out wait in the
// workgroup of the affected chains -- a missing far partial, band row or ring entry), so that a caller notices without
// reading the status word: logZ / alpha / beta / the gradient come back NaN, decode returns a negative total.
__shared__ int s_abort;

// pinned host word (or null) that an aborting launch raises: the host looks at it before it trusts a leased workspace again
__device__ unsigned* g_host_abort = nullptr;

__device__ __forceinline__ void set_error(unsigned* ctrl, unsigned code)
{
    __hip_atomic_store(ctrl + 1, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&g_dev_status, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    unsigned* const h = g_host_abort;
    if (h) __hip_atomic_store(h, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Bounded waiting: returns true when the caller must give up.  The first waiter to exceed its limit
// raises the error word; everybody else notices it within a few hundred polls and drains, so a
// protocol bug costs well under a second instead of hanging the GPU.
__device__ __forceinline__ bool spin_abort(unsigned* ctrl, int& spins, int limit, unsigned code)
{
    ++spins;
    if (spins > limit) { set_error(ctrl, code); s_abort = 1; return true; }
    if ((spins & 255) == 0 &&
        __hip_atomic_load(ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != CTRL_INIT) { s_abort = 1; return true; }
    return false;
}

// ---------------------------------------------------------------------------------------------
// SPINE workgroup: ring waves + loader wave + far-partial wave
// ---------------------------------------------------------------------------------------------
// The ring waves never issue a global LOAD: vector-memory results return in order, so one request queued
// behind a burst of HBM misses (several microseconds while the panels stream) would stall the dependent chain.
//   * the LOADER wave copies the band cells of row block kr -- four 16x16x4-chain tiles -- and its per-row
//     constants into LDS with asynchronous global->LDS loads, running up to NRBUF row blocks ahead of the ring;
//   * the FAR wave polls the panels' partials for the upcoming blocks (its polls are the only requests it has
//     in flight, so they see L2 latency) and leaves one combined value per (row, chain) in LDS;
//   * a ring wave reads its four tiles from LDS into registers at the start of its iteration.
// LDS ring: 128 positions x 4 chains x 8 bytes {u, seq = position+1}.  A position is published with ONE
// 8-byte DS write per lane by the four lanes r == 0 of the owning wave (after the broadcast every lane holds
// u[j] of its chain); consumers read their chain's 8 bytes and check seq.
//
// The step bodies are written for a lone wave (issue-bound):
//   * the diagonal step needs no lane predicates: the broadcast value is what gets published and the
//     next broadcast simply reads the lanes of the next row,
//   * row jj+1 receives its last term through logaddexp2(Vp, u + W) where Vp (everything but that term)
//     is refreshed one step ahead by the (M,S) push that runs beside it.
#ifndef SEMICRF_NLOADER
#define SEMICRF_NLOADER 1
#endif
constexpr int NLOADER = SEMICRF_NLOADER;              // loader waves per ring (a row block is RING tiles)
#ifndef SEMICRF_NRBUF
#define SEMICRF_NRBUF 4
#endif
constexpr int NRBUF = SEMICRF_NRBUF;                  // row-block buffers between the loader and the ring
constexpr int TILE_BYTES = PB * PB * RS * 4;          // 4096: [column u][row r][chain] floats
constexpr int NCONST = 3;                             // per-row constants: diagonal cell, noise, alpha (GRAD)
constexpr int LDS_TILES = 0;
constexpr int LDS_CONST = LDS_TILES + NRBUF * RING * TILE_BYTES;       // [NRBUF][NCONST][64] floats
constexpr int LDS_FAR = LDS_CONST + NRBUF * NCONST * 256;              // [8][64] x {value, key, seq, pad}
constexpr int NFAR = 8;                               // far-partial entries (blocks) between the far wave and the ring (>= RING)
constexpr int NPOS = 128;                             // positions kept in the LDS ring (>= the band, RING blocks)
static_assert(NFAR >= RING && NPOS >= RING * PB, "ring geometry");
constexpr int LDS_RING = LDS_FAR + NFAR * 64 * 16;                     // [NPOS][RS] x {u, seq}
#ifndef SEMICRF_STAGGER
#define SEMICRF_STAGGER 32          // s_sleep units (64 cycles) per block of distance before a panel wave starts on its known first task
#endif
#ifndef SEMICRF_GRAD_LAZY
#define SEMICRF_GRAD_LAZY 127        // s_sleep units between a gradient-sweep panel wave's polls for a tile that is not its task's newest
#endif
#ifndef SEMICRF_GRAD_HSTART
#define SEMICRF_GRAD_HSTART 12       // gradient sweep: the spine workgroups' spare waves stream from this row block on
#endif
#ifndef SEMICRF_LOADER_PACE
#define SEMICRF_LOADER_PACE 0       // s_sleep units between the loader's 1 KB band loads (0: a burst per row block)
#endif
#ifndef SEMICRF_LOADER_AUX
#define SEMICRF_LOADER_AUX 0        // cache policy bits of the loader's band loads: 1 sc0, 2 nt, 16 sc1
#endif
#ifndef SEMICRF_SPH
#define SEMICRF_SPH 1               // spines (4-chain rings with their loader and far wave) per spine workgroup: 1 or 2
#endif
#ifndef SEMICRF_FAR_EARLY
#define SEMICRF_FAR_EARLY 0         // 1: the far wave requests a block's first four partials before its near-tile work (measured: neutral)
#endif
#ifndef SEMICRF_FAR_PRIO
#define SEMICRF_FAR_PRIO 0          // s_setprio of the far wave(s) (2: above the ring's shadow phase; measured: neutral)
#endif
#ifndef SEMICRF_NFARW
#define SEMICRF_NFARW 2
#endif
// Far waves per spine workgroup: far wave f takes the blocks k = RING + f, RING + f + NFARW, ...  One far wave needs a device-scope
// round trip per block even when the partial has long been stored (issue the poll, wait for it: ~2 us behind the loader's
// traffic in the same compute unit) -- the block period of every hand-off-bound shape sat exactly there (2.1 - 2.2 us at 88
// chains); two of them take turns.
constexpr int NFARW = SEMICRF_NFARW;
constexpr int NNSLOT = 8;                             // near-tile slots (row blocks) between the loader and the far waves
constexpr int NEAR_SLOT_BYTES = NNEAR * TILE_BYTES + 256;              // NNEAR tiles + alpha of the row block (GRAD)
constexpr int LDS_NEAR = LDS_RING + NPOS * RS * 8;                      // [NNSLOT] near slots
constexpr int LDS_CTL = LDS_NEAR + (NNEAR > 0 ? NNSLOT * NEAR_SLOT_BYTES : 0);   // ready[NRBUF], cons[RING], nready[NNSLOT], fcons[NFARW] (ints)
// (the far wave reads the ring entries of block k - RING - n while the ring has written up to block k - 1)
static_assert(NPOS >= (FAR0 + 1) * PB && NNSLOT >= FAR0 + 1 && NRBUF + RING + NNSLOT + NFARW <= 64 && NNSLOT % NFARW == 0, "near geometry");
constexpr int LDS_DUMMY = LDS_CTL + 256;                               // sink of the non-writer lanes' ring stores
constexpr int LDS_SPINE_BYTES = LDS_DUMMY + (64 * 2 + PB * 8) * 4;

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

struct SpineBlk { float v[PB]; };

// lane -> (row of the block, chain of the ring) in the spine workgroup's ring, far and loader-constant code: a chain's 16 rows
// are one DPP row (lanes 16 ch .. 16 ch + 15), so that "row j of my chain" is a row_newbcast operand
__device__ __forceinline__ int spine_row(int lane) { return lane & 15; }
__device__ __forceinline__ int spine_chain(int lane) { return lane >> 4; }

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N)); }

// LDS word accessors for the flags (every poll must be a fresh DS read)
__device__ __forceinline__ int lds_flag_load(const int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_flag_store(int* p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// The same for the loader wave, as asm: the compiler orders every DS access it can see after ALL outstanding
// global->LDS loads (s_waitcnt vmcnt(0)), which would serialise the loader's pipeline; it counts its loads itself.
__device__ __forceinline__ unsigned lds_addr(const void* p)
{
    return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ int lds_flag_load_asm(const int* p)
{
    int v;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(lds_addr(p)) : "memory");
    return v;
}
__device__ __forceinline__ void lds_flag_store_asm(int* p, int v)
{
    asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr(p)), "v"(v) : "memory");
}
// hand-off entries that carry their own sequence number: the whole entry in ONE access, always a fresh read
__device__ __forceinline__ u64 lds_load64(const void* p)
{
    return __hip_atomic_load((const u64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_store64(void* p, float value, int seq)
{
    __hip_atomic_store((u64*)p, ((u64)(unsigned)seq << 32) | (u64)__float_as_uint(value), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_store128(void* p, float4 v)
{
    const f32x4_t w = {v.x, v.y, v.z, v.w};
    asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr(p)), "v"(w) : "memory");
}
__device__ __forceinline__ float4 lds_load128(const void* p)
{
    f32x4_t w;
    asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(lds_addr(p)) : "memory");
    return make_float4(w.x, w.y, w.z, w.w);
}

// ---- loader wave ---------------------------------------------------------------------------------------
template <int DIR, bool GRAD>
__device__ __forceinline__ void loader_role(const SweepParams& P, int sg, char* lds, int lid)
{
    const int T = P.T, B = P.B, K = P.K;
    unsigned* const ctrl = P.ctrl;
    const size_t Bs = (size_t)B;
    const float* const score = P.score;
    const float* const noise = P.noise;
    const float* const vfwd = P.vfwd;
    const int lane = threadIdx.x & 63;
    const int cbase = P.c0 + sg * GS;
    const size_t last4 = (size_t)T * T * Bs - 4;                 // last element offset a 16-byte load may start at
    // tile loads: instruction q covers columns 4q..4q+3, lane = (column within the group, row)
    const int tu = lane >> 4, tr = lane & 15;
    // constants: lane = (row, chain) like the ring waves
    const int cr = spine_row(lane), cch = spine_chain(lane);
    const int cc = cbase + cch < P.c1 ? cbase + cch : cbase;
    int* const ready = (int*)(lds + LDS_CTL);
    int* const cons = ready + NRBUF;
    int* const nready = cons + RING;
    const int* const fcons = nready + NNSLOT;
    constexpr int NL = RING * 4 + 2 + (GRAD ? 1 : 0) + NNEAR * 4 + ((GRAD && NNEAR > 0) ? 1 : 0);   // loads per row block (exactly, the waits count them)
    constexpr int LDEPTH = 3 * NL <= 63 ? 3 : 2;
    static_assert(2 * NL <= 63, "loader pipeline");

    const bool useband = SEMICRF_BANDX && P.band != nullptr;
    const float* const bandp = useband ? P.band + (size_t)(cbase / GS) * NBT * (TILE_BYTES / 4) + lane * 4 : nullptr;
    const size_t band_kr = (size_t)P.bandSpines * NBT * (TILE_BYTES / 4);
    // The copy's hand-off: NBT words per (32-chain group, row block), each {launch tag, row block} once its tile is in place.  The
    // loader looks at them with a SCALAR load (its own counter: the vector-memory counter of this wave counts LDS-DMA pieces exactly)
    // that takes two row blocks at once, request and wait in ONE asm statement: the compiler believes an asm's outputs are there when
    // the statement ends -- a first version requested a row block ahead and waited later, and the compiler copied the registers in
    // between and reused them while the load was still on its way (a memory aperture violation a microsecond later).
    typedef unsigned v16u __attribute__((ext_vector_type(16)));
    const int bKpad = K + 2;
    const unsigned* const bflags = useband ? P.bandFlags + (size_t)(cbase / GP) * bKpad * 8 : nullptr;
    const unsigned btag = P.tag;
    int ready_upto = -1;                       // row blocks <= ready_upto are known to be copied
    auto band_ready = [&](int kr) -> bool {
        const uintptr_t fa = (uintptr_t)(bflags + (size_t)kr * 8);                 // wave-uniform by construction; said explicitly for the "s" operand
        const u64 fp = (u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)fa) |
                       ((u64)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(fa >> 32)) << 32);
        v16u f;
        asm volatile("s_load_dwordx16 %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "=&s"(f) : "s"(fp) : "memory");
        // (a flag word is {launch tag, row block}: neither the fill nor a previous launch's words can pass for this row block's)
        const unsigned want0 = ((btag & 0xffffu) << 16) | (unsigned)kr, want1 = want0 + 1u;
        bool ok0 = true, ok1 = true;
#pragma unroll
        for (int i = 0; i < NBT; ++i) { ok0 = ok0 && f[i] == want0; ok1 = ok1 && f[8 + i] == want1; }
        if (ok0) ready_upto = ok1 ? kr + 1 : kr;
        return ok0;
    };
    auto issue = [&](int kr) {
        const int slot = kr % NRBUF;
        const int prow_t = kr * PB + tr < T ? kr * PB + tr : T - 1;
        const bool fromband = useband && kr >= P.bandK0;
        if (fromband) {
            const float* const bk = bandp + (size_t)kr * band_kr;
#pragma unroll
            for (int i = 0; i < RING; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)(bk + i * (TILE_BYTES / 4) + q * 256),
                                                     (lds_void_t*)(lds + LDS_TILES + (slot * RING + i) * TILE_BYTES + q * 1024), 16, 0, SEMICRF_BAND_LOAD_AUX);   // sc1: written by another CU in this launch
        } else
#pragma unroll
        for (int i = 0; i < RING; ++i) {
            const int kc = kr - (RING - 1) + i > 0 ? kr - (RING - 1) + i : 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                int pj = kc * PB + 4 * q + tu;
                pj = pj < prow_t ? pj : (prow_t > 0 ? prow_t - 1 : 0);       // only cells below the diagonal are used
                size_t off = cell_index<DIR>(prow_t, pj, T) * Bs + cbase;
                off = off < last4 ? off : last4;
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(score + off),
                                                 (lds_void_t*)(lds + LDS_TILES + (slot * RING + i) * TILE_BYTES + q * 1024), 16, 0, SEMICRF_LOADER_AUX);
                if (SEMICRF_LOADER_PACE > 0) __builtin_amdgcn_s_sleep(SEMICRF_LOADER_PACE);
            }
        }
        const int prow_c = kr * PB + cr < T ? kr * PB + cr : T - 1;
        const int frow = frame_of<DIR>(prow_c, T);
        char* cb = lds + LDS_CONST + slot * NCONST * 256;
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(score + ((size_t)frow * T + frow) * Bs + cc), (lds_void_t*)cb, 4, 0, 0);
        __builtin_amdgcn_global_load_lds((gbl_void_t*)(noise + (size_t)gap_of<DIR>(prow_c >= 1 ? prow_c : 1, T) * Bs + cc),
                                         (lds_void_t*)(cb + 256), 4, 0, 0);
        if (GRAD)
            __builtin_amdgcn_global_load_lds((gbl_void_t*)(vfwd + (size_t)frow * Bs + cc), (lds_void_t*)(cb + 512), 4, 0, 0);
        if (NNEAR > 0) {
            // the near tiles of row block kr: column blocks kr - RING - n (clamped: the first blocks have none, nobody reads them)
            char* const nb = lds + LDS_NEAR + (kr % NNSLOT) * NEAR_SLOT_BYTES;
            if (fromband) {
                const float* const bk = bandp + (size_t)kr * band_kr;
#pragma unroll
                for (int n = 0; n < NNEAR; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        __builtin_amdgcn_global_load_lds((gbl_void_t*)(bk + (RING + n) * (TILE_BYTES / 4) + q * 256),
                                                         (lds_void_t*)(nb + n * TILE_BYTES + q * 1024), 16, 0, 16);
            } else
#pragma unroll
            for (int n = 0; n < NNEAR; ++n) {
                const int kc = kr - RING - n > 0 ? kr - RING - n : 0;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    int pj = kc * PB + 4 * q + tu;
                    pj = pj < prow_t ? pj : (prow_t > 0 ? prow_t - 1 : 0);
                    size_t off = cell_index<DIR>(prow_t, pj, T) * Bs + cbase;
                    off = off < last4 ? off : last4;
                    __builtin_amdgcn_global_load_lds((gbl_void_t*)(score + off), (lds_void_t*)(nb + n * TILE_BYTES + q * 1024), 16, 0, SEMICRF_LOADER_AUX);
                }
            }
            if (GRAD)
                __builtin_amdgcn_global_load_lds((gbl_void_t*)(vfwd + (size_t)frow * Bs + cc), (lds_void_t*)(nb + NNEAR * TILE_BYTES), 4, 0, 0);
        }
    };
    auto publish = [&](int kr) {
        if (lane == 0) {
            lds_flag_store_asm(ready + kr % NRBUF, kr + 1);
            if (NNEAR > 0) lds_flag_store_asm(nready + kr % NNSLOT, kr + 1);
        }
    };

    // NLOADER loader waves take the row blocks round-robin (lid = this wave's index); each keeps LDEPTH of its own
    // row blocks in flight (the counter holds 63 operations)
    int prev2 = -1, prev1 = -1;                    // row blocks issued by this wave and not yet published (oldest first)
    for (int kr = lid; kr < K; kr += NLOADER) {
        if (kr >= NRBUF) {
            // the buffer's previous row block (kr - NRBUF) must have been read by its ring wave
            const int w = (kr - NRBUF) % RING;
            int spins = 0;
            while (lds_flag_load_asm(cons + w) < kr - NRBUF + 1) {
                __builtin_amdgcn_s_sleep(2);
                if (spin_abort(ctrl, spins, SPIN_LIMIT_LDS, 6)) return;
            }
        }
        if (NNEAR > 0 && kr >= NNSLOT && !(SEMICRF_PANEL_PROBES && (DBGF(P) & 9u))) {
            // ... and the near slot's previous row block (kr - NNSLOT) by the far wave
            int spins = 0;
            while (lds_flag_load_asm(fcons + (kr - NNSLOT - RING + NFARW * NNSLOT) % NFARW) < kr - NNSLOT + 1 && kr - NNSLOT >= RING) {
                __builtin_amdgcn_s_sleep(2);
                if (spin_abort(ctrl, spins, SPIN_LIMIT_LDS, 6)) return;
            }
        }
        if (useband && kr >= P.bandK0 && kr > ready_upto) {
            int spins = 0;
            while (!band_ready(kr)) {
                __builtin_amdgcn_s_sleep(8);
                if (spin_abort(ctrl, spins, SPIN_LIMIT, 16)) return;
                asm volatile("s_dcache_inv" ::: "memory");          // (a line of the scalar cache also holds the next row blocks' words: looked at too early, stale from then on)
            }
        }
        issue(kr);
        if (LDEPTH == 3) {
            if (prev2 >= 0) { wait_vmcnt<2 * NL>(); publish(prev2); }
            prev2 = prev1; prev1 = kr;
        } else {
            if (prev1 >= 0) { wait_vmcnt<NL>(); publish(prev1); }
            prev1 = kr;
        }
    }
    if (LDEPTH == 3 && prev2 >= 0) { wait_vmcnt<NL>(); publish(prev2); }
    wait_vmcnt<0>();
    if (prev1 >= 0) publish(prev1);
}

// max_push (common.h) as two selects instead of a compare-and-branch: larger value wins, on ties the smaller key
__device__ __forceinline__ void max_push_sel(float& best, int& key, float t, int k)
{
    const bool take = (t > best) | ((t == best) & (k < key));
    best = take ? t : best;
    key = take ? k : key;
}

// ---- far wave -------------------------------------------------------------------------------------------
// Per block k >= RING, one combined value per (row, chain) for the owner's diagonal phase: the NEAR tiles (k, k-RING-n) from LDS --
// cells staged by the loader, u from the ring's own LDS entries -- and the panels' partials of the far field (tiles 0 .. k-FAR0).
template <int MODE, int DIR, bool GRAD>
__device__ __forceinline__ void far_role(const SweepParams& P, int sg, char* lds, int fw)
{
    const int T = P.T, B = P.B, K = P.K;
    unsigned* const ctrl = P.ctrl;
    const u64* const farg = P.farg;
    const unsigned tag = P.tag;
    const size_t Bs = (size_t)B;
    const int lane = threadIdx.x & 63;
    const int r = spine_row(lane), ch = spine_chain(lane);
    const int c = P.c0 + sg * GS + ch;
    const bool cvalid = c < P.c1;
    const int cc = cvalid ? c : P.c0;
    float4* const far = (float4*)(lds + LDS_FAR);
    const bool cbase_is_first = sg == 0;
    const float* const ring = (const float*)(lds + LDS_RING);
    const float* const rd_base = ring + ch * 2;
    const int* const nready = (const int*)(lds + LDS_CTL) + NRBUF + RING;
    int* const fcons = (int*)(lds + LDS_CTL) + NRBUF + RING + NNSLOT + fw;          // this far wave's own counter
    float* const dScore = P.dScore;
    float gz = 0.f, lzc = 0.f;
    if (GRAD && NNEAR > 0 && cvalid) { gz = P.gout[(size_t)c * P.gstride] * P.gscale; lzc = P.logZ[c]; }
    __builtin_amdgcn_s_setprio(SEMICRF_FAR_PRIO);     // above the ring's shadow phase (1), below its diagonal phase (3)
    const bool fprobe = SEMICRF_PANEL_PROBES && (DBGF(P) & 16u) && cbase_is_first && lane == 0;
    // (NFARW far waves exist; a launch uses the first P.nfarw of them: SweepParams::nfarw)
    const int nfar = NNEAR > 0 ? NFARW : __builtin_amdgcn_readfirstlane(P.nfarw);
    if (fw >= nfar) return;
    for (int k = RING + fw; k < K; k += nfar) {
        if (fprobe && k < 64) P.ts[640 + k] = __builtin_amdgcn_s_memrealtime();
        const int prow = k * PB + r;
        const bool rvalid = cvalid && prow < T;
        const int prow_c = prow < T ? prow : T - 1;
        float aM = SEMICRF_NEG_INF, aS = 0.f;
        int aK = 0x7fffffff;
        // parts of block k: the panels' column parts of its far tiles 0 .. k-FAR0
        const int nparts = k >= FAR0 ? nparts_of(k - FAR0) : 0;
        u64 eq[4] = {0, 0, 0, 0};                       // the early request of parts 0..3 (see below)
        if (SEMICRF_FAR_EARLY && rvalid && nparts > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (i < nparts) eq[i] = load_granule(farg + ((size_t)i * T + prow) * Bs + c);
        }
        // ---- near tiles: cells (and alpha) of row block k into registers, then the slot goes back to the loader -------------
        float ncell[NNEAR > 0 ? NNEAR : 1][PB];
        float arow = 0.f;
        if (NNEAR > 0) {
            int spins = 0;
            while (lds_flag_load(nready + k % NNSLOT) != k + 1) {
                __builtin_amdgcn_s_sleep(1);
                if (spin_abort(ctrl, spins, SPIN_LIMIT_LDS, 11)) break;
            }
            const char* nb = lds + LDS_NEAR + (k % NNSLOT) * NEAR_SLOT_BYTES;
#pragma unroll
            for (int n = 0; n < NNEAR; ++n)
#pragma unroll
                for (int u = 0; u < PB; ++u) ncell[n][u] = *(const float*)(nb + n * TILE_BYTES + u * (PB * RS * 4) + (r * RS + ch) * 4);
            if (GRAD) arow = rvalid ? (((const float*)(nb + NNEAR * TILE_BYTES))[lane] - lzc) * LOG2E : 0.f;
            if (lane == 0) lds_flag_store(fcons, k + 1);          // in-order DS queue: after the reads above
        }
        if (fprobe && k < 64) P.ts[704 + k] = __builtin_amdgcn_s_memrealtime();
        // ---- near tiles (oldest column block first: a fixed merge order) ---------------------------------------------------------
        if (NNEAR > 0) {
            // GRAD stores as in the ring: wave-uniform base + one per-lane byte offset
            const int own0 = k * PB;
            const unsigned bvoff = DIR == 0 ? (unsigned)(((size_t)(prow_c - own0) * T * Bs + cc) * 4)
                                            : (unsigned)(((size_t)(T - 1 - prow_c) * Bs + cc) * 4);
#pragma unroll
            for (int n = NNEAR - 1; n >= 0; --n) {
                const int b = k - RING - n;
                if (b < 0) continue;
                float uq[PB];
#pragma unroll
                for (int u4 = 0; u4 < PB; u4 += 4) {
                    const int j = b * PB + u4;
                    const float* e = rd_base + (j % NPOS) * 8;
                    u64 w0, w1, w2, w3;
                    int spins = 0;
                    while (true) {
                        w0 = lds_load64(e); w1 = lds_load64(e + 2 * RS); w2 = lds_load64(e + 4 * RS); w3 = lds_load64(e + 6 * RS);
                        if (__all((int)(w0 >> 32) == j + 1 && (int)(w1 >> 32) == j + 2 && (int)(w2 >> 32) == j + 3 && (int)(w3 >> 32) == j + 4)) break;
                        __builtin_amdgcn_s_sleep(1);
                        if (spin_abort(ctrl, spins, SPIN_LIMIT_LDS, 12)) break;
                    }
                    uq[u4] = __uint_as_float((unsigned)w0); uq[u4 + 1] = __uint_as_float((unsigned)w1);
                    uq[u4 + 2] = __uint_as_float((unsigned)w2); uq[u4 + 3] = __uint_as_float((unsigned)w3);
                }
                if (fprobe && k < 64 && n == 0) P.ts[768 + k] = __builtin_amdgcn_s_memrealtime();
                if (MODE == 0) {
                    float t[PB];
#pragma unroll
                    for (int u = 0; u < PB; ++u) {
                        t[u] = fmaf(ncell[n][u], LOG2E, uq[u]);
                        if (GRAD && rvalid && P.bandWaves == 0) {
                            const int j = b * PB + u;
                            const size_t co = DIR == 0 ? ((size_t)own0 * T + (size_t)j) * Bs : (size_t)(T - 1 - j) * T * Bs;
                            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dScore + co), 0, 0x7fffffff, 0x00020000);
                            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(gz * fexp2(t[u] + arow)), rs, bvoff, 0, 0);
                        }
                    }
                    float mx = aM;
#pragma unroll
                    for (int u = 0; u < PB; u += 2) mx = fmaxf(mx, fmaxf(t[u], t[u + 1]));
                    float sum = aS * fexp2(aM - mx);
#pragma unroll
                    for (int u = 0; u < PB; ++u) sum += fexp2(t[u] - mx);
                    aM = mx; aS = sum;
                } else {
#pragma unroll
                    for (int u = 0; u < PB; ++u) max_push_sel(aM, aK, uq[u] + ncell[n][u], frame_of<DIR>(b * PB + u, T));
                }
            }
        }
        if (fprobe && k < 64) P.ts[832 + k] = __builtin_amdgcn_s_memrealtime();
        if (rvalid && nparts > 0) {
            // all parts are requested together (they complete in any order); the poll repeats for the missing ones.  The FIRST
            // request of the first four parts went out at the top of the iteration (eq[]): a request is a device-scope round
            // trip of ~2 us from this compute unit -- behind the loader's traffic -- whether or not the partial has long been
            // stored, and the near tiles' work hides it.
            for (int p0 = 0; p0 < nparts; p0 += 4) {
                const int np = nparts - p0 < 4 ? nparts - p0 : 4;
                u64 gq[4];
                bool have[4] = {false, false, false, false};
                int spins = 0;
                bool first = SEMICRF_FAR_EARLY && p0 == 0;
                while (true) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < np && !have[i])
                            gq[i] = first ? eq[i] : load_granule(farg + ((size_t)(p0 + i) * T + prow) * Bs + c);
                    first = false;
                    bool all = true;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (i < np && !have[i]) {
                            const bool ok = MODE == 0 ? (unsigned)(gq[i] >> 32) == tag : (unsigned)(gq[i] >> 48) == (tag & 0xffffu);
                            if (ok) have[i] = true;          // gq[i] keeps the granule: it is not requested again
                            else all = false;
                        }
                    if (all) break;
                    __builtin_amdgcn_s_sleep(1);
                    if (spin_abort(ctrl, spins, SPIN_LIMIT, 3)) break;
                }
                // the parts are merged in index order, not in the order they arrived in: the sum's last bit (and a tie
                // between two maxima) must not depend on timing
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < np && have[i]) {
                        if (MODE == 0) acc_push1(aM, aS, __uint_as_float((unsigned)gq[i]));
                        else max_push(aM, aK, __uint_as_float((unsigned)gq[i]), (int)((gq[i] >> 32) & 0xffffu));
                    }
            }
        }
        if (fprobe && k < 64) P.ts[896 + k] = __builtin_amdgcn_s_memrealtime();
        // (LSE: an empty accumulator -- rows past the end, ghost chains -- gives -inf + log2(0) = NaN, which nobody reads)
        const float val = MODE == 0 ? aM + flog2(aS) : aM;
        lds_store128(far + (k % NFAR) * 64 + lane, make_float4(val, __int_as_float(aK), __int_as_float(k + 1), 0.0f));   // one DS write: data + seq
        if (SEMICRF_PANEL_PROBES && (DBGF(P) & 16u) && cbase_is_first && lane == 0) P.ts[128 + k] = __builtin_amdgcn_s_memrealtime();   // chain probe
    }
}

// ---- ring waves -----------------------------------------------------------------------------------------
// The diagonal phase of the LSE sweeps (round 3).  Rows p = own0 .. own0+15 of the owner's block satisfy
//     y[r] = log2( 2^V[r] + sum_{j<r} 2^(y[j] + d[r][j]) ),   u[r] = y[r] + sp[r],
// V[r] = everything outside the block (shadow pushes + far partial, known when the phase starts), d[r][j] = log2 cell
// (r, j) + sp[j] (for j = r-1 the cell with the skip folded in, `wl`).  Walked in the log domain this is 16 dependent
// logaddexp steps of ~240 cycles each for a lone wave (two transcendentals and a cross-lane broadcast per step): 1.6 us per
// block, the pace of the whole head of the sweep.  Now:
//   1. a (max,+) pass over the block gives every row an exponent e[r] = max(Vmax[r], max_j e[j] + d[r][j]): the weight of the
//      best single path into row r, so y[r] - e[r] lies in [0, 16 log2 17];
//   2. with Y[r] = 2^(y[r] - e[r]) the recurrence is LINEAR with coefficients m[r][j] = 2^(e[j] + d[r][j] - e[r]) <= 1
//      (sixteen independent exponentials per lane): Y[r] = a[r] + sum_j Y[j] m[r][j], a[r] = S[r] 2^(Vmax[r] - e[r]) <= #terms;
//      nothing can overflow, what underflows is below 2^-24 of a term of size >= 1;
//   3. y[r] = e[r] + log2 Y[r].
// Both passes are 15 steps of TWO dependent vector instructions: a lane is (row r, chain) with the chain's 16 rows in one DPP
// row, so "the value of row j" is a row_newbcast operand (v_add_f32_dpp / v_mov_b32_dpp), no LDS round trip.
// tools/dpp_probe.hip measures the chains on their own.  The (max,+) sweeps keep their stepwise form (one add and one
// compare per step, bit-exact candidates).

template <int J>
__device__ __forceinline__ float row_bcast(float x)       // lane J of this lane's DPP row: row J of the block, same chain
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), 0x150 + J, 0xf, 0xf, false));
}
template <int J, int N, typename F>
__device__ __forceinline__ void static_for(F&& f)
{
    if constexpr (J < N) { f(IC<J>{}); static_for<J + 1, N>(f); }
}

template <int MODE, int DIR, bool GRAD, bool RINGBAND>
__device__ __forceinline__ void spine_role(const SweepParams& P, int sg, int ring_pos, char* lds)
{
    // kernel arguments are copied into locals: lambdas that capture the struct by reference make the
    // compiler spill it to scratch and reload fields inside the step loops
    const int T = P.T, B = P.B, K = P.K;
    const unsigned dbg = DBGF(P);
    unsigned* const ctrl = P.ctrl;
    u64* const ts = P.ts;
    unsigned* const ug = P.ug;
    float* const u_out = P.u_out;
    float* const last_out = P.last_out;
    int* const code = P.code;
    float* const dScore = P.dScore;
    float* const dNoise = P.dNoise;
    const u64* const pathg = P.pathg;
    float* const pathOut = P.pathOut;
    const unsigned ptag = P.tag;
    const int rw = __builtin_amdgcn_readfirstlane(ring_pos);                // position of the wave in the ring
    const int lane = threadIdx.x & 63;
    const int r = spine_row(lane), ch = spine_chain(lane);
    const int c = P.c0 + sg * GS + ch;
    const bool cvalid = c < P.c1;
    const int cc = cvalid ? c : P.c0;
    const size_t Bs = (size_t)B;
    const bool trace = (dbg & 16u) && sg == 0 && lane == 0;
    float* const ring = (float*)(lds + LDS_RING);
    const float* rd_base = ring + ch * 2;                                   // + (j % NPOS) * 8 floats
    const int* const ready = (const int*)(lds + LDS_CTL);
    int* const cons = (int*)(lds + LDS_CTL) + NRBUF;
    const float4* const far = (const float4*)(lds + LDS_FAR);
    float gz = 0.f, lzc = 0.f;
    float nadd = 0.f;     // logProb's backward: + gout on every gap (the path score's cum[T-1] term; the product, then the sum: two
                          // roundings, exactly what a separate `dNoise += gout` pass over the stored marginal gave)
    if (GRAD && cvalid) {                                                                                 // the only global loads of a ring wave
        const float go = P.gout[(size_t)c * P.gstride];
        gz = go * P.gscale; lzc = P.logZ[c]; nadd = P.noiseAdd * go;
    }
    // GRAD: who stores the marginals of the band (the cells the ring itself reads)?  The panel workgroups' band waves when the
    // launch has them (band_role: whole lines, off the ring's critical path); the ring waves themselves otherwise (short sequences)
    // (a template parameter: as a run-time flag it put a branch around every cell's store and cut the shadow batch into 16 pieces)
    constexpr int ringBand = GRAD && RINGBAND ? 1 : 0;
    __builtin_amdgcn_s_setprio(1);

    for (int k = rw; k < K; k += RING) {
        u64* ev = ts + T + (size_t)k * 8;          // debug events of this block (8 slots)
        if (trace) ev[0] = __builtin_readcyclecounter();
        const int prow = k * PB + r;
        const bool rvalid = cvalid && prow < T;
        const int prow_c = prow < T ? prow : T - 1;
        const int frow = frame_of<DIR>(prow_c, T);
        const int own0 = k * PB;
        const int slot = k % NRBUF;

        // ---- the row block's cells and constants, staged in LDS by the loader wave ----------------
        {
            int spins = 0;
            while (lds_flag_load(ready + slot) != k + 1) {
                __builtin_amdgcn_s_sleep(1);
                if (spin_abort(ctrl, spins, SPIN_LIMIT_LDS, 7)) break;
            }
        }
        const char* tb = lds + LDS_TILES + slot * RING * TILE_BYTES + (r * RS + ch) * 4;
        auto read_tile = [&](int i) -> SpineBlk {
            SpineBlk o;
#pragma unroll
            for (int u = 0; u < PB; ++u) o.v[u] = *(const float*)(tb + i * TILE_BYTES + u * (PB * RS * 4));
            return o;
        };
        SpineBlk X[RING];        // X[i]: columns of block k - (RING-1) + i (the last one is the own block)
#pragma unroll
        for (int i = 0; i < RING; ++i) X[i] = read_tile(i);
        const float* cst = (const float*)(lds + LDS_CONST + slot * NCONST * 256) + lane;
        const float dcell = cst[0], nzl = cst[64], vfl = cst[128];
        // first sub-diagonal cell of this lane's row: column r-1 of the own tile (last column of the previous tile for row 0)
        const float s1own = *(const float*)(lds + LDS_TILES + (slot * RING + RING - 1) * TILE_BYTES +
                                            ((((r > 0 ? r - 1 : 0) * PB) + r) * RS + ch) * 4);
        const float s1 = r == 0 ? X[RING - 2].v[PB - 1] : s1own;
        if (lane == 0) lds_flag_store(cons + rw, k + 1);          // in-order DS queue: after the reads above

        // GRAD stores: wave-uniform base (SGPR) + one per-lane byte offset
        //   DIR 0: cell(prow, j) = ((own0 + r)*T + j)*B          DIR 1: cell(prow, j) = ((T-1-j)*T + (T-1-prow))*B
        const unsigned bvoff = DIR == 0 ? (unsigned)(((size_t)(prow_c - own0) * T * Bs + cc) * 4)
                                        : (unsigned)(((size_t)(T - 1 - prow_c) * Bs + cc) * 4);
        auto col_off = [&](int j) -> size_t {
            return DIR == 0 ? ((size_t)own0 * T + (size_t)j) * Bs : (size_t)(T - 1 - j) * T * Bs;
        };
        auto grad_store = [&](int j, float v) {
#ifdef SEMICRF_NO_BAND_GSTORE
            (void)j; asm volatile("" ::"v"(v)); return;      // timing experiment: the band's marginals are computed but not stored
#endif
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(dScore + col_off(j)), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), rs, bvoff, 0, 0);
        };

        // ---- per-row constants ----------------------------------------------------------------
        float sp = 0.f;   // LSE: softplus2(diag); MAX: diag
        float nz = 0.f;   // noise between prow-1 and prow
        float wl = 0.f;   // LSE: log2(2^s[prow][prow-1] + 2^n): first sub-diagonal cell with the skip folded in
        float draw = 0.f;
        if (rvalid) {
            draw = dcell * LOG2E;
            sp = MODE == 0 ? softplus2(dcell) : dcell;
            if (prow >= 1) {
                nz = nzl;
                if (MODE == 0) {
                    const float a0 = s1 * LOG2E, b0 = nz * LOG2E;
                    wl = fmaxf(a0, b0) + flog2(1.0f + fexp2(-fabsf(a0 - b0)));
                }
            }
        }
        // GRAD: marginal(prow, j) = gz * exp2(t + arow) with t = u[j] + cell*log2e, arow = (alpha[frame] - logZ)*log2e
        const float arow = (GRAD && rvalid) ? (vfl - lzc) * LOG2E : 0.f;
        // LSE: the block's own triangle as log2 coefficients d[j] = cell(r, j) + sp[j] (column r-1: the cell with the skip
        // folded in; columns >= r: none) -- known at the start of the iteration, long before the phase that needs them
        float d[PB - 1];
        if (MODE == 0) {
            static_for<0, PB - 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const float cj = r > j + 1 ? X[RING - 1].v[j] * LOG2E : (r == j + 1 ? wl : SEMICRF_NEG_INF);
                d[j] = (rvalid ? cj : SEMICRF_NEG_INF) + row_bcast<j>(sp);
            });
        }
        if (trace) ev[1] = __builtin_readcyclecounter();

        float aM = SEMICRF_NEG_INF, aS = 0.f;
        int aK = 0x7fffffff;

        // wait for positions j .. j+3 in the ring and return this lane's chain (one check per four positions: a wave that
        // lags behind its ring mate catches up at the cost of the pushes alone).  Every entry is ONE 8-byte word {u, seq},
        // written with one DS store and read with ONE 64-bit load (lds_load64: the compiler splits a plain float2 load into
        // separate loads of value and seq when they are used apart -- the value read BEFORE its seq is a torn read);
        // the LSE owner writes all 16 positions of a block with one instruction, so each entry's own seq is checked.
        auto ring_get4 = [&](int j, float (&uo)[4]) {
            const float* e = rd_base + (j % NPOS) * 8;                         // j % 4 == 0: no wrap inside the group
            u64 w0, w1, w2, w3;
            auto fetch = [&]() { w0 = lds_load64(e); w1 = lds_load64(e + 2 * RS); w2 = lds_load64(e + 4 * RS); w3 = lds_load64(e + 6 * RS); };
            auto ok = [&]() {
                return (int)(w0 >> 32) == j + 1 && (int)(w1 >> 32) == j + 2 && (int)(w2 >> 32) == j + 3 && (int)(w3 >> 32) == j + 4;
            };
            fetch();
            if (!__all(ok())) {
                int spins = 0;
                while (true) {
                    __builtin_amdgcn_s_sleep(1);
                    fetch();
                    if (__all(ok())) break;
                    if (spin_abort(ctrl, spins, SPIN_LIMIT_LDS, 2)) break;
                }
            }
            uo[0] = __uint_as_float((unsigned)w0); uo[1] = __uint_as_float((unsigned)w1);
            uo[2] = __uint_as_float((unsigned)w2); uo[3] = __uint_as_float((unsigned)w3);
        };

        // ---------------- shadow phase: apply a block published by a ring mate ---------------------
        // `last`: block k-1, whose final column own0-1 is the first sub-diagonal cell of row 0 (skip folded in)
        auto shadow = [&](const SpineBlk& X, int b, bool last) {
            if (trace && b >= k - 3) ev[3 + (b - (k - 3))] = __builtin_readcyclecounter();
            if (MODE == 0) {
                // log-sum-exp: the block's 16 terms in one batch -- all terms first, ONE exact maximum, then 16 independent
                // exponentials relative to it (a push per term costs a dependent compare / select chain each: 16 x ~45 cycles of
                // a lone wave; the `last` block's batch sits on the ring's critical path, between a mate's publish and the own
                // diagonal phase)
                float t[PB];
#pragma unroll
                for (int u4 = 0; u4 < PB; u4 += 4) {
                    float uq[4];
                    ring_get4(b * PB + u4, uq);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int u = u4 + q;
                        const int j = b * PB + u;
                        float p = fmaf(X.v[u], LOG2E, uq[q]);
                        if (GRAD && ringBand && rvalid && j < prow) grad_store(j, gz * fexp2(p + arow));
                        if (last && u == PB - 1 && r == 0) {
                            if (GRAD && rvalid)       // noise marginal of the gap between prow-1 and prow
                                dNoise[(size_t)gap_of<DIR>(prow, T) * Bs + c] = gz * fexp2(uq[q] + nz * LOG2E + arow) + nadd;
                            p = uq[q] + wl;
                        }
                        t[u] = p;
                    }
                }
                float mx = aM;
#pragma unroll
                for (int u = 0; u < PB; u += 2) mx = fmaxf(mx, fmaxf(t[u], t[u + 1]));
                float sum = aS * fexp2(aM - mx);               // empty accumulator: 0 * exp2(-inf) = 0 (mx is finite: the terms are)
#pragma unroll
                for (int u = 0; u < PB; ++u) sum += fexp2(t[u] - mx);
                aM = mx; aS = sum;
            } else {
#pragma unroll
                for (int u4 = 0; u4 < PB; u4 += 4) {
                    float uq[4];
                    ring_get4(b * PB + u4, uq);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int u = u4 + q;
                        const int j = b * PB + u;
                        const float uv = uq[q];
                        const int key = frame_of<DIR>(j, T);
                        if (last && u == PB - 1 && r == 0) max_push_sel(aM, aK, uv + nz, -1);   // the skip candidate (key -1 wins every tie)
                        max_push_sel(aM, aK, uv + X.v[u], key);
                    }
                }
            }
        };

#pragma unroll
        for (int i = 0; i < RING - 1; ++i)
            if (k - (RING - 1) + i >= 0) shadow(X[i], k - (RING - 1) + i, i == RING - 2);

        // ---------------- diagonal phase: finalise the 16 positions of block k ------------------------
        __builtin_amdgcn_s_setprio(3);       // the dependent chain goes before everything else on this SIMD
        if (trace) ev[6] = __builtin_readcyclecounter();
        if (k >= RING && !(dbg & 1u)) {
            // combined far-field partial of this (row, chain), left in LDS by the far wave
            const float4* fe = far + (k % NFAR) * 64 + lane;
            float4 f = lds_load128(fe);                          // one 16-byte read: data and seq together
            if (!__all(__float_as_int(f.z) == k + 1)) {
                int spins = 0;
                while (true) {
                    __builtin_amdgcn_s_sleep(1);
                    f = lds_load128(fe);
                    if (__all(__float_as_int(f.z) == k + 1)) break;
                    if (spin_abort(ctrl, spins, SPIN_LIMIT_LDS, 8)) break;
                }
            }
            if (rvalid) {
                if (MODE == 0) acc_push1(aM, aS, f.x);
                else max_push_sel(aM, aK, f.x, __float_as_int(f.y));
            }
        }

        if (SEMICRF_PANEL_PROBES && trace) { ts[64 + k] = __builtin_amdgcn_s_memrealtime(); ev[2] = __builtin_readcyclecounter(); }   // chain probe: far partial in hand
        int mykey = -1;
        float mine = 0.f;                    // this lane's finished u (log2 units for LSE)
        if (MODE == 0) {
            // everything outside the block: S terms relative to their maximum mx (the first position has the empty path, weight 1;
            // rows past the end and ghost chains run on finite stand-ins, nobody reads them)
            float mx = aM, S = aS;
            if (prow == 0 || !rvalid) { mx = 0.0f; S = 1.0f; }
            // 1. exponents: the best single path into every row
            float e = mx;
            static_for<0, PB - 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                e = fmaxf(e, row_bcast<j>(e) + d[j]);      // lane j's e is final: it only ever took terms of rows < j
            });
            // 2. linear solve in units of 2^e.  The coefficient's exponent is formed as (e[j] - e[r]) + d[j]: the difference of
            // the two large numbers first (exact or nearly so), then the small one -- one rounding at the magnitude of d, where
            // the log-domain step rounded every term at the magnitude of u (ulp(1000) ~ 6e-5: 4e-5 relative in the sum)
            float acc = S * fexp2(mx - e);
            static_for<0, PB - 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                acc = fmaf(row_bcast<j>(acc), fexp2((row_bcast<j>(e) - e) + d[j]), acc);
            });
            // 3. back to the log domain; one 8-byte DS write per lane publishes the whole block to the ring mates
            mine = e + flog2(acc) + sp;
            lds_store64(ring + ch * 2 + ((own0 + r) % NPOS) * 8, mine, own0 + r + 1);
            __builtin_amdgcn_s_setprio(1);
            if (GRAD) {
                // marginals of the block's own triangle (unless the band waves write them) and of its gaps (off the critical path:
                // the block is out).  The gap in front of row r needs u of row r - 1: ONE row shift and one store for the 15 gaps
                // inside the block (until round 5: fifteen stores with four lanes each)
                if (ringBand) {
                    static_for<0, PB - 1>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        const float uj = row_bcast<j>(mine);
                        if (rvalid && r > j) grad_store(own0 + j, gz * fexp2(fmaf(X[RING - 1].v[j], LOG2E, uj) + arow));
                    });
                }
                const float um1 = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mine), 0x111, 0xf, 0xf, false));   // row_shr:1
                if (rvalid && r >= 1) dNoise[(size_t)gap_of<DIR>(prow, T) * Bs + c] = gz * fexp2(um1 + nz * LOG2E + arow) + nadd;
            }
        } else {
            // (max,+): the same walk over the block's triangle on DPP row broadcasts.  Row j's value is final after step j-1 (the
            // cells of columns >= r are masked to -inf), so all rows advance together: per step ONE broadcast of u[j] = best[j] +
            // relu-select(diag[j]) and two branch-free candidate selects -- the skip (key -1: it wins every tie) for row j+1 and
            // the interval (j, r) for the rows below.  Candidates are the same single fp32 adds as before (u + noise, u + cell)
            // and ties go to the smaller key: the decode stays bit-identical.  (The stepwise form -- an LDS broadcast, a DS
            // publish and two compare-branch pushes per position -- took 465 cycles per position, 3.1 us per block; the
            // log-sum-exp sweeps need 1.4 since round 3.)
            float best = aM;
            int key = aK;
            if (prow == 0 || !rvalid) { best = 0.0f; key = -1; }      // the first position has the empty path; rows past the end: stand-ins
            const float spr = sp > 0.0f ? sp : 0.0f;                   // s * (s > 0), reference :29, :49-51
            static_for<0, PB - 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                const float uj = row_bcast<j>(best + spr);
                const int keyj = frame_of<DIR>(own0 + j < T ? own0 + j : T - 1, T);
                const float ts = uj + nz;
                const bool c1 = (r == j + 1) & ((ts > best) | ((ts == best) & (key > -1)));
                best = c1 ? ts : best;
                key = c1 ? -1 : key;
                const float tc = uj + (r > j ? X[RING - 1].v[j] : SEMICRF_NEG_INF);
                const bool c2 = (tc > best) | ((tc == best) & (keyj < key));
                best = c2 ? tc : best;
                key = c2 ? keyj : key;
            });
            mine = best + spr;
            mykey = key;
            lds_store64(ring + ch * 2 + ((own0 + r) % NPOS) * 8, mine, own0 + r + 1);
            __builtin_amdgcn_s_setprio(1);
        }
        if (trace) ev[7] = __builtin_readcyclecounter();

        // ---- once per block: publish the 16 finished positions to HBM -----------------------------
        if (rvalid) {
            if (k == K - 1 && lds_flag_load(&s_abort) != 0) mine = __uint_as_float(0x7fc00000u);     // a wait timed out: poison the results
            if (k < K - FAR0 || (GRAD && ringBand == 0)) {      // the far field of block k' ends at block k' - FAR0: only the band waves (GRAD) read the last FAR0 blocks' u
                unsigned ub = __float_as_uint(mine);
                if (ub == U_EMPTY) ub = 0x7fc00000u;            // keep the one reserved pattern free (NaN input scores)
                __hip_atomic_store(ug + (size_t)prow * Bs + c, ub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const float sc = MODE == 0 ? LN2 : 1.0f;
            if (u_out) u_out[(size_t)frow * Bs + c] = mine * sc;
            if (last_out && prow == T - 1) last_out[c] = mine * sc;
            if (MODE == 0 && DIR == 0 && !GRAD && pathg && prow == T - 1) {
                // logProb: the chain's path score was left by a path wave long ago (it needs no u); the only global LOAD of a ring
                // wave, after its last publish
                u64 gr;
                int spins = 0;
                bool ok = true;
                while (true) {
                    gr = load_granule(pathg + c);
                    if ((unsigned)(gr >> 32) == ptag) break;
                    __builtin_amdgcn_s_sleep(8);
                    if (spin_abort(ctrl, spins, SPIN_LIMIT, 15)) { ok = false; break; }
                }
                pathOut[c] = ok ? __uint_as_float((unsigned)gr) - mine * sc : __uint_as_float(0x7fc00000u);
            }
            if (GRAD)      // diagonal: gout * exp(alpha + beta - logZ + s - 2 softplus(s))
                dScore[((size_t)frow * T + frow) * Bs + c] = gz * fexp2(arow + mine + draw - 2.0f * sp);
            if (MODE == 1) code[(size_t)c * T + frow] = (mykey + 1) | (sp > 0.0f ? 0x40000000 : 0);
        }
        if (SEMICRF_PANEL_PROBES && trace) ts[k] = __builtin_amdgcn_s_memrealtime();                 // chain probe: u published
    }
}

// ---------------------------------------------------------------------------------------------
// PANEL role (per wave)
// ---------------------------------------------------------------------------------------------
#ifndef SEMICRF_GRAD_AUX
#define SEMICRF_GRAD_AUX 2
#endif
constexpr int GRAD_AUX = SEMICRF_GRAD_AUX;   // nt: the gradient is written once
#ifndef SEMICRF_CELL_AUX
#define SEMICRF_CELL_AUX 2
#endif
constexpr int CELL_AUX = SEMICRF_CELL_AUX;   // nt: every cell is read once -- keep the stream from evicting the (re-read) u values from L2
#ifndef SEMICRF_PNS
#define SEMICRF_PNS 3
#endif
constexpr int PNS = SEMICRF_PNS;             // LDS stages per panel wave (tiles fetched ahead)
constexpr int PSTAGE_BYTES = 10240;          // 8 KB of cells + 2 KB of u values
#ifndef SEMICRF_PW_MAX
#define SEMICRF_PW_MAX 5
#endif
constexpr int PW_MAX = SEMICRF_PW_MAX;       // panel waves per workgroup (LDS: PW_MAX * PNS * PSTAGE_BYTES = 150 KB; four of them run unless the sweep is throughput-bound)
constexpr int LDS_PANEL_BYTES = PW_MAX * PNS * PSTAGE_BYTES;
// Spines per spine workgroup (round 4).  A spine -- ring of RING waves, its loader(s) and far wave(s), LDS_SPINE_BYTES of LDS -- is
// self-contained; SPH of them share a workgroup, i.e. a compute unit: the tail of the headline sweep is bound by the far field's
// streaming rate, ~26 GB/s per compute unit that is NOT a spine, and with one spine per unit 88 of the 256 were spines at
// NBatch = 352 (the panels alone, fed without any dependency, need 147 us there; the whole sweep took 175).
constexpr int SPH = SEMICRF_SPH;
// waves of one spine.  With one spine per workgroup only the FIRST far wave sits behind the loader; the others are the workgroup's
// LAST waves: as wave 6 a second far wave pushed the two streaming waves from SIMDs 2 and 3 to SIMDs 3 and 0 -- next to ring wave 0
// and the loader -- and the gradient sweep, whose spare waves stream from block 12 on, lost 2 - 7 % although it uses one far wave
constexpr int HWAVES = SPH == 1 ? RING + NLOADER + 1 : RING + NLOADER + NFARW;
constexpr int LDS_HALF = (LDS_SPINE_BYTES + 1023) / 1024 * 1024;              // LDS of one spine
constexpr int LDS_HYBRID_PANEL = SPH * LDS_HALF;                              // stages of a spine workgroup's panel waves
constexpr int HPW_MAX = (160 * 1024 - 512 - LDS_HYBRID_PANEL) / (PNS * PSTAGE_BYTES) < NT / 64 - SPH * HWAVES
                            ? (160 * 1024 - 512 - LDS_HYBRID_PANEL) / (PNS * PSTAGE_BYTES) : NT / 64 - SPH * HWAVES;
static_assert(SPH * HWAVES <= NT / 64 && SPH * LDS_HALF <= 160 * 1024 - 512 && (GP / GS) % SPH == 0, "wave roles");
constexpr int LDS_HYBRID_BYTES = LDS_HYBRID_PANEL + (HPW_MAX > 0 ? HPW_MAX : 0) * PNS * PSTAGE_BYTES;
constexpr int LDS_DYN_MAX2 = SPH * LDS_HALF > LDS_PANEL_BYTES ? SPH * LDS_HALF : LDS_PANEL_BYTES;
constexpr int LDS_DYN_BYTES = LDS_HYBRID_BYTES > LDS_DYN_MAX2 ? LDS_HYBRID_BYTES : LDS_DYN_MAX2;   // > half of the CU's 160 KB: one workgroup per CU
// A wave owns rows pi = 16k + 4*q4 + rr (rr < 4) of position block k for 32 chains.  lane = slot*8 + q8:
// q8 selects 4 of the 32 chains (8 consecutive lanes read one 128-byte line), slot selects the columns
// pj = 16m + slot + 8h (h < 2) of tile m.  Cells and u values of the next tiles are requested before tile m is
// processed.  The same mapping serves both directions (only cell_index differs).
// slow path of the panel accumulation: move the reference above tm when tm is within RESC_EARLY of the limit (see RESC_LIFT)
__device__ __forceinline__ void acc_lift(float& M, float& S, float tm)
{
    const bool need = tm > M + (RESC_HI - RESC_EARLY);                       // true for the empty accumulator (M = -inf)
    const float newM = need ? floorf(tm) + RESC_LIFT : M;
    const int sh = (int)fmaxf(M - newM, -4096.0f);                            // an integer; 0 when nothing moves
    S = __builtin_ldexpf(S, sh);                                              // exact; empty: S = 0 stays 0
    M = newM;
}

// ---- panel helpers (free functions: a lambda that calls another lambda keeps its closure in scratch once
// the body contains operations the optimiser treats as memory writes -- the global->LDS loads) ----------------
struct PanelGeom {
    int pirow[4];          // rows of the task, clamped to T-1
    unsigned voff;         // per-lane byte offset of the PROCESSING mapping (also the gradient stores)
    unsigned fvoff;        // per-lane byte offset of the FETCH mapping
    unsigned gvoff;        // per-lane byte offset into the u values
};

// element offset of the tile base
template <int DIR>
__device__ __forceinline__ size_t panel_tile_off(const PanelGeom& G, int m, int T, size_t Bs)
{
    return DIR == 0 ? ((size_t)G.pirow[0] * T + (size_t)m * PB) * Bs
                    : ((size_t)(T - 1 - (m * PB + 15)) * T + (size_t)(T - 1 - G.pirow[3])) * Bs;
}
// wave-uniform byte offset of cell group (rr, h) in the processing mapping
template <int DIR>
__device__ __forceinline__ unsigned panel_soff(const PanelGeom& G, int rr, int h, int T, size_t Bs)
{
    return DIR == 0 ? (unsigned)((((size_t)(G.pirow[rr] - G.pirow[0]) * T + 8 * h) * Bs) * 4)
                    : (unsigned)((((size_t)(1 - h) * 8 * T + (G.pirow[3] - G.pirow[rr])) * Bs) * 4);
}
// u of a tile's 16 columns (2 KB: the value is its own flag, U_EMPTY until published).
// COHERENT = false: ordinary cached loads.  u is written once and re-read by every task of the launch;
// device-scope (sc1) loads would fetch it through the fabric every time.  A stale cache line can only show
// U_EMPTY (the launch's fill, visible since the kernel boundary; 4-byte values are written atomically), which
// sends the reader to the COHERENT retry.
template <bool COHERENT>
__device__ __forceinline__ void panel_fetch_gran(__amdgpu_buffer_rsrc_t ursrc, char* stage, unsigned gvoff, int m, int B)
{
#pragma unroll
    for (int h = 0; h < 2; ++h) {       // positions 16m + 8h + slot, 4 chains (16 bytes) per lane
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane((m * PB + 8 * h) * B * 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(ursrc, (lds_void_t*)(stage + 8192 + h * 1024), 16, gvoff, so, 0, COHERENT ? 16 : 0);   // 16 = sc1
    }
}
// FETCH mapping (global -> LDS, 1 KB per instruction, lane-linear in LDS):
//   DIR 0: instruction e = (rr, h) fetches what its lanes process (8 columns x 128 B of one matrix row);
//   DIR 1: instruction e fetches columns pj = 16m + 2e + pjsub for the 4 rows pi (4 x 128 B contiguous per pj):
//          lane = (pisub, pjsub, q8), same tile base, lane part ((1-pjsub)*T + (pi_3 - pi_pisub))*B, soffset (14-2e)*T*B.
//          The processing lane reads its cells back from ((slot>>1) + 4h)*1024 + rr*256 + (slot&1)*128 + q8*16.
template <int DIR, int AUX>
__device__ __forceinline__ void panel_fetch_cells_aux(const float* score, const PanelGeom& G, char* stage, int m, int T, size_t Bs)
{
    // The tile base and the piece offsets are wave-uniform by construction; said explicitly, because a load whose descriptor
    // or offset the compiler takes for divergent (task state that passed a lane-dependent loop exit) is wrapped in a
    // waterfall loop -- ten of them per tile.
    const size_t toff = panel_tile_off<DIR>(G, m, T, Bs);
    const size_t toff_u = (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)toff) |
                          ((size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(toff >> 32)) << 32);
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(score + toff_u), 0, 0x7fffffff, 0x00020000);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const unsigned so = (unsigned)__builtin_amdgcn_readfirstlane(
            (int)(DIR == 0 ? panel_soff<DIR>(G, e >> 1, e & 1, T, Bs) : (unsigned)(((size_t)(14 - 2 * e) * T * Bs) * 4)));
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void_t*)(stage + e * 1024), 16, G.fvoff, so, 0, AUX);
    }
}
// nt (wave-uniform run-time choice; the cache policy is an immediate of the instruction): see cell_policy_nt
template <int DIR>
__device__ __forceinline__ void panel_fetch_cells(const float* score, const PanelGeom& G, char* stage, int m, int T, size_t Bs, bool nt)
{
    if (nt) panel_fetch_cells_aux<DIR, CELL_AUX>(score, G, stage, m, T, Bs);
    else panel_fetch_cells_aux<DIR, 0>(score, G, stage, m, T, Bs);
}
__device__ __forceinline__ void panel_read_gran(unsigned addr, v4u& g0, v4u& g1)
{
    asm volatile("ds_read_b128 %0, %2\n\tds_read_b128 %1, %2 offset:1024\n\ts_waitcnt lgkmcnt(0)"
                 : "=&v"(g0), "=&v"(g1)
                 : "v"(addr));
}
template <int DIR>
__device__ __forceinline__ void panel_read_cells(unsigned addr, v4u (&o)[8])
{
    constexpr int XR = DIR == 0 ? 2048 : 256, XH = DIR == 0 ? 1024 : 4096;      // LDS offset of (rr, h): rr*XR + h*XH
    asm volatile("ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%10\n\t"
                 "ds_read_b128 %2, %8 offset:%11\n\tds_read_b128 %3, %8 offset:%12\n\t"
                 "ds_read_b128 %4, %8 offset:%13\n\tds_read_b128 %5, %8 offset:%14\n\t"
                 "ds_read_b128 %6, %8 offset:%15\n\tds_read_b128 %7, %8 offset:%16\n\t"
                 "s_waitcnt lgkmcnt(0)"
                 : "=&v"(o[0]), "=&v"(o[1]), "=&v"(o[2]), "=&v"(o[3]), "=&v"(o[4]), "=&v"(o[5]), "=&v"(o[6]), "=&v"(o[7])
                 : "v"(addr), "n"(0 * XR + 0 * XH), "n"(0 * XR + 1 * XH), "n"(1 * XR + 0 * XH), "n"(1 * XR + 1 * XH),
                   "n"(2 * XR + 0 * XH), "n"(2 * XR + 1 * XH), "n"(3 * XR + 0 * XH), "n"(3 * XR + 1 * XH));
}
// wait until at most `younger` vector-memory operations are outstanding (rounded down to an encodable step)
__device__ __forceinline__ void panel_wait_younger(int y)
{
    if (y >= 44) wait_vmcnt<44>();
    else if (y >= 36) wait_vmcnt<36>();
    else if (y >= 28) wait_vmcnt<28>();
    else if (y >= 20) wait_vmcnt<20>();
    else if (y >= 10) wait_vmcnt<10>();
    else wait_vmcnt<0>();
}

// ---- task selection -----------------------------------------------------------------------------------------------
// Block k = FAR0 + q has q + 1 far tiles, cut at fixed columns into FULL parts of TPT tiles and one LAST part (see
// part_tiles), each split in four row quarters and 32-chain groups.  ONE queue in (block, part, chain group, row
// quarter) order: an atomic counter.  (Per-part earliest-deadline queues, full parts handed out early, next-task
// prefetch and "recent-tile" waves on the spine CUs were all measured slower, DESIGN.md section 3 "Round 2".)
struct PanelTask { int k, part, g, q4; };

// (block, part) number tt -> block k and column part
__device__ __forceinline__ void panel_tt_decode(int tt, PanelTask& t)
{
    if (tt < LEADT - 1) { t.part = 0; t.k = FAR0 + tt; return; }      // the first LEADT - 1 blocks: one part
    tt -= LEADT - 1;
    int a = 0;
    while (tt >= TPT * (a + 1) * (a + 2) / 2) ++a;          // group a: blocks with a+1 parts
    tt -= TPT * a * (a + 1) / 2;
    const int q = a * TPT + tt / (a + 1) + LEADT - 1;
    t.part = tt % (a + 1);
    t.k = FAR0 + q;
}
// queue position -> task
__device__ __forceinline__ void panel_task_decode(const SweepParams& P, int task, PanelTask& t)
{
    t.q4 = task & 3;
    const int t2 = task >> 2;
    t.g = t2 % P.nPanelGroups;
    panel_tt_decode(t2 / P.nPanelGroups, t);
}

__device__ __forceinline__ bool panel_next_task(const SweepParams& P, PanelTask& t)
{
    int task = 0;
    if ((threadIdx.x & 63) == 0) task = (int)(atomicAdd(P.ctrl + 2, 1u) + 1u);        // counters start at 0xffffffff (one 0xff fill)
    task = __builtin_amdgcn_readfirstlane(task) + P.taskBase;
    if (task >= P.nTasks) return false;
    panel_task_decode(P, task, t);
    return true;
}

template <int MODE, int DIR, bool GRAD>
__device__ __forceinline__ void panel_role(const SweepParams& P, char* lds, int wslot, int hist_role = 0, int first_task = -1)
{
    const int T = P.T, B = P.B;
    const int c0 = P.c0, c1 = P.c1;
    const unsigned dbg = DBGF(P);
    unsigned* const ctrl = P.ctrl;
    u64* const farg = P.farg;
    const float* const vfwd = P.vfwd;
    const float* const logZp = P.logZ;
    const float* const goutp = P.gout;
    float* const dScore = P.dScore;
    const int lane = threadIdx.x & 63;
    const int slot = lane >> 3, q8 = lane & 7;
    const size_t Bs = (size_t)B;
    const float* __restrict__ score = P.score;
    const unsigned tag = P.tag;
    const auto ursrc = __builtin_amdgcn_make_buffer_rsrc((void*)P.ug, 0, (int)((size_t)T * Bs * 4), 0x00020000);
    const bool cell_nt = __builtin_amdgcn_readfirstlane(P.cellNT) != 0, grad_nt = __builtin_amdgcn_readfirstlane(P.gradNT) != 0;

    // geometry of a task's tiles (see "addressing" below)
    auto geom_of = [&](const PanelTask& t, PanelGeom& G) {
        const int pb = t.k * PB + t.q4 * 4;
        const int cc = c0 + t.g * GP + q8 * 4;
        const int ccl = cc < c1 ? cc : c0;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) G.pirow[rr] = pb + rr < T ? pb + rr : T - 1;
        G.voff = DIR == 0 ? (unsigned)((slot * B + ccl) * 4) : (unsigned)(((size_t)(7 - slot) * T * Bs + ccl) * 4);
        const int f_pisub = lane >> 4, f_pjsub = (lane >> 3) & 1;
        const int f_pirow = pb + f_pisub < T ? pb + f_pisub : T - 1;
        G.fvoff = DIR == 0 ? G.voff : (unsigned)((((size_t)(1 - f_pjsub) * T + (G.pirow[3] - f_pirow)) * Bs + ccl) * 4);
        G.gvoff = (unsigned)((slot * B + ccl) * 4);
    };
    // vector-memory operations issued so far by this wave (a lower bound: the waits below may only under-count the operations
    // younger than the fetch they wait for), and its value right after each stage's fetch.  They run on across tasks.
    int issued = 0, mark0 = 0, mark1 = 0, mark2 = 0;
    // task timeline probe (probe build, dbg & 256; tools/task_trace.py): the wave that drew task 0 stamps 5 words per task at ts[3T/2 + 5n]
    bool tracer = false;
    int tn = 0;

    while (true) {
        // ---- next task: (k, part, g, q4); a task only waits on spine progress below k-3 ----
        PanelTask tk;
        if (first_task >= 0 && first_task < P.nTasks) {
            panel_task_decode(P, first_task, tk);
#if SEMICRF_STAGGER > 0
            // Staggered start (round 4): the ~700 panel waves of a launch all know their first task and used to request its first
            // tiles in the same microsecond -- 20 MB that the spines' loaders (and the first hand-offs) queued behind: the first 8
            // blocks took 20 - 27 us where the ring alone needs 11.  A wave whose first task belongs to block k waits roughly
            // until the ring can be there.
            for (int i = 0, n = tk.k - FAR0 < 48 ? tk.k - FAR0 : 48; i < n; ++i) __builtin_amdgcn_s_sleep(SEMICRF_STAGGER);
#endif
        } else if (!panel_next_task(P, tk)) break;
        first_task = -1;
        if (SEMICRF_PROBE_TASKS && (dbg & 256u) && tk.k == RING && tk.part == 0 && tk.g == 0 && tk.q4 == 0) tracer = true;
        u64* const tsp = P.ts + (3 * T) / 2 + 5 * tn;
        const bool tr = SEMICRF_PROBE_TASKS && tracer && lane == 0 && 5 * tn + 5 <= T / 2;
        if (tr) tsp[0] = __builtin_amdgcn_s_memrealtime();
        const int q = tk.k - FAR0;                                  // the newest far tile of this block
        int m0, m1;
        part_tiles(q, tk.part, m0, m1);                             // tiles m0 .. m1-1 of the q+1 panel tiles of block k
        const int k = tk.k, part = tk.part, g = tk.g, q4 = tk.q4;
        const int pbase = k * PB + q4 * 4;
        if (pbase >= T) continue;                               // rows past the end (last block)
        const int c = c0 + g * GP + q8 * 4;
        const bool cvalid = c < c1;
        const int cl = cvalid ? c : c0;                         // clamped chain for addresses

        float aM[4][4], aS[4][4];
        int aK[4][4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
            for (int i = 0; i < 4; ++i) { aM[rr][i] = SEMICRF_NEG_INF; aS[rr][i] = 0.f; aK[rr][i] = 0x7fffffff; }

        // GRAD: marginal(pi, pj) = gz * exp2(t + arow[rr]), arow = (alpha[frame(pi)] - logZ) * log2e
        float arow[4][4], gz[4];
        if (GRAD) {
            float lz[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = c + i < c1;
                lz[i] = ok ? logZp[c + i] : 0.f;
                gz[i] = ok ? goutp[(size_t)(c + i) * P.gstride] * P.gscale : 0.f;
            }
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int pi = pbase + rr;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float vv = (c + i < c1 && pi < T) ? vfwd[(size_t)frame_of<DIR>(pi, T) * Bs + c + i] : 0.f;
                    arow[rr][i] = (vv - lz[i]) * LOG2E;
                }
            }
        }
        // ---- addressing -----------------------------------------------------------------------------------
        // Buffer addressing: one VGPR byte offset per lane (constant over the task), everything that depends
        // on (tile, rr, h) is wave-uniform and lives in the SGPR base / soffset -- no per-load 64-bit VALU math.
        //   DIR 0: cell(pi, pj) = (pi*T + pj)*B: tile base at (pi_0, 16m), lane part slot*B, soffset ((pi_rr-pi_0)*T + 8h)*B
        //   DIR 1: cell(pi, pj) = ((T-1-pj)*T + (T-1-pi))*B: tile base at (pi_3, 16m+15), lane part (7-slot)*T*B,
        //          soffset ((1-h)*8*T + (pi_3-pi_rr))*B            (pi_rr = min(pbase+rr, T-1))
        PanelGeom G;
        geom_of(tk, G);
        char* const stage0 = lds + wslot * (PNS * PSTAGE_BYTES);
        const unsigned rdbase = lds_addr(stage0) + (DIR == 0 ? (unsigned)lane * 16u
                                                             : (unsigned)((slot >> 1) * 1024 + (slot & 1) * 128 + q8 * 16));
        const unsigned grbase = lds_addr(stage0) + 8192u + (unsigned)lane * 16u;
        const bool probe_stream = SEMICRF_PANEL_PROBES && (dbg & 32u);
        const bool probe_nowait = SEMICRF_PANEL_PROBES && (dbg & 4u);
        const bool probe_nou = SEMICRF_PANEL_PROBES && (dbg & 64u);       // math on whatever the stage holds, no u traffic
        const int nst = GRAD ? 2 * (T - pbase < 4 ? T - pbase : 4) : 0;    // gradient stores per tile (a lower bound)

        // A task that has met an unpublished u runs at the spine's frontier: the u copies it prefetched three tiles ahead
        // are stale by construction.  From then on it fetches the NEXT tile's u again (device scope) while it works on
        // the current tile, so that a published value is found on the first look instead of two round trips later.
        bool frontier = false;
#pragma unroll
        for (int i = 0; i < PNS; ++i)
            if (m0 + i < m1) {
                panel_fetch_cells<DIR>(score, G, stage0 + i * PSTAGE_BYTES, m0 + i, T, Bs, cell_nt);
                if (!probe_stream && !probe_nou) panel_fetch_gran<false>(ursrc, stage0 + i * PSTAGE_BYTES, G.gvoff, m0 + i, B);
                issued += 10;
                if (i == 0) mark0 = issued; else if (i == 1) mark1 = issued; else mark2 = issued;
            }

        for (int m = m0, s = 0; m < m1; ++m, s = (s + 1 == PNS ? 0 : s + 1)) {
            char* const stage = stage0 + s * PSTAGE_BYTES;
            // ---- wait for the stage, move it to registers ------------------------------------------------
            const bool tprobe = SEMICRF_PANEL_PROBES && (dbg & 16u) && g == 0 && q4 == 0 && m == q && lane == 0 && k < 64 && T >= 1024;
            if (tprobe) P.ts[1536 + k] = __builtin_amdgcn_s_memrealtime();
            panel_wait_younger(issued - (s == 0 ? mark0 : (s == 1 ? mark1 : mark2)));
            if (tprobe) P.ts[1600 + k] = __builtin_amdgcn_s_memrealtime();
            int npoll = 0;
            if (tr && m == m0) tsp[1] = __builtin_amdgcn_s_memrealtime();
            if (SEMICRF_PROBE_HIST && (dbg & 1024u) && lane == 0) {
                // activity histogram (tools/activity_hist.py): tiles taken per 4 us bucket, panel and spare waves apart
                u64* const hb = P.ts + (3 * T) / 2;
                const u64 now = __builtin_amdgcn_s_memrealtime();
                // (s_memrealtime counts per XCD: every XCD's tiles are bucketed against the start of ITS first workgroup)
                const u64 t0 = __hip_atomic_load(hb + 129 + (__builtin_amdgcn_s_getreg(6164) & 7), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                u64 b = now > t0 ? (now - t0) / 400 : 0;
                if (b > 63) b = 63;
                atomicAdd((unsigned long long*)(hb + 1 + 64 * hist_role + b), 1ull);
            }
            v4u xo[8];
            panel_read_cells<DIR>(rdbase + (unsigned)(s * PSTAGE_BYTES), xo);
            const bool refill = m + PNS < m1;
            if (probe_stream) {          // streaming probe: touch the data, nothing else
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    aS[0][0] += __uint_as_float(xo[e].x) + __uint_as_float(xo[e].y) + __uint_as_float(xo[e].z) + __uint_as_float(xo[e].w);
            } else {
                v4u g0, g1;      // [h]: u of 4 chains at position 16m + 8h + slot
                panel_read_gran(grbase + (unsigned)(s * PSTAGE_BYTES), g0, g1);
                // a value that has not been published yet reads as U_EMPTY (from the launch's memset, possibly out of a
                // stale cache line): fetch again with device-scope loads until the spine has published block m
                if (!probe_nowait) {
                    const bool v0 = c < c1, v1 = c + 1 < c1, v2 = c + 2 < c1, v3 = c + 3 < c1;
                    int spins = 0;
                    while (true) {
                        const bool ok = (g0.x != U_EMPTY || !v0) && (g0.y != U_EMPTY || !v1) && (g0.z != U_EMPTY || !v2) &&
                                        (g0.w != U_EMPTY || !v3) && (g1.x != U_EMPTY || !v0) && (g1.y != U_EMPTY || !v1) &&
                                        (g1.z != U_EMPTY || !v2) && (g1.w != U_EMPTY || !v3);
                        if (__all(ok)) break;
                        frontier = true;
                        // The task's newest tile (m == q) is on the spine's critical path: poll tightly.  Waves blocked
                        // further ahead advance one tile per spine block; the forward sweeps want them prompt as well
                        // (lazy: 240 us, prompt: 212 us), the bandwidth-bound gradient sweep wants the fabric quiet
                        // (prompt: 505-525 us, lazy: 465-472 us).
                        if (m == q) __builtin_amdgcn_s_sleep(2);
                        else if (GRAD) {
                            // (round 5: the lazy poll pays only where the gradient sweep is bandwidth-bound; with few chains it is
                            // hand-off-bound like the forward sweep -- T=691 x 96: 122 -> 112 us, T=1024 x 88: 222 -> 216 with the short one)
                            if (P.gradLazyShort) __builtin_amdgcn_s_sleep(16);
                            else __builtin_amdgcn_s_sleep(SEMICRF_GRAD_LAZY);
                        }
                        else __builtin_amdgcn_s_sleep(16);
                        if (spin_abort(ctrl, spins, SPIN_LIMIT, 5)) break;
                        ++npoll;
                        // One round trip per poll: the 2 KB tile is fetched again together with a light probe -- one word per
                        // lane, the block's last position for the first of the lane's four chains (a ring publishes its
                        // 16 positions x 4 chains with one store instruction).  The LDS copy is only read back and checked
                        // when every ring of the tile shows a value in the probe; urgent polls (the task's newest tile)
                        // fetch the tile with every probe, lazy ones only after a successful probe.
                        if (m == q) panel_fetch_gran<true>(ursrc, stage, G.gvoff, m, B);
                        const unsigned pw = __builtin_amdgcn_raw_buffer_load_b32(ursrc, (unsigned)(cl * 4), (unsigned)((m * PB + PB - 1) * B * 4), 16);
                        if (!__all(pw != U_EMPTY || !cvalid)) continue;
                        if (m != q) panel_fetch_gran<true>(ursrc, stage, G.gvoff, m, B);
                        wait_vmcnt<0>();
                        panel_read_gran(grbase + (unsigned)(s * PSTAGE_BYTES), g0, g1);
                    }
                }
                if (frontier && m + 1 < m1) {
                    const int sn = s + 1 == PNS ? 0 : s + 1;
                    panel_fetch_gran<true>(ursrc, stage0 + sn * PSTAGE_BYTES, G.gvoff, m + 1, B);
                    issued += 2;
                    if (sn == 0) mark0 = issued; else if (sn == 1) mark1 = issued; else mark2 = issued;
                }
                if (SEMICRF_PANEL_PROBES && (dbg & 16u) && g == 0 && q4 == 0 && m == q && lane == 0) P.ts[192 + k] = __builtin_amdgcn_s_memrealtime();   // chain probe: newest tile's u seen
                // chains past the end of the batch (a ragged last quad reads the NEXT position's first chains there, published or
                // not; a quad that lies past the end altogether reads the first chains, waited for or not): their u is set to 0,
                // so that what they contribute to the wave-wide rescale test -- and with it the other chains' reference points
                // and the last bits of their sums -- does not depend on timing
                if (c >= c1) { g0.x = 0u; g1.x = 0u; }
                if (c + 1 >= c1) { g0.y = 0u; g1.y = 0u; }
                if (c + 2 >= c1) { g0.z = 0u; g1.z = 0u; }
                if (c + 3 >= c1) { g0.w = 0u; g1.w = 0u; }
                const float uv[2][4] = {{__uint_as_float(g0.x), __uint_as_float(g0.y), __uint_as_float(g0.z), __uint_as_float(g0.w)},
                                        {__uint_as_float(g1.x), __uint_as_float(g1.y), __uint_as_float(g1.z), __uint_as_float(g1.w)}};

                if (MODE == 0) {
                    // The accumulation with half the vector instructions (the panels are issue- and power-bound next
                    // to their loads): the exponent argument t - M = cell*log2e + (u - M) is ONE packed fma per two chains
                    // on top of one packed subtract, the overflow test is a max3 tree over those arguments, the sums are
                    // packed adds: 17 plain instructions + 8 exps per row of 8 cells instead of 35 + 8.
                    // GRAD (round 5: the same packed form, and ONE exponential per cell): the marginal gz 2^(t + arow) is the
                    // accumulator's own term p = 2^(t - M) times the per-(row, chain) factor gz 2^(M + arow) -- four exponentials
                    // per row of eight cells instead of eight more.  (Until then the gradient sweep ran the unpacked form with
                    // two exponentials per cell: a frontier task's tile took 1.3 us of math against 0.3 in the forward sweep,
                    // and the ring's block period in the head of the sweep IS that task's time per tile.  M + arow <= ~97: M sits
                    // RESC_LIFT above the largest term when it is set and 2^(term + arow) is a marginal; a cell whose term
                    // lies 2^-30 below that one underflows -- as it does in the sum -- where the reference holds a marginal below
                    // 1e-9.)  The beta values are bit-identical to the plain beta sweep's.
                    typedef float v2f __attribute__((ext_vector_type(2)));
                    const v2f l2e = {LOG2E, LOG2E};
                    const v2f u2[2][2] = {{{uv[0][0], uv[0][1]}, {uv[0][2], uv[0][3]}}, {{uv[1][0], uv[1][1]}, {uv[1][2], uv[1][3]}}};
                    const auto gs = __builtin_amdgcn_make_buffer_rsrc((void*)(GRAD ? dScore + panel_tile_off<DIR>(G, m, T, Bs) : nullptr), 0,
                                                                      0x7fffffff, 0x00020000);
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr) {
                        v2f e[2][2];
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const v4u xv = xo[rr * 2 + h];
                            const v2f x2[2] = {{__uint_as_float(xv.x), __uint_as_float(xv.y)}, {__uint_as_float(xv.z), __uint_as_float(xv.w)}};
#pragma unroll
                            for (int j = 0; j < 2; ++j) {
                                const v2f mj = {aM[rr][2 * j], aM[rr][2 * j + 1]};
                                e[h][j] = __builtin_elementwise_fma(x2[j], l2e, u2[h][j] - mj);      // M = -inf (empty): +inf
                            }
                        }
                        const float emax = fmaxf(fmaxf(fmaxf(e[0][0].x, e[0][0].y), fmaxf(e[0][1].x, e[0][1].y)),
                                                 fmaxf(fmaxf(e[1][0].x, e[1][0].y), fmaxf(e[1][1].x, e[1][1].y)));
                        if (__any(emax > RESC_HI)) {
                            // some accumulator's reference point is too low (always on the first tile): move it up
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float x0 = __uint_as_float(i == 0 ? xo[rr * 2].x : (i == 1 ? xo[rr * 2].y : (i == 2 ? xo[rr * 2].z : xo[rr * 2].w)));
                                const float x1 = __uint_as_float(i == 0 ? xo[rr * 2 + 1].x : (i == 1 ? xo[rr * 2 + 1].y : (i == 2 ? xo[rr * 2 + 1].z : xo[rr * 2 + 1].w)));
                                const float t0 = fmaf(x0, LOG2E, uv[0][i]), t1 = fmaf(x1, LOG2E, uv[1][i]);
                                acc_lift(aM[rr][i], aS[rr][i], fmaxf(t0, t1));
                                const float e0 = t0 - aM[rr][i], e1 = t1 - aM[rr][i];
                                if (i & 1) { e[0][i >> 1].y = e0; e[1][i >> 1].y = e1; } else { e[0][i >> 1].x = e0; e[1][i >> 1].x = e1; }
                            }
                        }
                        v2f mg[2][2];                                   // GRAD: the marginals of the row's two cell groups
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            v2f sj = {aS[rr][2 * j], aS[rr][2 * j + 1]};
                            const v2f p0 = {fexp2(e[0][j].x), fexp2(e[0][j].y)};
                            const v2f p1 = {fexp2(e[1][j].x), fexp2(e[1][j].y)};
                            sj = (sj + p0) + p1;
                            aS[rr][2 * j] = sj.x; aS[rr][2 * j + 1] = sj.y;
                            if (GRAD) {
                                const v2f mj = {aM[rr][2 * j], aM[rr][2 * j + 1]};
                                const v2f aj = {arow[rr][2 * j], arow[rr][2 * j + 1]};
                                const v2f fa = mj + aj;
                                const v2f gzj = {gz[2 * j], gz[2 * j + 1]};
                                const v2f fx = {fexp2(fa.x), fexp2(fa.y)};
                                const v2f gF = gzj * fx;
                                mg[0][j] = gF * p0;
                                mg[1][j] = gF * p1;
                            }
                        }
                        if (GRAD) {
#pragma unroll
                            for (int h = 0; h < 2; ++h) {
                                if (cvalid && pbase + rr < T) {
                                    const unsigned so = panel_soff<DIR>(G, rr, h, T, Bs);
                                    v4u gv;
                                    gv.x = __float_as_uint(mg[h][0].x);
                                    gv.y = __float_as_uint(mg[h][0].y);
                                    gv.z = __float_as_uint(mg[h][1].x);
                                    gv.w = __float_as_uint(mg[h][1].y);
#ifdef SEMICRF_NO_PANEL_GSTORE
                                    asm volatile("" ::"v"(gv));          // timing ablation: the far field's marginals are computed, not stored
                                    continue;
#endif
                                    if (c + 3 < c1) {
                                        if (grad_nt) __builtin_amdgcn_raw_buffer_store_b128(gv, gs, G.voff, so, GRAD_AUX);
                                        else __builtin_amdgcn_raw_buffer_store_b128(gv, gs, G.voff, so, 0);
                                    }
                                    else {                                  // ragged tail of the chain range
                                        __builtin_amdgcn_raw_buffer_store_b32(gv.x, gs, G.voff, so, 0);
                                        if (c + 1 < c1) __builtin_amdgcn_raw_buffer_store_b32(gv.y, gs, G.voff + 4, so, 0);
                                        if (c + 2 < c1) __builtin_amdgcn_raw_buffer_store_b32(gv.z, gs, G.voff + 8, so, 0);
                                    }
                                }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (GRAD) issued += nst;
                } else {
#pragma unroll
                    for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const v4u xv = xo[rr * 2 + h];
                            const float xe[4] = {__uint_as_float(xv.x), __uint_as_float(xv.y), __uint_as_float(xv.z), __uint_as_float(xv.w)};
                            const int key = frame_of<DIR>(m * PB + slot + 8 * h, T);
#pragma unroll
                            for (int i = 0; i < 4; ++i) max_push_seq<DIR>(aM[rr][i], aK[rr][i], uv[h][i] + xe[i], key);
                        }
                }
            }
            if (tprobe) { P.ts[1664 + k] = __builtin_amdgcn_s_memrealtime(); P.ts[1792 + k] = (u64)npoll; }
            // ---- refill the stage PNS tiles ahead ----------------------------------------------------------
            if (refill) {
                panel_fetch_cells<DIR>(score, G, stage, m + PNS, T, Bs, cell_nt);
                if (!probe_stream && !probe_nou) panel_fetch_gran<false>(ursrc, stage, G.gvoff, m + PNS, B);
                issued += 10;
                if (s == 0) mark0 = issued; else if (s == 1) mark1 = issued; else mark2 = issued;
            }
        }
        if (tr) { tsp[2] = __builtin_amdgcn_s_memrealtime(); tsp[4] = (u64)(m1 - m0) | ((u64)k << 8) | ((u64)(frontier ? 1 : 0) << 16) | ((u64)part << 20); }
        // Nothing that WRITES LDS is in flight here: every stage of the task was waited for before it was read, the coherent
        // re-fetch of a tile's u is only issued for a tile that is still to come (and waited for there), a refill only for a tile
        // m + PNS < m1.  What may be outstanding are the gradient stores of the last tiles -- and waiting for their
        // acknowledgements (0.7 us, measured) sat in front of the reduction of EVERY newest tile, on the ring's critical path.
        if (!(GRAD && MODE == 0)) wait_vmcnt<0>();
        (void)frontier;
        if (SEMICRF_PANEL_PROBES && (dbg & 16u) && g == 0 && q4 == 0 && m1 == q + 1 && lane == 0 && k < 64 && T >= 1024)
            P.ts[1728 + k] = __builtin_amdgcn_s_memrealtime();

        // ---- reduce over the 8 column slots (lane bits 3..5) --------------------------------------------------------------
        // The task's last tile is the newest one: what follows sits on the ring's critical path (16 hand-off rounds per
        // sweep at T=1024: a microsecond here is 16 in the total).  The 16 accumulators of every lane go
        // through LDS once (the wave's stages are idle now) and each lane merges the 8 slot values of its two results with
        // ONE exact maximum and 8 independent exps, instead of a 3-stage shuffle tree with two dependent exps per stage.
        const bool b5 = (lane & 32) != 0, b4 = (lane & 16) != 0, b3 = (lane & 8) != 0;
        u64* fbase = farg + (size_t)part * T * Bs;
        const int pi = pbase + (b5 ? 2 : 0) + (b4 ? 1 : 0);
        {
            constexpr int RSTR = 65 * 8;                                  // bytes between accumulators (65 lanes: spreads the banks)
            char* const rb = stage0;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *(float2*)(rb + (rr * 4 + i) * RSTR + lane * 8) =
                        make_float2(aM[rr][i], MODE == 0 ? aS[rr][i] : __int_as_float(aK[rr][i]));
            const int orow = (b5 ? 2 : 0) + (b4 ? 1 : 0);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int oi = (b3 ? 2 : 0) + e;
                const char* src = rb + (orow * 4 + oi) * RSTR + q8 * 8;
                float2 v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = *(const float2*)(src + j * 64);            // slot j, same chain quad
                const int cc = c + oi;
                u64 gr;
                if (MODE == 0) {
                    float mx = v[0].x;
#pragma unroll
                    for (int j = 1; j < 8; ++j) mx = fmaxf(mx, v[j].x);
                    float sum = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) sum += v[j].y == 0.0f ? 0.0f : v[j].y * fexp2(v[j].x - mx);   // empty: (-inf, 0)
                    gr = make_granule(tag, mx + flog2(sum));
                } else {
                    float bm = v[0].x;
                    int bk = __float_as_int(v[0].y);
#pragma unroll
                    for (int j = 1; j < 8; ++j) max_push(bm, bk, v[j].x, __float_as_int(v[j].y));
                    gr = make_granule_key(tag, bm, bk);
                }
                if (pi < T && cc < c1 && !(SEMICRF_PANEL_PROBES && (dbg & 32u))) store_granule(fbase + (size_t)pi * Bs + cc, gr);
            }
        }
        if (SEMICRF_PANEL_PROBES && (dbg & 16u) && g == 0 && m1 == q + 1 && lane == 0 && k < 64)
            P.ts[256 + 64 * q4 + k] = __builtin_amdgcn_s_memrealtime();     // chain probe: partial stored (per row quarter)
        if (tr) tsp[3] = __builtin_amdgcn_s_memrealtime();
        ++tn;
    }
}

// ---------------------------------------------------------------------------------------------
// ZERO role (GRAD): the dense gradient's upper triangle (begin > end) is exact zeros.  Spare waves of the panel
// workgroups write it while the sweep runs (row e: columns e+1..T-1, contiguous over all chains); rows are handed
// out in pairs (e, T-2-e) of equal total length.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void zero_row(float* __restrict__ dScore, int e, int T, int B, int lane)
{
    const size_t n = (size_t)(T - 1 - e) * B;                 // floats to clear in this row
    float* rowp = dScore + ((size_t)e * T + e + 1) * B;
    const size_t lead = (4 - (((uintptr_t)rowp >> 2) & 3)) & 3;          // floats up to 16-byte alignment
    const size_t head = lead < n ? lead : n;
    const size_t n4 = (n - head) / 4;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4* v = (f32x4*)(rowp + head);
    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = lane; i < n4; i += 64) __builtin_nontemporal_store(z, v + i);
    if ((size_t)lane < head) rowp[lane] = 0.f;
    const size_t tail0 = head + n4 * 4;
    if (tail0 + lane < n) rowp[tail0 + lane] = 0.f;
}
__device__ __forceinline__ void zero_role(const SweepParams& P)
{
    const int T = P.T, B = P.B;
    float* const dScore = P.dScore;
    unsigned* const ctrl = P.ctrl;
    const int lane = threadIdx.x & 63;
    const int npairs = T / 2;                                  // rows 0 .. T-2 in pairs (e, T-2-e)
    while (true) {
        int p = 0;
        if (lane == 0) p = (int)(atomicAdd(ctrl + 3, 1u) + 1u);
        p = __builtin_amdgcn_readfirstlane(p);
        if (p >= npairs) break;
        const int e2 = T - 2 - p;
        if (p <= e2) zero_row(dScore, p, T, B, lane);
        if (p < e2) zero_row(dScore, e2, T, B, lane);
    }
}

// ---------------------------------------------------------------------------------------------
// BAND role (GRAD): the marginals of the band, written by spare waves of the panel workgroups
// ---------------------------------------------------------------------------------------------
// Until round 5 the ring waves stored the marginals of the cells they read themselves (the band: the diagonal tile and the
// FAR0 - 1 tiles left of it) -- one 4-byte store per lane and cell, 64 store instructions per block in the middle of the
// dependent chain, every one a 16-byte piece of 16 different lines: the ring of the gradient sweep needed 3.1 us per block where
// the forward ring needs 1.4, and the whole head of the sweep ran at that pace (T=1024 x 352: 357 us with, 295 us without
// those stores).  The band's marginals need nothing the ring has not published: marginal(pi, pj) = gz exp2(u[pj] + cell log2e +
// arow[pi]) -- the same expression, bit for bit, the panels evaluate for the far field.  So they are a task of their own:
// (column block m, row block k = m + dk, 32-chain group g, row quarter q4), ready as soon as the ring has published block m,
// handed out in m order to `bandWaves` otherwise idle waves of every panel workgroup.  A wave reads the tile's cells once more
// (+12 % reads: 4 rows x 16 columns x 128 bytes, register loads in 512-byte runs), its 16 x 4 u values and writes whole lines.
// Nobody waits for these waves: the ring's only global stores are u, the diagonal and the noise gradient.
constexpr int CTRL_BANDQ = 32;        // [32]: band task queue head (a line of its own)

template <int DIR>
__device__ __forceinline__ void band_role(const SweepParams& P)
{
    const int T = P.T, B = P.B, K = P.K, c0 = P.c0, c1 = P.c1;
    const size_t Bs = (size_t)B;
    unsigned* const ctrl = P.ctrl;
    const float* __restrict__ score = P.score;
    float* const dScore = P.dScore;
    const int lane = threadIdx.x & 63;
    const int q8 = lane & 7, pjsub = (lane >> 3) & 1, pisub = lane >> 4;
    const int nG = P.nPanelGroups;
    const int nTasks = K * FAR0 * nG * 4;
    const size_t last4 = (size_t)T * T * Bs - 4;
    const auto ursrc = __builtin_amdgcn_make_buffer_rsrc((void*)P.ug, 0, (int)((size_t)T * Bs * 4), 0x00020000);
    const bool grad_nt = __builtin_amdgcn_readfirstlane(P.gradNT) != 0;
    while (true) {
        int task = 0;
        if (lane == 0) task = (int)(atomicAdd(ctrl + CTRL_BANDQ, 1u) + 1u);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= nTasks) break;
        const int q4 = task & 3, t2 = task >> 2;
        const int g = t2 % nG, t3 = t2 / nG;
        const int dk = t3 % FAR0, m = t3 / FAR0, k = m + dk;
        const int pbase = k * PB + q4 * 4;
        if (k >= K || pbase >= T) continue;
#if defined(SEMICRF_BAND_ABL) && SEMICRF_BAND_ABL == 2
        continue;                                                    // timing ablation: the band tasks are only drawn
#endif
        const int pi = pbase + pisub;
        const bool rowok = pi < T;
        const int pic = rowok ? pi : T - 1;
        const int c = c0 + g * GP + q8 * 4;
        const int cl = c < c1 ? c : c0;
        const bool v0 = c < c1, v1 = c + 1 < c1, v2 = c + 2 < c1, v3 = c + 3 < c1;
        float gz[4], ar[4];
        {
            const int fr = frame_of<DIR>(pic, T);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = c + i < c1;
                const float lz = ok ? P.logZ[c + i] : 0.f;
                gz[i] = ok ? P.gout[(size_t)(c + i) * P.gstride] * P.gscale : 0.f;
                ar[i] = ok ? (P.vfwd[(size_t)fr * Bs + c + i] - lz) * LOG2E : 0.f;
            }
        }
        v4u x[8], uu[8];
        size_t off[8];                                               // element offset of the cell (loads: clamped into the tensor)
        unsigned uoff[8];
        bool use[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int pj = m * PB + 2 * e + pjsub;
            use[e] = rowok && pj < pi;
            const int pjc = pj < pic ? pj : (pic > 0 ? pic - 1 : 0);       // the diagonal tile: cells at and above the diagonal are not used
            off[e] = cell_index<DIR>(pic, pjc, T) * Bs + cl;
            uoff[e] = (unsigned)(((size_t)pjc * Bs + cl) * 4);
            const v4u_a4 t = *(const v4u_a4*)(score + (off[e] < last4 ? off[e] : last4));
            x[e] = t;
            uu[e] = __builtin_amdgcn_raw_buffer_load_b128(ursrc, uoff[e], 0, 0);
        }
        // u not published yet (or a stale line of the launch's start): again with device scope until the ring is there
        {
            int spins = 0;
            while (true) {
                bool ok = true;
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    ok = ok && (uu[e].x != U_EMPTY || !v0) && (uu[e].y != U_EMPTY || !v1) && (uu[e].z != U_EMPTY || !v2) && (uu[e].w != U_EMPTY || !v3);
                if (__all(ok)) break;
                __builtin_amdgcn_s_sleep(64);
                if (spin_abort(ctrl, spins, SPIN_LIMIT, 14)) break;
#pragma unroll
                for (int e = 0; e < 8; ++e) uu[e] = __builtin_amdgcn_raw_buffer_load_b128(ursrc, uoff[e], 0, 16);     // 16 = sc1
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (!use[e] || !v0) continue;
            v4u gv;
            gv.x = __float_as_uint(gz[0] * fexp2(fmaf(__uint_as_float(x[e].x), LOG2E, __uint_as_float(uu[e].x)) + ar[0]));
            gv.y = __float_as_uint(gz[1] * fexp2(fmaf(__uint_as_float(x[e].y), LOG2E, __uint_as_float(uu[e].y)) + ar[1]));
            gv.z = __float_as_uint(gz[2] * fexp2(fmaf(__uint_as_float(x[e].z), LOG2E, __uint_as_float(uu[e].z)) + ar[2]));
            gv.w = __float_as_uint(gz[3] * fexp2(fmaf(__uint_as_float(x[e].w), LOG2E, __uint_as_float(uu[e].w)) + ar[3]));
            float* const dst = dScore + off[e];
#if defined(SEMICRF_BAND_ABL) && SEMICRF_BAND_ABL == 1
            asm volatile("" ::"v"(gv));                              // timing ablation: computed, not stored
            continue;
#endif
            if (v3) {
                const v4u_a4 ga = gv;
                if (grad_nt) __builtin_nontemporal_store(ga, (v4u_a4*)dst);
                else *(v4u_a4*)dst = ga;
            }
            else {                                                   // ragged tail of the chain range
                dst[0] = __uint_as_float(gv.x);
                if (v1) dst[1] = __uint_as_float(gv.y);
                if (v2) dst[2] = __uint_as_float(gv.z);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// PATH role (forward sweep of logProb): the path scores, by one otherwise idle wave per workgroup
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void path_role(const SweepParams& P)
{
    const int nb = P.c1 - P.c0;
    while (true) {
        int i = 0;
        if ((threadIdx.x & 63) == 0) i = (int)(atomicAdd(P.ctrl + CTRL_PATHQ, 1u) + 1u);
        i = __builtin_amdgcn_readfirstlane(i);
        if (i >= nb) break;
        const int c = P.c0 + i;
        const double r = path_score_wave(P.score, P.noise, P.T, P.B, P.pathK, P.pathPairs, P.pathOffsets, c);
        // Every lane stores the (same) granule.  With `if (lane == 0) store` here the compiler threaded this branch into the next
        // iteration's `if (lane == 0) draw` and ran the readfirstlane of the draw -- and everything behind it -- a second time with
        // lane 0 masked off: an endless loop on chain c0 (gfx950, ROCm 7.2; found as a hang of the whole launch).
        __builtin_amdgcn_wave_barrier();
        store_granule(P.pathg + c, make_granule(P.tag, (float)r));
    }
}

// Workgroup index -> role ticket (spines: tickets < nSpine = the chain group; panel workgroups: nSpine + a dense rank).
// The eight rings of a 32-chain panel group read the 16-byte pieces of the SAME 128-byte lines of the band, and their far
// waves take partials from the same tasks: they belong on ONE XCD (one L2).  Block b is observed to run on XCD b % 8 (a
// placement HIP does not promise: only speed depends on it), so panel group g gets the first blocks of XCD g % 8.  Groups
// dealt to the XCDs by block index alone -- every ring of a group on a different XCD -- cost 216-228 us instead of 185; the
// earlier first-come ticket (an atomic per workgroup) clustered them by luck of arrival.
__host__ __device__ __forceinline__ int wg_ticket(int nSpine, int grid, int b)
{
    constexpr int X = 8, SPG = GP / GS / SPH;                // XCDs, spine WORKGROUPS per panel group (nSpine counts workgroups here)
    const int ngroups = (nSpine + SPG - 1) / SPG;
    int S[X], W[X];
    bool fits = true;
#pragma unroll
    for (int x = 0; x < X; ++x) {
        W[x] = x < grid ? (grid - x + X - 1) / X : 0;        // workgroups of the launch on XCD x
        int sx = 0;
        for (int g = x; g < ngroups; g += X) sx += nSpine - g * SPG < SPG ? nSpine - g * SPG : SPG;
        S[x] = sx;
        fits = fits && sx <= W[x];
    }
    if (!fits) return b;                                     // too few workgroups per XCD: plain order
    const int x = b % X, slot = b / X;
    int Sx = 0, Px = 0;
#pragma unroll
    for (int i = 0; i < X; ++i) if (i == x) { Sx = S[i]; Px = W[i] - S[i]; }
    (void)Px;
    if (slot < Sx) return (x + X * (slot / SPG)) * SPG + slot % SPG;       // ring slot % SPG of panel group x + 8 (slot / SPG)
    const int ps = slot - Sx;                                // panel workgroups: dense rank, level by level across the XCDs
    int rank = 0;
#pragma unroll
    for (int i = 0; i < X; ++i) {
        const int pn = W[i] - S[i];
        rank += ps < pn ? ps : pn;
        if (i < x && pn > ps) ++rank;
    }
    return nSpine + rank;
}

// ---------------------------------------------------------------------------------------------
// kernel
// ---------------------------------------------------------------------------------------------
// Launched with enough dynamic LDS that only ONE workgroup fits a compute unit, so with grid <= #CUs every
// workgroup is resident.  A SPINE workgroup (ticket < nSpine): waves 0-3 ring, wave 4 loader, wave 5 far wave,
// waves 6.. panel waves (hybridPanelWaves of them); the other workgroups: `panelWaves` panel waves.
template <int MODE, int DIR, bool GRAD>
__global__ __launch_bounds__(NT) void persist_sweep_kernel(SweepParams P)
{
    extern __shared__ __attribute__((aligned(16))) char s_dyn[];
    __shared__ int s_exit;
    // A leased workspace that an earlier launch left with its error word up (the host fills it again as soon as it has seen the
    // pinned abort word, but launches it had already enqueued arrive first): this launch gives up at once -- its waits see the
    // word within a few polls, its outputs are poisoned like the aborted launch's own.
    if (threadIdx.x == 0) {
        const unsigned e1 = __hip_atomic_load(P.ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned gen = P.selfclean ? __hip_atomic_load(P.ctrl + CTRL_GEN, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : P.expect_gen;
        s_abort = e1 != CTRL_INIT ? 1 : 0;
        // a "clean" lease that is not in the state its last launch must have left (somebody else wrote to the workspace, or a launch
        // the host never saw): nothing in it can be trusted
        if (gen != P.expect_gen) { set_error(P.ctrl, 13u); s_abort = 1; }
        s_exit = 0;
    }
    // flags and sequence numbers start at 0
    for (int h = 0; h < SPH; ++h)
        for (int i = threadIdx.x; i < (LDS_DUMMY - LDS_FAR) / 4; i += NT) ((int*)(s_dyn + h * LDS_HALF + LDS_FAR))[i] = 0;
    __syncthreads();
    // Roles by workgroup index: every workgroup of the launch is resident (one per CU, the grid never exceeds the CUs) and
    // the dispatcher starts them in index order, so a workgroup still only waits on earlier ones.  (An atomic ticket cost
    // every workgroup a device-scope round trip before it could do anything, on a line that hundreds of waves hit with
    // their first task draws at the same moment.)
    const int nSpineWG = (P.nSpine + SPH - 1) / SPH;
    const int ticket = wg_ticket(nSpineWG, (int)gridDim.x, (int)blockIdx.x);
    const int wave = (int)(threadIdx.x >> 6);
    if (P.ug_other && wave == NT / 64 - 1) {
        // Leased workspace: the u values of the PREVIOUS launch go back to U_EMPTY here -- the two launches use alternate
        // buffers, so this one's own values can stay until the next launch: nobody has to know when the last reader of a u is
        // through (rounds 2-4: the rings cleared their group's u at the end of the sweep, which ruled out any reader that may
        // finish after the ring).  Every workgroup clears one slice with its last wave (idle in every role), in whole 16-byte pieces.
        const size_t n = (size_t)P.T * P.B, n4 = n / 4;
        const size_t per = (n4 + gridDim.x - 1) / gridDim.x, i0 = per * blockIdx.x, i1 = i0 + per < n4 ? i0 + per : n4;
        const v4u e = {U_EMPTY, U_EMPTY, U_EMPTY, U_EMPTY};
        for (size_t i = i0 + (threadIdx.x & 63); i < i1; i += 64) ((v4u*)P.ug_other)[i] = e;
        if (blockIdx.x == 0 && (threadIdx.x & 63) < (int)(n - n4 * 4)) P.ug_other[n4 * 4 + (threadIdx.x & 63)] = U_EMPTY;
    }
    // clock probe (probe build, debug flag 128): shader cycles and 100 MHz ticks of one panel workgroup over the launch
    const bool clk = SEMICRF_PANEL_PROBES && (DBGF(P) & 128u) && ticket == nSpineWG + 3 && threadIdx.x == 0;
    if (clk) { P.ts[600] = __builtin_readcyclecounter(); P.ts[601] = __builtin_amdgcn_s_memrealtime(); }
    if (SEMICRF_PROBE_HIST && (DBGF(P) & 1024u) && threadIdx.x == 0) {        // activity histogram: t0 = the first workgroup's start, per XCD
        const unsigned long long now = (unsigned long long)__builtin_amdgcn_s_memrealtime();
        atomicMin((unsigned long long*)(P.ts + (3 * P.T) / 2), now);
        atomicMin((unsigned long long*)(P.ts + (3 * P.T) / 2 + 129 + (__builtin_amdgcn_s_getreg(6164) & 7)), now);     // hwreg(HW_REG_XCC_ID, 0, 4)
    }
    if (ticket < nSpineWG) {
        // chain group = ticket: neighbouring groups read neighbouring 16-byte pieces of the same sectors, and
        // measured fetch traffic is 3x lower this way than with groups spread 8 tickets apart (0.13 vs 0.36 GB for
        // the band at T=1024, NBatch=352)
        const int half = wave / HWAVES, hw = wave % HWAVES;       // which spine of the workgroup, which wave of that spine
        const int sg = ticket * SPH + half;
        char* const lds_h = s_dyn + half * LDS_HALF;
        if (half >= SPH) {
            const int xw = wave - SPH * HWAVES;
            if (SPH == 1 && NFARW > 1 && wave >= NT / 64 - (NFARW - 1)) {
                // far waves 1 .. NFARW - 1 of the (one) spine (they leave at once in launches that use fewer: far_role)
                if (ticket < P.nSpine && !(DBGF(P) & 9u)) far_role<MODE, DIR, GRAD>(P, ticket, s_dyn, 1 + wave - (NT / 64 - (NFARW - 1)));
            } else if (!(DBGF(P) & 2u) && xw < P.hybridPanelWaves) {
                // Spare waves stream tiles like the panel workgroups do (their stages lie behind the spines' LDS) -- but only
                // once the sweep is bound by the far field: during the first blocks the ring sets the pace and a streaming
                // wave on its CU only slows it down.  They wait until the (first) ring has taken row block hybridStart.
                {
                    const int* cons = (const int*)(s_dyn + LDS_CTL) + NRBUF;
                    int spins = 0;
                    while (lds_flag_load(cons + (P.hybridStart % RING)) < P.hybridStart + 1 && P.hybridStart < P.K &&
                           !(SEMICRF_PANEL_PROBES && (DBGF(P) & 8u))) {
                        __builtin_amdgcn_s_sleep(127);
                        if (spin_abort(P.ctrl, spins, SPIN_LIMIT_LDS, 10)) break;
                    }
                }
                panel_role<MODE, DIR, GRAD>(P, s_dyn + LDS_HYBRID_PANEL, xw, 1);
            } else if (MODE == 0 && DIR == 0 && !GRAD && P.pathPairs != nullptr && P.pathSpine && xw == P.hybridPanelWaves) {
                path_role(P);
            }
        } else if (sg >= P.nSpine) {
            // (an odd number of spines: the last workgroup's second one has no chains)
        } else if (hw < RING) {
            if (!(DBGF(P) & 8u)) {
                if (GRAD && P.bandWaves == 0) spine_role<MODE, DIR, GRAD, GRAD>(P, sg, hw, lds_h);
                else spine_role<MODE, DIR, GRAD, false>(P, sg, hw, lds_h);
            }
        } else if (hw < RING + NLOADER) {
            if (!(DBGF(P) & 8u)) loader_role<DIR, GRAD>(P, sg, lds_h, hw - RING);
        } else {
            if (!(DBGF(P) & 9u)) far_role<MODE, DIR, GRAD>(P, sg, lds_h, hw - (RING + NLOADER));
        }
    } else {
        if (!(DBGF(P) & 2u) && wave < P.panelWaves)
            panel_role<MODE, DIR, GRAD>(P, s_dyn, wave, 0, P.taskBase > 0 ? (ticket - nSpineWG) * P.panelWaves + wave : -1);
        else if (GRAD && wave - P.panelWaves >= 0 && wave - P.panelWaves < P.zeroWaves) zero_role(P);
        else if (GRAD && wave - P.panelWaves - P.zeroWaves >= 0 && wave - P.panelWaves - P.zeroWaves < P.bandWaves) band_role<DIR>(P);
        else if (SEMICRF_BANDX && wave >= NT / 64 - P.copyWaves && (ticket - nSpineWG) % SEMICRF_BANDX_WGSTRIDE == 0) copy_role<DIR>(P);
        else if (MODE == 0 && DIR == 0 && !GRAD && P.pathPairs != nullptr && wave == P.panelWaves) path_role(P);
    }
    if (clk) { P.ts[602] = __builtin_readcyclecounter(); P.ts[603] = __builtin_amdgcn_s_memrealtime(); }
    if (P.selfclean) {
        // The last wave out puts the control words back (unless a wait timed out: the error word stays for the backtrack
        // kernel, the host has been told through g_host_abort and fills the workspace before its next use).  Counted per
        // workgroup (LDS first) and in a line of its own: one device-scope atomic per WAVE on the line of the ticket
        // counter -- 1500 idle waves leave at the very start -- delayed every workgroup's ticket: +21 us.
        int left = 0;
        if ((threadIdx.x & 63) == 0) left = atomicAdd(&s_exit, 1) + 1;
        left = __builtin_amdgcn_readfirstlane(left);
        if (left == NT / 64) {
            unsigned before = 0;
            if ((threadIdx.x & 63) == 0) before = atomicAdd(P.ctrl + CTRL_EXIT, 1u) + 1u;      // workgroups that left before this one
            before = (unsigned)__builtin_amdgcn_readfirstlane((int)before);
            if (before + 1u == gridDim.x &&
                __hip_atomic_load(P.ctrl + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == CTRL_INIT) {
                for (int i = threadIdx.x & 63; i < (int)CTRL_WORDS; i += 64)
                    __hip_atomic_store(P.ctrl + i, i == CTRL_GEN ? P.tag : CTRL_INIT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// BAND COPY (SEMICRF_BANDX): the band in spine-major order
// ---------------------------------------------------------------------------------------------
// A spine reads its band as 16-byte pieces (its 4 chains) of 128-byte lines: 64 lines per 1 KB load, 1024 line fills per row
// block through its compute unit's L1.  In the copy a spine's tile is 4 KB contiguous -- [column][row][4 chains], the loader's
// LDS image -- so the same load touches 8 lines.  One copy unit = (row block kr, 32-chain group g, band tile t, column quarter
// q): 64 cells x 128 bytes in, 1 KB to each of the group's 8 spines out; lane = (cell of 8, chain quad): 8 lanes read one whole
// line, and the 8 lanes of a quad write 128 contiguous bytes.  Cells and clamps are exactly the loader's own (bit-identical LDS).
template <int DIR>
__device__ __forceinline__ void band_copy_load(const float* __restrict__ score, int T, int B, int kr, int g, int t, int q, int lane,
                                               v4u_a4 (&v)[8])
{
    const int q8 = lane & 7, cs = lane >> 3;
    const int cbase = (g * (GP / GS) + q8) * GS;             // the batch's spine number x 4
    const size_t Bs = (size_t)B;
    const size_t last4 = (size_t)T * T * Bs - 4;
    const int kc = kr - t > 0 ? kr - t : 0;                  // tile t: column block kr - t (the ring's tiles, newest first, then the near tiles)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int ci = q * 64 + it * 8 + cs;                 // cell of the tile in LDS order: column (ci >> 4), row (ci & 15)
        const int col = ci >> 4, tr = ci & 15;
        const int prow_t = kr * PB + tr < T ? kr * PB + tr : T - 1;
        int pj = kc * PB + col;
        pj = pj < prow_t ? pj : (prow_t > 0 ? prow_t - 1 : 0);
        size_t off = cell_index<DIR>(prow_t, pj, T) * Bs + (cbase < B ? cbase : 0);
        off = off < last4 ? off : last4;
        v[it] = *(const v4u_a4*)(score + off);
    }
}
__device__ __forceinline__ void band_copy_store(float* __restrict__ band, int B, int bandSpines, int kr, int g, int t, int q, int lane,
                                                const v4u_a4 (&v)[8])
{
    const int q8 = lane & 7, cs = lane >> 3;
    const int sgG = g * (GP / GS) + q8;
    if (sgG * GS >= B) return;
    // loader tile index i: column block kr - (RING-1) + i for the ring's tiles, RING + n for near tile n (column block kr - RING - n)
    const int ti = t < RING ? RING - 1 - t : t;
    // write-through stores (sc1): the reader is another compute unit of the same launch, possibly on another XCD
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)band, 0, 0x7fffffff, 0x00020000);
    const unsigned voff = (unsigned)(((((size_t)kr * bandSpines + sgG) * NBT + ti) * (TILE_BYTES / 4) + (q * 64 + cs) * 4) * 4);
#pragma unroll
    for (int it = 0; it < 8; ++it) __builtin_amdgcn_raw_buffer_store_b128(v[it], rs, voff + it * 128, 0, 16);
}

// COPY role: tasks (row block kr >= bandK0, band tile t, 32-chain group g) in that order from an atomic queue; a task is the tile's
// four quarters, then -- all of its stores acknowledged -- the launch tag in the tile's flag word.
template <int DIR>
__device__ __forceinline__ void copy_role(const SweepParams& P)
{
    const int K = P.K, nG = P.nPanelGroups, g0 = P.c0 / GP;
    const int nTasks = (K - P.bandK0) * NBT * nG;
    const int lane = threadIdx.x & 63;
    while (true) {
        int task = 0;
        if (lane == 0) task = (int)(atomicAdd(P.ctrl + CTRL_COPYQ, 1u) + 1u);
        task = __builtin_amdgcn_readfirstlane(task);
        if (task >= nTasks) break;
        const int g = task % nG, t = (task / nG) % NBT, kr = P.bandK0 + task / (nG * NBT);
        // the tile's four quarters, two of them in flight (a quarter is eight 16-byte loads per lane: a wave with one quarter in flight
        // needs 12 us per tile, and the first row blocks of the copy are wanted 6 us into the sweep)
        v4u_a4 va[8], vb[8];
        band_copy_load<DIR>(P.score, P.T, P.B, kr, g0 + g, t, 0, lane, va);
        band_copy_load<DIR>(P.score, P.T, P.B, kr, g0 + g, t, 1, lane, vb);
        band_copy_store(P.band, P.B, P.bandSpines, kr, g0 + g, t, 0, lane, va);
        band_copy_load<DIR>(P.score, P.T, P.B, kr, g0 + g, t, 2, lane, va);
        band_copy_store(P.band, P.B, P.bandSpines, kr, g0 + g, t, 1, lane, vb);
        band_copy_load<DIR>(P.score, P.T, P.B, kr, g0 + g, t, 3, lane, vb);
        band_copy_store(P.band, P.B, P.bandSpines, kr, g0 + g, t, 2, lane, va);
        band_copy_store(P.band, P.B, P.bandSpines, kr, g0 + g, t, 3, lane, vb);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // (every lane stores the same word: a lane-0 branch here would be threaded into the next draw's, see path_role)
        __builtin_amdgcn_wave_barrier();
        __hip_atomic_store(P.bandFlags + ((size_t)(g0 + g) * (K + 2) + kr) * 8 + t, ((P.tag & 0xffffu) << 16) | (unsigned)kr, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
    }
}

// in-sweep band copy: for which problems, and what it needs in the workspace (flags first: they are part of the launch's fill)
static bool band_in_sweep(int T, int B)
{
    const int K = (T + PB - 1) / PB;
    return SEMICRF_BANDX && B <= SEMICRF_BANDX_MAXB && K > SEMICRF_BANDX_K0 + FAR0;
}
static size_t band_flag_bytes(int T, int B)
{
    if (!band_in_sweep(T, B)) return 0;
    return align_up((size_t)((T + PB - 1) / PB + 2) * ((B + GP - 1) / GP) * 8 * sizeof(unsigned));      // [group][K + 2 row blocks][8]
}
static size_t band_copy_bytes(int T, int B)
{
    if (!band_in_sweep(T, B)) return 0;
    return align_up((size_t)((T + PB - 1) / PB) * ((B + GS - 1) / GS) * NBT * TILE_BYTES);
}

constexpr size_t CTRL_BYTES = MAX_CHUNKS * CTRL_WORDS * sizeof(unsigned);

// Cache policy of the panels' cell stream (round 6).  Every cell is read once per sweep, so the stream was non-temporal from
// round 1 on (it must not evict the re-read u values from the L2s) -- right for the headline shape (0.74 GB per sweep), wrong in two
// cases, both measured with the sweeps in the order a step runs them (tools/nt_probe.py: forward, gradient sweep, decode on one
// tensor; profiles/r06_nt_policy.md):
//   * the score tensor fits the 256 MB memory-side cache next to what else the step touches (lower triangle <= SEMICRF_NT_MIN_MB):
//     the cells were just written by the scorer or read by the previous sweep of the step, and a non-temporal line is not kept
//     there.  Gradient sweep T=1024 x 88 244 -> 190 us, 691 x 180 210 -> 168, 691 x 192 147 -> 137, 1024 x 96 183 -> 168, 512 x 352 119 ->
//     114; forward 142 -> 132 / 116 -> 105 / 97 -> 96 / 128 -> 123 / 85 -> 85; decode level.  (With 1 GiB of unrelated traffic in front
//     of every step the forward sweep -- then the first reader of cold cells -- loses 5 - 10 us of it again; the gradient sweep keeps
//     its gain.)  Above ~250 MB ordinary loads lose everywhere an aligned tensor was tried (T=691 x 384: forward 129 -> 138,
//     gradient sweep 200 -> 239; T=1024 x 352: 194 -> 216, 331 -> 357).
//   * 4 NBatch is no multiple of 128 bytes (the reference's own chain counts: 88, 90, 360): every 32-chain piece straddles two lines
//     and the neighbour group's task fetches the same two -- rocprofv3 FETCH_SIZE at T=691: 635 MB for 360 chains (1.84 x the
//     algorithmic 345 MB) against 388 MB for 384 (1.05 x).  An ordinary line waits in the L2 / the memory-side cache for its
//     second reader: gradient sweep T=691 x 360 325 -> 269 us (278 -> 255 in a loop of gradient sweeps alone), T=2048 x 88 665 -> 583.
//     Only the gradient sweep: a loop of forward sweeps at T=691 x 360 -- the first reader of a tensor that does not fit the cache --
//     takes 132 us non-temporal and 145 with ordinary loads, the decode sweep 167 and 184.
// The marginals' stores stay non-temporal always (ordinary stores push the score tensor out from under the next sweep: decode
// 159 -> 168 at T=1024 x 88, 98 -> 112 at 512 x 352).
#ifndef SEMICRF_NT_MIN_MB
#define SEMICRF_NT_MIN_MB 240
#endif
#ifndef SEMICRF_GRAD_NT_MIN_MB
#define SEMICRF_GRAD_NT_MIN_MB 0
#endif
#ifndef SEMICRF_NFARW2_MAXB
#define SEMICRF_NFARW2_MAXB 192     // forward / decode launches of at most this many chains ...
#endif
#ifndef SEMICRF_NFARW2_MINT
#define SEMICRF_NFARW2_MINT 1024    // ... and at least this many frames use two far waves per spine
#endif
#ifndef SEMICRF_NT_MISALIGNED
#define SEMICRF_NT_MISALIGNED 0     // 1: non-temporal also for the gradient sweep of tensors whose chain axis is no multiple of 32 (rounds 1-5)
#endif
static bool cell_policy_nt(int T, int nb, int min_mb)
{
    const double mb = 4.0 * nb * ((double)T * (T + 1) / 2) / 1e6;       // the lower triangle this launch streams
    return mb >= (double)min_mb;
}

static int max_parts(int T)
{
    const int K = (T + PB - 1) / PB;
    return K > FAR0 ? nparts_of(K - 1 - FAR0) : 1;
}

// writes the exact zeros of the upper triangle (begin > end) of the dense gradient: row e, columns e+1..T-1
__global__ __launch_bounds__(256) void zero_upper_kernel(float* __restrict__ dScore, int T, int B)
{
    const int e = blockIdx.y;
    const size_t n = (size_t)(T - 1 - e) * B;                 // floats to clear in this row
    float* rowp = dScore + ((size_t)e * T + e + 1) * B;
    const size_t lead = (4 - (((uintptr_t)rowp >> 2) & 3)) & 3;          // floats up to 16-byte alignment
    const size_t head = lead < n ? lead : n;
    const size_t n4 = (n - head) / 4;
    float4* v = (float4*)(rowp + head);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256)
        v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (blockIdx.x == 0) {
        if (threadIdx.x < head) rowp[threadIdx.x] = 0.f;
        const size_t tail0 = head + n4 * 4;
        if (tail0 + threadIdx.x < n) rowp[tail0 + threadIdx.x] = 0.f;
    }
}

// the workspace fill (see launch_persist_sweep_impl for why it is not hipMemsetAsync)
__global__ __launch_bounds__(256) void fill_ff_kernel(v4u* __restrict__ p, size_t n16)
{
    const v4u e = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0xffffffffu};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = e;
}

// exact zeros above the diagonal (begin > end) of a [T][T][B] tensor
void launch_zero_upper(float* X, int T, int B, hipStream_t stream)
{
    if (T < 2) return;
    int gx = (int)(((size_t)T * B / 4 + 255) / 256);
    if (gx > 8) gx = 8;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL(zero_upper_kernel, dim3(gx, T - 1), dim3(256), 0, stream, X, T, B);
}

constexpr int MAX_DEVICES = 64;
static int current_device()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEVICES) dev = 0;
    return dev;
}
static int device_cus()
{
    static std::atomic<int> ncu[MAX_DEVICES];                    // 0: not asked yet (per device: a process may drive several)
    const int dev = current_device();
    int n = ncu[dev].load();
    if (n == 0) {
        int v = 0;
        n = 256;
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        ncu[dev].store(n);
    }
    return n;
}

size_t persist_workspace_bytes(int T, int B)
{
    return CTRL_BYTES + align_up((size_t)2 * T * sizeof(u64)) + (size_t)max_parts(T) * align_up((size_t)T * B * sizeof(u64)) +
           2 * align_up((size_t)T * B * sizeof(unsigned)) +         // two u buffers: consecutive launches into a leased workspace alternate
           align_up((size_t)B * sizeof(u64)) +                      // path-score granules (logProb as one launch)
           band_flag_bytes(T, B) + band_copy_bytes(T, B);            // the spine-major band copy of launches with few chains (not filled)
}

// Even NBatch: the loader's 16-byte global->LDS loads and the panels' 16-byte loads need 8-byte aligned
// addresses (4-byte aligned ones, i.e. odd NBatch, return wrong data); chains past the end of the range are masked.
// (An odd NBatch runs too: the 16-byte accesses to four neighbouring chains are then only 4-byte aligned, which the
// global->LDS loads and the 16-byte stores take -- bit-identical results, 271 vs 184 us at T=1024, NBatch=351 / 352: every
// 128-byte piece straddles two lines -- where a padded copy of the tensor cost 1.0 ms.)
bool persist_supported(int T, int B)
{
    return B >= 2 && T >= 2 && T < 65535 && (long long)T * B * 64 < (1ll << 31) &&      // 32-bit buffer offsets
           (B + GS - 1) / GS <= MAX_CHUNKS * (device_cus() / 2 > 0 ? device_cus() / 2 : 1);      // chain chunks of at most half the CUs' worth of rings
}

static unsigned next_tag()
{
    static std::atomic<unsigned> counter{0};
    const unsigned lo = (counter.fetch_add(1) % 65534u) + 1u;   // 1..65534: never the workspace's fill pattern 0xffff
    return (lo << 16) | lo;                                      // both 16-bit halves nonzero
}

// Launch-geometry knobs of the development tools (tools/bench_sweep.py): the environment is read only by a library built
// with -DSEMICRF_DEBUG_BUILD=1 (once, at the first launch); the release library ignores it (-1 = the built-in choice).
struct Knobs { int hybrid_waves, hybrid_start, panel_waves, zero_waves, band_waves, nfarw; };
static Knobs read_knobs()
{
#if defined(SEMICRF_DEBUG_BUILD) && SEMICRF_DEBUG_BUILD
    auto get = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : -1; };
    return Knobs{get("SEMICRF_HYBRID_PANEL_WAVES"), get("SEMICRF_HYBRID_START"), get("SEMICRF_PANEL_WAVES"), get("SEMICRF_ZERO_WAVES"),
                 get("SEMICRF_BAND_WAVES"), get("SEMICRF_NFARW_RT")};
#else
    return Knobs{-1, -1, -1, -1, -1, -1};
#endif
}

struct GradArgs {
    const float* vfwd; const float* logZ; const float* gout; float* dScore; float* dNoise; int gstride; float gscale;
    int keep_upper;       // the cells begin > end of dScore hold zeros already (SEMICRF_GRAD_UPPER_IS_ZERO): not written
    float noise_add;      // dNoise += noise_add * gout on every gap (logProb's backward), 0: nothing
};
struct PathArgs { const int* pairs; const int* offsets; int K; float* out; };

template <int MODE, int DIR, bool GRAD>
static void launch_one(const SweepParams& P, int grid, hipStream_t stream)
{
    static std::atomic<bool> attr_set[MAX_DEVICES];              // the attribute is per device
    const int dev = current_device();
    if (!attr_set[dev].load()) {
        (void)hipFuncSetAttribute((const void*)persist_sweep_kernel<MODE, DIR, GRAD>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS_DYN_BYTES);
        attr_set[dev].store(true);
    }
    hipLaunchKernelGGL((persist_sweep_kernel<MODE, DIR, GRAD>), dim3(grid), dim3(NT), LDS_DYN_BYTES, stream, P);
}

// mode 0 = LSE, 1 = MAX.  ws must hold persist_workspace_bytes().  Enqueues a memset + one kernel per chain chunk.
// lease: 0 = ordinary workspace (filled here, every launch); 1 = leased, not known to be clean (filled here, the launch
// cleans up after itself); 2 = leased and left clean by the previous launch (no fill).  lease_tag: the workspace's own
// launch count (consecutive launches in one workspace must not share a granule tag), 0 = the process-wide counter.
static int launch_persist_sweep_impl(int mode, int dir, const float* score, const float* noise, int T, int B,
                                     float* u_out, float* last_out, int* code, void* ws, hipStream_t stream,
                                     const GradArgs* grad, int lease, unsigned lease_tag, const PathArgs* path = nullptr)
{
    SweepParams P;
    P.pathPairs = nullptr; P.pathOffsets = nullptr; P.pathK = 0; P.pathOut = nullptr; P.pathg = nullptr; P.pathSpine = 0; P.noiseAdd = 0.0f;
    P.vfwd = nullptr; P.logZ = nullptr; P.gout = nullptr; P.dScore = nullptr; P.dNoise = nullptr; P.gstride = 1; P.gscale = 1.0f;
    if (grad) {
        P.vfwd = grad->vfwd; P.logZ = grad->logZ; P.gout = grad->gout; P.dScore = grad->dScore; P.dNoise = grad->dNoise;
        P.gstride = grad->gstride; P.gscale = grad->gscale; P.noiseAdd = grad->noise_add;
    }
    P.score = score; P.noise = noise; P.T = T; P.B = B; P.K = (T + PB - 1) / PB;
    P.band = nullptr; P.bandFlags = nullptr; P.bandSpines = (B + GS - 1) / GS; P.bandGroups = (B + GP - 1) / GP; P.bandK0 = 0; P.copyWaves = 0;
    P.tag = lease_tag ? (lease_tag % 65534u) + 1u : next_tag();
    P.dbg = 0u;
    P.selfclean = 0;
#if SEMICRF_PANEL_PROBES
    // timing ablations (results are wrong when set): only the probe build reads them
    if (const char* dbg = getenv("SEMICRF_DEBUG_FLAGS")) P.dbg = (unsigned)atoi(dbg);
    if (SEMICRF_CT_DBG >= 0) P.dbg = (unsigned)SEMICRF_CT_DBG;
#endif
    char* w = (char*)ws;
    P.ts = (u64*)(w + CTRL_BYTES);
    const size_t ts_bytes = align_up((size_t)2 * T * sizeof(u64));
    P.farg = (u64*)(w + CTRL_BYTES + ts_bytes);
    const size_t ug_off = CTRL_BYTES + ts_bytes + (size_t)max_parts(T) * align_up((size_t)T * B * sizeof(u64));
    // leased workspaces: launches alternate between the two u buffers, each puts the other one back to U_EMPTY (see the kernel)
    const size_t ug_bytes = align_up((size_t)T * B * sizeof(unsigned));
    const unsigned upar = lease != 0 ? (lease_tag & 1u) : 0u;
    P.ug = (unsigned*)(w + ug_off + upar * ug_bytes);
    P.ug_other = nullptr;
    const size_t band_off = ug_off + 2 * ug_bytes + align_up((size_t)B * sizeof(u64));
    const bool band_ok = band_in_sweep(T, B);
    if (path && mode == 0 && dir == 0 && !grad) {
        // (offsets is never null here; pairs may be when K == 0 -- the role only looks at pairs[k] for k < K)
        P.pathPairs = path->pairs ? path->pairs : path->offsets; P.pathOffsets = path->offsets; P.pathK = path->K; P.pathOut = path->out;
        P.pathg = (u64*)(w + ug_off + 2 * ug_bytes);
    }
    P.u_out = u_out; P.last_out = last_out; P.code = code;
    // ONE fill: every word of the workspace starts as 0xffffffff -- u reads U_EMPTY, far-field granules carry a tag no
    // launch uses, the counters return 0 after their first increment, the error word reads CTRL_INIT
    // (probe build, flag 512: the u values of the previous launch in the same workspace stay -- panels alone on real values)
    // (the band copy's scratch is not filled: its flag words, which are, say what is valid)
    const size_t fill_bytes = (SEMICRF_PANEL_PROBES && (P.dbg & 512u)) ? ug_off : persist_workspace_bytes(T, B) - band_copy_bytes(T, B);
    if (P.dbg != 0u) lease = lease ? 1 : 0;                       // timing ablations leave anything behind
    // (a kernel of our own, not hipMemsetAsync: as fast, and a captured memset node of this size breaks the SECOND replay of an
    // instantiated HIP graph -- wrong results or a memory fault -- while this kernel replays correctly)
    if (lease != 2) hipLaunchKernelGGL(fill_ff_kernel, dim3(1024), dim3(256), 0, stream, (v4u*)ws, (fill_bytes + 15) / 16);
    P.selfclean = lease != 0 && P.dbg == 0u;
    if (P.selfclean) P.ug_other = (unsigned*)(w + ug_off + (1u - upar) * ug_bytes);
    // what ctrl[CTRL_GEN] must read when the kernel starts: the previous launch's tag (a clean lease) or the fill
    P.expect_gen = (lease == 2 && P.selfclean) ? ((lease_tag - 1u) % 65534u) + 1u : CTRL_INIT;
    static const Knobs knobs = read_knobs();

    // Chain chunks: at most a quarter of the CUs host spines in one launch (every workgroup must be resident
    // and the panels need the rest of the chip).
    const int ncu = device_cus();
    const int nSpineTotal = (B + GS - 1) / GS;
    const int maxSpine = ncu / 2 > 0 ? ncu / 2 : 1;          // (more rings per launch instead of a second chunk: no faster, 438 vs 445 us at NBatch=600)
    const int nchunks = (nSpineTotal + maxSpine - 1) / maxSpine;
    if (nchunks > MAX_CHUNKS) return 1;
    // chunks start on a panel-group boundary (32 chains = 128 bytes): a chunk that starts in the middle of a line makes every
    // 128-byte piece of its panels straddle two (T=1024, NBatch=600 as 2 x 300 chains: 488 us forward, 1680 us gradient sweep)
    int perChunk = (nSpineTotal + nchunks - 1) / nchunks;
    perChunk = (perChunk + GP / GS - 1) / (GP / GS) * (GP / GS);
    if (perChunk > maxSpine) perChunk = maxSpine / (GP / GS) * (GP / GS) > 0 ? maxSpine / (GP / GS) * (GP / GS) : maxSpine;
    for (int ci = 0; ci < nchunks; ++ci) {
        P.c0 = ci * perChunk * GS;
        P.c1 = P.c0 + perChunk * GS < B ? P.c0 + perChunk * GS : B;
        if (P.c0 >= P.c1) break;
        const int nb = P.c1 - P.c0;
        P.nSpine = (nb + GS - 1) / GS;
        P.nPanelGroups = (nb + GP - 1) / GP;
        // panel tasks per chain group: block k = FAR0 + q has nparts_of(q) column parts, each split in 4 row quarters
        long long ntask = 0;
        for (int q = 0; q < P.K - FAR0; ++q) ntask += nparts_of(q);
        ntask *= 4;
        P.nTasks = (int)(ntask * P.nPanelGroups);
        P.ctrl = (unsigned*)w + (size_t)ci * CTRL_WORDS;
        // Panel waves.  Every CU streams (HBM bandwidth is limited per CU by the misses it can keep in flight):
        // a spine workgroup carries NT/64 - RING panel waves, the other workgroups `panelWaves`.  More waves
        // mean longer memory queues, and the spine's band loads and hand-offs wait in the same queues; the far
        // field grows with T^2 and the spine's chain with T, so longer sequences get more panel waves, and
        // the gradient sweep (which also stores a tile per tile loaded) a few more.
        const int nSpineWG = (P.nSpine + SPH - 1) / SPH;
        int nPanelWG = ncu - nSpineWG;
        if (nPanelWG < 0) nPanelWG = 0;
        // The spine workgroups' two spare waves stream tiles too, from row block hybridStart on: while the ring sets
        // the pace (the first third of the blocks) a streaming wave on its CU only slows it down; afterwards the sweep is
        // bound by the far field and every CU helps.  Forward, T=1024: 205 us (start at 24-32) vs 214 (from the start)
        // vs 217 (never); T=691, NBatch=360: 158 vs 177 vs 173; T=2048: 651 vs 677 (from the start) vs 788 (never).
        // The gradient sweep is bandwidth-bound almost from the start.
        int hpw = HPW_MAX > 0 ? HPW_MAX : 0;
        if (knobs.hybrid_waves >= 0 && knobs.hybrid_waves <= HPW_MAX) hpw = knobs.hybrid_waves;
        // (round 2, after the panel math got cheaper: later is better -- T=691: 143-150 us from block 28 vs 160 from 16;
        // T=512: 89 from 20 vs 92 from 12; T=1024: 185 from 40 vs 188 from 32; T=2048: 583-598 from 32-48 vs 602-617 from 64-80)
        int hstart = grad ? SEMICRF_GRAD_HSTART : P.K * 5 / 8;
        hstart = hstart < 8 ? 8 : (hstart > 40 ? 40 : hstart);
        if (knobs.hybrid_start >= 0) hstart = knobs.hybrid_start;
        P.hybridStart = hstart;
        P.hybridPanelWaves = hpw;
        // Five panel waves per workgroup where the sweep is bound by the far field's throughput (long sequences, many
        // chains: T=2048, NBatch=352 forward 612 -> 562 us, gradient sweep 1620 -> 1550; T=1024: 185 -> 178), four where the
        // hand-off chain sets the pace (T=691, 90 chains: 103 vs 115 us with five; T=1024, 88 chains: 151 vs 160).
        // tools/stream_probe.hip shows the same per CU: with the panels' math and task boundaries 4 waves x 3 stages stream
        // 22.7 GB/s, 5 x 3 24.8, 6 x 2 29.6, 8 x 2 35.3 -- more waves, not deeper stages, hide a wave's math and boundaries.
        int pw = (T >= 1024 && nb >= 256) ? PW_MAX : (PW_MAX < 4 ? PW_MAX : 4);
        if (knobs.panel_waves > 0) pw = knobs.panel_waves;
        if (pw < 1) pw = 1;
        if (pw > PW_MAX) pw = PW_MAX;
        P.panelWaves = pw;
        // the zero upper triangle of the gradient: by spare waves of the first chunk's panel workgroups, or (no
        // panel workgroups: short sequences) by its own kernel
        int zw = 0;
        if (grad && ci == 0 && !grad->keep_upper) {
            zw = 2;
            if (knobs.zero_waves >= 0) zw = knobs.zero_waves;
            if (zw > NT / 64 - pw) zw = NT / 64 - pw;
            if (nPanelWG <= 0 || P.nTasks == 0) zw = 0;
            if (zw == 0 && T > 1) {
                launch_zero_upper(grad->dScore, T, B, stream);
            }
        }
        P.zeroWaves = zw;
        // the band's marginals: three spare waves of every panel workgroup (none: the ring waves store them).  One / two / three /
        // four / five waves: 446 / 355 / 344 / 354 / 363 us at T=1024 x 352, 272 / 205 / 184-189 / 189 / 196 at T=691 x 384 (one box).
        int bw = 0;
        if (grad && nPanelWG > 0 && P.nTasks > 0) {
            bw = 3;
            if (knobs.band_waves >= 0) bw = knobs.band_waves;
            if (bw > NT / 64 - pw - zw) bw = NT / 64 - pw - zw;
        }
        P.bandWaves = bw;
        P.gradLazyShort = nb < 256 ? 1 : 0;
        // Two far waves taking turns halve the far wave's share of a block's period (one device-scope round trip per block and wave):
        // worth 3 - 4 % where the forward / decode sweeps are hand-off-bound, i.e. with few chains; with many chains, and in the gradient
        // sweep, the second poller costs the fabric more than it returns (T=1024 x 352: 178 vs 176 us, gradient sweep 347 vs 337)
        P.nfarw = (!grad && nb <= SEMICRF_NFARW2_MAXB && T >= SEMICRF_NFARW2_MINT && NFARW >= 2) ? 2 : 1;
        if (knobs.nfarw >= 1 && knobs.nfarw <= NFARW) P.nfarw = knobs.nfarw;
        P.cellNT = cell_policy_nt(T, B, SEMICRF_NT_MIN_MB) ? 1 : 0;          // (by the whole batch: chain chunks stream the same tensor)
        if (!SEMICRF_NT_MISALIGNED && grad && B % 32 != 0) P.cellNT = 0;      // straddling pieces: the line waits for its second reader
        P.gradNT = cell_policy_nt(T, B, SEMICRF_GRAD_NT_MIN_MB) ? 1 : 0;
        if (P.nTasks == 0) nPanelWG = 0;
        P.pathSpine = nPanelWG == 0 ? 1 : 0;        // no panel workgroups (short sequences): the spine workgroups' spare wave
        // the band as a spine-major copy, made by two spare waves of every panel workgroup while the sweep runs
        P.band = nullptr; P.copyWaves = 0;
        // (not in the gradient sweep: neutral at 88 - 96 chains, 3 % slower at 176 -- its stores leave the copy no quiet fabric)
        if (band_ok && !grad && nchunks == 1 && nPanelWG > 0 && NT / 64 - pw - zw - bw - 1 >= SEMICRF_BANDX_WAVES) {
            P.bandFlags = (unsigned*)(w + band_off);
            P.band = (float*)(w + band_off + band_flag_bytes(T, B));
            P.bandK0 = SEMICRF_BANDX_K0;
            P.copyWaves = SEMICRF_BANDX_WAVES;
        }
        const int grid = nSpineWG + nPanelWG;
        // the panel workgroups' waves know their first task (the first draws of ~700 waves all hit one counter at the start)
        P.taskBase = nPanelWG * P.panelWaves;
        if (P.taskBase > P.nTasks) P.taskBase = P.nTasks;
        if (grad) launch_one<0, 1, true>(P, grid, stream);
        else if (mode == 0 && dir == 0) launch_one<0, 0, false>(P, grid, stream);
        else if (mode == 0 && dir == 1) launch_one<0, 1, false>(P, grid, stream);
        else if (mode == 1 && dir == 0) launch_one<1, 0, false>(P, grid, stream);
        else launch_one<1, 1, false>(P, grid, stream);
    }
    return 0;
}

int launch_persist_sweep(int mode, int dir, const float* score, const float* noise, int T, int B, float* u_out,
                         float* last_out, int* code, void* ws, hipStream_t stream, int lease, unsigned lease_tag)
{
    return launch_persist_sweep_impl(mode, dir, score, noise, T, B, u_out, last_out, code, ws, stream, nullptr, lease, lease_tag);
}

// logZ and logProb = path - logZ in one launch (the path scores by spare waves while the sweep runs)
int launch_persist_logprob_fwd(const float* score, const float* noise, int T, int B, float* u_out, float* logZ, const int* pairs,
                               int K, const int* offsets, float* logProb, void* ws, hipStream_t stream, int lease, unsigned lease_tag)
{
    const PathArgs pa{pairs, offsets, K, logProb};
    return launch_persist_sweep_impl(0, 0, score, noise, T, B, u_out, logZ, nullptr, ws, stream, nullptr, lease, lease_tag, &pa);
}

// Fused backward: beta sweep + marginals (dScore fully written incl. the zero upper triangle, dNoise).
int launch_persist_logz_bwd(const float* score, const float* noise, const float* v, const float* logZ,
                            const float* gout, int T, int B, float* dScore, float* dNoise, float* q_out, void* ws,
                            hipStream_t stream, int lease, unsigned lease_tag, int gstride, float gscale, int keep_upper, float noise_add)
{
    GradArgs ga{v, logZ, gout, dScore, dNoise, gstride, gscale, keep_upper, noise_add};
    return launch_persist_sweep_impl(0, 1, score, noise, T, B, q_out, nullptr, nullptr, ws, stream, &ga, lease, lease_tag);
}

// host-side view of the workgroup -> role map (tests/test_abi.py checks that it is a permutation for every launch shape)
int persist_wg_ticket(int nSpine, int grid, int b) { return wg_ticket(nSpine, grid, b); }
int persist_spine_wgs_per_group() { return GP / GS / SPH; }

// the error word of every chain chunk of a sweep launched into `pws` (0xffffffff = no wait timed out)
const unsigned* persist_error_words(void* pws, int* n, int* stride)
{
    *n = MAX_CHUNKS; *stride = (int)CTRL_WORDS;
    return (const unsigned*)pws + 1;
}

// Tell the sweep kernels of the current device where the host's abort word lives (pinned, mapped memory).  Synchronises.
int persist_set_host_abort_word(unsigned* devptr)
{
    static std::atomic<bool> done[MAX_DEVICES];
    const int dev = current_device();
    if (done[dev].load()) return 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_host_abort), &devptr, sizeof(devptr)) != hipSuccess) return 1;
    done[dev].store(true);
    return 0;
}

int read_and_clear_device_status()
{
    unsigned v = 0, z = 0;
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_dev_status), sizeof(v)) != hipSuccess) return -1;
    if (v != 0 && hipMemcpyToSymbol(HIP_SYMBOL(g_dev_status), &z, sizeof(z)) != hipSuccess) return -1;
    return (int)v;
}

}  // namespace semicrf
