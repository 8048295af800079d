// scorer_tiles.h -- what the LDS-tiled interval-score kernels share (scorer_mfma.hip, scorer_tiled.hip): the slot layout of S's
// chain axis and the chain quads the work items are made of.
#pragma once
#include "common.h"

namespace semicrf {

// Slot layout of the chain axis (interval_score_fwd_p, include/semicrf_hip.h): the C chains come in groups of `group`
// (the symbols of one segment), each group owns `pitch` >= group SLOTS of S's chain axis; the slots group..pitch-1 of
// every group are ghosts and read zero.  With pitch a multiple of 32 every 32-slot piece of a cell is one aligned 128-byte
// line -- what the CRF kernels want (T=691: 90 symbols at pitch 96 run 22 % faster than at pitch 90) -- while the scorer
// only multiplies the real chains.  Items are quads of REAL chains (ceil(group / 4) per group: no all-ghost items, the
// static schedule stays balanced); the last quad of a group also writes the zeros of the group's ghost tail.
// group == pitch == C: the plain contiguous layout.
struct SlotGeom {
    int group, pitch, qps, nrq;          // quads per group, real quads in total
};
__host__ __device__ inline SlotGeom slot_geom(int C, int group, int pitch)
{
    SlotGeom g;
    g.group = group; g.pitch = pitch;
    g.qps = (group + 3) / 4;
    g.nrq = (C / group) * g.qps;
    return g;
}
struct QuadInfo {
    int c4;      // first slot of the quad (S's chain index)
    int ck;      // its first chain (q / k / diag index)
    int nr;      // real chains in it (0: padding item, nothing to do)
    int tz;      // ghost slots behind it that this item zero-fills (a multiple of 4)
};
__device__ __forceinline__ QuadInfo quad_info(const SlotGeom& g, int rq)
{
    QuadInfo o;
    if (rq >= g.nrq) { o.c4 = 0; o.ck = 0; o.nr = 0; o.tz = 0; return o; }
    const int seg = rq / g.qps, qd = rq - seg * g.qps;
    o.c4 = seg * g.pitch + qd * 4;
    o.ck = seg * g.group + qd * 4;
    o.nr = g.group - qd * 4 < 4 ? g.group - qd * 4 : 4;
    const int tail = g.pitch - g.qps * 4;
    o.tz = (qd == g.qps - 1 && tail > 0) ? tail : 0;
    return o;
}


}  // namespace semicrf
