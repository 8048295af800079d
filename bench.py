#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native semi-CRF interval layer.

Metric (BASELINE.json): semi-CRF logProb+backward steps/sec at T=1024, NBatch=352.
A "step" is one pass of the hot path over one batch of synthetic input that is already
resident in HBM: NeuralSemiCRFInterval(score, noise).logProb(intervals) forward, then backward
of the train.py-shaped loss (-logProb.sum()/NBatch-segments, train.py:187-189), which writes the
dense [T,T,NBatch] score gradient and the [T-1,NBatch] noise gradient.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  The chain (NBatch) axis is the unit of sharding: every rank owns its own
[T,T,352] problem (weak scaling: per-GPU work fixed), there is no data-path collective; with
N > 1 each step also issues the fused 3-float loss/length/batch all-reduce of train.py:215-217
over RCCL.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the log-partition forward sweep (semicrf_logz_fwd): algorithmic bytes
                  4*B*(T(T+1)/2 + T-1) per launch / average launch time, HIP events on the launch stream.
  cpu_baseline -- the torch-CPU op-loop port of the reference's forward_backward (oracle/oracle.py),
                  timed on this box's host cores on a bounded sample (rank 0, N=1 only).
  extra        -- forward-only and decode rates (decode: T=2048, NBatch=352, forcedStartPos=[4]*NBatch).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def algorithmic_bytes_logz_fwd(T: int, B: int) -> int:
    L = T * (T + 1) // 2
    return 4 * B * (L + T - 1)


def log(msg: str) -> None:
    sys.stderr.write(f"[bench {time.strftime('%H:%M:%S')}] {msg}\n")
    sys.stderr.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--T", type=int, default=1024)
    ap.add_argument("--nbatch", type=int, default=352)
    ap.add_argument("--impl", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--cpu-sample-nbatch", type=int, default=88)
    ap.add_argument("--cpu-threads", type=int, default=32)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an AMD GPU (no CPU fallback exists)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod

    import importlib
    from transkun_amd import CRF, _lib, synth
    from transkun_amd.dist import fused_loss_allreduce, max_over_ranks
    nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")   # the module, not the class
    _lib.set_impl(args.impl)

    T, B = args.T, args.nbatch
    seed = 1234 + 1000 * rank
    score, noise = synth.crf_inputs(T, B, seed, dev, "randn")
    intervals = synth.synthetic_intervals(T, B, seed=seed)
    pairs, offsets = nsci.pack_intervals(intervals, T, B, dev)
    score.requires_grad_(); noise.requires_grad_()
    nseg = max(B // 88, 1)

    def step():
        score.grad = None; noise.grad = None
        lp = nsci._LogProb.apply(score, noise, pairs, offsets)          # == crf.logProb(intervals), pre-packed
        loss = -lp.sum() / nseg                                          # train.py:187
        if dist is not None:
            fused_loss_allreduce(loss, float(T), float(nseg))            # train.py:215-217, fused to one [3] over RCCL
        loss.backward()

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    log(f"inputs ready: T={T} B={B} world={world} cpu_count={os.cpu_count()}")
    for _ in range(args.warmup):
        step()
    sync_all()
    log("warmup done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed, dev)
    value = world * args.steps / elapsed
    log(f"timed region done: {elapsed / args.steps * 1e3:.3f} ms/step")

    # ---- roofline of the dominant kernel: the log-partition forward sweep -------------------------
    s_d, n_d = score.detach(), noise.detach()
    nrep = max(args.steps, 10)
    for _ in range(2):
        nsci._logz_fwd_raw(s_d, n_d, want_v=True)
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(dev)
    e0.record()
    for _ in range(nrep):
        nsci._logz_fwd_raw(s_d, n_d, want_v=True)
    e1.record()
    torch.cuda.synchronize(dev)
    fwd_ms = e0.elapsed_time(e1) / nrep
    abytes = algorithmic_bytes_logz_fwd(T, B)
    achieved = abytes / (fwd_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if os.path.exists(tpath):
        try:
            traffic = json.load(open(tpath)).get(f"logz_fwd_T{T}_B{B}_bytes")
        except Exception:
            traffic = None
    roofline = {"bound": "hbm", "kernel": "semicrf_logz_fwd (log-partition forward sweep)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                "algorithmic_bytes": abytes, "us_per_launch": round(fwd_ms * 1e3, 2)}

    log(f"logz_fwd: {fwd_ms * 1e3:.1f} us/launch = {achieved:.1f} GB/s algorithmic")
    extra = {}
    if not args.no_extra and rank == 0:
        # fwd+bwd split, API-level call with Python lists, and decode (BASELINE configs[2])
        e0.record()
        for _ in range(5):
            score.grad = None; noise.grad = None
            lp = CRF.NeuralSemiCRFInterval(score, noise).logProb(intervals)
            (-lp.sum() / nseg).backward()
        e1.record(); torch.cuda.synchronize(dev)
        extra["api_logprob_fwd_bwd_ms_with_list_marshalling"] = round(e0.elapsed_time(e1) / 5, 3)
        extra["logz_fwd_us"] = round(fwd_ms * 1e3, 2)
        log("api-level loop done; decode next")
        Td, Bd = 2048, 352
        sd, nd = synth.crf_inputs(Td, Bd, 1234, dev, "randn")
        crf_d = CRF.NeuralSemiCRFInterval(sd, nd)
        start = [4] * Bd
        crf_d.decode(forcedStartPos=start)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        nd_rep = 3
        for _ in range(nd_rep):
            crf_d.decode(forcedStartPos=start)
        torch.cuda.synchronize(dev)
        dt = (time.perf_counter() - t1) / nd_rep
        extra["decode_T2048_B352_ms_end_to_end_python_lists"] = round(dt * 1e3, 3)
        extra["decode_segments_per_s_end_to_end"] = round((Bd / 88) / dt, 2)
        # device part only: Viterbi sweep + backtrack + pack, packed pairs left in HBM (HIP events)
        st_t = torch.tensor(start, dtype=torch.int32, device=dev)
        for _ in range(2):
            nsci._viterbi_raw(sd, nd, st_t, False)
        e0.record()
        for _ in range(5):
            nsci._viterbi_raw(sd, nd, st_t, False)
        e1.record(); torch.cuda.synchronize(dev)
        dk = e0.elapsed_time(e1) / 5 * 1e-3
        extra["decode_T2048_B352_ms_device"] = round(dk * 1e3, 3)
        extra["decode_segments_per_s_device"] = round((Bd / 88) / dk, 1)
        extra["decode_chains_per_s_device"] = round(Bd / dk, 1)
        del sd, nd, crf_d
        # the upstream T x T interval-score construction (SURVEY 8 "next" row), same NBatch and T, D = 256: kernels only
        try:
            from transkun_amd import _lib
            from transkun_amd.scorer import _interval_score_raw
            lib = _lib.load()
            Cq, Dq = B, 256
            qq = synth.hash_normal(Cq * T * Dq, 5, dev).view(Cq, T, Dq)
            kk = synth.hash_normal(Cq * T * Dq, 6, dev).view(Cq, T, Dq)
            dd = synth.hash_normal(Cq * T, 7, dev).view(Cq, T)
            for _ in range(2):
                Sq, _ = _interval_score_raw(qq, kk, dd, T, Cq, Dq, 1.0 / 16, 0, False)
            e0.record()
            for _ in range(5):
                Sq, _ = _interval_score_raw(qq, kk, dd, T, Cq, Dq, 1.0 / 16, 0, False)
            e1.record(); torch.cuda.synchronize(dev)
            extra["interval_score_fwd_ms"] = round(e0.elapsed_time(e1) / 5, 3)
            dq = torch.empty_like(qq); dk2 = torch.empty_like(kk); ddg = torch.empty_like(dd)
            nws = int(lib.interval_score_bwd_workspace_bytes(Cq, T, Dq))
            wsq = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
            def _bwd():
                _lib.check(lib.interval_score_bwd_ws(_lib.ptr(Sq), _lib.ptr(qq), _lib.ptr(kk), Cq, T, Dq, Dq, Dq, 1.0 / 16, 0,
                                                     _lib.ptr(dq), _lib.ptr(dk2), _lib.ptr(ddg), Dq, Dq, 1, _lib.ptr(wsq), nws,
                                                     _lib.stream_of(Sq)), "interval_score_bwd_ws")
            for _ in range(2):
                _bwd()
            e0.record()
            for _ in range(5):
                _bwd()
            e1.record(); torch.cuda.synchronize(dev)
            extra["interval_score_bwd_ms"] = round(e0.elapsed_time(e1) / 5, 3)
            extra["interval_score_config"] = f"T={T}, chains={Cq}, D={Dq}, exact-fp32 MFMA, lower triangle"
            del qq, kk, dd, Sq, dq, dk2, ddg, wsq
        except Exception as ex:                       # the headline line must not depend on the extras
            extra["interval_score_error"] = repr(ex)[:200]

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import oracle as cpu_port          # checker/baseline leg only
        avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        ncores = max(1, min(avail, args.cpu_threads))     # the op-loop's small per-row ops do not scale past ~32 threads
        torch.set_num_threads(ncores)
        log(f"cpu baseline on {ncores} of {avail} host threads")
        Bs = min(args.cpu_sample_nbatch, B)
        sc = s_d[:, :, :Bs].contiguous().cpu(); nc = n_d[:, :Bs].contiguous().cpu()
        gout = torch.full((Bs,), -1.0 / nseg)
        t1 = time.perf_counter()
        reps = 0
        while True:
            logz, grad, gn = cpu_port.oploop_forward_backward(sc, nc)
            ds = grad * gout; dn = gn * gout                             # ComputeLogZFasterGrad.backward :472
            reps += 1
            if time.perf_counter() - t1 > 8.0 or reps >= 3:
                break
        dt = (time.perf_counter() - t1) / reps
        del ds, dn, grad
        # steps/s of the full NBatch workload, scaled from the chain sample (chains are independent)
        cpu_baseline = {"value": round((Bs / B) / dt, 5), "unit": "steps/s", "cores": int(torch.get_num_threads()),
                        "kind": "port",
                        "sample": f"torch-CPU op-loop port of forward_backward + backward multiply, T={T}, "
                                  f"{Bs} of {B} chains, {reps} reps, {dt:.2f}s each; scaled by {Bs}/{B}",
                        "cpu_model": _cpu_model()}

    if rank == 0:
        line = {
            "metric": "semi-CRF logProb+backward steps/sec at T=1024, NBatch=352" if (T, B) == (1024, 352)
                      else f"semi-CRF logProb+backward steps/sec at T={T}, NBatch={B}",
            "value": round(value, 3), "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"NeuralSemiCRFInterval.logProb fwd+bwd, T={T}, NBatch={B} per GPU, fp32, "
                                   f"exact-hash randn-like scores, synthetic interval lists (pre-packed, resident)",
                       "T": T, "NBatch": B, "impl": args.impl,
                       "parallelism": f"chains sharded, {world} rank(s), no data-path collective"
                                      + ("; [3] fp32 loss all-reduce per step over RCCL" if world > 1 else "")},
            "roofline": roofline, "cpu_baseline": cpu_baseline, "extra": extra,
        }
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _cpu_model() -> str:
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


if __name__ == "__main__":
    main()
