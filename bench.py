#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native semi-CRF interval layer.

Metric (BASELINE.json): semi-CRF logProb+backward steps/sec at T=1024, NBatch=352.
A "step" is one pass of the hot path over one batch of synthetic input that is already resident in HBM, through
the PUBLIC API: NeuralSemiCRFInterval(score, noise).logProb(intervals) with Python interval lists (packed to the
device inside the timed region, as a caller of the reference would pay it), then backward of the train.py-shaped
loss (-logProb.sum()/segments, train.py:187-189), which writes the dense [T,T,NBatch] score gradient and the
[T-1,NBatch] noise gradient.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One process per GPU.  Launched without torchrun and N > 1, bench.py spawns its N ranks itself
(torch.multiprocessing, MASTER_ADDR=127.0.0.1); it never prints an N=1 line for N>1: the number of ranks in the
line is the world size the process group reports.  The chain (NBatch) axis is the unit of sharding: every rank owns its
own [T,T,352] problem (weak scaling: per-GPU work fixed), there is no data-path collective; with N > 1 each step also
issues the fused 3-float loss/length/batch all-reduce of train.py:215-217 over RCCL.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the log-partition forward sweep (semicrf_logz_fwd): algorithmic bytes 4*B*(T(T+1)/2 + T-1) per
                  launch / average launch time, HIP events on the launch stream.
  cpu_baseline -- the torch-CPU op-loop port of the reference's forward_backward (oracle/oracle.py), timed on this
                  box's host cores at the best of {8,16,32} threads (rank 0, N=1 only).
  extra        -- API-level variants (pre-packed intervals; the reference's two-node evalPath + computeLogZ pattern),
                  decode (T=2048, NBatch=352, randn and model-like scores: device and end to end with Python lists),
                  the segment-shaped path at the model's real shape (T=691, 90 symbols, 1 and 4 segments: scorer + CRF
                  logProb fwd+bwd, decode + attribute features) and the train.py-shaped step (scorer + fused CRF
                  log_prob -> (loss/50).backward() -> [3] all-reduce -> flat gradient all-reduce of 13.61 M parameters).
"""
from __future__ import annotations

import argparse
import hashlib
import gc
import json
import os
import socket
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--T", type=int, default=1024)
    ap.add_argument("--nbatch", type=int, default=352)
    ap.add_argument("--impl", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two rocprofv3 --pmc passes behind roofline.traffic")
    ap.add_argument("--cpu-sample-nbatch", type=int, default=88)
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): every rank its own [T,T,NBatch] problem; strong: the NBatch chains of ONE problem are cut "
                         "across the ranks with transkun_amd.dist.shard_chains (352 -> 44 chains per GPU at 8), value = problems/s")
    ap.add_argument("--backend", default="nccl", help="process-group backend (nccl = RCCL; gloo only for --selftest-launch)")
    ap.add_argument("--selftest-launch", action="store_true",
                    help="CPU-only check of the launcher: form the process group, all-reduce, print the world size seen")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------------------------
# launcher
# ---------------------------------------------------------------------------------------------------------------------

def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _spawn_entry(local_rank: int, world: int, port: int, argv):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["WORLD_SIZE"] = str(world)
    os.environ["RANK"] = str(local_rank)
    os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    worker(parse_args(argv))


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # not under torchrun: create the N ranks here (one process per GPU)
        import torch.multiprocessing as mp
        if not args.selftest_launch and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but this node has {torch.cuda.device_count()} GPU(s)")
        port = _free_port()
        log(f"spawning {args.gpus} ranks (MASTER_ADDR=127.0.0.1 port {port})")
        mp.spawn(_spawn_entry, args=(args.gpus, port, argv), nprocs=args.gpus, join=True)
        return
    worker(args)


def worker(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher created WORLD_SIZE={world} rank(s)")
    dist = None
    if args.selftest_launch:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an AMD GPU (no CPU fallback exists)")
        if local_rank >= torch.cuda.device_count():
            raise SystemExit(f"bench.py: rank {rank} wants GPU {local_rank}, the node has {torch.cuda.device_count()}")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.selftest_launch:
            dist_mod.init_process_group(args.backend, rank=rank, world_size=world)
        else:
            dist_mod.init_process_group(args.backend, rank=rank, world_size=world, device_id=dev)
        dist = dist_mod
    seen_world = dist.get_world_size() if dist is not None else 1
    if seen_world != args.gpus:
        raise SystemExit(f"bench.py: process group reports {seen_world} ranks, --gpus {args.gpus}")
    if args.selftest_launch:
        t = torch.ones(1)
        if dist is not None:
            dist.all_reduce(t)
        if rank == 0:
            from transkun_amd.dist import shard_chains
            shards = [shard_chains(args.nbatch, seen_world, r) if args.scaling == "strong" else (0, args.nbatch) for r in range(seen_world)]
            print(json.dumps({"selftest": "launch", "n_gpus": seen_world, "allreduce_sum": float(t.item()),
                              "backend": args.backend if dist is not None else None, "scaling": args.scaling,
                              "chains_per_rank": [e - b for b, e in shards]}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    import importlib
    from transkun_amd import CRF, _lib, synth
    from transkun_amd.dist import fused_loss_allreduce, max_over_ranks
    nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")   # the module, not the class
    _lib.set_impl(args.impl)

    T, B = args.T, args.nbatch
    B_total = B
    if args.scaling == "strong" and seen_world > 1:
        # SURVEY 8e's stand-alone form: ONE [T,T,NBatch] problem, its chains cut across the ranks (each rank holds its own shard's
        # scores; chains are independent, so there is still no data-path collective) -- the regime where a rank's sweep is
        # hand-off-bound (44-88 chains), which the weak curve does not show
        from transkun_amd.dist import shard_chains
        cb, ce = shard_chains(B, seen_world, rank)
        B = ce - cb
    seed = 1234 + 1000 * rank
    intervals = synth.synthetic_intervals(T, B, seed=seed)               # (host work first: the device does not sit idle between
    score, noise = synth.crf_inputs(T, B, seed, dev, "randn")            # the generation of its inputs and the warmup steps)
    score.requires_grad_(); noise.requires_grad_()
    nseg = max(B // 88, 1)

    def step():
        score.grad = None; noise.grad = None
        lp = CRF.NeuralSemiCRFInterval(score, noise).logProb(intervals)  # the public call, Python lists in
        loss = lp.sum() * (-1.0 / nseg)                                  # train.py:187 (-logp.sum(-1).mean()) as one reduction and one scale
        if dist is not None:
            fused_loss_allreduce(loss, float(T), float(nseg))            # train.py:215-217, fused to one [3] over RCCL
        loss.backward()

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    log(f"inputs ready: T={T} B={B} world={seen_world} ({'RCCL' if dist is not None else 'single process'}) cpu_count={os.cpu_count()}")
    for _ in range(args.warmup):
        step()
    sync_all()
    log("warmup done")
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    elapsed_local = time.perf_counter() - t0
    elapsed = max_over_ranks(elapsed_local, dev)
    value = (1 if args.scaling == "strong" else seen_world) * args.steps / elapsed
    log(f"timed region done: {elapsed / args.steps * 1e3:.3f} ms/step")
    # the same K steps four more times (each bracketed like the timed region): the spread of the headline on THIS box.  `value`
    # stays the first region's.
    repeats = [elapsed / args.steps * 1e3]
    for _ in range(4 if not args.no_extra else 0):
        sync_all()
        tr = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync_all()
        repeats.append(max_over_ranks(time.perf_counter() - tr, dev) / args.steps * 1e3)
    # N > 1: what the scaling number is made of (the first multi-GPU run should explain itself): every rank's own time per step,
    # and the [3] loss all-reduce on its own
    diag = None
    if dist is not None:
        diag = {}
        try:
            mine = torch.tensor([elapsed_local / args.steps * 1e3], dtype=torch.float64, device=dev)
            allr = [torch.zeros_like(mine) for _ in range(seen_world)]
            dist.all_gather(allr, mine)
            diag["per_rank_ms_per_step"] = [round(float(x.item()), 4) for x in allr]
            probe = torch.zeros((), device=dev)
            for _ in range(3):
                fused_loss_allreduce(probe, 1.0, 1.0)
            torch.cuda.synchronize(dev); dist.barrier(); torch.cuda.synchronize(dev)
            tA = time.perf_counter()
            for _ in range(20):
                fused_loss_allreduce(probe, 1.0, 1.0)
            torch.cuda.synchronize(dev)
            diag["loss_allreduce3_us"] = round(max_over_ranks((time.perf_counter() - tA) / 20, dev) * 1e6, 2)
        except Exception as ex:
            diag["diag_error"] = repr(ex)[:200]

    # (the shader clock climbs for tens of milliseconds once the device is busy: two warm calls + five timed ones of a 1-2 ms kernel
    # measure the climb -- the exact scorer backward reads 2.40 ms that way and 2.12 from the tenth call on -- so the matrix-pipe
    # kernels below are timed over 20 calls after 8)
    def ev_time(fn, n, warm=2):
        for _ in range(warm):
            fn()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(dev)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / n        # ms

    # ---- roofline of the dominant kernel: the log-partition forward sweep -------------------------
    s_d, n_d = score.detach(), noise.detach()
    nrep = max(args.steps, 10)
    fwd_ms = ev_time(lambda: nsci._logz_fwd_raw(s_d, n_d, want_v=True), nrep)
    abytes = algorithmic_bytes_logz_fwd(T, B)
    achieved = abytes / (fwd_ms * 1e-3) / 1e9
    traffic, traffic_note = (None, "")
    if rank == 0 and args.gpus == 1 and not args.no_live_traffic and args.impl == 0:
        traffic, traffic_note = _traffic_live(T, B)
    if traffic is None:
        live_note = traffic_note
        traffic, traffic_note = _traffic_for(T, B)
        if live_note:
            traffic_note += " (live measurement not available: " + live_note + ")"
    roofline = {"bound": "hbm", "kernel": "semicrf_logz_fwd (log-partition forward sweep)",
                "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                "algorithmic_bytes": abytes, "us_per_launch": round(fwd_ms * 1e3, 2)}
    log(f"logz_fwd: {fwd_ms * 1e3:.1f} us/launch = {achieved:.1f} GB/s algorithmic")

    extra = {}
    if len(repeats) > 1:
        srt = sorted(repeats)
        extra["headline_ms_per_step_repeats"] = [round(v, 4) for v in repeats]
        extra["headline_steps_per_s_min_median_max"] = [round(seen_world * 1e3 / srt[-1], 1), round(seen_world * 1e3 / srt[len(srt) // 2], 1),
                                                        round(seen_world * 1e3 / srt[0], 1)]
    if not args.no_extra:
        try:
            _extras(extra, args, dev, rank, dist, score, noise, intervals, nseg, ev_time, fwd_ms)
        except Exception as ex:                       # the headline line must not depend on the extras
            extra["extras_error"] = repr(ex)[:300]
            log("extras failed: " + repr(ex))

    cpu_baseline = None
    if rank == 0 and seen_world == 1 and not args.no_cpu_baseline:
        cpu_baseline = _cpu_baseline(args, s_d, n_d, T, B, nseg)

    if rank == 0:
        line = {
            "metric": "semi-CRF logProb+backward steps/sec at T=1024, NBatch=352" if (T, B_total) == (1024, 352)
                      else f"semi-CRF logProb+backward steps/sec at T={T}, NBatch={B_total}",
            "value": round(value, 3), "unit": "steps/s", "n_gpus": seen_world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4), "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"NeuralSemiCRFInterval(score, noise).logProb(intervals) fwd+bwd through the public API "
                                   f"(Python interval lists packed inside the timed region), T={T}, NBatch={B} per GPU"
                                   + (f" (= NBatch {B_total} cut across {seen_world} ranks)" if args.scaling == "strong" else "") + ", "
                                   f"fp32, exact-hash randn-like scores resident in HBM",
                       "T": T, "NBatch": B, "impl": args.impl,
                       "parallelism": f"chains sharded, {seen_world} rank(s) seen by the process group, no data-path collective"
                                      + ("; [3] fp32 loss all-reduce per step over RCCL" if seen_world > 1 else "")},
            "roofline": roofline, "cpu_baseline": cpu_baseline,
            # BASELINE.json's second metric ("decode segments/sec", configs[2]: T=2048, NBatch=352, forcedStartPos set) as a named field
            "decode": extra.pop("_decode", None), "extra": extra,
        }
        if diag is not None:
            line["multi_gpu_diagnostics"] = diag
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# ---------------------------------------------------------------------------------------------------------------------
# pieces
# ---------------------------------------------------------------------------------------------------------------------

def _kernel_source_sha() -> str:
    h = hashlib.sha256()
    for f in ("persist.hip", "common.h"):
        with open(os.path.join(ROOT, "transkun_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _traffic_live(T: int, B: int):
    """HBM bytes per launch of the forward sweep, measured IN THIS RUN: two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    cannot share one; counters + kernel trace only) around tools/bench_sweep.py --ops fwd in a child process, mean over the
    dispatches of persist_sweep_kernel<0, 0, false>; FETCH_SIZE is in KB and counts the 128-byte requests of wide reads as
    64 bytes on gfx950 (x2, MI355X_MICROARCH.md, HBM / rocprofv3 section), WRITE_SIZE is in KB.  (None, reason) when the
    profiler is not there or a pass fails -- the caller then falls back to the committed, source-stamped constant."""
    import csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "no rocprofv3 on this box"
    vals = {}
    tmp = tempfile.mkdtemp(prefix="semicrf_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, ctr)
            cmd = [exe, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", out, "--", sys.executable,
                   os.path.join(ROOT, "tools", "bench_sweep.py"), "--ops", "fwd", "--n", "5", "--T", str(T), "--B", str(B)]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=180)
            except Exception as ex:
                return None, f"rocprofv3 {ctr} pass: {type(ex).__name__}"
            if r.returncode != 0:
                return None, f"rocprofv3 {ctr} pass: exit code {r.returncode}"
            files = glob.glob(os.path.join(out, "**", "*_counter_collection.csv"), recursive=True)
            got = []
            for f in files:
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == ctr and "persist_sweep_kernel<0, 0, false>" in row.get("Kernel_Name", ""):
                        got.append(float(row["Counter_Value"]))
            if not got:
                return None, f"rocprofv3 {ctr} pass: no dispatch of the forward sweep in the output"
            vals[ctr] = sum(got) / len(got)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    nbytes = int(vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024)
    return nbytes, ("measured in this run: rocprofv3 --pmc FETCH_SIZE (KB, x2: gfx950 counts the 128-byte requests of wide reads as "
                    "64) + --pmc WRITE_SIZE (KB), separate passes around tools/bench_sweep.py --ops fwd --n 5, mean per dispatch")


def _traffic_for(T: int, B: int):
    """HBM bytes per launch of the forward sweep from the round's rocprofv3 --pmc passes (profiles/traffic_latest.json,
    FETCH_SIZE x2 gfx950 correction + WRITE_SIZE).  The file is stamped with the hash of the kernel source it was
    measured on; a stale stamp gives null instead of an old number."""
    tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
    if not os.path.exists(tpath):
        return None, "no PMC pass recorded"
    try:
        d = json.load(open(tpath))
    except Exception:
        return None, "unreadable traffic file"
    if d.get("kernel_source_sha16") != _kernel_source_sha():
        return None, "PMC pass predates the current persist.hip (stamp mismatch)"
    return d.get(f"logz_fwd_T{T}_B{B}_bytes"), "rocprofv3 --pmc FETCH_SIZE (x2) + WRITE_SIZE of this kernel source, see profiles/"


def _cpu_baseline(args, s_d, n_d, T, B, nseg):
    from oracle import oracle as cpu_port          # checker/baseline leg only
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    Bs = min(args.cpu_sample_nbatch, B)
    sc = s_d[:, :, :Bs].contiguous().cpu(); nc = n_d[:, :Bs].contiguous().cpu()
    gout = torch.full((Bs,), -1.0 / nseg)

    def one(sc_, nc_, go_):
        t1 = time.perf_counter()
        logz, grad, gn = cpu_port.oploop_forward_backward(sc_, nc_)
        ds = grad * go_; dn = gn * go_                             # ComputeLogZFasterGrad.backward :472
        dt = time.perf_counter() - t1
        del ds, dn, grad
        return dt

    # thread count: the op-loop's small per-row ops stop scaling early; take the best of a short probe
    probe = {}
    for nt in (8, 16, 32):
        if nt > avail:
            continue
        torch.set_num_threads(nt)
        one(sc, nc, gout)
        probe[nt] = min(one(sc, nc, gout) for _ in range(2))
    best = min(probe, key=probe.get) if probe else max(1, min(avail, 8))
    torch.set_num_threads(best)
    log(f"cpu baseline thread probe on {Bs} chains: " + ", ".join(f"{k}t {v:.2f}s" for k, v in probe.items()) + f" -> {best} threads")
    # the full workload (all B chains), a bounded number of repetitions
    sc = s_d.contiguous().cpu(); nc = n_d.contiguous().cpu()
    gout = torch.full((B,), -1.0 / nseg)
    times = []
    t_all = time.perf_counter()
    while len(times) < 3 and time.perf_counter() - t_all < 20.0:
        times.append(one(sc, nc, gout))
    dt = min(times)
    # beside it: the PRODUCT's own host kernels (csrc/cpu_ops.cpp behind the same Python API on CPU tensors, OpenMP over the chains)
    # -- not the baseline (it is this repo's code, not the reference's algorithm as written), but what a CPU-only user of the
    # drop-in gets
    product = None
    try:
        from transkun_amd import CRF as _CRF, synth as _synth
        iv = _synth.synthetic_intervals(T, B, seed=1234)
        nthreads = max(1, min(avail, _physical_cores() or avail, 32))        # (its OpenMP loops have B / 16 blocks of work)
        torch.set_num_threads(nthreads)
        sc_p = sc.clone().requires_grad_(); nc_p = nc.clone().requires_grad_()
        pt = []
        t_all = time.perf_counter()
        while len(pt) < 3 and time.perf_counter() - t_all < 15.0:
            sc_p.grad = None; nc_p.grad = None
            t1 = time.perf_counter()
            lp = _CRF.NeuralSemiCRFInterval(sc_p, nc_p).logProb(iv)
            (lp.sum() * (-1.0 / nseg)).backward()
            pt.append(time.perf_counter() - t1)
        product = {"value": round(1.0 / min(pt), 4), "unit": "steps/s", "best_of": len(pt), "cores": nthreads,
                   "what": "NeuralSemiCRFInterval(score, noise).logProb(intervals) forward + backward on CPU tensors: this library's host "
                           "kernels (cpu_ops.cpp, OpenMP over chains, all logical CPUs), same workload"}
        del sc_p, nc_p
    except Exception as ex:
        product = {"error": repr(ex)[:200]}
    return {"value": round(1.0 / dt, 5), "unit": "steps/s", "cores": int(best), "kind": "port", "product_host_kernels": product,
            "sample": f"torch-CPU op-loop port of forward_backward + backward multiply, T={T}, all {B} chains, "
                      f"best of {len(times)} reps ({dt:.2f}s); thread probe on {Bs} chains: "
                      + ", ".join(f"{k}t={v:.2f}s" for k, v in probe.items()),
            "cpu_model": _cpu_model(), "physical_cores": _physical_cores(), "logical_cpus": os.cpu_count(),
            "cores_note": "cores = the threads the timed run used (the fastest of the probe); physical_cores / logical_cpus = what the box has"}


def _extras(extra, args, dev, rank, dist, score, noise, intervals, nseg, ev_time, fwd_ms):
    import importlib
    from transkun_amd import CRF, _lib, attributes, synth
    nsci = importlib.import_module("transkun_amd.CRF.NeuralSemiCRFInterval")
    T, B = args.T, args.nbatch
    single = dist is None
    extra["logz_fwd_us"] = round(fwd_ms * 1e3, 2)
    if single and rank == 0:
        # ---- API-level variants of the step -------------------------------------------------------------------
        pairs, offsets = nsci.pack_intervals(intervals, T, B, dev)

        def prepacked():
            score.grad = None; noise.grad = None
            lp = nsci._LogProb.apply(score, noise, pairs, offsets)
            (-lp.sum() / nseg).backward()

        def two_nodes():          # the reference's unchanged call pattern, ModelTransformer.py:263-265
            score.grad = None; noise.grad = None
            crf = CRF.NeuralSemiCRFInterval(score, noise)
            lp = crf.evalPath(intervals) - crf.computeLogZ()
            (-lp.sum() / nseg).backward()

        extra["logprob_fwd_bwd_ms_prepacked_intervals"] = round(ev_time(prepacked, 10), 3)
        extra["api_evalPath_plus_computeLogZ_ms"] = round(ev_time(two_nodes, 5), 3)
        t1 = time.perf_counter()
        for _ in range(20):
            nsci.pack_intervals(intervals, T, B, dev)
        torch.cuda.synchronize(dev)
        extra["pack_intervals_ms"] = round((time.perf_counter() - t1) / 20 * 1e3, 3)
        # BASELINE configs[1]: "NeuralSemiCRFInterval logProb fwd+bwd, T=1024, NBatch=88 (one segment x 88 pitches)" -- the same public call at
        # that shape (a parity-test case, not the headline; here for its time and the roofline fraction of its forward sweep)
        if (T, B) == (1024, 352):
            s2, n2 = synth.crf_inputs(1024, 88, 4321, dev, "randn")
            s2.requires_grad_(); n2.requires_grad_()
            iv2 = synth.synthetic_intervals(1024, 88, seed=4321)

            def config2():
                s2.grad = None; n2.grad = None
                lp = CRF.NeuralSemiCRFInterval(s2, n2).logProb(iv2)
                (lp.sum() * -1.0).backward()
            c2_ms = ev_time(config2, 20, warm=3)
            s2d, n2d = s2.detach(), n2.detach()
            c2_fwd = ev_time(lambda: nsci._logz_fwd_raw(s2d, n2d, want_v=True), 20)
            extra["config2_T1024_B88"] = {"logprob_fwd_bwd_ms": round(c2_ms, 4), "steps_per_s": round(1e3 / c2_ms, 1),
                                          "logz_fwd_us": round(c2_fwd * 1e3, 1),
                                          "logz_fwd_frac_of_8TBs": round(algorithmic_bytes_logz_fwd(1024, 88) / (c2_fwd * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                          "note": "hand-off-bound (64 dependent blocks; the ring alone needs 86 us); cells loaded with the ordinary cache "
                                                  "policy at this size (csrc/persist.hip: cell_policy_nt)"}
            del s2, n2, s2d, n2d
        log("api-level variants done; decode next")

        # ---- decode (BASELINE configs[2]): T=2048, NBatch=352, forcedStartPos=[4]*NBatch ---------------------------
        Td, Bd = 2048, 352
        start = [4] * Bd
        st_t = torch.tensor(start, dtype=torch.int32, device=dev)
        # ("model" first: timed BEHIND the randn case, whose three discarded results are 2 M tuples, the model case's 35 k tuples took 6.5 ms per
        # call instead of 1.7 -- CPython's allocator handing fragmented arenas back and forth, tools/decode_host_probe.py)
        for kind in ("model", "randn"):
            sd, nd = synth.crf_inputs(Td, Bd, 1234, dev, kind)
            crf_d = CRF.NeuralSemiCRFInterval(sd, nd)
            res = crf_d.decode(forcedStartPos=start)
            nint = sum(len(x) for x in res)
            del res
            gc.collect()                 # (a full collection of the previous phase's objects inside the timed calls cost 10 ms in one run)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(3):
                crf_d.decode(forcedStartPos=start)
            torch.cuda.synchronize(dev)
            dt = (time.perf_counter() - t1) / 3
            dk = ev_time(lambda: nsci._viterbi_raw(sd, nd, st_t, False), 5) * 1e-3
            t1 = time.perf_counter()
            for _ in range(3):
                crf_d.decode_packed(forcedStartPos=start)
            dpk = (time.perf_counter() - t1) / 3
            tag = f"decode_T2048_B352_{kind}"
            extra[tag + "_ms_end_to_end_packed_arrays"] = round(dpk * 1e3, 3)      # decode_packed: the same path as int32 arrays
            extra[tag + "_intervals"] = nint
            extra[tag + "_ms_end_to_end_python_lists"] = round(dt * 1e3, 3)
            extra[tag + "_segments_per_s_end_to_end"] = round((Bd / 88) / dt, 2)
            extra[tag + "_ms_device"] = round(dk * 1e3, 3)
            extra[tag + "_segments_per_s_device"] = round((Bd / 88) / dk, 1)
            if kind == "randn":
                dbytes = algorithmic_bytes_logz_fwd(Td, Bd)          # the Viterbi sweep reads the same cells as the forward sweep
                extra["_decode"] = {
                    "config": f"NeuralSemiCRFInterval.decode(forcedStartPos=[4]*{Bd}), T={Td}, NBatch={Bd} (= {Bd // 88} segments of 88 chains), "
                              f"randn scores ({nint} intervals); decoded intervals bit-identical to the reference (tests)",
                    "segments_per_s_device": round((Bd / 88) / dk, 1), "ms_device": round(dk * 1e3, 3),
                    "segments_per_s_api": round((Bd / 88) / dt, 2), "ms_api_python_lists": round(dt * 1e3, 3),
                    "ms_api_packed_arrays": round(dpk * 1e3, 3),
                    "roofline": {"bound": "hbm", "kernel": "semicrf_viterbi (sweep + backtrack + pack)", "algorithmic_bytes": dbytes,
                                 "achieved": round(dbytes / dk / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(dbytes / dk / 1e9 / HBM_PEAK_GBS, 4)}}
            del sd, nd, crf_d
        log("decode done; interval scorer next")

        # ---- the upstream T x T interval-score construction, same NBatch and T, D = 256: kernels only ----------
        from transkun_amd.scorer import _interval_score_raw
        lib = _lib.load()
        Cq, Dq = B, 256
        qq = synth.hash_normal(Cq * T * Dq, 5, dev).view(Cq, T, Dq)
        kk = synth.hash_normal(Cq * T * Dq, 6, dev).view(Cq, T, Dq)
        dd = synth.hash_normal(Cq * T, 7, dev).view(Cq, T)
        Sq, _ = _interval_score_raw(qq, kk, dd, T, Cq, Dq, 1.0 / 16, 0, False)
        extra["interval_score_fwd_ms"] = round(ev_time(lambda: _interval_score_raw(qq, kk, dd, T, Cq, Dq, 1.0 / 16, 0, False), 20, warm=8), 3)
        # (opt-in: three exact bf16 limbs per operand, six limb products on the bf16 matrix instructions -- fp32-grade, not bit-identical)
        extra["interval_score_fwd_bf16x3_ms"] = round(ev_time(lambda: _interval_score_raw(qq, kk, dd, T, Cq, Dq, 1.0 / 16, 0, 4), 20, warm=8), 3)
        dq = torch.empty_like(qq); dk2 = torch.empty_like(kk); ddg = torch.empty_like(dd)
        nws = int(lib.interval_score_bwd_workspace_bytes(Cq, T, Dq))
        wsq = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)

        def _bwd():
            _lib.check(lib.interval_score_bwd_ws(_lib.ptr(Sq), _lib.ptr(qq), _lib.ptr(kk), Cq, T, Dq, Dq, Dq, 1.0 / 16, 0,
                                                 _lib.ptr(dq), _lib.ptr(dk2), _lib.ptr(ddg), Dq, Dq, 1, _lib.ptr(wsq), nws,
                                                 _lib.stream_of(Sq)), "interval_score_bwd_ws")
        extra["interval_score_bwd_ms"] = round(ev_time(_bwd, 20, warm=8), 3)

        def _bwd3():            # opt-in: the two products on the three-limb bf16 kernels (length_scaling | SEMICRF_LEN_BF16X3 = 16)
            _lib.check(lib.interval_score_bwd_ws(_lib.ptr(Sq), _lib.ptr(qq), _lib.ptr(kk), Cq, T, Dq, Dq, Dq, 1.0 / 16, 16,
                                                 _lib.ptr(dq), _lib.ptr(dk2), _lib.ptr(ddg), Dq, Dq, 1, _lib.ptr(wsq), nws,
                                                 _lib.stream_of(Sq)), "interval_score_bwd_ws")
        extra["interval_score_bwd_bf16x3_ms"] = round(ev_time(_bwd3, 20, warm=8), 3)
        # the scorer is the path's matrix-bound kernel: its own roofline object (the line's `roofline` is the HBM-bound sweep)
        sflop = 2.0 * Cq * (T * (T + 1) / 2) * Dq                           # lower triangle only
        extra["scorer_roofline"] = {
            "bound": "mfma", "kernel": "interval_score_tiled_kernel<4> (exact fp32, v_mfma_f32_32x32x2_f32)", "unit": "TFLOP/s",
            "achieved": round(sflop / (extra["interval_score_fwd_ms"] * 1e-3) / 1e12, 2), "peak": 157.3,
            "frac": round(sflop / (extra["interval_score_fwd_ms"] * 1e-3) / 1e12 / 157.3, 4), "algorithmic_flop": sflop,
            "note": "through the pooled score tensor: the zeros above the diagonal are written once per buffer, not in the timed calls; counters: profiles/r06_derived.json "
                    "(matrix pipe busy fraction, clock under load; r05_derived.json when this round's pass is missing); cycle stamps: tools/tiled_probe.py -- bound by the CU's "
                    "vector-memory address path (DESIGN.md section 3)",
            "backward_frac": round(2 * sflop / (extra["interval_score_bwd_ms"] * 1e-3) / 1e12 / 157.3, 4),
            # the opt-in three-limb kernels run on the BF16 pipe: six bf16 limb products per fp32 product, priced against the dense bf16
            # peak (2.5 PFLOP/s), algorithmic flops only (the forward also multiplies the part of its diagonal tiles above the diagonal)
            "bf16x3": {"bound": "mfma", "unit": "TFLOP/s (bf16, executed = 6 x algorithmic)", "peak": 2500.0,
                       "forward_achieved": round(6 * sflop / (extra["interval_score_fwd_bf16x3_ms"] * 1e-3) / 1e12, 1),
                       "forward_frac": round(6 * sflop / (extra["interval_score_fwd_bf16x3_ms"] * 1e-3) / 1e12 / 2500.0, 4),
                       "backward_achieved": round(12 * sflop / (extra["interval_score_bwd_bf16x3_ms"] * 1e-3) / 1e12, 1),
                       "backward_frac": round(12 * sflop / (extra["interval_score_bwd_bf16x3_ms"] * 1e-3) / 1e12 / 2500.0, 4),
                       "forward_speedup_vs_exact_fp32": round(extra["interval_score_fwd_ms"] / extra["interval_score_fwd_bf16x3_ms"], 3),
                       "backward_speedup_vs_exact_fp32": round(extra["interval_score_bwd_ms"] / extra["interval_score_bwd_bf16x3_ms"], 3),
                       "note": "backward = the repack kernel + both products; the pipe alone sustains 2.0-2.2 PFLOP/s on these boxes (profiles/r05_mfma_peak.txt)"}}
        extra["interval_score_config"] = f"T={T}, chains={Cq}, D={Dq}, exact-fp32 MFMA, lower triangle"
        del qq, kk, dd, Sq, dq, dk2, ddg, wsq
        log("interval scorer done; segment-shaped path next")

        # ---- the segment-shaped path at the model's real shape (BASELINE configs[3], scorer + CRF part) ---------------
        from transkun_amd.fused import scorer_crf_logprob
        from transkun_amd.scorer import ScaledInnerProductIntervalScorer
        Ts, P, D = 691, 90, 256
        m = ScaledInnerProductIntervalScorer(D, 1).to(dev)
        for N in (1, 4):
            ctx = (synth.hash_normal(N * P * Ts * D, 11, dev).view(N, P, Ts, D) * 0.5).requires_grad_()
            iv = synth.synthetic_intervals(Ts, N * P, seed=11)

            def seg_step(fused, projection="merged"):
                m.zero_grad(); ctx.grad = None
                if fused:
                    lp = scorer_crf_logprob(m, ctx, iv, projection=projection)
                else:
                    S, b = m(ctx)
                    lp = CRF.NeuralSemiCRFInterval(S.flatten(-2, -1), b.flatten(-2, -1)).logProb(iv)
                (-lp.view(N, -1).sum(-1).mean() / 50).backward()

            def seg_decode():
                with torch.no_grad():
                    S, b = m(ctx)
                    pr, offs = nsci._viterbi_raw(S.flatten(-2, -1), b.flatten(-2, -1), None, False)
                    return attributes.attribute_input_packed(ctx.detach(), pr, offs)

            tag = f"segment_T691_P90_N{N}"
            extra[tag + "_scorer_crf_logprob_fwd_bwd_ms_fused"] = round(ev_time(lambda: seg_step(True), 5), 3)
            extra[tag + "_scorer_crf_logprob_fwd_bwd_ms_fused_separate_projection"] = round(ev_time(lambda: seg_step(True, "separate"), 5), 3)
            extra[tag + "_scorer_crf_logprob_fwd_bwd_ms_unfused"] = round(ev_time(lambda: seg_step(False), 5), 3)
            extra[tag + "_scorer_decode_features_ms_device"] = round(ev_time(seg_decode, 5), 3)
            m.contraction = "bf16x3"
            extra[tag + "_scorer_crf_logprob_fwd_bwd_ms_fused_bf16x3"] = round(ev_time(lambda: seg_step(True), 5), 3)
            extra[tag + "_scorer_decode_features_ms_device_bf16x3"] = round(ev_time(seg_decode, 5), 3)
            m.contraction = "fp32"
            del ctx
        # ---- BASELINE configs[3] end to end: 16 s of audio -> frames (T = 691) -> log-mel (fp32) -> backbone in bf16 (a stand-in
        # with the reference Backbone's interface; the real one is out of scope) -> fp32 ctx -> scorer + CRF logProb (fused route)
        from transkun_amd.frontend import MelSpectrum, makeFrame, normalize_gain
        audio = synth.hash_normal(2 * 705600, 41, dev).view(1, 2, 705600) * 0.1
        mel = MelSpectrum(4096, 30, 8000, 229, 44100, nExtraWins=5, log=True, toMono=True).to(dev)
        backbone = synth.StandInBackbone().to(dev).to(torch.bfloat16)
        iv1 = synth.synthetic_intervals(Ts, P, seed=12)

        def full_forward():
            with torch.no_grad():
                feat = mel(normalize_gain(makeFrame(audio, 1024, 4096)))
                c = backbone(feat.to(torch.bfloat16)).float()
                return scorer_crf_logprob(m, c, iv1)
        extra["segment_full_forward_standin_backbone_ms"] = round(ev_time(full_forward, 5), 3)
        extra["segment_full_forward_config"] = ("16 s @ 44.1 kHz stereo -> makeFrame (T=691) -> 6-window log-mel (229 bands, fp32) -> stand-in "
                                                "backbone in bf16 -> fp32 ctx [1,90,691,256] -> Linear + interval scorer + CRF logProb (forward only)")
        del audio, mel, backbone
        # ---- the transcription segment loop: decode -> heads -> events -> next forced start, F recordings in
        # lock step, incomplete-event merge on the host; the shipped geometry (16 s segments, 8 s hop: T = 691, 90 symbols) ----
        from transkun_amd.transcribe import SegmentTranscriber
        torch.manual_seed(0)                                          # the heads are random-init: the event count depends on them
        tr = SegmentTranscriber(D).to(dev).eval()
        n_audio = int(56.0 * tr.fs)                                   # 56 s of audio: 9 segments per recording
        for Fn in (1, 4):
            plan = tr.segment_plan(n_audio)
            nseg_f = len(plan["begins"])
            ctxs = [(synth.hash_normal(P * plan["nFrame"] * D, 31 + i, dev).view(1, P, plan["nFrame"], D) * 0.5) for i in range(3)]
            fns = [(lambda i, T, f=f: ctxs[(i + f) % 3]) for f in range(Fn)]
            tr.transcribe_many(fns, [n_audio] * Fn)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            ev = tr.transcribe_many(fns, [n_audio] * Fn)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t1
            tag = f"transcribe_loop_T691_P90_F{Fn}"
            extra[tag + "_segments_per_s_end_to_end"] = round(Fn * nseg_f / dt, 1)
            extra[tag + "_ms_per_step"] = round(dt / nseg_f * 1e3, 3)
            extra[tag + "_events"] = sum(len(x) for x in ev)
        del tr, ctxs
        log("segment-shaped path done; train-shaped step next")

    # ---- train.py-shaped step (BASELINE configs[4]): every rank, its own 4 segments ---------------------------------
    from transkun_amd.trainstep import SegmentModel, train_step
    Ts, P, D, N = 691, 90, 256, 4
    torch.manual_seed(0)
    model = SegmentModel(D).to(dev)
    ctx = synth.hash_normal(N * P * Ts * D, 21 + rank, dev).view(N, P, Ts, D) * 0.5
    iv = synth.synthetic_intervals(Ts, N * P, seed=21 + rank)
    ncoll = [0]
    from transkun_amd.dist import FlatGradBucket
    bucket = FlatGradBucket(model.parameters())      # gradients live in one persistent flat buffer; reduce-scatter + all-gather

    ctx.requires_grad_()          # the backbone's output: its gradient (dctx) is part of the step ...

    def tstep():
        ctx.grad = None           # ... but it is an intermediate tensor in the model: no accumulation into a leaf's .grad (255 MB add)
        _, ncoll[0] = train_step(model, ctx, iv, bucket=bucket)

    def sync_all():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    TS_WARM, TS_N = 5, 20         # (steady state: see ev_time's note on the clock's climb)
    for _ in range(TS_WARM):
        tstep()
    sync_all()
    t1 = time.perf_counter()
    for _ in range(TS_N):
        tstep()
    sync_all()
    from transkun_amd.dist import max_over_ranks
    dt = max_over_ranks((time.perf_counter() - t1) / TS_N, dev)
    world = dist.get_world_size() if dist is not None else 1
    extra["train_step_ms"] = round(dt * 1e3, 3)
    extra["train_step_segments_per_s"] = round(world * N / dt, 2)
    # the same step with the opt-in three-limb bf16 kernels (fp32-grade, not bit-identical -- the headline and train_step_ms stay exact
    # fp32): scorer.contraction = "bf16x3" (forward contraction and backward products) and "bf16x3-bwd" (backward products only: at
    # this shape the exact forward kernel is the faster one)
    # ... "bf16x3-train": the backward products and the merged projection's two NN GEMMs (csrc/proj_gemm3.hip); "bf16x3-all": everything
    for cname, key in (("bf16x3", "train_step_ms_bf16x3"), ("bf16x3-bwd", "train_step_ms_bf16x3_bwd"), ("bf16x3-train", "train_step_ms_bf16x3_train"),
                       ("bf16x3-all", "train_step_ms_bf16x3_all")):
        model.scorer.contraction = cname
        for _ in range(TS_WARM):
            tstep()
        sync_all()
        t1 = time.perf_counter()
        for _ in range(TS_N):
            tstep()
        sync_all()
        extra[key] = round(max_over_ranks((time.perf_counter() - t1) / TS_N, dev) * 1e3, 3)
    model.scorer.contraction = "fp32"
    extra["train_step_config"] = (f"per rank: 4 segments x 90 symbols x T=691, D=256: Linear + interval scorer + fused CRF log_prob, "
                                  f"(loss/50).backward(), one [3] all-reduce, gradient exchange of "
                                  f"{sum(p.numel() for p in model.parameters()) / 1e6:.2f} M fp32 parameters from a persistent flat bucket: "
                                  + (f"{ncoll[0]} collective(s) per step (reduce-scatter + all-gather over RCCL, started by the backward pass "
                                     f"on a side stream), {bucket.bytes_per_rank / 1e6:.1f} MB sent per rank" if ncoll[0] else
                                     "no exchange at one rank (N > 1: reduce-scatter + all-gather over RCCL, started by the backward pass on a side stream)")
                                  + "; backbone out of scope (ctx is the input)")
    extra["train_step_collectives"] = int(ncoll[0])
    extra["train_step_exchange_bytes_per_rank"] = int(bucket.bytes_per_rank)
    if dist is not None:
        # the gradient exchange on its own (reduce-scatter + all-gather of the flat bucket, in place, current stream)
        try:
            for _ in range(2):
                bucket.exchange()
            sync_all()
            t2 = time.perf_counter()
            for _ in range(5):
                bucket.exchange()
            torch.cuda.synchronize(dev)
            extra["train_step_exchange_ms"] = round(max_over_ranks((time.perf_counter() - t2) / 5, dev) * 1e3, 3)
        except Exception as ex:
            extra["train_step_exchange_error"] = repr(ex)[:200]


def _physical_cores():
    """Distinct (physical id, core id) pairs of /proc/cpuinfo: the box's physical cores (SMT siblings counted once)."""
    try:
        seen, phys, core = set(), None, None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    seen.add((phys, core))
                phys = core = None
        return len(seen) or None
    except Exception:
        return None


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
