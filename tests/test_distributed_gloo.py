"""world_size-2 gloo test (CPU) of the N>1 path: chain sharding, the fused loss all-reduce and the
max-over-ranks timing used by bench.py.  Chain independence -- the property that lets the batch axis shard
with no data-path collective -- is checked with the CPU oracle (test infrastructure)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import oracle as o
    from transkun_amd import synth
    from transkun_amd.dist import fused_loss_allreduce, max_over_ranks, shard_chains
    T, B = 40, 24
    score, noise = synth.crf_inputs(T, B, 77, "cpu")
    b0, b1 = shard_chains(B, world, rank)
    s_sh, n_sh = score[:, :, b0:b1].contiguous().numpy(), noise[:, b0:b1].contiguous().numpy()
    _, logz = o.alpha(s_sh, n_sh)
    dec = o.viterbi(s_sh, n_sh)
    loss = torch.tensor(float(-logz.sum()))
    stats = fused_loss_allreduce(loss, total_len=T * (b1 - b0), n_batch=b1 - b0)
    tmax = max_over_ranks(0.5 + rank, "cpu")
    np.save(os.path.join(out_dir, f"logz_{rank}.npy"), logz)
    torch.save({"stats": stats, "tmax": tmax, "range": (b0, b1), "dec": dec}, os.path.join(out_dir, f"r{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_allreduce(tmp_path):
    from oracle import oracle as o
    from transkun_amd import synth
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    T, B = 40, 24
    score, noise = synth.crf_inputs(T, B, 77, "cpu")
    _, logz_full = o.alpha(score.numpy(), noise.numpy())
    dec_full = o.viterbi(score.numpy(), noise.numpy())
    parts = [torch.load(os.path.join(tmp_path, f"r{r}.pt"), weights_only=False) for r in range(world)]
    ranges = [p["range"] for p in parts]
    assert ranges[0][0] == 0 and ranges[-1][1] == B and ranges[0][1] == ranges[1][0]
    assert all((b - a) % 4 == 0 for a, b in ranges)
    logz_cat = np.concatenate([np.load(os.path.join(tmp_path, f"logz_{r}.npy")) for r in range(world)])
    assert np.allclose(logz_cat, logz_full, rtol=1e-6)                   # chains are independent
    assert sum((p["dec"] for p in parts), []) == dec_full
    for p in parts:
        assert abs(float(p["stats"][0]) - float(-logz_full.sum())) < 1e-2   # SUM over ranks
        assert float(p["stats"][1]) == T * B and float(p["stats"][2]) == B
        assert p["tmax"] == pytest.approx(1.5)                             # MAX over ranks


def test_shard_chains_properties():
    from transkun_amd.dist import shard_chains
    for n, w in ((352, 8), (352, 1), (360, 8), (90, 4), (88, 3), (7, 2)):
        spans = [shard_chains(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= (4 if n % 4 == 0 else 1)


def _grad_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transkun_amd.dist import allreduce_gradients_flat
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(7, 5), torch.nn.Linear(5, 3))
    m[0].bias.requires_grad_(False)                                   # frozen parameters are skipped
    x = torch.full((4, 7), float(rank + 1))
    m(x).sum().backward()
    local = [p.grad.clone() for p in m.parameters() if p.requires_grad]
    n = allreduce_gradients_flat(m.parameters(), bucket_bytes=64)     # tiny buckets: several collectives
    torch.save({"local": local, "reduced": [p.grad.clone() for p in m.parameters() if p.requires_grad], "n": n},
               os.path.join(out_dir, f"g{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_allreduce(tmp_path):
    """SUM (no divide) of every trainable parameter's gradient across ranks, in flat buckets
    (the semantics of TrainUtil.average_gradients, TrainUtil.py:36-48)."""
    world = 2
    mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(os.path.join(str(tmp_path), f"g{k}.pt")) for k in range(world)]
    assert r[0]["n"] == r[1]["n"] and r[0]["n"] >= 2
    for i in range(len(r[0]["local"])):
        want = r[0]["local"][i] + r[1]["local"][i]
        assert torch.allclose(r[0]["reduced"][i], want) and torch.allclose(r[1]["reduced"][i], want)


def test_bench_launcher_creates_its_ranks():
    """`python bench.py --gpus 2` without torchrun must create two ranks by itself and report the world size the
    process group saw; a launcher/--gpus mismatch must refuse instead of printing a 1-rank line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch", "--backend", "gloo"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["allreduce_sum"] == 2.0
    assert line["scaling"] == "weak" and line["chains_per_rank"] == [352, 352]
    # --scaling strong: the 352 chains of ONE problem cut across the ranks (44 per GPU at 8)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch", "--backend", "gloo",
                        "--scaling", "strong"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["scaling"] == "strong" and line["chains_per_rank"] == [176, 176]
    env2 = dict(env, WORLD_SIZE="1", RANK="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--selftest-launch"],
                       capture_output=True, text=True, timeout=120, env=env2)
    assert r.returncode != 0 and "WORLD_SIZE=1" in (r.stderr + r.stdout)


def _cpu_log_prob(scorer, ctx, intervals):
    """CPU stand-in for scorer -> CRF -> logProb with the same signature and output shape ([N*P]); only the exchange
    around it is under test here (the HIP path itself is covered by the -m gpu tests)."""
    y = scorer.map[0](ctx)                                   # [N,P,T,2D+1]
    return -(y ** 2).mean(dim=(-1, -2)).reshape(-1)


def _train_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transkun_amd import synth
    from transkun_amd.trainstep import SegmentModel, train_step
    torch.manual_seed(0)                                      # same replica on every rank (train.py:69-73 loads one checkpoint)
    model = SegmentModel(size=8, total_params=5000)
    N, P, T = 2, 3, 10
    ctx = synth.hash_normal(N * P * T * 8, 50 + rank, "cpu").view(N, P, T, 8)
    stats, ncoll = train_step(model, ctx, None, seconds_per_segment=16.0, log_prob=_cpu_log_prob, bucket_bytes=4096)
    # the same step without any exchange, for the expected sums
    local = SegmentModel(size=8, total_params=5000)
    local.load_state_dict(model.state_dict())
    logp = _cpu_log_prob(local.scorer, ctx, None).view(N, -1)
    loss = -logp.sum(-1).mean()
    (loss / 50).backward()
    torch.save({"stats": stats, "ncoll": ncoll, "loss": float(loss),
                "grads": [p.grad.clone() for p in model.parameters()],
                "local": [(p.grad.clone() if p.grad is not None else torch.full_like(p, 1e-3)) for p in local.parameters()]},
               os.path.join(out_dir, f"t{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_train_shaped_step_two_ranks(tmp_path):
    """train.py:186-189 + :214-229 over two ranks: the [3] stats are summed, every parameter's gradient is the SUM of the
    ranks' local gradients (no divide, TrainUtil.py:44-49), exchanged as a few flat buckets."""
    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"t{r}.pt"), weights_only=False) for r in range(world)]
    for p in parts:
        assert float(p["stats"][0]) == pytest.approx(sum(q["loss"] for q in parts), rel=1e-6)
        assert float(p["stats"][1]) == 16.0 * 2 * world and float(p["stats"][2]) == world
        assert 1 < p["ncoll"] < 10                                   # buckets, not one message per parameter
        for i, g in enumerate(p["grads"]):
            want = sum(q["local"][i] for q in parts)
            assert torch.allclose(g, want, rtol=1e-6, atol=1e-9)


def _bucket_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transkun_amd import synth
    from transkun_amd.dist import FlatGradBucket
    from transkun_amd.trainstep import SegmentModel, train_step
    torch.manual_seed(0)
    model = SegmentModel(size=8, total_params=5000)
    bucket = FlatGradBucket(model.parameters())
    ptrs = [p.grad.data_ptr() for p in model.parameters()]
    N, P, T = 2, 3, 10
    out = []
    for step in range(2):                                     # twice: the buffer persists, zero() replaces grad = None
        ctx = synth.hash_normal(N * P * T * 8, 50 + rank + 10 * step, "cpu").view(N, P, T, 8)
        stats, ncoll = train_step(model, ctx, None, log_prob=_cpu_log_prob, bucket=bucket)
        local = SegmentModel(size=8, total_params=5000)
        local.load_state_dict(model.state_dict())
        logp = _cpu_log_prob(local.scorer, ctx, None).view(N, -1)
        (-logp.sum(-1).mean() / 50).backward()
        out.append({"ncoll": ncoll, "grads": [p.grad.clone() for p in model.parameters()],
                    "local": [(p.grad.clone() if p.grad is not None else torch.full_like(p, 1e-3)) for p in local.parameters()],
                    "same_storage": [p.grad.data_ptr() for p in model.parameters()] == ptrs})
    torch.save(out, os.path.join(out_dir, f"b{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_bucket_two_ranks(tmp_path):
    """FlatGradBucket: .grad of every parameter is a view of one persistent buffer (same storage step after step), the
    hook-started exchange sums over the ranks (one all_reduce over gloo; reduce-scatter + all-gather over RCCL, covered by
    tests/test_gpu_parity.py::test_rccl_world1_train_step on the GPU)."""
    world = 2
    mp.spawn(_bucket_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"b{r}.pt"), weights_only=False) for r in range(world)]
    for step in range(2):
        for p in parts:
            assert p[step]["ncoll"] == 1 and p[step]["same_storage"]
            for i, g in enumerate(p[step]["grads"]):
                want = sum(q[step]["local"][i] for q in parts)
                assert torch.allclose(g, want, rtol=1e-6, atol=1e-9)


def _emulated_reduce_scatter(output, input, op=None, group=None, async_op=False):
    """reduce_scatter_tensor on a backend that has none (gloo): the SUM of everybody's `input`, my slice of it into `output`
    -- which the bucket passes as a VIEW of `input` (in place), exactly as it does over RCCL."""
    full = input.clone()
    dist.all_reduce(full, op=dist.ReduceOp.SUM, group=group)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    output.copy_(full.view(world, -1)[rank])


def _rs_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from transkun_amd import synth
    from transkun_amd.dist import FlatGradBucket
    from transkun_amd.trainstep import SegmentModel, train_step
    torch.manual_seed(0)
    model = SegmentModel(size=8, total_params=5003)           # an odd total: the flat buffer is padded to world * 64 elements
    bucket = FlatGradBucket(model.parameters())
    # drive the branch the RCCL backend takes (slice ownership flat.view(world, -1)[rank], in-place reduce-scatter, in-place
    # all-gather) over gloo: the backend NAME is faked, reduce_scatter_tensor is emulated, all_gather_into_tensor is gloo's own
    bucket._backend = lambda: "nccl"
    real_rs = dist.reduce_scatter_tensor
    dist.reduce_scatter_tensor = _emulated_reduce_scatter
    N, P, T = 2, 3, 10
    out = []
    try:
        for step in range(2):
            ctx = synth.hash_normal(N * P * T * 8, 70 + rank + 10 * step, "cpu").view(N, P, T, 8)
            if step == 1:
                for p in model.parameters():                  # what optimizer.zero_grad(set_to_none=True) leaves behind
                    p.grad = None
            stats, ncoll = train_step(model, ctx, None, log_prob=_cpu_log_prob, bucket=bucket)
            local = SegmentModel(size=8, total_params=5003)
            local.load_state_dict(model.state_dict())
            logp = _cpu_log_prob(local.scorer, ctx, None).view(N, -1)
            (-logp.sum(-1).mean() / 50).backward()
            out.append({"ncoll": ncoll, "bytes": bucket.bytes_per_rank, "rebound": bucket.rebound,
                        "grads": [p.grad.clone() for p in model.parameters()],
                        "views": all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket._views)),
                        "local": [(p.grad.clone() if p.grad is not None else torch.full_like(p, 1e-3)) for p in local.parameters()]})
    finally:
        dist.reduce_scatter_tensor = real_rs
    torch.save(out, os.path.join(out_dir, f"rs{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_grad_bucket_reduce_scatter_branch_two_ranks(tmp_path):
    """The reduce-scatter + all-gather branch of FlatGradBucket.exchange (the one RCCL takes) at world size 2, through a faked
    backend name over gloo -- so that the slice arithmetic is not executed for the first time on an 8-GPU node -- and the
    rebinding of gradients that `p.grad = None` / zero_grad(set_to_none=True) moved out of the flat buffer."""
    world = 2
    mp.spawn(_rs_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    parts = [torch.load(os.path.join(tmp_path, f"rs{r}.pt"), weights_only=False) for r in range(world)]
    for step in range(2):
        for p in parts:
            assert p[step]["ncoll"] == 2 and p[step]["bytes"] > 0 and p[step]["views"]
            for i, g in enumerate(p[step]["grads"]):
                want = sum(q[step]["local"][i] for q in parts)
                assert torch.allclose(g, want, rtol=1e-6, atol=1e-9)
    assert parts[0][0]["rebound"] == 0 and parts[0][1]["rebound"] == 3          # step 1 found every .grad gone and put it back


def test_flat_grad_bucket_guards_single_process():
    """No process group: the aliasing checks and the step bookkeeping of FlatGradBucket."""
    from transkun_amd.dist import FlatGradBucket
    torch.manual_seed(0)
    m = torch.nn.Linear(4, 3)
    b = FlatGradBucket(m.parameters())
    x = torch.ones(2, 4)
    # (1) gradients that autograd wrote into fresh tensors (after p.grad = None) are moved back before the exchange
    b.zero(); b.arm()
    for p in m.parameters():
        p.grad = None
    m(x).sum().backward()
    assert b.wait() == 0 and b.rebound == 2
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(b.params, b._views))
    assert torch.allclose(m.weight.grad, torch.full((3, 4), 2.0)) and torch.allclose(b.flat[:12], torch.full((12,), 2.0))
    # (2) micro-batches: unarmed backward passes accumulate, arm() before the last one, one exchange
    b.zero()
    m(x).sum().backward()
    b.arm()
    m(x).sum().backward()
    assert b.wait() == 0
    assert torch.allclose(m.weight.grad, torch.full((3, 4), 4.0))
    # (3) a second backward pass behind ONE arm(): its gradients came after the exchange -> wait() raises
    b.zero(); b.arm()
    m(x).sum().backward()
    m(x).sum().backward()
    with pytest.raises(RuntimeError, match="after this step's exchange"):
        b.wait()
    # (4) wait() without arm(): the exchange happens there; a next step is clean again
    b.zero()
    m(x).sum().backward()
    assert b.wait() == 0 and b._exchanged
