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
