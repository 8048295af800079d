"""Where a wave of the bf16 contraction kernel spends its cycles (probe build: python -c "from transkun_amd import _build;
_build.build_variant('sprobe', ['SEMICRF_SCORE_PROBE=1'])"; SEMICRF_LIB=transkun_amd/_variants/sprobe/libsemicrf_hip.so).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import synth
from transkun_amd.scorer import _interval_score_raw
dev = torch.device("cuda:0")
T, C, D = (int(sys.argv[1]), int(sys.argv[2]), 256) if len(sys.argv) > 2 else (1024, 352, 256)
y = synth.hash_normal(C * T * (2 * D + 1), 5, dev).view(C, T, 2 * D + 1)
q, k, dg = y[..., :D].contiguous(), y[..., D:2 * D].contiguous(), y[..., 2 * D].contiguous()
for _ in range(3):
    S, _ = _interval_score_raw(q, k, dg, T, C, D, 1.0 / 16, 0, 2 | 4)
torch.cuda.synchronize()
ncu = torch.cuda.get_device_properties(dev).multi_processor_count
nw = ncu // 64 * 64 * 8
p = S.view(-1)[C:C + nw * 4].view(nw, 4).double().cpu()
tot = p.sum(1)
names = ["barrier wait", "reads+matrix+split+stores", "fetch", "epilogue"]
print(f"T={T} C={C}: {nw} waves, cycles per wave: mean {tot.mean():.3e} (min {tot.min():.3e}, max {tot.max():.3e})")
for i, n in enumerate(names):
    print(f"  {n:28s} {p[:, i].mean():.3e}  = {100 * p[:, i].mean() / tot.mean():.1f} %   (min {p[:, i].min():.3e} max {p[:, i].max():.3e})")
nt = (T + 127) // 128; items = nt * (nt + 1) // 2 * ((C + 3) // 4 + 7) // 8 * 8 / (ncu // 64 * 64)
print(f"  ~{items:.1f} items per workgroup, {items * 4 * D / 32:.0f} half iterations: {p[:, 1].mean() / (items * 4 * D / 32):.0f} cycles of body, {p[:, 0].mean() / (items * 4 * D / 32):.0f} of barrier wait, {p[:, 2].mean() / (items * 4 * D / 32):.0f} of fetch per half iteration; {p[:, 3].mean() / items:.0f} per epilogue")
