"""Train-shaped step scorer -> CRF logProb -> backward: unfused (dense dS) vs fused (scorer_crf_logprob).  GPU box only."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transkun_amd import CRF, synth
from transkun_amd.fused import scorer_crf_logprob
from transkun_amd.scorer import ScaledInnerProductIntervalScorer
dev = torch.device("cuda:0")
for (N, P, T) in ((4, 88, 1024), (4, 90, 691), (4, 96, 691), (1, 90, 691), (1, 96, 691)):
    m = ScaledInnerProductIntervalScorer(256).to(dev)
    with torch.no_grad(): m.map[0].weight.mul_(0.3)
    ctx0 = torch.randn(N, P, T, 256, device=dev) * 0.5
    iv = synth.synthetic_intervals(T, N * P, seed=7)
    def step(fused):
        m.zero_grad(); ctx = ctx0.clone().requires_grad_()
        if fused: lp = scorer_crf_logprob(m, ctx, iv, projection=fused)
        else:
            S, b = m(ctx); lp = CRF.NeuralSemiCRFInterval(S.flatten(-2, -1), b.flatten(-2, -1)).logProb(iv)
        (-lp.sum() / 50).backward()
    for fused in (False, "separate", "merged"):
        for _ in range(2): step(fused)
        torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); t0 = time.perf_counter()
        for _ in range(5): step(fused)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"N={N} P={P} T={T} {('fused/' + fused) if fused else 'unfused':14s}: {dt*1e3:7.2f} ms per fwd+bwd step, peak memory {torch.cuda.max_memory_allocated()/2**30:.2f} GiB", flush=True)
