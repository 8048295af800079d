"""Where the host time of SegmentTranscriber.transcribe_many goes (cProfile, cumulative), F recordings of 56 s."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from transkun_amd import synth
from transkun_amd.transcribe import SegmentTranscriber
dev = torch.device("cuda:0")
Fn = int(sys.argv[1]) if len(sys.argv) > 1 else 4
D, P = 256, 90
torch.manual_seed(0)
tr = SegmentTranscriber(D).to(dev).eval()
n_audio = int(56.0 * tr.fs)
plan = tr.segment_plan(n_audio)
ctxs = [(synth.hash_normal(P * plan["nFrame"] * D, 31 + i, dev).view(1, P, plan["nFrame"], D) * 0.5) for i in range(3)]
fns = [(lambda i, T, f=f: ctxs[(i + f) % 3]) for f in range(Fn)]
tr.transcribe_many(fns, [n_audio] * Fn); torch.cuda.synchronize()
for rep in range(4):
    t = time.perf_counter(); ev = tr.transcribe_many(fns, [n_audio] * Fn); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"F={Fn}: {dt*1e3:.1f} ms for {len(plan['begins'])} steps, {sum(len(e) for e in ev)} events, {Fn*len(plan['begins'])/dt:.0f} segments/s")
    del ev
pr = cProfile.Profile(); pr.enable(); tr.transcribe_many(fns, [n_audio] * Fn); torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
