"""Where does interval_score_tile3_kernel with a row constant (the merged projection + scorer.contraction 'bf16x3-all') differ from the
exact kernel?  T691_P90 golden inputs; per-cell errors of both kernels against a float64 evaluation of the same formula."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from segment_common import SEGMENT_CASES, segment_inputs
from transkun_amd import _lib
from transkun_amd.fused import merged_weights
from transkun_amd.scorer import _interval_score_raw, proj_forward, slot_pitch, QPAD, BF16X3
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "T691_P90"
N, P, T, D = SEGMENT_CASES[name][:4]
ctx, W, bias, iv, gout, starts = segment_inputs(name, dev)
C = N * P; size = D
Wm, bm = merged_weights(W, bias, D)
x3 = ctx.reshape(C, T, size).contiguous()
for prec in (0, 1):
    zc = proj_forward(x3.view(-1, size), Wm, bm, size, Wt=getattr(Wm, "_semicrf_T", None), prec=prec).view(C, T, size + QPAD)
    z64 = (x3.double().view(-1, size) @ Wm.double().t() + bm.double()).view(C, T, size + QPAD)
    print(f"projection prec={prec}: max |zc - fp64| main {float((zc[..., :size].double() - z64[..., :size]).abs().max()):.3e} "
          f"c {float((zc[..., size].double() - z64[..., size]).abs().max()):.3e} diag {float((zc[..., size+1].double() - z64[..., size+1]).abs().max()):.3e}"
          f"  (max |c| {float(z64[..., size].abs().max()):.3f}, max |z| {float(z64[..., :size].abs().max()):.3f})")
zc = proj_forward(x3.view(-1, size), Wm, bm, size, Wt=getattr(Wm, "_semicrf_T", None), prec=0).view(C, T, size + QPAD)
qs = 1.0 / math.sqrt(D)
pitch = slot_pitch(P, T, size, N)
res = {}
for tag, fs in (("exact", 2), ("bf16x3", 2 | BF16X3)):
    S, nz = _interval_score_raw(zc[..., :size], x3, zc[..., size + 1], T, C, size, qs, 0, fs, P, pitch, rowc=zc[..., size])
    res[tag] = S.clone()
torch.cuda.synchronize()
tri = torch.tril(torch.ones(T, T, dtype=torch.bool, device=dev))
e = torch.arange(T, device=dev).view(T, 1); b = torch.arange(T, device=dev).view(1, T)
ln = (e - b).abs().double()
for c in (0, 1, 45, 89):
    n, p = divmod(c, P); slot = n * pitch + p
    z = zc[c].double(); x = x3[c].double()
    ref = qs * ((z[:, :size] @ x.t()) + z[:, size].view(T, 1)) * ln + torch.diag(z[:, size + 1])
    for tag in ("exact", "bf16x3"):
        d = (res[tag][:, :, slot].double() - ref).abs() * tri
        i = int(d.argmax()); ee, bb = divmod(i, T)
        bound = 2.0 ** -21 * qs * ln * (z[:, :size].abs() @ x.abs().t())
        print(f"chain {c} {tag}: max |S - fp64| {float(d.max()):.3e} at (e={ee}, b={bb}; e-b={ee-bb}) mean {float(d.sum() / tri.sum()):.3e}; "
              f"max err/bound {float((d / (bound + 1e-30) * tri).max()):.2f}; mean signed {float(((res[tag][:, :, slot].double() - ref) * tri).sum() / tri.sum()):.3e}")
d = ((res["exact"] - res["bf16x3"]).abs() * tri.unsqueeze(-1))
print("exact vs bf16x3: max", float(d.max()), "mean", float(d.sum() / (tri.sum() * d.shape[2])))
# error by 32-block row / column classes of the worst chain
dd = d[:, :, 0]
print("by row mod 128 //32:", [float(dd[r::1][(torch.arange(T, device=dev) % 128 // 32) == r4].max()) for r4 in range(4) for r in [0]][:4])
print("status", _lib.device_status())
ps = d.amax(dim=(0, 1))
print("per-slot max diff:", [f"{float(v):.2g}" for v in ps])
w = int(ps.argmax()); dw = d[:, :, w]; i = int(dw.argmax()); ee, bb = divmod(i, T)
print("worst slot", w, "at e,b", ee, bb, "exact", float(res["exact"][ee, bb, w]), "bf16x3", float(res["bf16x3"][ee, bb, w]))
bad = (dw > 1e-2)
print("bad cells in worst slot:", int(bad.sum()), "rows", sorted(set((bad.nonzero()[:, 0] // 32).tolist()))[:40], "cols", sorted(set((bad.nonzero()[:, 1] // 32).tolist()))[:40])
rows = bad.nonzero()
print("sample bad cells:", rows[:10].tolist())
