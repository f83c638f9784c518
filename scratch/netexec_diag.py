import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, numpy as np
from test_gpu_netexec import tta_model, lidar_of, rel_err
from latte_amd import synth
from oracle import ts_ref
cs, fs = [], []
for j in range(3):
    pts, inten = synth.make_scan(2048, j, beams=16)
    c, f, inds, inv = ts_ref.quantize_scan(pts, inten, scale=20)
    cs.append(c); fs.append(f)
coords, feats = ts_ref.sparse_collate(cs, fs)
scans = dict(coords=coords, feats=feats)
m = tta_model(1.0); net = m.net_3d
lidar = lidar_of(scans)
torch.manual_seed(3)
out = net(lidar); wgt = torch.randn_like(out)
(out * wgt).sum().backward()
got = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
for p in net.parameters(): p.grad = None
ref = net._forward_modules(lidar)
print("fwd rel", rel_err(out, ref))
(ref * wgt).sum().backward()
want = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
for n in want:
    print(f"{n:45s} {rel_err(got[n], want[n]):.2e}  |g| {float(want[n].norm()):.3e}")
# module path twice: run-to-run noise of the module path itself (atomics?)
for p in net.parameters(): p.grad = None
ref2 = net._forward_modules(lidar)
(ref2 * wgt).sum().backward()
w2 = {n: p.grad.clone() for n, p in net.named_parameters() if p.grad is not None}
print("module path vs itself: worst", max(rel_err(w2[n], want[n]) for n in want))
