import sys; sys.path.insert(0,'.')
import torch, numpy as np
from tests.test_gpu_spvcnn import make_model
from oracle.spvcnn_ref import SPVCNNRef
from latte_amd import synth, sparse_modules as spnn
from latte_amd.tensor import SparseTensor
from oracle import ts_ref
cs, fs = [], []
for j in range(3):
    pts, inten = synth.make_scan(2048, j, beams=16)
    c, f, inds, inv = ts_ref.quantize_scan(pts, inten, 20)
    cs.append(c); fs.append(f)
coords, feats = ts_ref.sparse_collate(cs, fs)
m = make_model(0.5)
ref = SPVCNNRef(m.state_dict())
with torch.no_grad(): ref(coords, torch.from_numpy(feats))
mg = m.cuda()
acts = {}
names = {mod: n[len('net_3d.'):] for n, mod in mg.named_modules() if isinstance(mod, spnn.BatchNorm)}
orig = spnn.BatchNorm.bn_feats
def rec(self, feats, residual=None, relu=False):
    y = orig(self, feats, residual=residual, relu=relu)
    acts[names[self]] = y
    return y
spnn.BatchNorm.bn_feats = rec
with torch.no_grad():
    mg({"lidar": SparseTensor(torch.from_numpy(feats).cuda(), torch.from_numpy(coords).cuda(), 1)})
tot = 0
for n, y in acts.items():
    r = ref.acts[n]
    yc = y.cpu()
    flips = int(((yc > 0) != (r > 0)).sum())
    tot += flips
    ez = int((r == 0).sum())
    print(f"{n:32s} flips={flips} maxabs={float((yc-r).abs().max()):.2e} zeros_ref={ez} zeros_gpu={int((yc==0).sum())} n={r.numel()}")
print('total flips', tot)
