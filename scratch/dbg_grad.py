import sys; sys.path.insert(0,'.')
import torch, numpy as np
from tests.test_gpu_spvcnn import make_model
from oracle.spvcnn_ref import SPVCNNRef
from latte_amd import synth
from latte_amd.tensor import SparseTensor
from oracle import ts_ref
cs, fs = [], []
for j in range(3):
    pts, inten = synth.make_scan(2048, j, beams=16)
    c, f, inds, inv = ts_ref.quantize_scan(pts, inten, 20)
    cs.append(c); fs.append(f)
coords, feats = ts_ref.sparse_collate(cs, fs)
m = make_model(0.5)
names = [f"{mn}.{pn}"[len("net_3d."):] for mn, mod in m.named_modules() if isinstance(mod, torch.nn.BatchNorm1d) for pn,_ in mod.named_parameters()]
g = torch.Generator().manual_seed(0)
tgt = torch.randint(0, 9, (coords.shape[0],), generator=g)
ref = SPVCNNRef(m.state_dict())
ps = ref.parameters_requiring_grad(names)
out = ref(coords, torch.from_numpy(feats))
out['feats'].retain_grad()
loss = torch.nn.functional.cross_entropy(out['seg_logit'], tgt) + out['seg_logit2'].square().mean()
loss.backward()
mg = m.cuda(); mg.requires_grad_(False)
for mod in mg.modules():
    if "norm" in type(mod).__name__.lower(): mod.requires_grad_(True)
lidar = SparseTensor(torch.from_numpy(feats).cuda(), torch.from_numpy(coords).cuda(), 1)
got = mg({"lidar": lidar})
got['feats'].retain_grad()
l2 = torch.nn.functional.cross_entropy(got["seg_logit"], tgt.cuda()) + got["seg_logit2"].square().mean()
l2.backward()
print('feats grad err', float((got['feats'].grad.cpu()-out['feats'].grad).norm()/out['feats'].grad.norm()))
gp = dict(mg.named_parameters())
for n, pr in zip(names, ps):
    pg = gp["net_3d."+n].grad.cpu()
    print(f"{n:40s} {float((pg-pr.grad).norm()/(pr.grad.norm()+1e-12)):.2e}  |g|={float(pr.grad.norm()):.3e}")
