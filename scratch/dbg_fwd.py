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
def rel(a,b): return float((a.double()-b.double()).norm()/(b.double().norm()+1e-30))
with torch.no_grad():
    o32 = SPVCNNRef(m.state_dict())(coords, torch.from_numpy(feats))
    o64 = SPVCNNRef(m.state_dict(), dtype=torch.float64)(coords, torch.from_numpy(feats))
    mg = m.cuda()
    got = mg({"lidar": SparseTensor(torch.from_numpy(feats).cuda(), torch.from_numpy(coords).cuda(), 1)})
for k in ('feats','seg_logit'):
    print(k, 'cpu32 vs 64', rel(o32[k], o64[k]), 'gpu vs 64', rel(got[k].cpu(), o64[k]), 'gpu vs cpu32', rel(got[k].cpu(), o32[k]))
