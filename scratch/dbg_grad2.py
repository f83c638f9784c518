import sys; sys.path.insert(0,'.')
import torch, numpy as np
from tests.test_gpu_spvcnn import make_model
from latte_amd import synth, functional as LF, ops
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
orig_conv_bwd = LF._SparseConv.backward
def conv_bwd(ctx, gout):
    res = orig_conv_bwd(ctx, gout)
    if res[0] is not None:
        feats_, weight = ctx.saved_tensors
        km = ctx.kmap
        nb, ns = km.pairs()
        x = feats_.detach().cpu().requires_grad_(True)
        w = weight.detach().cpu(); w = w if w.dim()==3 else w.unsqueeze(0)
        nout = km.n_in if ctx.transposed else km.n_out
        with torch.enable_grad():
            y = ts_ref.conv3d(x, w, nb.cpu().numpy(), ns.cpu().numpy(), nout, transposed=ctx.transposed)
            y.backward(gout.cpu())
        print(f"conv bwd K={km.K} T={ctx.transposed} {tuple(w.shape)} rel={rel(res[0].cpu(), x.grad):.2e}")
    return res
LF._SparseConv.backward = staticmethod(conv_bwd)
orig_bn_bwd = LF._BatchNorm.backward
def bn_bwd(ctx, dy):
    res = orig_bn_bwd(ctx, dy)
    x, y, mean, var, gamma = ctx.saved_tensors
    xc = x.detach().cpu().requires_grad_(True); gc = gamma.detach().cpu().requires_grad_(True)
    bc = torch.zeros_like(gc).requires_grad_(True)
    g = dy.cpu()
    if ctx.relu: g = g * (y.cpu() > 0)
    with torch.enable_grad():
        yy = torch.nn.functional.batch_norm(xc, None, None, gc, bc, True, 0.0, ctx.eps)
        yy.backward(g)
    print(f"bn bwd {tuple(x.shape)} relu={ctx.relu} res={ctx.has_res} dx={rel(res[0].cpu(), xc.grad) if res[0] is not None else -1:.2e} dgam={rel(res[1].cpu(), gc.grad):.2e} dbeta={rel(res[2].cpu(), bc.grad):.2e}" + (f" dres={rel(res[3].cpu(), g):.2e}" if res[3] is not None else ""))
    return res
LF._BatchNorm.backward = staticmethod(bn_bwd)
for name, F_, in (("vox", LF._Voxelize), ("devox", LF._Devoxelize)):
    pass
g = torch.Generator().manual_seed(0)
tgt = torch.randint(0, 9, (coords.shape[0],), generator=g)
mg = m.cuda(); mg.requires_grad_(False)
for mod in mg.modules():
    if "norm" in type(mod).__name__.lower(): mod.requires_grad_(True)
lidar = SparseTensor(torch.from_numpy(feats).cuda(), torch.from_numpy(coords).cuda(), 1)
got = mg({"lidar": lidar})
l2 = torch.nn.functional.cross_entropy(got["seg_logit"], tgt.cuda()) + got["seg_logit2"].square().mean()
l2.backward()
