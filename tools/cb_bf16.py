#!/usr/bin/env python3
"""bf16x3 split-precision conv kernel against the fp32 default, per U-Net level at the a2d2 sizes: time and error."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import ops, synth
from latte_amd import sparse_modules as spnn
from latte_amd.quantize import quantize_scan_gpu
from latte_amd.point_voxel import initial_voxelize
from latte_amd.tensor import PointTensor, SparseTensor

dev = torch.device("cuda", 0)
coords = []
for b in range(6):
    p, i = synth.make_scan(131072, b, beams=64, full_360=False)
    c, f, _, _ = quantize_scan_gpu(torch.from_numpy(p).to(dev), torch.from_numpy(i).to(dev))
    coords.append(torch.cat([c, torch.full((c.shape[0], 1), b, dtype=torch.int32, device=dev)], 1))
C4 = torch.cat(coords, 0)
x = initial_voxelize(PointTensor(torch.zeros(C4.shape[0], 4, device=dev), C4.float()), 0.05, 0.05)

def us_of(fn, reps=10):
    fn(); fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best

s = 1
levels = {1: [(96, 96), (32, 32)], 2: [(96, 96), (32, 32)], 4: [(128, 128), (64, 64)], 8: [(256, 256)], 16: [(256, 256)]}
variants = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3000]
while True:
    km, _ = spnn.build_kernel_map(x, 3, 1, 1)
    plan = km.plan()
    V = km.results.shape[1]
    M = int((km.results >= 0).sum())
    for cin, cout in levels[s]:
        feats = torch.randn(V, cin, device=dev)
        w = torch.randn(27, cin, cout, device=dev) / (27 * cin) ** 0.5
        p32, p16 = ops.pack_weights(w, 0), ops.pack_weights(w, 4)
        ref = ops.spconv_plan(feats, p32, plan, 27, V, cout)
        t32 = us_of(lambda: ops.spconv_plan(feats, p32, plan, 27, V, cout))
        for v in variants:
            try:
                got = ops.spconv_plan(feats, p16, plan, 27, V, cout, variant=v)
            except Exception as e:
                print(json.dumps(dict(stride=s, C=(cin, cout), variant=v, err=str(e)[:80])))
                continue
            err = float((got - ref).norm() / ref.norm())
            emax = float((got - ref).abs().max() / ref.abs().max())
            t = us_of(lambda: ops.spconv_plan(feats, p16, plan, 27, V, cout, variant=v))
            fl = 2.0 * M * cin * cout
            print(json.dumps(dict(stride=s, V=V, C=(cin, cout), variant=v, us_fp32=round(t32, 1), us=round(t, 1),
                                  speedup=round(t32 / t, 2), tf_equiv=round(fl / t / 1e6, 1), rel_err=err, max_err=emax)), flush=True)
    if s == 16:
        break
    _, oc = spnn.build_kernel_map(x, 2, 2, 1)
    nx = SparseTensor(torch.zeros(oc.shape[0], 4, device=dev), oc, 2 * s)
    nx.cmaps, nx.kmaps = x.cmaps, x.kmaps
    nx.cmaps[nx.stride] = oc
    x, s = nx, 2 * s
