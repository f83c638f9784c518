"""CPU study: MFMA strip utilisation of the mask-sorted order at every U-Net level of the bench workload."""
import numpy as np, sys
sys.path.insert(0, '.')
from latte_amd import synth
from oracle import ts_ref
w = synth.workload("nuscenes_34k_w5")
coords = []
for b in range(w["B"]):
    p, i = synth.make_scan(w["n"], b, beams=w["beams"], full_360=w["full_360"])
    c, f, _, _ = ts_ref.quantize_scan(p, i)
    coords.append(np.concatenate([c, np.full((c.shape[0], 1), b, np.int32)], 1))
C = np.concatenate(coords, 0).astype(np.int32)
ts = 1
np.save('scratch/lvl_masks_dummy.npy', np.zeros(1))
while ts <= 16:
    _, _, oc, res = ts_ref.build_kmap(C, ts, 3, 1)
    K, V = res.shape
    valid = res >= 0
    mask = (valid.astype(np.int64) << np.arange(K)[:, None]).sum(0)
    np.save(f'scratch/lvl_masks_{ts}.npy', mask)
    M = int(valid.sum())
    def util(order, rows=16):
        v = valid[:, order]
        Vp = (V + rows - 1) // rows * rows
        v = np.concatenate([v, np.zeros((K, Vp - V), bool)], 1).reshape(K, -1, rows)
        act = v.any(2).sum()
        return M / (act * rows)
    o_mask = np.argsort(mask, kind='stable')
    nuniq = len(np.unique(mask))
    print(f"stride {ts}: V={V} M/V={M/V:.2f} distinct masks {nuniq}; util storage {util(np.arange(V)):.3f} mask-sort {util(o_mask):.3f}", flush=True)
    if ts < 16:
        C = ts_ref.spdownsample(C, 2, ts) if False else ts_ref.build_kmap(C, ts, 2, 2)[2]
    ts *= 2
