#!/usr/bin/env python3
"""1x1 'conv' = dense [N,Cin] x [Cin,Cout] through the sparse-conv kernel with an identity one-offset plan (statistics in
its epilogue) against torch.mm (hipBLASLt) + the separate statistics pass, at the point-branch / shortcut shapes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from latte_amd import ops


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(3):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, 1e3 * e0.elapsed_time(e1) / reps)
    return best


def main():
    dev = torch.device("cuda", 0)
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 310120
    for n, cin, cout in [(N, 32, 256), (N, 256, 128), (N, 128, 96), (N, 128, 96), (N // 2, 128, 96), (N // 5, 192, 128),
                         (N // 12, 384, 256), (N // 5, 32, 64), (N // 12, 64, 128)]:
        x = torch.randn(n, cin, device=dev)
        w = torch.randn(1, cin, cout, device=dev) * 0.05
        nbr = torch.arange(n, dtype=torch.int32, device=dev).view(1, n)
        plan = ops.kmap_plan(nbr)
        packed = ops.pack_weights(w, 0)
        ref = x @ w[0]
        out = ops.spconv_plan(x, packed, plan, 1, n, cout)
        err = float((out - ref).abs().max() / ref.abs().max())
        t_mm = timeit(lambda: torch.mm(x, w[0]))
        t_stats = timeit(lambda: ops.bn_stats(ref))
        t_conv = timeit(lambda: ops.spconv_plan(x, packed, plan, 1, n, cout))
        gamma = torch.ones(cout, device=dev); beta = torch.zeros(cout, device=dev)
        t_comp = timeit(lambda: ops.conv_bn_fwd(x, packed, plan, 1, n, cout, gamma, beta, 1e-5, None, True))
        t_sep = timeit(lambda: ops.bn_apply(torch.mm(x, w[0]), *ops.bn_stats(ref), gamma, beta, 1e-5, relu=True))
        print(f"N={n} {cin}->{cout}: mm {t_mm:.1f} us, stats pass {t_stats:.1f} us, conv kernel {t_conv:.1f} us (err {err:.1e}); "
              f"composite conv+stats+bn {t_comp:.1f} us vs mm + apply (stats not counted) {t_sep:.1f} us", flush=True)


if __name__ == "__main__":
    main()
