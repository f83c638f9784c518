#!/usr/bin/env python3
"""What a dense fp32 GEMM reaches on this GPU through torch (hipBLASLt / rocBLAS): a practical reference point for the
fp32 MFMA peak the sparse conv is priced against."""
import torch
torch.backends.cuda.matmul.allow_tf32 = False
for n in (2048, 4096, 8192):
    a = torch.randn(n, n, device="cuda"); b = torch.randn(n, n, device="cuda")
    for _ in range(3): c = a @ b
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    reps = 10
    for _ in range(reps): c = a @ b
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print(f"sgemm {n}^3: {ms:.3f} ms, {2 * n ** 3 / ms / 1e9:.1f} TFLOP/s", flush=True)
