"""lt_bn_stats_static -- the entry point that allocates nothing (per-block partial sums in a file-scope device array of
the library) -- against lt_bn_stats (caller workspace), through the C-ABI, two calls in a row, the second one smaller.
Same kernels, same launch geometry => bit-equal mean / var."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _both(x):
    from latte_amd import _lib
    lib = _lib.load()
    V, C = x.shape
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    ws = torch.empty(int(lib.lt_bn_workspace_bytes(V, C)), dtype=torch.uint8, device=x.device)
    m0, v0 = torch.empty(C, device=x.device), torch.empty(C, device=x.device)
    m1, v1 = torch.full((C,), float("nan"), device=x.device), torch.full((C,), float("nan"), device=x.device)
    _lib.call("lt_bn_stats", _p(x), V, C, _p(ws), ws.numel(), _p(m0), _p(v0), stream)
    _lib.call("lt_bn_stats_static", _p(x), V, C, _p(m1), _p(v1), stream)
    torch.cuda.synchronize()
    return (m0, v0), (m1, v1)


def _static_only(x, m, v):
    from latte_amd import _lib
    V, C = x.shape
    _lib.call("lt_bn_stats_static", _p(x), V, C, _p(m), _p(v), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


def test_bn_stats_static_matches_workspace_variant_twice():
    g = torch.Generator().manual_seed(7)
    big = (torch.randn(200_003, 96, generator=g) * 2 + 3).cuda()
    small = (torch.randn(1_531, 32, generator=g) - 5).cuda()       # second call: fewer rows, fewer channels, fewer blocks
    for x in (big, small):
        (m0, v0), (m1, v1) = _both(x)
        assert torch.equal(m0, m1) and torch.equal(v0, v1)
        # the call allocates nothing: neither the device's free memory (hipMemGetInfo) nor torch's allocator moves
        # across further calls (the first call on the device, above, is where the symbol address was looked up)
        torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        res0, alloc0 = torch.cuda.memory_reserved(), torch.cuda.memory_allocated()
        for _ in range(3):
            _static_only(x, m1, v1)
        torch.cuda.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0 and torch.cuda.memory_reserved() == res0 and torch.cuda.memory_allocated() == alloc0
        assert torch.equal(m0, m1) and torch.equal(v0, v1)
        ref_m, ref_v = x.double().mean(0), x.double().var(0, unbiased=False)
        assert torch.allclose(m1.double(), ref_m, rtol=1e-5, atol=1e-6)
        assert torch.allclose(v1.double(), ref_v, rtol=1e-4, atol=1e-7)


def test_bn_stats_static_rejects_more_channels_than_its_scratch_holds():
    from latte_amd import _lib
    lib = _lib.load()
    x = torch.zeros(64, 512, device="cuda")
    m, v = torch.empty(512, device="cuda"), torch.empty(512, device="cuda")
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.lt_bn_stats_static(_p(x), 64, 512, _p(m), _p(v), stream)
    assert rc == -4 and b"exceeds" in lib.lt_last_error()
