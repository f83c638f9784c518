"""The boundary used from plain C (tests/abi/abi_smoke.c: gcc, include/latte_hip*.h, the HIP runtime's C API -- no torch,
no C++): sphash against the 64-bit FNV-1a of torchsparse, softmax + normalised entropy (latte/models/losses.py:10-19),
batch statistics through the entry point that allocates nothing, and the error convention."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "latte_amd", "abi_smoke")


@pytest.mark.gpu
def test_plain_c_consumer_of_the_abi():
    assert os.path.exists(EXE), "latte_amd/abi_smoke is built by __graft_entry__.build() (make -C latte_amd/csrc)"
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    assert "abi_smoke: OK" in r.stdout


def test_plain_c_consumer_links_without_a_gpu():
    """on a box without a GPU the program loads (liblatte_hip.so and the HIP runtime resolve) and stops at device discovery"""
    if not os.path.exists(EXE):
        pytest.skip("latte_amd/abi_smoke not built")
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the gpu-marked test runs the program")
    r = subprocess.run([EXE], capture_output=True, text=True, timeout=120)
    assert r.returncode in (2, 4), (r.returncode, r.stdout[-500:], r.stderr[-500:])
