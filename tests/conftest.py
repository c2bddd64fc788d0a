import os
import sys
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by `pytest -m gpu` on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def gpu_available():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def emul():
    """CPU emulation build of the HIP sources (tests/emul) -- kernel-logic tests without a GPU."""
    so = os.path.join(ROOT, "tests", "emul", "libzkhip_emul.so")
    srcs = [os.path.join(ROOT, "ethsnarks_amd", "csrc", f) for f in os.listdir(os.path.join(ROOT, "ethsnarks_amd", "csrc")) if f.endswith(("pp",))]
    srcs.append(os.path.join(ROOT, "tests", "emul", "hip_emul.h"))
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "emul"), "-s", "-j5"])
    return so


@pytest.fixture(scope="session")
def hip():
    """the product library on a real GPU; loud failure when it is missing"""
    from ethsnarks_amd import prover
    prover._lib = None
    prover._lib_path_loaded = None
    prover.load_library()          # raises ImportError if libzkhip.so was not built
    assert b"gfx950" in prover._lib.zk_version()
    if prover.device_count() < 1:
        pytest.fail("libzkhip.so loaded but no HIP device is visible")
    return prover
