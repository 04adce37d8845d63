import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _device_count() -> int:
    from halo2_experiments_amd import _lib
    return _lib.load().hm_device_count()


def pytest_collection_modifyitems(config, items):
    if not any("gpu" in item.keywords for item in items):
        return
    try:
        have = _device_count() > 0
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no HIP device visible (GPU tests run with -m gpu on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _init_torch_gpu_first():
    """On a GPU box initialise torch's HIP context before the library makes its first HIP call."""
    try:
        if _device_count() > 0:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
                torch.zeros(1, device="cuda")
    except Exception:
        pass
    yield


@pytest.fixture(scope="session")
def cref():
    """The C restatement oracle (built on first use)."""
    from oracle import cpu_ref
    cpu_ref.build()
    return cpu_ref


@pytest.fixture(scope="session")
def pyref():
    from oracle import bn256_ref
    return bn256_ref


@pytest.fixture(scope="session")
def golden():
    return {name: np.load(os.path.join(GOLDEN, name + ".npz")) for name in ("field", "curve", "msm", "ntt")}


def g1_equal(jac12: np.ndarray, affine8: np.ndarray) -> bool:
    """Compare a normalised G1 (x, y, 1)/(0,0,0) with an expected affine point ((0,0) = identity)."""
    jac12 = np.asarray(jac12, dtype=np.uint64).reshape(12)
    affine8 = np.asarray(affine8, dtype=np.uint64).reshape(8)
    if not jac12[8:].any():
        return not affine8.any()
    return bool(np.array_equal(jac12[:8], affine8))
