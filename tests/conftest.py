import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
ORACLE_LIB = os.path.join(ROOT, "oracle", "_build", "libssf_oracle.so")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_present():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    """gpu-marked tests need a real MI355X: on a box without one they are skipped, not failed (the product has no CPU
    fallback: ssf_create returns SSF_ERR_NO_DEVICE there)."""
    if _gpu_present():
        return
    skip = pytest.mark.skip(reason="no HIP device on this box (gpu-marked tests run with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def _make(path):
    r = subprocess.run(["make", "-C", path], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout


@pytest.fixture(scope="session")
def oracle_lib():
    """The CPU oracle (checker).  Tests are one of the three places allowed to load it."""
    from supersurfel_fusion_amd import binding
    if not os.path.exists(ORACLE_LIB):
        _make(os.path.join(ROOT, "oracle"))
    return binding.Library(ORACLE_LIB)


@pytest.fixture(scope="session")
def product_lib():
    """The HIP product library; loads (and registers its gfx950 code objects) on a CPU box too,
    but ssf_create needs a GPU.  No fallback: a missing .so is an error."""
    from supersurfel_fusion_amd import binding
    if not os.path.exists(binding.PRODUCT_LIB):
        _make(os.path.join(ROOT, "supersurfel_fusion_amd", "csrc"))
    return binding.load_product()


@pytest.fixture(scope="session")
def lab_lib():
    """The laboratory build of the product sources (-DSSF_EXPERIMENTS: measurement arms, environment switches)"""
    from supersurfel_fusion_amd import binding
    if not os.path.exists(os.path.join(ROOT, "supersurfel_fusion_amd", "csrc", "variants", "lab", "libssf_hip.so")):
        _make(os.path.join(ROOT, "supersurfel_fusion_amd", "csrc"))
    return binding.load_lab()
