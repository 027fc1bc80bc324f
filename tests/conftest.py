import importlib
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (REPO, os.path.join(REPO, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_box():
    if not os.path.exists("/dev/kfd"):
        return False
    try:
        import torch
        return bool(torch.cuda.is_available())
    except Exception:
        return False


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    """On a box WITH a GPU every test carries the `gpu` marker, so `pytest -m gpu` there also runs the host-side
    parity tests (OBJ ingest, C-ABI, the oracle against its fixtures, the gloo world-2 reduce): the box that renders
    is the box whose libc / locale / compiler those have to pass on.  Without a GPU (the build container) only the
    tests that need one carry it, and `-m "not gpu"` runs the rest."""
    if _gpu_box():
        for it in items:
            if "gpu" not in it.keywords:
                it.add_marker(pytest.mark.gpu)


@pytest.fixture(scope="session")
def pt():
    """The product package (directory name has '-', hence importlib)."""
    mod = importlib.import_module("single-file-vulkan-pathtracing_amd")
    so = os.path.join(os.path.dirname(mod.__file__), "libpt_amd.so")
    host = os.path.join(os.path.dirname(mod.__file__), "libpt_host.so")
    if not (os.path.exists(so) and os.path.exists(host)):
        mod.build()
    return mod


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle binding (test infrastructure)."""
    from oracle import pt_oracle
    pt_oracle.lib()
    return pt_oracle


@pytest.fixture(scope="session")
def cornell_arrays(pt):
    return pt.load_obj(pt.ASSET_CORNELL)


@pytest.fixture(scope="session")
def cornell_oracle(orc, cornell_arrays):
    return orc.Scene(*cornell_arrays)


@pytest.fixture(scope="session")
def gpu_ctx(pt):
    ctx = pt.Context(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def cornell_gpu(pt, gpu_ctx, cornell_arrays):
    sc = pt.Scene(gpu_ctx, *cornell_arrays)
    yield sc
    sc.close()
