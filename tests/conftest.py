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


class _WavefrontByDefault:
    """The product package as the tests see it.  Since API version 5 pt_params_default returns PT_PIPELINE_AUTO, which renders LDS-class
    scenes (the Cornell box, the instanced grid) with the fused kernel.  This suite was written against the wavefront pipeline as the
    default and most of its tests are ABOUT that pipeline (queues, sample groups, pipelines on streams, term logs, ray sorting), so here
    `default_params` names PT_PIPELINE_WAVEFRONT wherever a test does not name a pipeline itself; everything else is the module's own.
    PT_PIPELINE_AUTO has its own tests (test_auto_pipeline_*, the full-size frames through AUTO), which pass `pipeline=pt.PIPELINE_AUTO`
    or use `pt.library_default_params`; __graft_entry__.smoke() and bench.py use the library's real default."""

    def __init__(self, mod):
        self.__dict__["_mod"] = mod

    def __getattr__(self, name):
        return getattr(self._mod, name)

    def default_params(self, **kw):
        kw.setdefault("pipeline", self._mod.PIPELINE_WAVEFRONT)
        return self._mod.default_params(**kw)

    def library_default_params(self, **kw):
        return self._mod.default_params(**kw)


@pytest.fixture(scope="session")
def pt():
    """The product package (directory name has '-', hence importlib)."""
    mod = importlib.import_module("single-file-vulkan-pathtracing_amd")
    so = os.path.join(os.path.dirname(mod.__file__), "libpt_amd.so")
    host = os.path.join(os.path.dirname(mod.__file__), "libpt_host.so")
    if not (os.path.exists(so) and os.path.exists(host)):
        mod.build()
    return _WavefrontByDefault(mod)


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


@pytest.fixture(params=["wavefront", "auto"])
def any_pipeline(request, pt):
    """tests that are not ABOUT the queues run through both the wavefront pipeline (what `pt.default_params` names in this suite) and what a
    caller of pt_params_default really gets (PT_PIPELINE_AUTO: the fused kernels for the scenes that live in LDS).  -> the `pipeline=` value."""
    return pt.PIPELINE_WAVEFRONT if request.param == "wavefront" else pt.PIPELINE_AUTO
