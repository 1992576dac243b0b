import os
import sys

import pytest

try:        # torch first: its bundled ROCm runtime then serves the whole process (see pathtracer_amd/core.py)
    import torch  # noqa: F401
except ImportError:
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a gfx950 GPU (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle
    if not os.path.exists(oracle.LIB_PATH):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    return oracle


@pytest.fixture(scope="session")
def bunny():
    """CBbunny: positions, aux, indices (as loaded), plus the product-built BVH."""
    import pathtracer_amd as pt
    pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", "CBbunny.ply"))
    nodes, ridx, depth = pt.build_bvh(pos, idx)
    return dict(positions=pos, aux=aux, indices_in=idx, nodes=nodes, indices=ridx, depth=depth)


@pytest.fixture(scope="session")
def monkey():
    import pathtracer_amd as pt
    pos, aux, idx = pt.load_ply(os.path.join(ROOT, "tests", "golden", "models", "CBmonkey.ply"))
    nodes, ridx, depth = pt.build_bvh(pos, idx)
    return dict(positions=pos, aux=aux, indices_in=idx, nodes=nodes, indices=ridx, depth=depth)


@pytest.fixture(scope="session")
def courtyard_small():
    import pathtracer_amd as pt
    pos, aux, idx = pt.scenes.courtyard(60000, seed=3)
    nodes, ridx, depth = pt.build_bvh(pos, idx)
    return dict(positions=pos, aux=aux, indices_in=idx, nodes=nodes, indices=ridx, depth=depth)


@pytest.fixture(scope="session", params=["records64", "records48", "treelets"])
def gpu_ctx(request):
    """One context per record layout: every GPU test runs on the 64 B records, on the 48 B records and on the 128 B
    two-level treelet records."""
    import pathtracer_amd as pt
    from pathtracer_amd import core
    ctx = pt.PathTracer(0)
    ctx.set_option(core.OPT_RECORD_LAYOUT, {"records64": 0, "records48": 1, "treelets": 2}[request.param])
    yield ctx
    ctx.close()
