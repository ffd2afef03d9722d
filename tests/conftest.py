import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(params=[1, 2], ids=["tile16", "tile8"])
def tile_mode(request):
    """Both tilings of the time-unrolled kernels (16-agent workgroups on v_mfma_f32_16x16x4, 8-agent workgroups on
    v_mfma_f32_4x4x1_16B; include/socialways_hip.h: sw_set_tile_mode) - by default the library picks by batch size."""
    from socialways_amd import _lib as L
    L.load().sw_set_tile_mode(request.param)
    yield request.param
    L.load().sw_set_tile_mode(0)
