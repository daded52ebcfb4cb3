import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "variants: measured-and-rejected kernel forms of csrc/variants/ in the A/B library "
                                       ".ab_libs/variants.so -- not product kernels: run only with -m variants")


def pytest_collection_modifyitems(config, items):
    """tests of the rejected kernel forms (marker `variants`) stay out of every run that does not name the marker: the
    driver's `-m gpu` count is product kernels only, and the A/B library they load is not part of the product."""
    if "variants" in (config.getoption("markexpr") or ""):
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("variants") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
