import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the CPU oracle is a small-op workload: spread over the hundreds of hardware threads of a GPU host, torch's intra-op pool
    # thrashes (measured: 131 s per oracle step at 256 threads against 0.4 s at 32)
    import torch
    torch.set_num_threads(min(os.cpu_count() or 1, 32))


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for it in items:
        if 'gpu' in it.keywords:
            it.add_marker(skip)


@pytest.fixture(autouse=True)
def _fresh_dropout_epoch():
    """The dropout seed epoch of libqagnn_hip (advanced by every hipGraph replay of qagnn_amd.graphed) is process-wide device state:
    tests that predict keep masks from their seeds need it at 0."""
    import torch
    if torch.cuda.is_available():
        from qagnn_amd import ops
        k = ops._K
        if k is not None and hasattr(k, 'seed_epoch_set'):
            k.seed_epoch_set(0)
    yield
