import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "dab-radio_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu tests are skipped (not failed) where no GPU exists, e.g. when the whole suite is run on CPU
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "reference_vectors.npz"))


def pytest_sessionstart(session):
    """DAB_FUZZ_OFFSET=k (development, tools/fuzz_suite.sh): every integer seed handed to numpy.random.default_rng is shifted by k x 1000003,
    so that the parity tests that build their inputs and their oracle answers on the fly run on other data than the committed seeds'.
    (Tests against committed golden vectors do not draw their inputs from a generator and are unaffected; a test that asserts a property
    of ITS seed's data -- "at least three uncorrectable super frames" -- may need a look when it fails under an offset.)"""
    k = int(os.environ.get("DAB_FUZZ_OFFSET", "0") or 0)
    if k:
        import numpy as np
        plain = np.random.default_rng

        def shifted(seed=None, *a, **kw):
            if isinstance(seed, (int, np.integer)):
                seed = int(seed) + k * 1000003
            return plain(seed, *a, **kw)

        shifted.plain = plain                      # (tests/fig_ensemble.py regenerates a FIXTURE's capture: it takes the shift out for that)
        np.random.default_rng = shifted
