import hashlib
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
for p in (REPO, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def sha1(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


_cache = {}


def load_golden(name):
    """Fixture dict + the seeded inputs it was generated from (checksum verified)."""
    if name in _cache:
        return _cache[name]
    import golden_inputs as gi
    z = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    X, Q = getattr(gi, name + "_inputs")()
    assert str(z["inputs_sha1"]) == sha1(X) + sha1(Q), "seeded inputs drifted from fixture " + name
    _cache[name] = (z, X, Q)
    return _cache[name]


@pytest.fixture(scope="session")
def golden():
    return load_golden


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
