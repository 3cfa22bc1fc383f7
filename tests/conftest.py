import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")
    # Stress runs: D4W_SEED_SHIFT=k shifts every integer seed handed to numpy.random.default_rng (and torch.manual_seed) by k,
    # so the seeded random cases of the whole suite become different cases.  Off by default: the committed runs stay pinned.
    shift = int(os.environ.get("D4W_SEED_SHIFT", "0"))
    if shift:
        import numpy as np
        orig = np.random.default_rng

        def shifted(seed=None, *a, **k):
            return orig(seed + shift if isinstance(seed, int) else seed, *a, **k)
        np.random.default_rng = shifted
        try:
            import torch
            tms = torch.manual_seed
            torch.manual_seed = lambda s: tms(int(s) + shift)
        except Exception:
            pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load
