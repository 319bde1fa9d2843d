import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden", "qwen2vl_tiny_golden.npz")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    return dict(np.load(GOLDEN))


@pytest.fixture(scope="session")
def tiny_models():
    """{config name: (cfg, weights, fixture prefix)} with the seeds make_golden.py used."""
    from karanta_ocr_amd.config import CONFIGS
    from karanta_ocr_amd.weights import random_weights

    out = {}
    for name, seed in (("tiny", 1234), ("tiny-gqa", 4321)):
        cfg = CONFIGS[name]
        out[name] = (cfg, random_weights(cfg, seed), name.replace("-", "_") + "__")
    return out
