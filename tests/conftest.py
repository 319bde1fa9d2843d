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


GOLDEN_25 = os.path.join(ROOT, "tests", "golden", "qwen2_5vl_tiny_golden.npz")   # Qwen2.5-VL variant (make_golden_qwen2_5.py)


@pytest.fixture(scope="session")
def golden():
    out = dict(np.load(GOLDEN))
    out.update(np.load(GOLDEN_25))
    return out


@pytest.fixture(scope="session")
def tiny_models():
    """{config name: (cfg, weights, fixture prefix)} with the seeds make_golden.py used."""
    from karanta_ocr_amd.config import CONFIGS
    from karanta_ocr_amd.weights import random_weights

    out = {}
    for name, seed in (("tiny", 1234), ("tiny-gqa", 4321), ("tiny-2.5", 2525)):
        cfg = CONFIGS[name]
        out[name] = (cfg, random_weights(cfg, seed), name.replace("-", "_").replace(".", "_") + "__")
    return out
