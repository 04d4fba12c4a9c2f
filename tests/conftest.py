import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    config.addinivalue_line("markers", "slow: long-running CPU test")


def load_case(name):
    js = json.loads((GOLDEN / f"{name}.json").read_text())
    npz = None
    p = GOLDEN / f"{name}.npz"
    if p.exists():
        npz = np.load(p)
    return js, npz


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
