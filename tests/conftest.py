import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "tests", ROOT / "oracle"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def bits_equal(a: np.ndarray, b: np.ndarray) -> bool:
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    if a.shape != b.shape or a.dtype != b.dtype:
        return False
    if a.dtype == np.float32:
        return bool(np.array_equal(a.view(np.uint32), b.view(np.uint32)))
    return bool(np.array_equal(a, b))


def describe_diff(a: np.ndarray, b: np.ndarray) -> str:
    if a.shape != b.shape:
        return f"shape {a.shape} vs {b.shape}"
    if a.dtype == np.float32:
        neq = a.view(np.uint32) != b.view(np.uint32)
        n = int(neq.sum())
        if n == 0:
            return "identical"
        first = int(np.argmax(neq))
        return f"{n}/{a.size} words differ, first at {first}: {a.flat[first]!r} vs {b.flat[first]!r}, max|d|={np.max(np.abs(a - b)):.3e}"
    return f"{int((a != b).sum())}/{a.size} differ"


def rms(x: np.ndarray) -> float:
    x = np.asarray(x, dtype=np.float64)
    return float(np.sqrt(np.mean(x * x))) if x.size else 0.0


@pytest.fixture(scope="session")
def golden():
    return lambda name: np.load(GOLDEN / name, allow_pickle=False)
