"""pytest configuration: registers the `gpu` marker and puts the host-side
package directory (huffman-gpu_amd/, not an importable name) on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "huffman-gpu_amd"), os.path.join(ROOT, "tests"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import json
    import numpy as np
    g = os.path.join(ROOT, "tests", "golden")
    with open(os.path.join(g, "golden.json")) as f:
        meta = json.load(f)
    period = np.fromfile(os.path.join(g, "g1_period.bin"), dtype=np.uint8)
    table = np.fromfile(os.path.join(g, "g1_table.bin"), dtype=np.uint32)
    counts = np.fromfile(os.path.join(g, "g1_seg_counts.bin"), dtype=np.uint32)
    vec = np.load(os.path.join(g, "diff_vectors.npz"))
    return {"meta": meta, "period": period, "cw": table[:256].copy(), "cl": table[256:].copy(),
            "seg_counts": counts, "vec": vec}
