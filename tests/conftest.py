"""pytest config: registers the ``gpu`` marker and shared fixture helpers."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # libmaskbit_hip.so is a build artefact (git-ignored): compile it if this checkout has none yet (hipcc cross-compiles
    # gfx950 without a GPU).  If that is impossible the tests that need it fail loudly -- there is no fallback.
    try:
        from maskbit_amd import build as mb_build
        if mb_build.needs_build():
            mb_build.build()
    except Exception as e:  # noqa: BLE001
        print(f"[conftest] could not build libmaskbit_hip.so: {e}", file=sys.stderr)


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def golden_weights(z):
    return {k[2:]: torch.from_numpy(v) for k, v in z.items() if k.startswith("w.")}


@pytest.fixture(scope="session")
def golden():
    return load_golden
