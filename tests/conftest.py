"""Shared test plumbing: markers, import paths, golden fixtures."""
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / '3d-point-cloud-segmentation-using-2d-img-segmentation_amd'
GOLDEN = ROOT / 'tests' / 'golden'
for p in (str(ROOT), str(PKG)):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run on the GPU box with -m gpu)')
    # a fresh checkout has no built library (it is git-ignored): build it once (hipcc cross-compiles without a GPU)
    if not (PKG / 'f3d' / 'libf3d_hip.so').is_file():
        import subprocess
        subprocess.run(['make', '-C', str(PKG / 'csrc')], check=True, capture_output=True)


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(GOLDEN / f'{name}.npz', allow_pickle=False)
    return load
