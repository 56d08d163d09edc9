"""pytest configuration: `gpu` marker, repo root on sys.path, golden loader."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


# Every device block the library hands out during the tests is pre-filled with NaN bytes
# (PS_POISON, read at the first allocation): the FFT passes skip rows that are known to be zero
# on the writing AND the reading side, and a mismatch between the two would otherwise only show
# when the caching allocator recycles a dirty block.
os.environ.setdefault('PS_POISON', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu)')
    # the shared library is a build artefact (git-ignored): build it if this checkout has none
    # (hipcc cross-compiles gfx950 without a GPU)
    lib = os.path.join(ROOT, 'parasitoids_amd', 'libparasitoid_hip.so')
    if not os.path.exists(lib):
        import subprocess
        subprocess.run(['make', '-j4', '-C', os.path.join(ROOT, 'parasitoids_amd', 'csrc')], check=False)


@pytest.fixture(scope='session')
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = np.load(os.path.join(GOLDEN, name + '.npz'))
        return cache[name]
    return load


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN
