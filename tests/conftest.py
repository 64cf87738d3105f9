import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "minimax-speech_amd")
for p in (ROOT, PKG, os.path.join(PKG, "speech"), os.path.join(PKG, "dac-vae")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle runs inside these tests: give torch the cores this job really has (a 1-GPU box hands out ~16 of the
    # machine's cores; torch's default of one thread per machine core oversubscribes them 8x and the composed-path oracle
    # then takes minutes instead of seconds)
    import torch
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(n, 16)))


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
