import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's fp32 matmuls on the CPUs this process is really allotted (a one-GPU box: 16 of the host's 256; torch's default pool of
    # 128 threads ran them 3 x slower)
    from oracle import use_allotted_cpu_threads
    use_allotted_cpu_threads()


@pytest.fixture(scope="session")
def bssd_golden():
    with open(os.path.join(GOLDEN_DIR, "bssd_golden.json")) as f:
        return {c["name"]: c for c in json.load(f)}


@pytest.fixture(scope="session")
def trie_golden():
    with open(os.path.join(GOLDEN_DIR, "trie_golden.json")) as f:
        return {c["name"]: c for c in json.load(f)}
