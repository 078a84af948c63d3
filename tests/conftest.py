import gzip
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
READ_MATCHER_GOLDENS = ["toy_f8_l5_c2", "s300_f30_l12_c3", "ref150_f150_l14_c11", "msa8_f50_c4",
                        "msa_gaps_f40_c5", "p6_f100_c17", "pacbio_f100_l30_c6"]
GENERIC_GOLDENS = ["generic_finite", "generic_infinite"]
ALL_MODEL_GOLDENS = READ_MATCHER_GOLDENS + GENERIC_GOLDENS

_cache = {}


def load_golden(name):
    if name not in _cache:
        with gzip.open(os.path.join(GOLDEN_DIR, name + ".json.gz"), "rb") as f:
            _cache[name] = json.loads(f.read().decode())
    return _cache[name]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
