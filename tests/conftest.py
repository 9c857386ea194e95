import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def synth_sd():
    """Synthetic state_dict (seed 0) in the reference's key layout, built from the committed key inventory."""
    from speinet_amd.synth import state_dict_template, synth_state_dict
    return synth_state_dict(state_dict_template(), seed=0)
