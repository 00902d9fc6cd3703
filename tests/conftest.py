import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def compiled_model():
    from mujoco_robot_environments_amd.model import compile as MC
    A = MC.compile_scene()
    return A, MC.to_blob(A)


@pytest.fixture(scope="session")
def oracle_model(compiled_model):
    from oracle import oracle as O
    return O.Model(compiled_model[1])
