import importlib.util
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Import the product package; its directory name has a hyphen, so load it by path."""
    name = "vectordb_from_scratch_amd"
    if name in sys.modules:
        return sys.modules[name]
    pkg_dir = os.path.join(ROOT, "vectordb-from-scratch_amd")
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(pkg_dir, "__init__.py"), submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def vdb():
    return load_package()


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(GOLDEN, "reference_known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden_cases():
    return np.load(os.path.join(GOLDEN, "flat_cases.npz"), allow_pickle=False)
