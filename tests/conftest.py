import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """No test may sit for ever (a loop that never ends costs GPU-box minutes by the thousand): ten minutes each, where the
    pytest-timeout plugin is installed."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session", autouse=True)
def _native_libraries():
    """Build what is missing (the GPU box receives prebuilt .so files; the build container builds)."""
    import beast_mcmc_amd as bm
    build = __import__("importlib").import_module("beast-mcmc_amd.build")
    if not (os.path.exists(bm.beagle.ENGINE_LIB) and os.path.exists(bm.beagle.HOST_LIB)):
        build.build_all()
    oracle = os.path.join(ROOT, "oracle", "liboracle_beagle.so")
    if not os.path.exists(oracle):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    yield


@pytest.fixture(scope="session")
def oracle_lib():
    import helpers
    return helpers.oracle_library()


@pytest.fixture(scope="session")
def engine_lib():
    import beast_mcmc_amd as bm
    return bm.beagle.engine()
