import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # Multi-process GPU tests start their ranks from a fork server that is created HERE, before anything in this process has
    # touched the GPU: a process that has initialised HIP must not fork + exec children (the GPU box refuses that), and the
    # server's children are plain forks of a process that never saw the device.
    import multiprocessing as mp
    from multiprocessing import forkserver
    mp.get_context("forkserver")
    forkserver.ensure_running()


@pytest.fixture(scope="session")
def clean_process_context():
    """multiprocessing context whose children descend from the fork server started in pytest_configure."""
    import multiprocessing as mp
    return mp.get_context("forkserver")


@pytest.fixture(scope="session")
def pkg():
    import gmupt_pkg
    return gmupt_pkg.load()


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def device(pkg):
    dev = pkg.capi.Device(0)
    yield dev
    dev.close()


@pytest.fixture(scope="session")
def cornell_scene(pkg):
    return pkg.scenes.build_scene(pkg.scenes.cornell_mesh())


@pytest.fixture(scope="session")
def soup_scene(pkg):
    return pkg.scenes.build_scene(pkg.scenes.random_triangles_mesh(2000, seed=1))


@pytest.fixture(scope="session")
def spheres_small_scene(pkg):
    # 12 icospheres at subdivision 2: 320 tris each, with glass / metal / diffuse materials
    return pkg.scenes.build_scene(pkg.scenes.spheres_mesh(n_spheres=12, subdiv=2, seed=7, floor_quads=4))
