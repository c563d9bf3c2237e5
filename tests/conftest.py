import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def scene_dir(tmp_path_factory):
    return str(tmp_path_factory.mktemp("scenes"))


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure); built on demand with g++."""
    from oracle import oracle
    oracle.build()
    return oracle


def make_scene(scene_dir, name, **kw):
    from raytracing_amd import scenes
    tag = name + "".join("_%s%s" % (k, v) for k, v in sorted(kw.items()))
    path = os.path.join(scene_dir, tag + ".gltf")
    if not os.path.exists(path):
        scenes.generate(name, path, **kw)
    return path
