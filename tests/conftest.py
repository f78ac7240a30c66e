import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENES = os.path.join(ROOT, "scenes")
LIVER_XML = os.path.join(SCENES, "Liver-SingleMesh", "mitsuba3", "scene.xml")
PARENCHYMA_XML = os.path.join(SCENES, "Parenchyma", "mitsuba3", "scene_temp.xml")      # scene.xml is LiverRenderer.py's template ("360:0.2464" placeholders)
GLISSON_XML = os.path.join(SCENES, "GlissonCapsule", "mitsuba3", "scene_temp.xml")
REALTIME_XML = os.path.join(SCENES, "Liver-SingleMesh-Realtime", "mitsuba3", "scene.xml")
MULTIMESH_XML = os.path.join(SCENES, "Liver-MultiMesh", "mitsuba3", "scene.xml")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def mi():
    import liverrenderer_amd
    return liverrenderer_amd


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.lib()
    return _orc


@pytest.fixture(scope="session")
def cornell(mi):
    return mi.load_dict(mi.cornell_box())


@pytest.fixture(scope="session")
def liver_small(mi):
    return mi.load_file(LIVER_XML, integrator="volpath", spp=16, res_width=128, res_height=72)
