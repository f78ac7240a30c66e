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
MULTIMESH_FULL_XML = os.path.join(SCENES, "Liver-MultiMesh", "mitsuba3", "scene_temp.xml")   # both meshes, both tissue media, envmap: what liver-multimesh.png shows


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


def layer_scene_variant(name):
    """The scene the reference's committed GlissonCapsule / Parenchyma renders were made from, rebuilt from the committed
    scene_temp.xml: GlissonCapsule under a constant white environment (the committed file has the envmap); Parenchyma under the
    envmap block the committed file keeps in a comment, emitters visible.  Returns (xml, base_dir)."""
    import re
    base = os.path.join(SCENES, name, "mitsuba3")
    xml = open(os.path.join(base, "scene_temp.xml")).read()
    if name == "GlissonCapsule":
        xml, n = re.subn(r'<emitter type="envmap">.*?</emitter>', '<emitter type="constant"><rgb name="radiance" value="1.0 1.0 1.0"/></emitter>', xml, flags=re.S)
    else:
        env = ('<emitter type="envmap"><string name="filename" value="cavidade_latitude.exr"/><float name="scale" value="2.5"/><transform name="to_world">'
               '<translate x="-3" y="3" z="4"/><scale value="1.0"/><rotate x="0.57735" y="0.57735" z="0.57735" angle="180"/></transform></emitter>')
        xml, n = re.subn(r'<emitter id="Environment-constant".*?</emitter>', env, xml, flags=re.S)
        xml, m = re.subn(r'<boolean name="hide_emitters" value="true"\s*/>', '<boolean name="hide_emitters" value="false"/>', xml)
        assert m == 1
    assert n == 1
    return xml, base
