"""SURVEY.md 8f row 1 on the GPU: the fork's bio transport (`biovolpath` / `biovolpath06` integrators, `liver` /
`parenchyma` / `glissonCapsule` media; docs/BIO_TRANSPORT_SPEC.md) through the C ABI against the CPU oracle, lane by lane
and bit for bit, with equal loop-trip and shadow-ray counts.  The liver scenes load with their files' OWN defaults (no
`integrator=` override)."""
import numpy as np
import pytest

from conftest import LIVER_XML, PARENCHYMA_XML, GLISSON_XML, REALTIME_XML, MULTIMESH_FULL_XML, ROOT, layer_scene_variant
from test_parity_gpu import assert_lanes_equal, center_lane, film_close

pytestmark = pytest.mark.gpu


def test_liver_singlemesh_own_defaults_bit_exact(mi, orc):
    sc = mi.load_file(LIVER_XML, spp=16, res_width=256, res_height=144)          # biovolpath + liver medium, box filter, rgba
    assert sc.desc.integrator.type == 3 and sc.desc.media[0].type == 1
    o = orc.OrcScene(sc)
    g = assert_lanes_equal(sc, o, 0, 256 * 144 * 16)
    assert g[:, 3].min() == 1.0                                                  # envmap visible: every ray is valid
    assert (g[:, :3] > 0).any(axis=1).mean() > 0.5
    inside = g[center_lane(sc, 16):center_lane(sc, 16) + 256 * 16, :3]           # the centre row crosses the liver: one-hot transport
    assert ((inside > 0).sum(axis=1) == 1).mean() > 0.1
    assert_lanes_equal(sc, o, center_lane(sc, 16), 1 << 15, seed=5, max_depth=40, rr_depth=2)
    assert_lanes_equal(sc, o, center_lane(sc, 16, 0.3), 1 << 14, seed=1, hide_emitters=True)


@pytest.mark.parametrize("path", [PARENCHYMA_XML, GLISSON_XML], ids=["parenchyma", "glissoncapsule"])
def test_layer_scenes_own_defaults_bit_exact(mi, orc, path):
    sc = mi.load_file(path, spp=16, res_width=192, res_height=108)               # biovolpath06, ld sampler, tent filter
    assert sc.desc.integrator.type == 4 and sc.desc.sampler_type == 1
    o = orc.OrcScene(sc)
    g = assert_lanes_equal(sc, o, 0, 192 * 108 * 16)
    assert (g[:, :3] > 0).any(axis=1).mean() > 0.05
    assert_lanes_equal(sc, o, center_lane(sc, 16), 1 << 14, seed=3, max_depth=65)
    # the same files through biovolpath (JIT reading: parenchyma's absorbers do not absorb there) and with the independent sampler
    assert_lanes_equal(sc, o, center_lane(sc, 16, 0.4), 1 << 14, integrator="biovolpath")
    img, raw = sc.render(return_raw=True, spp=4)
    assert film_close(raw, o.render(return_raw=True, spp=4)[1]).all()


def test_realtime_scene_own_defaults(mi, orc):
    sc = mi.load_file(REALTIME_XML, res_width=192, res_height=108)              # the file's own integrator, 1 spp
    o = orc.OrcScene(sc)
    assert_lanes_equal(sc, o, 0, 192 * 108 * sc.spp)


def bio_xml(medium, integrator, boundary, sampler="independent", spectral="true", sensor_inside=False, md=12, rr=5, hide="false"):
    """A cube of tissue on a checkerboard floor under an area light and a dim sky: NEE marches through the bio medium (null
    boundary: several steps), smooth and delta surfaces, layers from `tissueDepth` (limits scaled to the cube)."""
    coeffs = "".join(f'<float name="sigma_{k}{l}_{c}" value="{v:.4f}"/>' for k, base in (("collagen", 0.9), ("elastin", 0.5))
                     for l in range(1, 5) for c, v in zip("RGB", (base * l, base * l * 0.6 + 0.1, base * (5 - l) * 0.4)))
    limits = '<float name="layer1Limit" value="0.1"/><float name="layer2Limit" value="0.2"/><float name="layer3Limit" value="0.35"/><float name="layer4Limit" value="0.6"/>'
    par = ('<rgb name="sigma_blood" value="0.3, 0.9, 1.1"/><rgb name="sigma_bile" value="0.02, 0.0, 0.3"/><rgb name="sigma_lipid_water" value="0.05, 0.01, 0.2"/>'
           '<float name="sigma_hepatocity" value="9.5"/>')
    body = {"liver": coeffs + limits + par, "parenchyma": par, "glissonCapsule": coeffs + limits}[medium]
    spec = f'<boolean name="has_spectral_extinction" value="{spectral}"/><rgb name="sigma_t" value="0.4, 0.2, 0.6"/>'
    sens = '<ref id="tissue"/>' if sensor_inside else ""
    origin = "0.2, 0.1, 0.3" if sensor_inside else "3, 2.5, 4"
    bs = {"null": '<bsdf type="null"/>', "dielectric": '<bsdf type="dielectric"><float name="int_ior" value="1.38"/><float name="ext_ior" value="1"/></bsdf>'}[boundary]
    return f"""<scene version="3.0.0">
  <integrator type="{integrator}"><integer name="max_depth" value="{md}"/><integer name="rr_depth" value="{rr}"/><boolean name="hide_emitters" value="{hide}"/></integrator>
  <medium type="{medium}" id="tissue">{body}{spec}<phase type="hg"><float name="g" value="0.4"/></phase></medium>
  <sensor type="perspective"><float name="fov" value="40"/>
    <transform name="to_world"><lookat origin="{origin}" target="0, 0, 0" up="0, 1, 0"/></transform>
    <sampler type="{sampler}"><integer name="sample_count" value="16"/></sampler>
    <film type="hdrfilm"><integer name="width" value="64"/><integer name="height" value="48"/><rfilter type="box"/></film>
    {sens}
  </sensor>
  <shape type="cube">{bs}<ref name="interior" id="tissue"/></shape>
  <shape type="rectangle"><transform name="to_world"><scale value="6"/><rotate x="1" angle="-90"/><translate y="-1.001"/></transform>
    <bsdf type="diffuse"><texture name="reflectance" type="checkerboard"><transform name="to_uv"><scale x="8" y="8"/></transform></texture></bsdf></shape>
  <shape type="rectangle"><transform name="to_world"><scale value="0.7"/><rotate x="1" angle="90"/><translate y="3.5"/></transform>
    <emitter type="area"><rgb name="radiance" value="20, 18, 15"/></emitter></shape>
  <emitter type="constant"><rgb name="radiance" value="0.3, 0.4, 0.6"/></emitter>
</scene>"""


@pytest.mark.parametrize("medium", ["liver", "parenchyma", "glissonCapsule"])
@pytest.mark.parametrize("integrator", ["biovolpath", "biovolpath06"])
def test_bio_cube_bit_exact(mi, orc, medium, integrator):
    for boundary, sampler, spectral, inside, kw in [("null", "independent", "true", False, {}),
                                                    ("dielectric", "ldsampler", "false", False, dict(max_depth=40, rr_depth=2)),     # (ld + unbounded depth never ends: a pixel's
                                                    # 16 sample values may all lie below the Russian-roulette bound 0.95)
                                                    ("dielectric", "independent", "false", False, dict(max_depth=-1, rr_depth=1, seed=2)),
                                                    ("null", "ldsampler", "true", True, dict(seed=7)),
                                                    ("dielectric", "independent", "true", False, dict(hide_emitters=True, max_depth=3))]:
        sc = mi.load_string(bio_xml(medium, integrator, boundary, sampler, spectral, inside))
        o = orc.OrcScene(sc)
        g = assert_lanes_equal(sc, o, 0, 64 * 48 * 16, **kw)
        assert np.isfinite(g).all()
        if integrator == "biovolpath" and boundary == "null" and not inside:
            assert sc.stats()["n_shadow"] > 16 * 48 * 16          # emitter sampling from the floor marches through the cube


def test_bio_params_change_the_render(mi, orc):
    """element coefficients reach the device: another hepatocyte coefficient, other lanes; a bad value is rejected"""
    sc = mi.load_string(bio_xml("parenchyma", "biovolpath", "null"))
    a = sc.render_samples(0, 64 * 48 * 16)
    sc2 = mi.load_string(bio_xml("parenchyma", "biovolpath", "null").replace('"sigma_hepatocity" value="9.5"', '"sigma_hepatocity" value="2.5"'))
    b = sc2.render_samples(0, 64 * 48 * 16)
    assert (a != b).any()
    assert_lanes_equal(sc2, orc.OrcScene(sc2), 0, 64 * 48 * 16)


def test_layer_scene_reference_renders_on_the_device(mi, orc):
    """The reference's own GlissonCapsule renders (tight) and Parenchyma renders (loose) against the HIP render: the device side
    of tests/test_bio_oracle.py's goldens, at 256 spp."""
    from test_bio_oracle import interior_colour, layer_golden, environment_only
    xml, base = layer_scene_variant("GlissonCapsule")
    env = environment_only(mi, orc, xml, base)
    img = mi.load_string(xml, base_dir=base, spp=256, res_width=240, res_height=135, integrator="biovolpath").render().astype(np.float64)[..., :3]
    for dev, tol in (("gpu", 0.004), ("cpu", 0.008)):                            # observed 0.05 % / 0.3 %
        ours, ref, iou, bg = interior_colour(img, layer_golden("GlissonCapsule", dev), env)
        assert iou > 0.99 and bg < 1e-3 and np.allclose(ours, ref, rtol=tol), (dev, ours, ref)
    xml, base = layer_scene_variant("Parenchyma")
    env = environment_only(mi, orc, xml, base)
    img = mi.load_string(xml, base_dir=base, spp=256, res_width=240, res_height=135).render().astype(np.float64)[..., :3]
    ours, ref, iou, bg = interior_colour(img, layer_golden("Parenchyma", "cpu"), env)
    assert iou > 0.99 and bg < 1e-3 and np.allclose(ours, ref, rtol=0.2), (ours, ref)


def test_liver_multimesh_full_scene(mi, orc):
    """Liver-MultiMesh/scene_temp.xml with its own defaults (biovolpath, two nested meshes, glissonCapsule + parenchyma media, ld
    sampler, tent filter): lanes bit for bit against the oracle, and the HIP render against the reference's own render of this
    scene (liver-multimesh.png; interior colour within 1.5 %: the fixture that decides the JIT reading of parenchyma.cpp)."""
    import os
    from test_bio_oracle import interior_colour, environment_only
    sc = mi.load_file(MULTIMESH_FULL_XML, spp=16, res_width=192, res_height=108)
    assert sc.desc.integrator.type == 3 and sc.desc.n_media == 2 and sc.desc.sampler_type == 1
    o = orc.OrcScene(sc)
    g = assert_lanes_equal(sc, o, 0, 192 * 108 * 16)
    assert (g[:, :3] > 0).any(axis=1).mean() > 0.5
    assert_lanes_equal(sc, o, center_lane(sc, 16), 1 << 14, seed=3, max_depth=40, rr_depth=2)
    base = os.path.dirname(MULTIMESH_FULL_XML); xml = open(MULTIMESH_FULL_XML).read()
    golden = np.load(os.path.join(ROOT, "tests", "golden", "reference_liver_multimesh_down8.npy")).astype(np.float64)
    env = environment_only(mi, orc, xml, base)
    img = mi.load_string(xml, base_dir=base, spp=256, res_width=240, res_height=135).render().astype(np.float64)[..., :3]
    ours, ref, iou, bg = interior_colour(img, golden, env)
    assert iou > 0.99 and bg < 1e-3 and np.allclose(ours, ref, rtol=0.015), (ours, ref)


def test_parenchyma_traverse_parameters(mi, orc):
    """mi.traverse on a `parenchyma` medium (src/media/parenchyma.cpp:154-160): the absorbers' coefficients and sigma_hepatocity are
    parameters; an update reaches the device (lanes change and stay bit-identical to the oracle on the updated description)."""
    sc = mi.load_file(PARENCHYMA_XML, spp=16, res_width=96, res_height=54)
    p = mi.traverse(sc)
    mid = sc.desc.media[0].id.decode()
    for k in ("sigma_blood.value", "sigma_bile.value", "sigma_lipid_water.value", "sigma_hepatocity", "scale"):
        assert f"{mid}.{k}" in p
    assert np.allclose(p[f"{mid}.sigma_hepatocity"], sc.desc.media[0].sigma_hepatocity)
    before = sc.render_samples(0, 96 * 54 * 16)
    p[f"{mid}.sigma_hepatocity"] = 12.5
    p[f"{mid}.sigma_blood.value"] = [0.3, 0.05, 0.02]
    p.update()
    assert sc.desc.media[0].sigma_hepatocity == 12.5 and list(sc.desc.media[0].sigma_blood) == pytest.approx([0.3, 0.05, 0.02])
    after = assert_lanes_equal(sc, orc.OrcScene(sc), 0, 96 * 54 * 16)
    assert (before != after).any()
    with pytest.raises(RuntimeError, match="unknown parameter"):
        mi.load_file(LIVER_XML).param_set("LiverMedium.sigma_blood.value", [1, 1, 1])     # `liver` traverses scale, albedo, sigma_t only
