"""Randomised scenes: every combination of the supported plugins a seed happens to draw (BSDFs, media, phase functions,
emitters, filters, samplers, integrators, sensor inside a medium, hide_emitters ...) must give lanes that are bit-identical
to the oracle, with equal loop-trip and shadow-ray counts.  This is the net for rare branches (it is the kind of test that
found the gfx950 code-generation bug documented in DESIGN.md)."""
import os

import numpy as np
import pytest

from conftest import ROOT
from test_parity_gpu import assert_lanes_equal, film_close

pytestmark = pytest.mark.gpu
ASSETS = os.path.join(ROOT, "scenes", "assets")


def random_scene_xml(seed):
    r = np.random.default_rng(seed)
    pick = lambda *a: a[int(r.integers(len(a)))]
    f3 = lambda lo, hi: ", ".join(f"{x:.4g}" for x in r.uniform(lo, hi, 3))
    integrator = pick("path", "volpath", "volpath", "prbvolpath")
    use_media = integrator != "path"
    media, shapes = [], []
    n_media = int(r.integers(1, 3)) if use_media else 0
    for m in range(n_media):
        phase = pick('<phase type="isotropic"/>', f'<phase type="hg"><float name="g" value="{r.uniform(-0.8, 0.8):.3f}"/></phase>')
        media.append(f'<medium type="homogeneous" id="m{m}"><rgb name="sigma_t" value="{f3(0.2, 2.5)}"/><rgb name="albedo" value="{f3(0.3, 1.0)}"/>'
                     f'<float name="scale" value="{r.uniform(0.5, 2):.3f}"/><boolean name="has_spectral_extinction" value="{pick("true", "false")}"/>'
                     f'<boolean name="sample_emitters" value="{pick("true", "true", "false")}"/>{phase}</medium>')
    sensor_medium = ""
    haze = use_media and integrator == "volpath" and r.random() < 0.3       # prbvolpath: "TODO: support sensors inside media"
    if haze:
        media.append('<medium type="homogeneous" id="haze"><float name="sigma_t" value="0.04"/><float name="albedo" value="0.7"/></medium>')
        sensor_medium = '<ref id="haze"/>'
    ext = '<ref name="exterior" id="haze"/>' if haze else ""
    def bsdf(allow_null):
        kinds = ["diffuse", "diffuse_tex", "dielectric", "bump"] + (["null"] if allow_null else [])
        k = pick(*kinds)
        if k == "diffuse": return f'<bsdf type="diffuse"><rgb name="reflectance" value="{f3(0.1, 0.9)}"/></bsdf>'
        if k == "diffuse_tex": return ('<bsdf type="diffuse"><texture name="reflectance" type="checkerboard"><transform name="to_uv">'
                                        f'<scale x="{r.uniform(2, 9):.2f}" y="{r.uniform(2, 9):.2f}"/></transform></texture></bsdf>')
        if k == "dielectric": return f'<bsdf type="dielectric"><float name="int_ior" value="{r.uniform(1.1, 1.7):.3f}"/><float name="ext_ior" value="1"/></bsdf>'
        if k == "bump":
            nested = pick('<bsdf type="diffuse"/>', '<bsdf type="dielectric"/>')
            return (f'<bsdf type="bumpmap"><float name="scale" value="{r.uniform(0.002, 0.02):.4f}"/><texture name="texture" type="bitmap">'
                    f'<string name="filename" value="{ASSETS}/tissue_n.png"/></texture>{nested}</bsdf>')
        return '<bsdf type="null"/>'
    for k in range(int(r.integers(1, 4))):
        tr = (f'<transform name="to_world"><scale value="{r.uniform(0.4, 1.1):.3f}"/><rotate x="{r.random():.3f}" y="{r.random():.3f}" z="{r.random() + 0.1:.3f}" angle="{r.uniform(0, 90):.1f}"/>'
              f'<translate x="{r.uniform(-1.6, 1.6):.3f}" y="{r.uniform(-0.4, 1.0):.3f}" z="{r.uniform(-1.6, 1.6):.3f}"/></transform>')
        inside = f'<ref name="interior" id="m{int(r.integers(n_media))}"/>' if n_media and r.random() < 0.8 else ""
        mesh = pick("cube", "cube", "obj")
        geom = '<shape type="cube">' if mesh == "cube" else f'<shape type="obj"><string name="filename" value="{ASSETS}/liver1.obj"/>'
        if mesh == "obj": tr = tr.replace("<scale value=", '<translate x="38" y="23" z="38"/><scale value="0.05"/><scale value=')   # liver1 centred, ~1.5 units
        shapes.append(f'{geom}{tr}{bsdf(bool(inside))}{inside}{ext}</shape>')
    shapes.append(f'<shape type="rectangle"><transform name="to_world"><scale value="6"/><rotate x="1" angle="-90"/><translate y="-1.2"/></transform>{bsdf(False)}{ext}</shape>')
    emitters = []
    if r.random() < 0.6:
        emitters.append(f'<shape type="rectangle"><transform name="to_world"><scale value="{r.uniform(0.3, 1.2):.3f}"/><rotate x="1" angle="90"/>'
                        f'<translate x="{r.uniform(-1, 1):.3f}" y="{r.uniform(2.5, 4):.3f}" z="{r.uniform(-1, 1):.3f}"/></transform>'
                        f'<emitter type="area"><rgb name="radiance" value="{f3(5, 25)}"/></emitter>{ext}</shape>')
    env = pick("none", "constant", "envmap") if emitters else pick("constant", "envmap")
    if env == "constant": emitters.append(f'<emitter type="constant"><rgb name="radiance" value="{f3(0.2, 1.2)}"/></emitter>')
    if env == "envmap": emitters.append(f'<emitter type="envmap"><string name="filename" value="{ASSETS}/cavidade_latitude.exr"/><float name="scale" value="{r.uniform(0.5, 3):.3f}"/>'
                                        f'<transform name="to_world"><rotate y="1" angle="{r.uniform(0, 360):.1f}"/></transform></emitter>')
    rf = pick("box", "gaussian", "tent")
    sampler = pick("independent", "independent", "ldsampler")
    alpha = pick("rgb", "rgba")
    spp = pick(16, 16, 16, 16, 12, 7, 3, 1) if sampler != "ldsampler" else 16     # non-power-of-two counts take the division path
    fw, fh = (48, 40) if r.random() < 0.6 else (int(r.integers(17, 70)), int(r.integers(9, 50)))
    crop = ""
    if r.random() < 0.25:
        cw, ch = int(r.integers(1, fw + 1)), int(r.integers(1, fh + 1))
        cx, cy = int(r.integers(0, fw - cw + 1)), int(r.integers(0, fh - ch + 1))
        crop = (f'<integer name="crop_offset_x" value="{cx}"/><integer name="crop_offset_y" value="{cy}"/>'
                f'<integer name="crop_width" value="{cw}"/><integer name="crop_height" value="{ch}"/>')
    spass = pick(0, 0, 4, 8) if (integrator != "prbvolpath" and spp == 16) else 0   # multi-pass renders (samples_per_pass)
    spass_xml = f'<integer name="samples_per_pass" value="{spass}"/>' if spass else ""
    xml = f"""<scene version="3.0.0">
  <integrator type="{integrator}"><integer name="max_depth" value="{pick(-1, 3, 6, 12)}"/><integer name="rr_depth" value="{pick(1, 3, 5)}"/>
    <boolean name="hide_emitters" value="{pick("false", "false", "true")}"/>{spass_xml}</integrator>
  {''.join(media)}
  <sensor type="perspective"><float name="fov" value="{r.uniform(30, 60):.2f}"/>
    <transform name="to_world"><lookat origin="{r.uniform(2.5, 4):.3f}, {r.uniform(1, 3):.3f}, {r.uniform(2.5, 4.5):.3f}" target="0, 0, 0" up="0, 1, 0"/></transform>
    <sampler type="{sampler}"><integer name="sample_count" value="{spp}"/><integer name="seed" value="{int(r.integers(0, 5))}"/></sampler>
    <film type="hdrfilm"><integer name="width" value="{fw}"/><integer name="height" value="{fh}"/>{crop}<string name="pixel_format" value="{alpha}"/><rfilter type="{rf}"/></film>
    {sensor_medium}
  </sensor>
  {''.join(shapes)}
  {''.join(emitters)}
</scene>"""
    return xml, integrator


def random_scene_xml_r2(seed, tmpdir):
    """Round-2 families: `biovolpath` / `biovolpath06` on liver / parenchyma / glissonCapsule media,
    `volpath` / `volpathmis` on heterogeneous (grid-volume) and homogeneous media, with the same random shapes, BSDFs, emitters,
    filters, samplers, crops and passes as above.  Grid volumes are written to `tmpdir`."""
    import liverrenderer_amd as mi
    r = np.random.default_rng(10_000 + seed)
    pick = lambda *a: a[int(r.integers(len(a)))]
    f3 = lambda lo, hi: ", ".join(f"{x:.4g}" for x in r.uniform(lo, hi, 3))
    integrator = pick("biovolpath", "biovolpath06", "volpath", "volpathmis", "volpathmis")
    bio = integrator.startswith("bio")
    smis = pick("true", "false")
    n_media = int(r.integers(1, 3))
    media = []
    for m in range(n_media):
        phase = pick('<phase type="isotropic"/>', f'<phase type="hg"><float name="g" value="{r.uniform(-0.8, 0.8):.3f}"/></phase>')
        common = (f'<rgb name="sigma_t" value="{f3(0.2, 2.0)}"/><boolean name="has_spectral_extinction" value="{pick("true", "false")}"/>'
                  f'<boolean name="sample_emitters" value="{pick("true", "true", "false")}"/><float name="scale" value="{r.uniform(0.5, 2):.3f}"/>{phase}')
        if bio:
            kind = pick("liver", "parenchyma", "glissonCapsule")
            lim = np.sort(r.uniform(0.03, 0.7, 4))
            coeffs = "".join(f'<float name="sigma_{k}{l}_{c}" value="{r.uniform(0.05, 3.0):.4f}"/>' for k in ("collagen", "elastin") for l in range(1, 5) for c in "RGB")
            limits = "".join(f'<float name="layer{i + 1}Limit" value="{lim[i]:.4f}"/>' for i in range(4))
            par = (f'<rgb name="sigma_blood" value="{f3(0.01, 1.2)}"/><rgb name="sigma_bile" value="{f3(0.0, 0.4)}"/><rgb name="sigma_lipid_water" value="{f3(0.0, 0.3)}"/>'
                   f'<float name="sigma_hepatocity" value="{r.uniform(0.5, 300):.3f}"/>')
            body = {"liver": coeffs + limits + par, "parenchyma": par, "glissonCapsule": coeffs + limits}[kind]
            media.append(f'<medium type="{kind}" id="m{m}">{body}{common}</medium>')
        elif r.random() < 0.6:
            res = tuple(int(x) for x in r.integers(1, 9, 3))
            grid = (r.random(res) ** pick(1, 3)).astype(np.float32) * np.float32(r.uniform(0.3, 2.0))
            if r.random() < 0.3: grid[r.random(res) < 0.4] = 0                      # empty voxels: null collisions all the way
            if grid.max() == 0: grid.flat[0] = 0.5                                  # (a grid whose maximum is zero is rejected at scene creation: majorant 0)
            vol = os.path.join(str(tmpdir), f"fuzz_{seed}_{m}.vol"); mi.write_volume_grid(vol, grid)
            media.append(f'<medium type="heterogeneous" id="m{m}"><volume name="sigma_t" type="gridvolume"><string name="filename" value="{vol}"/>'
                         f'<transform name="to_world"><scale value="{r.uniform(4, 7):.3f}"/><translate x="-3" y="-2.5" z="-3"/></transform></volume>'
                         f'<rgb name="albedo" value="{f3(0.3, 1.0)}"/><float name="scale" value="{r.uniform(0.5, 4):.3f}"/>'
                         f'<boolean name="has_spectral_extinction" value="{pick("true", "false")}"/><boolean name="sample_emitters" value="{pick("true", "true", "false")}"/>{phase}</medium>')
        else:
            media.append(f'<medium type="homogeneous" id="m{m}"><rgb name="albedo" value="{f3(0.3, 1.0)}"/>{common}</medium>')
    sensor_medium, ext = "", ""
    if r.random() < 0.25:                                                             # the sensor sits inside medium 0
        sensor_medium = '<ref id="m0"/>'; ext = '<ref name="exterior" id="m0"/>'
    def bsdf(allow_null):
        kinds = ["diffuse", "diffuse_tex", "dielectric", "bump"] + (["null", "null"] if allow_null else [])
        k = pick(*kinds)
        if k == "diffuse": return f'<bsdf type="diffuse"><rgb name="reflectance" value="{f3(0.1, 0.9)}"/></bsdf>'
        if k == "diffuse_tex": return ('<bsdf type="diffuse"><texture name="reflectance" type="checkerboard"><transform name="to_uv">'
                                        f'<scale x="{r.uniform(2, 9):.2f}" y="{r.uniform(2, 9):.2f}"/></transform></texture></bsdf>')
        if k == "dielectric": return f'<bsdf type="dielectric"><float name="int_ior" value="{r.uniform(1.1, 1.7):.3f}"/><float name="ext_ior" value="1"/></bsdf>'
        if k == "bump":
            return (f'<bsdf type="bumpmap"><float name="scale" value="{r.uniform(0.002, 0.02):.4f}"/><texture name="texture" type="bitmap">'
                    f'<string name="filename" value="{ASSETS}/tissue_n.png"/></texture><bsdf type="dielectric"/></bsdf>')
        return '<bsdf type="null"/>'
    shapes = []
    for k in range(int(r.integers(1, 4))):
        tr = (f'<transform name="to_world"><scale value="{r.uniform(0.4, 1.1):.3f}"/><rotate x="{r.random():.3f}" y="{r.random():.3f}" z="{r.random() + 0.1:.3f}" angle="{r.uniform(0, 90):.1f}"/>'
              f'<translate x="{r.uniform(-1.6, 1.6):.3f}" y="{r.uniform(-0.4, 1.0):.3f}" z="{r.uniform(-1.6, 1.6):.3f}"/></transform>')
        inside = f'<ref name="interior" id="m{int(r.integers(n_media))}"/>' if r.random() < 0.9 else ""
        mesh = pick("cube", "cube", "obj")
        geom = '<shape type="cube">' if mesh == "cube" else f'<shape type="obj"><string name="filename" value="{ASSETS}/liver1.obj"/>'
        if mesh == "obj": tr = tr.replace("<scale value=", '<translate x="38" y="23" z="38"/><scale value="0.05"/><scale value=')
        shapes.append(f'{geom}{tr}{bsdf(bool(inside))}{inside}{ext}</shape>')
    shapes.append(f'<shape type="rectangle"><transform name="to_world"><scale value="6"/><rotate x="1" angle="-90"/><translate y="-1.2"/></transform>{bsdf(False)}{ext}</shape>')
    emitters = []
    if r.random() < 0.6:
        emitters.append(f'<shape type="rectangle"><transform name="to_world"><scale value="{r.uniform(0.3, 1.2):.3f}"/><rotate x="1" angle="90"/>'
                        f'<translate x="{r.uniform(-1, 1):.3f}" y="{r.uniform(2.5, 4):.3f}" z="{r.uniform(-1, 1):.3f}"/></transform>'
                        f'<emitter type="area"><rgb name="radiance" value="{f3(5, 25)}"/></emitter>{ext}</shape>')
    env = pick("none", "constant", "envmap") if emitters else pick("constant", "envmap")
    if env == "constant": emitters.append(f'<emitter type="constant"><rgb name="radiance" value="{f3(0.2, 1.2)}"/></emitter>')
    if env == "envmap": emitters.append(f'<emitter type="envmap"><string name="filename" value="{ASSETS}/cavidade_latitude.exr"/><float name="scale" value="{r.uniform(0.5, 3):.3f}"/>'
                                        f'<transform name="to_world"><rotate y="1" angle="{r.uniform(0, 360):.1f}"/></transform></emitter>')
    rf = pick("box", "gaussian", "tent")
    sampler = pick("independent", "independent", "ldsampler")
    spp = pick(16, 16, 16, 12, 7, 3, 1) if sampler != "ldsampler" else 16
    fw, fh = (48, 40) if r.random() < 0.6 else (int(r.integers(17, 70)), int(r.integers(9, 50)))
    spass = pick(0, 0, 4, 8) if spp == 16 else 0
    spass_xml = f'<integer name="samples_per_pass" value="{spass}"/>' if spass else ""
    # (ld sampler + unbounded depth: a pixel's 16 sample values may all lie below the Russian-roulette bound, the loop never ends)
    max_depth = pick(3, 6, 12, 30) if sampler == "ldsampler" else pick(-1, 3, 6, 12)
    smis_xml = f'<boolean name="use_spectral_mis" value="{smis}"/>' if integrator == "volpathmis" else ""
    xml = f"""<scene version="3.0.0">
  <integrator type="{integrator}"><integer name="max_depth" value="{max_depth}"/><integer name="rr_depth" value="{pick(1, 3, 5)}"/>
    <boolean name="hide_emitters" value="{pick("false", "false", "true")}"/>{spass_xml}{smis_xml}</integrator>
  {''.join(media)}
  <sensor type="perspective"><float name="fov" value="{r.uniform(30, 60):.2f}"/>
    <transform name="to_world"><lookat origin="{r.uniform(2.5, 4):.3f}, {r.uniform(1, 3):.3f}, {r.uniform(2.5, 4.5):.3f}" target="0, 0, 0" up="0, 1, 0"/></transform>
    <sampler type="{sampler}"><integer name="sample_count" value="{spp}"/><integer name="seed" value="{int(r.integers(0, 5))}"/></sampler>
    <film type="hdrfilm"><integer name="width" value="{fw}"/><integer name="height" value="{fh}"/><string name="pixel_format" value="{pick("rgb", "rgba")}"/><rfilter type="{rf}"/></film>
    {sensor_medium}
  </sensor>
  {''.join(shapes)}
  {''.join(emitters)}
</scene>"""
    return xml, integrator


# 20540: a multi-pass tent-filter biovolpath render whose lanes overflow (non-finite film values: same pattern on both sides)
@pytest.mark.parametrize("seed", list(range(24)) + [20540])
def test_random_scene_r2_bit_exact(mi, orc, tmp_path, seed):
    xml, integrator = random_scene_xml_r2(seed, tmp_path)
    sc = mi.load_string(xml)
    o = orc.OrcScene(sc)
    h, w, _ = sc.film_shape()
    per_pass = min(sc.spp, sc.desc.samples_per_pass or sc.spp)
    assert_lanes_equal(sc, o, 0, w * h * per_pass, seed=seed)
    if seed % 4 == 0 or sc.desc.samples_per_pass:
        raw = sc.render(return_raw=True, seed=seed)[1]
        assert film_close(raw, o.render(return_raw=True, seed=seed)[1]).all()


# 20059, 21481: adjoint under a crop window (the oracle's ray set-up once differed from the forward pass in the last bit there);
# 20756: a multi-pass tent-filter render with non-finite lanes (their zero-weight products must stay inside their own footprint);
# 120770: max_depth -1, ld sampler with 16 samples, albedo 0.99: a path leaves a leaky mesh with its medium flag set and walks an infinite medium; its
#         pixel's sixteen 1-D sample values all lie below 0.95, so Russian roulette never stops it: ends at depth 65535 (device.hip, resolve) on both sides
@pytest.mark.parametrize("seed", list(range(24)) + [20059, 21481, 20756, 120770])
def test_random_scene_bit_exact(mi, orc, seed):
    xml, integrator = random_scene_xml(seed)
    sc = mi.load_string(xml)
    o = orc.OrcScene(sc)
    h, w, _ = sc.film_shape()
    per_pass = min(sc.spp, sc.desc.samples_per_pass or sc.spp)
    assert_lanes_equal(sc, o, 0, w * h * per_pass, seed=seed)                # the per-lane hook addresses the lanes of one pass
    if integrator == "prbvolpath":                              # the adjoint too: gradients equal up to summation order
        h, w, c = sc.film_shape()
        grad = np.random.default_rng(seed).random((h, w, c)).astype(np.float32) / (h * w * c)
        gg, gc = sc.render_backward(grad, seed=seed), o.render_backward(grad, seed=seed)
        for k in ("sigma_t", "albedo"):
            assert np.abs(gg[k] - gc[k]).max() <= 3e-4 * max(np.abs(gc[k]).max(), 1e-7), (k, gg[k], gc[k])
        assert abs(gg["g"] - gc["g"]) <= 3e-4 * max(abs(gc["g"]), 1e-6) + 1e-9
    if (seed % 4 == 0 or sc.desc.samples_per_pass) and integrator != "prbvolpath":   # the film path too (all filters, all passes)
        raw = sc.render(return_raw=True, seed=seed)[1]
        assert film_close(raw, o.render(return_raw=True, seed=seed)[1]).all()


def random_scene_prb_het(seed, tmpdir):
    """Round-3 family: the PRB adjoint on the round-2 volume scenes (heterogeneous and homogeneous media mixed, null boundaries, every
    emitter kind, both samplers): the first `volpath` scene of the round-2 generator from `seed` on, with the integrator swapped."""
    k = seed
    while True:
        xml, integrator = random_scene_xml_r2(k, tmpdir)
        if integrator == "volpath": break
        k += 100_000
    return xml.replace('<integrator type="volpath">', '<integrator type="prbvolpath">')


@pytest.mark.parametrize("seed", range(16))
def test_random_scene_prb_het(mi, orc, tmp_path, seed):
    sc = mi.load_string(random_scene_prb_het(seed, tmp_path)); o = orc.OrcScene(sc)
    h, w, c = sc.film_shape()
    assert_lanes_equal(sc, o, 0, w * h * sc.spp, seed=seed)
    grad = np.random.default_rng(seed).random((h, w, c)).astype(np.float32) / (h * w * c)
    gg, gc = sc.render_backward(grad, seed=seed), o.render_backward(grad, seed=seed)
    for k in ("sigma_t", "albedo"):
        assert np.abs(gg[k] - gc[k]).max() <= 3e-4 * max(np.abs(gc[k]).max(), 1e-7), (k, gg[k], gc[k])
    assert abs(gg["g"] - gc["g"]) <= 3e-4 * max(abs(gc["g"]), 1e-6) + 1e-9
