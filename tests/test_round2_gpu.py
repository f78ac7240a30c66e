"""Round-2 additions to the GPU parity suite: the JIT reading of `path`'s last trip, per-medium PRB gradients, wide
reconstruction filters, bitmap reflectance rejection, the pinned fog render of the reference."""
import os

import numpy as np
import pytest

from conftest import ROOT
from test_parity_gpu import assert_lanes_equal, film_close, fog_xml

pytestmark = pytest.mark.gpu


def test_path_last_trip_jit_reading_multi_pass(mi, orc):
    """path.cpp:227-231: a JIT lane never takes the `dr::none_or<false>(active_next)` exit: six more sampler values are drawn
    (they shift the PCG32 state the next pass starts from) and valid_ray |= si.is_valid() && !Null (alpha at max_depth = 1
    on ordinary surfaces)."""
    d = mi.cornell_box(); d['sensor']['film'].update({'width': 64, 'height': 64, 'pixel_format': 'rgba', 'rfilter': {'type': 'box'}})
    d['sensor']['sampler'] = {'type': 'independent', 'sample_count': 16}
    d['integrator'].update({'samples_per_pass': 4, 'max_depth': 1})
    sc = mi.load_dict(d); o = orc.OrcScene(sc)
    g = assert_lanes_equal(sc, o, 0, 64 * 64 * 4)
    assert g[:, 3].mean() > 0.9                                        # every camera ray meets a wall: valid, whatever it hit
    img, raw = sc.render(return_raw=True, seed=3)
    oimg, oraw = o.render(return_raw=True, seed=3)
    assert film_close(raw, oraw).all() and np.allclose(img, oimg, rtol=2e-4, atol=2e-5)
    for md in (2, 3):                                                  # deeper paths end on the same exit
        assert_lanes_equal(sc, o, 0, 64 * 64 * 4, max_depth=md, seed=md)
        raw = sc.render(return_raw=True, seed=1, max_depth=md)[1]
        assert film_close(raw, o.render(return_raw=True, seed=1, max_depth=md)[1]).all()


def test_backward_per_medium(mi, orc):
    """lrt_render_backward differentiates ONE medium's parameters (opts->grad_medium); -1 sums all media into one set."""
    xml = fog_xml(md="8", rf="box", exterior='<ref name="exterior" id="haze"/>').replace('type="volpath"', 'type="prbvolpath"')
    xml = xml.replace('<float name="sigma_t" value="0.05"/>', '<float name="sigma_t" value="0.4"/>')
    sc = mi.load_string(xml); o = orc.OrcScene(sc)
    assert sc.desc.n_media == 2
    h, w, c = sc.film_shape()
    grad = np.random.default_rng(5).random((h, w, c)).astype(np.float32) / (h * w * c)
    res = {}
    for m in (0, 1, -1):
        gg, gc = sc.render_backward(grad, seed=4, medium=m), o.render_backward(grad, seed=4, medium=m)
        for k in ("sigma_t", "albedo"):
            assert np.abs(gg[k] - gc[k]).max() <= 3e-4 * max(np.abs(gc[k]).max(), 1e-7), (m, k, gg[k], gc[k])
        assert abs(gg["g"] - gc["g"]) <= 3e-4 * max(abs(gc["g"]), 1e-6) + 1e-9
        res[m] = gg
    assert np.abs(res[0]["sigma_t"]).max() > 0 and np.abs(res[1]["sigma_t"]).max() > 0 and res[1]["g"] == 0.0   # haze is isotropic
    for k in ("sigma_t", "albedo"):
        assert np.allclose(res[0][k] + res[1][k], res[-1][k], rtol=2e-3, atol=1e-7)
    dflt = sc.render_backward(grad, seed=4)                            # the Python default is the sum over all media
    for k in ("sigma_t", "albedo"):
        assert np.allclose(dflt[k], res[-1][k], rtol=2e-3, atol=1e-7)
    with pytest.raises(RuntimeError, match="grad_medium"):
        sc.render_backward(grad, medium=2)


@pytest.mark.parametrize("rf", ['<rfilter type="gaussian"><float name="stddev" value="1.0"/></rfilter>',
                                '<rfilter type="tent"><float name="radius" value="4"/></rfilter>',
                                '<rfilter type="gaussian"><float name="stddev" value="1.6"/></rfilter>'])
def test_wide_filter_footprints(mi, orc, rf):
    """Footprints of more than 64 pixels (9 x 9 and 13 x 13): k_splat_lanes walks them in chunks of 64 cells."""
    xml = fog_xml(md="6", rf="box").replace('<rfilter type="box"/>', rf)
    sc = mi.load_string(xml); o = orc.OrcScene(sc)
    assert sc.desc.film.rfilter_param in (1.0, 4.0, pytest.approx(1.6))
    img, raw = sc.render(return_raw=True, seed=1, spp=8)
    oimg, oraw = o.render(return_raw=True, seed=1, spp=8)
    assert film_close(raw, oraw).all()
    assert np.allclose(img, oimg, rtol=3e-4, atol=2e-5)
    # the PRB adjoint normalises by the same (weights-only) film
    pxml = xml.replace('type="volpath"', 'type="prbvolpath"')
    ps = mi.load_string(pxml); po = orc.OrcScene(ps)
    h, w, c = ps.film_shape()
    grad = np.random.default_rng(2).random((h, w, c)).astype(np.float32) / (h * w * c)
    gg, gc = ps.render_backward(grad, seed=2, spp=4), po.render_backward(grad, seed=2, spp=4)
    for k in ("sigma_t", "albedo"):
        assert np.abs(gg[k] - gc[k]).max() <= 3e-4 * max(np.abs(gc[k]).max(), 1e-7), (k, gg[k], gc[k])


def test_cornell_fog_reference_render(mi):
    """the reference's own volpath fog render pins the HIP path too.  Same pipeline as the fixture: render at
    1080x1080 (64 spp), box-average 8x8 in linear.  Interior mean within 5 %, fog-only border within 4 %, shape correlation."""
    from test_oracle_pins import cornell_fog_scene
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_cornell_box_fog_1080_down8.npy")).astype(np.float64)
    sc = cornell_fog_scene(mi, 1080, 64)
    img = sc.render(seed=0).astype(np.float64)[..., :3]
    c = np.clip(img, 0, 1).reshape(135, 8, 135, 8, 3).mean((1, 3))
    ok = (g < 0.9).all(-1) & (c < 0.9).all(-1)
    assert np.allclose(c[ok].mean(0), g[ok].mean(0), rtol=0.05), (c[ok].mean(0) / g[ok].mean(0))
    border = np.zeros((135, 135), bool); border[:, :3] = True; border[:, -3:] = True
    assert np.allclose(c[border].mean(0), g[border].mean(0), rtol=0.04), (c[border].mean(0) / g[border].mean(0))
    assert np.corrcoef(c[ok].ravel(), g[ok].ravel())[0, 1] > 0.99


def test_liver_singlemesh_bio_reference_render(mi):
    """The reference's committed scalar_rgb render of Liver-SingleMesh (biovolpath + liver medium, 128 spp) against the HIP render
    of scene.xml with the file's own integrator and medium at the same resolution and sample count: liver-interior mean colour
    (all in-tissue transport) within 1.5 % per channel, per-pixel agreement inside the liver."""
    import re
    from scipy.ndimage import binary_erosion
    from conftest import LIVER_XML
    g = np.load(os.path.join(ROOT, "tests", "golden", "reference_liver_singlemesh_cpu_down8.npy")).astype(np.float64)
    sc = mi.load_file(LIVER_XML, spp=128, res_width=1920, res_height=1080)
    assert sc.desc.integrator.type == 3
    img = sc.render(seed=0).astype(np.float64)[..., :3].reshape(135, 8, 240, 8, 3).mean((1, 3))
    env = mi.load_string(re.sub(r'<shape type="obj".*?</shape>', '', open(LIVER_XML).read(), flags=re.S),
                         base_dir=os.path.dirname(LIVER_XML), spp=4, res_width=240, res_height=135)
    E = np.clip(env.render().astype(np.float64)[..., :3], 0, 1)
    mg, mo = np.abs(g - E).max(-1) > 0.08, np.abs(np.clip(img, 0, 1) - E).max(-1) > 0.08
    assert (mg & mo).sum() / (mg | mo).sum() > 0.98
    inner = binary_erosion(mg & mo, iterations=6)
    ours, ref = img[inner].mean(0), g[inner].mean(0)
    assert np.allclose(ours, ref, rtol=0.015), (ours / ref)
    assert np.abs(img - g)[inner].mean() < 0.08 * g[inner].mean()


from scene_gen import het_xml  # noqa: E402  (shared with bench.py)


@pytest.mark.parametrize("case", ["null", "dielectric-ld", "nonspectral", "two-media"])
def test_heterogeneous_medium_bit_exact(mi, orc, tmp_path, case):
    """grid-volume medium with delta tracking; null collisions keep the surface interaction found earlier
    (carried in the record's hit stream), emitter sampling marches through null collisions (volpath.cpp:238-259,468-505)."""
    rng = np.random.default_rng(3)
    grid = (0.05 + rng.random((12, 10, 8)) ** 3).astype(np.float32)
    vol = os.path.join(str(tmp_path), "smoke.vol"); mi.write_volume_grid(vol, grid)
    kw = {}
    if case == "null": xml = het_xml(vol)
    elif case == "dielectric-ld": xml = het_xml(vol, sampler="ldsampler", boundary="dielectric"); kw = dict(max_depth=30, rr_depth=2)
    elif case == "nonspectral": xml = het_xml(vol, spectral="false"); kw = dict(seed=3)
    else:
        hom = '<medium type="homogeneous" id="fog"><rgb name="sigma_t" value="0.5, 0.3, 0.8"/><rgb name="albedo" value="0.8, 0.8, 0.9"/></medium>'
        xml = het_xml(vol, extra_medium=hom).replace('<shape type="rectangle"><transform name="to_world"><scale value="6"/>',
            '<shape type="cube"><transform name="to_world"><scale value="0.5"/><translate x="2" y="-0.4"/></transform><bsdf type="null"/><ref name="interior" id="fog"/></shape>'
            '<shape type="rectangle"><transform name="to_world"><scale value="6"/>')
    sc = mi.load_string(xml); o = orc.OrcScene(sc)
    g = assert_lanes_equal(sc, o, 0, 64 * 48 * 16, **kw)
    assert np.isfinite(g).all() and sc.stats()["n_iter"] > 2 * 64 * 48 * 16
    raw = sc.render(return_raw=True, **kw)[1]
    assert film_close(raw, o.render(return_raw=True, **kw)[1]).all()
    if case == "null":
        with pytest.raises(RuntimeError, match="NotImplementedError"):       # the bio integrators need bio media (medium.cpp:83-90)
            sc.render_samples(0, 64, integrator="biovolpath")


@pytest.mark.parametrize("case", ["null", "dielectric-ld", "two-media"])
def test_prb_through_heterogeneous_media(mi, orc, tmp_path, case):
    """VERDICT r2 missing 1 / next 3: lrt_render_backward on scenes with a heterogeneous medium (prbvolpath.py:178-196 null collisions
    on the path, :404-415 ratio tracking on the emitter march).  Primal lanes bit for bit against the oracle, gradients against the
    oracle's (which test_oracle_pins.py pins by finite differences), two tile shards summing to the unsharded gradients."""
    import scene_gen
    vol = os.path.join(str(tmp_path), "smoke.vol"); mi.write_volume_grid(vol, scene_gen.smoke_grid())
    kw = dict(seed=3)
    if case == "null": xml = het_xml(vol, md=8)
    elif case == "dielectric-ld": xml = het_xml(vol, sampler="ldsampler", boundary="dielectric", md=10); kw = dict(seed=1, rr_depth=2)
    else: xml = scene_gen.two_media_xml(vol, hom='<medium type="homogeneous" id="fog"><rgb name="sigma_t" value="0.5, 0.3, 0.8"/><rgb name="albedo" value="0.8, 0.8, 0.9"/></medium>', md=8)
    sc = mi.load_string(xml.replace('type="volpath"', 'type="prbvolpath"')); o = orc.OrcScene(sc)
    h, w, c = sc.film_shape()
    assert_lanes_equal(sc, o, 0, w * h * sc.spp, **kw)
    grad = np.random.default_rng(11).random((h, w, c)).astype(np.float32) / (h * w * c)
    media = (0, 1, -1) if case == "two-media" else (0,)
    for m in media:
        gg, gc = sc.render_backward(grad, medium=m, **kw), o.render_backward(grad, medium=m, **kw)
        for k in ("sigma_t", "albedo"):
            assert np.abs(gg[k] - gc[k]).max() <= 3e-4 * max(np.abs(gc[k]).max(), 1e-7), (case, m, k, gg[k], gc[k])
        assert abs(gg["g"] - gc["g"]) <= 3e-4 * max(abs(gc["g"]), 1e-6) + 1e-9
        assert np.abs(gg["sigma_t"]).max() > 0 and np.abs(gg["albedo"]).max() > 0
    full = sc.render_backward(grad, medium=0, **kw)
    parts = [sc.render_backward(grad, medium=0, tile_rank=r, tile_count=2, **kw) for r in (0, 1)]
    for k in ("sigma_t", "albedo"):
        assert np.allclose(parts[0][k] + parts[1][k], full[k], rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("smis", ["true", "false"])
def test_volpathmis_bit_exact(mi, orc, tmp_path, smis):
    """volpathmis (volpathmis.cpp:127-699) on the device against the oracle: weight matrices in the path record, null collisions,
    emitter sampling with both weight sets, hide_emitters, ld sampler, the liver scene."""
    from conftest import LIVER_XML
    integ = f'<integrator type="volpathmis"><boolean name="use_spectral_mis" value="{smis}"/>'
    rng = np.random.default_rng(3)
    grid = (0.05 + rng.random((12, 10, 8)) ** 3).astype(np.float32)
    vol = os.path.join(str(tmp_path), "smoke.vol"); mi.write_volume_grid(vol, grid)
    hom = '<medium type="homogeneous" id="fog"><rgb name="sigma_t" value="0.9, 0.3, 1.6"/><rgb name="albedo" value="0.8, 0.8, 0.9"/><boolean name="has_spectral_extinction" value="false"/></medium>'
    two = het_xml(vol, extra_medium=hom).replace('<shape type="rectangle"><transform name="to_world"><scale value="6"/>',
        '<shape type="cube"><transform name="to_world"><scale value="0.5"/><translate x="2" y="-0.4"/></transform><bsdf type="null"/><ref name="interior" id="fog"/></shape>'
        '<shape type="rectangle"><transform name="to_world"><scale value="6"/>')
    scenes = [(fog_xml(), {}), (fog_xml(rf="box", sensor_medium='<ref id="haze"/>', exterior='<ref name="exterior" id="haze"/>', md="8"), dict(seed=2)),
              (two, dict(rr_depth=2)), (het_xml(vol, sampler="ldsampler", boundary="dielectric"), dict(max_depth=20)),
              (fog_xml(env='<emitter type="constant"><rgb name="radiance" value="0.3, 0.4, 0.6"/></emitter>', rf="box"), dict(hide_emitters=True, max_depth=4))]
    for xml, kw in scenes:
        sc = mi.load_string(xml.replace('<integrator type="volpath">', integ)); o = orc.OrcScene(sc)
        h, w, _ = sc.film_shape()
        g = assert_lanes_equal(sc, o, 0, w * h * sc.spp, **kw)
        assert np.isfinite(g).all()
    raw = sc.render(return_raw=True)[1]
    assert film_close(raw, o.render(return_raw=True)[1]).all()
    d = mi.cornell_box(); d['integrator'] = {'type': 'volpathmis', 'max_depth': 6, 'use_spectral_mis': smis == "true"}
    d['sensor']['film'].update({'width': 64, 'height': 64}); d['sensor']['sampler']['sample_count'] = 8
    for s in (mi.load_dict(d), mi.load_file(LIVER_XML, integrator="volpathmis", spp=8, res_width=192, res_height=108)):
        h, w, _ = s.film_shape()
        assert_lanes_equal(s, orc.OrcScene(s), 0, w * h * 8)
