"""BASELINE.json's configurations at their FULL sizes (full resolution AND full sample count) on the device.

The oracle cannot render 10^9 samples in a test, but a lane's value depends on nothing but its global lane id, so:
  * windows of lanes spread over the whole lane space of the full-size render (first lanes, last lanes, in between; ids beyond 2^30)
    are compared bit for bit with the oracle, together with the loop-trip and shadow-ray counts;
  * the full render is checked through size-independent properties: every sample is splatted exactly once (the box filter's weight
    channel is an exact integer sum, so `W == spp` per pixel is a checksum), the counters add up, the image is finite, and the films
    of two tile shards add up to the full film (tolerance: float-atomic order, 8e-5 relative to the pixel's weight).
C2 and C3 at reduced sample counts live in test_parity_gpu.py; C5's adjoint at full size is below."""
import numpy as np
import pytest

from conftest import LIVER_XML, PARENCHYMA_XML, MULTIMESH_XML, MULTIMESH_FULL_XML
from test_parity_gpu import bits

pytestmark = pytest.mark.gpu

W, H = 1920, 1080
CASES = {
    # name: (scene file, load_file keywords, expected integrator id, box filter with alpha?)
    "c3_volpath_512": (LIVER_XML, dict(integrator="volpath", spp=512), 1, True),
    "c3_bio_512": (LIVER_XML, dict(spp=512), 3, True),
    "multimesh_own_defaults_256": (MULTIMESH_FULL_XML, dict(spp=256), 3, False),
    "parenchyma_own_defaults_256": (PARENCHYMA_XML, dict(spp=256), 4, False),
    "c4_scene_xml_1024": (MULTIMESH_XML, dict(spp=1024), 3, False),
}


def film_close(a, b, wcol=-1):
    tol = 8e-5 * np.maximum(np.abs(b[..., wcol:]), 1.0)
    return np.abs(a - b) <= tol + 1e-6 * np.abs(b)


@pytest.mark.parametrize("name", sorted(CASES))
def test_full_size_lane_windows_and_film_properties(mi, orc, name):
    xml, kw, integ, box_alpha = CASES[name]
    sc = mi.load_file(xml, res_width=W, res_height=H, **kw)
    assert sc.desc.integrator.type == integ
    spp = sc.spp
    assert spp == kw["spp"]
    n_lanes = W * H * spp
    assert n_lanes > (1 << 28)
    o = orc.OrcScene(sc)
    n = 1536
    # windows: the first lanes, the image centre (where the tissue is), 3/4 down, a window that straddles a pixel boundary, the last lanes
    starts = [0, (H // 2 * W + W // 2) * spp - n // 2, int(0.62 * H) * W * spp + (W // 3) * spp + 7, n_lanes - n]
    for lane0 in starts:
        g = sc.render_samples(lane0, n)
        c = o.render_samples(lane0, n)
        same = (bits(g) == bits(c)).all(axis=1)
        assert same.all(), f"{name}: {(~same).sum()} of {n} lanes differ from lane {lane0 + int(np.argmin(same))}"
        st = sc.stats()
        assert st["n_iter"] == o.last_stats["n_iter"] and st["n_shadow"] == o.last_stats["n_shadow_needed"]
    img, raw = sc.render(return_raw=True)
    st = sc.stats()
    assert st["n_samples"] == n_lanes and st["n_iter"] >= n_lanes
    assert np.isfinite(img).all() and np.isfinite(raw).all() and (img[..., :3] >= 0).all()
    wsum = raw[..., -1].astype(np.float64).sum()
    if sc.desc.film.rfilter == 0:
        assert (raw[..., -1] == spp).all(), f"{name}: a pixel's weight is not its sample count"       # every sample splatted exactly once
        if raw.shape[-1] == 5: assert (raw[..., 3] <= spp).all() and raw[..., 3].max() == spp           # alpha counts valid paths
    else:
        # tent filter (radius 1): a sample's weights over its 2x2..3x3 footprint; away from the border the total mass is n_lanes * E[sum of weights]
        assert wsum > 0 and abs(wsum / n_lanes - raw[300:780, 600:1300, -1].astype(np.float64).mean() / spp) < 2e-3 * wsum / n_lanes
    # the films of two tile shards add up to the full film
    a = sc.render(return_raw=True, tile_rank=0, tile_count=2)[1].astype(np.float64) + sc.render(return_raw=True, tile_rank=1, tile_count=2)[1].astype(np.float64)
    ok = film_close(a, raw.astype(np.float64))
    assert ok.all(), f"{name}: {(~ok).sum()} film values differ between the sharded and the full render"
    # same seed, same film (up to the order of the float atomics); another seed, another film
    again = sc.render(return_raw=True)[1]
    assert film_close(again.astype(np.float64), raw.astype(np.float64)).all()
    other = sc.render(return_raw=True, seed=1)[1]
    assert not np.array_equal(other[..., :3], raw[..., :3])


def test_full_size_c2_cornell_256(mi, orc):
    """BASELINE config C2 at full size: mi.cornell_box() at 1080 x 1080, `path`, 256 spp, Gaussian filter."""
    d = mi.cornell_box(); d["sensor"]["film"].update({"width": 1080, "height": 1080}); d["sensor"]["sampler"]["sample_count"] = 256
    sc = mi.load_dict(d)
    n_lanes = 1080 * 1080 * 256
    o = orc.OrcScene(sc)
    for lane0 in (0, n_lanes // 2 + 12345, n_lanes - 2048):
        g = sc.render_samples(lane0, 2048); c = o.render_samples(lane0, 2048)
        assert (bits(g) == bits(c)).all()
        assert sc.stats()["n_iter"] == o.last_stats["n_iter"] and sc.stats()["n_shadow"] == o.last_stats["n_shadow_needed"]
    img, raw = sc.render(return_raw=True)
    assert sc.stats()["n_samples"] == n_lanes and np.isfinite(img).all() and (img >= 0).all()
    a = sc.render(return_raw=True, tile_rank=0, tile_count=2)[1].astype(np.float64) + sc.render(return_raw=True, tile_rank=1, tile_count=2)[1].astype(np.float64)
    assert film_close(a, raw.astype(np.float64)).all()


def test_full_size_c5_prb_gradients_shard_sum(mi):
    """BASELINE config C5 at full size: the PRB adjoint on Parenchyma, 1920 x 1080, 256 spp.  The gradients of two tile shards add up to
    the unsharded gradients (the sums run in double precision; float-atomic order only enters through the weight film)."""
    sc = mi.load_file(PARENCHYMA_XML, integrator="prbvolpath", spp=256, res_width=W, res_height=H)
    h, w, _ = sc.film_shape(); C = sc.raw_channels()
    grad = np.ones((h, w, C - 1), np.float32) / (h * w * (C - 1))
    full = sc.render_backward(grad, spp=sc.spp, seed=3)
    assert sc.stats()["n_samples"] == w * h * sc.spp
    parts = [sc.render_backward(grad, spp=sc.spp, seed=3, tile_rank=r, tile_count=2) for r in (0, 1)]
    for key in ("sigma_t", "albedo"):
        tot = np.asarray(parts[0][key], np.float64) + np.asarray(parts[1][key], np.float64)
        assert np.all(np.isfinite(tot)) and np.allclose(tot, full[key], rtol=2e-4, atol=1e-9), (key, tot, full[key])
    assert np.isclose(parts[0]["g"] + parts[1]["g"], full["g"], rtol=2e-4, atol=1e-9)
    assert np.any(np.asarray(full["sigma_t"]) != 0) and np.any(np.asarray(full["albedo"]) != 0)
