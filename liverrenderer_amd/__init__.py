"""liverrenderer_amd -- host-side mirror of the reference's Python entry points
(`mi.set_variant`, `mi.load_file`, `mi.load_dict`, `mi.cornell_box`, `mi.render`,
`mi.traverse`; src/python/python/util.py:394-702) over the `hip_ad_rgb` C ABI
(include/liverrt.h).  All arithmetic of the render path runs in the HIP kernels
of libliverrt.so; this module only marshals arguments.
"""
import ctypes as C
import math
import os

import numpy as np

from . import _lib
from ._lib import make_opts

__all__ = ["set_variant", "variant", "variants", "load_file", "load_string", "load_dict", "cornell_box", "render",
           "traverse", "Scene", "ScalarTransform4f", "render_stats", "write_volume_grid", "Bitmap", "Struct", "util", "read_image", "write_exr", "write_png"]

_VARIANT = "hip_ad_rgb"


def variants():
    return ["hip_ad_rgb"]


def set_variant(name):
    """mi.set_variant(): only `hip_ad_rgb` exists in this back-end."""
    global _VARIANT
    if name not in variants():
        raise ImportError(f"Requested an unsupported variant \"{name}\". The following variants are available: "
                          + ", ".join(variants()) + ".")
    _VARIANT = name


def variant():
    return _VARIANT


class ScalarTransform4f:
    """Subset of mi.ScalarTransform4f used by scene dictionaries (right-multiplying chain)."""

    def __init__(self, m=None):
        self.matrix = np.eye(4) if m is None else np.array(m, dtype=np.float64).reshape(4, 4)

    def __matmul__(self, o):
        return ScalarTransform4f(self.matrix @ o.matrix)

    def translate(self, v):
        m = np.eye(4); m[:3, 3] = v
        return ScalarTransform4f(self.matrix @ m)

    def scale(self, v):
        v = np.broadcast_to(np.asarray(v, dtype=np.float64), (3,))
        return ScalarTransform4f(self.matrix @ np.diag([v[0], v[1], v[2], 1.0]))

    def rotate(self, axis, angle):
        x, y, z = [float(a) for a in axis]
        a = math.radians(angle); s, c = math.sin(a), math.cos(a); t = 1 - c
        m = np.eye(4)
        m[:3, :3] = [[c + x * x * t, x * y * t - z * s, x * z * t + y * s],
                     [y * x * t + z * s, c + y * y * t, y * z * t - x * s],
                     [z * x * t - y * s, z * y * t + x * s, c + z * z * t]]
        return ScalarTransform4f(self.matrix @ m)

    def look_at(self, origin, target, up):
        o, tg, up = [np.asarray(v, dtype=np.float64) for v in (origin, target, up)]
        d = tg - o; d /= np.linalg.norm(d)
        left = np.cross(up, d); left /= np.linalg.norm(left)
        nup = np.cross(d, left)
        m = np.eye(4); m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, nup, d, o
        return ScalarTransform4f(self.matrix @ m)


def cornell_box():
    """mi.cornell_box() (src/python/python/util.py:567-702): the same scene dictionary."""
    T = ScalarTransform4f
    return {
        'type': 'scene',
        'integrator': {'type': 'path', 'max_depth': 8},
        'sensor': {
            'type': 'perspective', 'fov_axis': 'smaller', 'near_clip': 0.001, 'far_clip': 100.0, 'focus_distance': 1000,
            'fov': 39.3077,
            'to_world': T().look_at(origin=[0, 0, 3.90], target=[0, 0, 0], up=[0, 1, 0]),
            'sampler': {'type': 'independent', 'sample_count': 64},
            'film': {'type': 'hdrfilm', 'width': 256, 'height': 256, 'rfilter': {'type': 'gaussian'},
                     'pixel_format': 'rgb', 'component_format': 'float32'},
        },
        'white': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.885809, 0.698859, 0.666422]}},
        'green': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.105421, 0.37798, 0.076425]}},
        'red': {'type': 'diffuse', 'reflectance': {'type': 'rgb', 'value': [0.570068, 0.0430135, 0.0443706]}},
        'light': {
            'type': 'rectangle',
            'to_world': T().translate([0.0, 0.99, 0.01]).rotate([1, 0, 0], 90).scale([0.23, 0.19, 0.19]),
            'bsdf': {'type': 'ref', 'id': 'white'},
            'emitter': {'type': 'area', 'radiance': {'type': 'rgb', 'value': [18.387, 13.9873, 6.75357]}},
        },
        'floor': {'type': 'rectangle', 'to_world': T().translate([0.0, -1.0, 0.0]).rotate([1, 0, 0], -90),
                  'bsdf': {'type': 'ref', 'id': 'white'}},
        'ceiling': {'type': 'rectangle', 'to_world': T().translate([0.0, 1.0, 0.0]).rotate([1, 0, 0], 90),
                    'bsdf': {'type': 'ref', 'id': 'white'}},
        'back': {'type': 'rectangle', 'to_world': T().translate([0.0, 0.0, -1.0]), 'bsdf': {'type': 'ref', 'id': 'white'}},
        'green-wall': {'type': 'rectangle', 'to_world': T().translate([1.0, 0.0, 0.0]).rotate([0, 1, 0], -90),
                       'bsdf': {'type': 'ref', 'id': 'green'}},
        'red-wall': {'type': 'rectangle', 'to_world': T().translate([-1.0, 0.0, 0.0]).rotate([0, 1, 0], 90),
                     'bsdf': {'type': 'ref', 'id': 'red'}},
        'small-box': {'type': 'cube', 'to_world': T().translate([0.335, -0.7, 0.38]).rotate([0, 1, 0], -17).scale(0.3),
                      'bsdf': {'type': 'ref', 'id': 'white'}},
        'large-box': {'type': 'cube',
                      'to_world': T().translate([-0.33, -0.4, -0.28]).rotate([0, 1, 0], 18.25).scale([0.3, 0.61, 0.3]),
                      'bsdf': {'type': 'ref', 'id': 'white'}},
    }


# ------------------------------------------------------------- dict -> XML
_TAGS = {
    "scene": "scene", "path": "integrator", "volpath": "integrator", "prbvolpath": "integrator",
    "biovolpath": "integrator", "biovolpath06": "integrator", "volpathmis": "integrator",
    "perspective": "sensor", "independent": "sampler", "ldsampler": "sampler", "hdrfilm": "film",
    "box": "rfilter", "gaussian": "rfilter", "tent": "rfilter",
    "diffuse": "bsdf", "dielectric": "bsdf", "bumpmap": "bsdf", "null": "bsdf",
    "bitmap": "texture", "checkerboard": "texture",
    "homogeneous": "medium", "liver": "medium", "parenchyma": "medium", "glissonCapsule": "medium", "heterogeneous": "medium",
    "gridvolume": "volume",
    "isotropic": "phase", "hg": "phase",
    "obj": "shape", "rectangle": "shape", "cube": "shape",
    "area": "emitter", "envmap": "emitter", "constant": "emitter",
}


def _esc(s):
    return str(s).replace("&", "&amp;").replace('"', "&quot;").replace("<", "&lt;").replace(">", "&gt;")


def _fmt(v):
    return repr(float(v))


def _dict_to_xml(name, d, out, indent, top_ids):
    typ = d.get("type")
    if typ == "ref":
        out.append(f'{indent}<ref id="{_esc(d["id"])}"' + (f' name="{_esc(name)}"' if name and name not in ("bsdf",) else "") + "/>")
        return
    if typ in ("rgb", "srgb", "spectrum"):
        v = d["value"]
        v = ", ".join(_fmt(x) for x in (v if hasattr(v, "__len__") else [v]))
        out.append(f'{indent}<rgb name="{_esc(name)}" value="{v}"/>')
        return
    if typ not in _TAGS:
        raise RuntimeError(f'load_dict(): unsupported plugin type "{typ}"')
    tag = _TAGS[typ]
    attrs = "" if tag == "scene" else f' type="{_esc(typ)}"'
    if tag == "scene":
        attrs += ' version="3.0.0"'
    if name is not None and tag != "scene":
        if indent == "    " and tag in ("bsdf", "medium", "texture", "shape", "emitter"):
            attrs += f' id="{_esc(name)}"'
        elif tag not in ("integrator", "sensor", "sampler", "film", "rfilter", "phase") and not (tag == "bsdf" and name == "bsdf") \
                and not (tag == "emitter" and name == "emitter"):
            attrs += f' name="{_esc(name)}"'
    out.append(f"{indent}<{tag}{attrs}>")
    sub = indent + "    "
    for k, v in d.items():
        if k == "type":
            continue
        if isinstance(v, dict):
            _dict_to_xml(k, v, out, sub, top_ids)
        elif isinstance(v, ScalarTransform4f):
            m = " ".join(_fmt(x) for x in v.matrix.reshape(-1))
            out.append(f'{sub}<transform name="{_esc(k)}"><matrix value="{m}"/></transform>')
        elif isinstance(v, bool):
            out.append(f'{sub}<boolean name="{_esc(k)}" value="{"true" if v else "false"}"/>')
        elif isinstance(v, (int, np.integer)):
            out.append(f'{sub}<integer name="{_esc(k)}" value="{int(v)}"/>')
        elif isinstance(v, (float, np.floating)):
            out.append(f'{sub}<float name="{_esc(k)}" value="{_fmt(v)}"/>')
        elif isinstance(v, str):
            out.append(f'{sub}<string name="{_esc(k)}" value="{_esc(v)}"/>')
        elif isinstance(v, (list, tuple, np.ndarray)):
            out.append(f'{sub}<rgb name="{_esc(k)}" value="{", ".join(_fmt(x) for x in v)}"/>')
        else:
            raise RuntimeError(f'load_dict(): unsupported value for key "{k}": {type(v)}')
    out.append(f"{indent}</{tag}>")


def dict_to_xml(d):
    if d.get("type") != "scene":
        raise RuntimeError("load_dict(): the top-level dictionary must have type 'scene'")
    out = []
    # objects that are referenced must be declared before their first use
    refd = set()

    def scan(x):
        if isinstance(x, dict):
            if x.get("type") == "ref":
                refd.add(x["id"])
            for v in x.values():
                scan(v)
    scan(d)
    ordered = {"type": "scene"}
    for k, v in d.items():
        if k in refd:
            ordered[k] = v
    for k, v in d.items():
        if k not in ordered:
            ordered[k] = v
    _dict_to_xml(None, ordered, out, "", refd)
    return "\n".join(out)


# ------------------------------------------------------------------ scenes
class Scene:
    """Handle to a loaded scene (wraps `lrt_scene*`)."""

    def __init__(self, handle):
        self._h = C.c_void_p(handle)
        self._lib = _lib.lib()

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                self._lib.lrt_scene_free(self._h)
                self._h = C.c_void_p(None)
        except Exception:
            pass

    # -- description -----------------------------------------------------
    @property
    def desc(self):
        return self._lib.lrt_scene_desc_get(self._h).contents

    def film_shape(self):
        f = self.desc.film
        return f.crop_height, f.crop_width, (4 if f.has_alpha else 3)

    def raw_channels(self):
        return 5 if self.desc.film.has_alpha else 4

    @property
    def spp(self):
        return self.desc.sample_count

    def medium_ids(self):
        d = self.desc
        return [d.media[i].id.decode() for i in range(d.n_media)]

    # -- rendering ---------------------------------------------------------
    def render(self, spp=0, seed=0, integrator=None, max_depth=None, rr_depth=None, hide_emitters=None,
               tile_rank=0, tile_count=1, device=0, return_raw=False):
        h, w, c = self.film_shape()
        img = np.empty((h, w, c), dtype=np.float32)
        raw = np.empty((h, w, self.raw_channels()), dtype=np.float32) if return_raw else None
        o = make_opts(integrator, max_depth, rr_depth, hide_emitters, spp, seed, tile_rank, tile_count, device)
        _lib.check(self._lib.lrt_render(self._h, C.byref(o), raw.ctypes.data if return_raw else None, img.ctypes.data))
        return (img, raw) if return_raw else img

    def render_multi(self, devices, spp=0, seed=0, integrator=None, max_depth=None, rr_depth=None, hide_emitters=None, return_raw=False):
        """lrt_render_multi: one process, the image tile-sharded over `devices` (a list of HIP ordinals), films summed by one RCCL
        all-reduce, developed on the first device.  Same image as render()."""
        h, w, c = self.film_shape()
        img = np.empty((h, w, c), dtype=np.float32)
        raw = np.empty((h, w, self.raw_channels()), dtype=np.float32) if return_raw else None
        o = make_opts(integrator, max_depth, rr_depth, hide_emitters, spp, seed)
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        _lib.check(self._lib.lrt_render_multi(self._h, C.byref(o), len(devices), ids, raw.ctypes.data if return_raw else None, img.ctypes.data))
        return (img, raw) if return_raw else img

    def render_backward_multi(self, grad_image, devices, **kw):
        """lrt_render_backward_multi: the PRB adjoint sharded over `devices`, gradients reduced with one all-reduce."""
        g = np.ascontiguousarray(grad_image, dtype=np.float32)
        o = make_opts(kw.get("integrator"), kw.get("max_depth"), kw.get("rr_depth"), kw.get("hide_emitters"), kw.get("spp", 0), kw.get("seed", 0), grad_medium=kw.get("medium", -1))
        ids = (C.c_int * len(devices))(*[int(d) for d in devices])
        out = _lib.ParamGrads()
        _lib.check(self._lib.lrt_render_backward_multi(self._h, C.byref(o), len(devices), ids, g.ctypes.data, C.byref(out)))
        return {"sigma_t": np.array(out.d_sigma_t[:], dtype=np.float32), "albedo": np.array(out.d_albedo[:], dtype=np.float32), "g": float(out.d_g)}

    def render_to_device(self, film_ptr, image_ptr=None, **kw):
        """Render into caller-provided DEVICE buffers (e.g. torch tensors' data_ptr())."""
        o = make_opts(kw.get("integrator"), kw.get("max_depth"), kw.get("rr_depth"), kw.get("hide_emitters"), kw.get("spp", 0),
                      kw.get("seed", 0), kw.get("tile_rank", 0), kw.get("tile_count", 1), kw.get("device", 0), True)
        _lib.check(self._lib.lrt_render(self._h, C.byref(o), C.c_void_p(film_ptr), C.c_void_p(image_ptr) if image_ptr else None))

    def develop(self, film_raw=None, film_ptr=None, image_ptr=None):
        if film_ptr is not None:
            _lib.check(self._lib.lrt_film_develop(self._h, C.c_void_p(film_ptr), C.c_void_p(image_ptr), 1))
            return None
        h, w, c = self.film_shape()
        film_raw = np.ascontiguousarray(film_raw, dtype=np.float32)
        img = np.empty((h, w, c), dtype=np.float32)
        _lib.check(self._lib.lrt_film_develop(self._h, film_raw.ctypes.data, img.ctypes.data, 0))
        return img

    def render_samples(self, lane_begin, n, **kw):
        out = np.empty((n, 4), dtype=np.float32)
        o = make_opts(kw.get("integrator"), kw.get("max_depth"), kw.get("rr_depth"), kw.get("hide_emitters"), kw.get("spp", 0),
                      kw.get("seed", 0), 0, 1, kw.get("device", 0))
        _lib.check(self._lib.lrt_render_samples(self._h, C.byref(o), int(lane_begin), int(n), out.ctypes.data))
        return out

    def render_backward(self, grad_image, **kw):
        """PRB adjoint.  `medium`: index of the medium whose sigma_t / albedo / g the gradients refer to; the default -1 sums the adjoint
        over all media into one parameter set (the round-1 meaning; lrt_render_opts.grad_medium in the C ABI, where a zero-initialised
        struct selects medium 0)."""
        g = np.ascontiguousarray(grad_image, dtype=np.float32)
        o = make_opts(kw.get("integrator"), kw.get("max_depth"), kw.get("rr_depth"), kw.get("hide_emitters"), kw.get("spp", 0),
                      kw.get("seed", 0), kw.get("tile_rank", 0), kw.get("tile_count", 1), kw.get("device", 0), grad_medium=kw.get("medium", -1))
        out = _lib.ParamGrads()
        _lib.check(self._lib.lrt_render_backward(self._h, C.byref(o), g.ctypes.data, C.byref(out)))
        return {"sigma_t": np.array(out.d_sigma_t[:], dtype=np.float32), "albedo": np.array(out.d_albedo[:], dtype=np.float32),
                "g": float(out.d_g)}

    def stats(self):
        s = _lib.RenderStats()
        _lib.check(self._lib.lrt_render_stats_get(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _lib.RenderStats._fields_}

    def trace(self, o, d, tmax=None, any_hit=False):
        o = np.ascontiguousarray(o, dtype=np.float32); d = np.ascontiguousarray(d, dtype=np.float32)
        n = o.shape[0]
        tmax = np.full(n, np.finfo(np.float32).max, dtype=np.float32) if tmax is None else np.ascontiguousarray(tmax, dtype=np.float32)
        cols = [np.ascontiguousarray(a) for a in (o[:, 0], o[:, 1], o[:, 2], d[:, 0], d[:, 1], d[:, 2], tmax)]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32); prim = np.empty(n, np.uint32)
        FP = C.POINTER(C.c_float)
        rays = _lib.RaysSoA(*[c.ctypes.data_as(FP) for c in cols])
        hits = _lib.HitsSoA(t.ctypes.data_as(FP), u.ctypes.data_as(FP), v.ctypes.data_as(FP), prim.ctypes.data_as(C.POINTER(C.c_uint32)))
        _lib.check(self._lib.lrt_trace(self._h, C.byref(rays), C.byref(hits), n, int(any_hit)))
        return t, u, v, prim

    # -- parameters (mi.traverse) -------------------------------------------
    def param_set(self, key, value):
        v = np.atleast_1d(np.asarray(value, dtype=np.float32))
        _lib.check(self._lib.lrt_param_set(self._h, key.encode(), v.ctypes.data_as(C.POINTER(C.c_float)), int(v.size)))

    def param_get(self, key, n=3):
        v = np.zeros(n, dtype=np.float32)
        _lib.check(self._lib.lrt_param_get(self._h, key.encode(), v.ctypes.data_as(C.POINTER(C.c_float)), n))
        return v


def math_eval(fn, x, y=None, device=0):
    """lrt_math_eval (test hook): the device's transcendental kernels, one value per lane.  Returns (out, out2)."""
    L = _lib.lib()
    x = np.ascontiguousarray(x, np.float32); y = x if y is None else np.ascontiguousarray(y, np.float32)
    out = np.empty_like(x); out2 = np.empty_like(x)
    FP = C.POINTER(C.c_float)
    _lib.check(L.lrt_math_eval(int(fn), x.ctypes.data_as(FP), y.ctypes.data_as(FP), x.size, out.ctypes.data_as(FP), out2.ctypes.data_as(FP), int(device)))
    return out, out2


class SceneParameters(dict):
    """mi.traverse(scene): dict of differentiable medium parameters; assignments are pushed by update()."""

    def __init__(self, scene):
        super().__init__()
        self._scene = scene
        for mid in scene.medium_ids():
            super().__setitem__(f"{mid}.sigma_t.value", scene.param_get(f"{mid}.sigma_t.value", 3))
            super().__setitem__(f"{mid}.albedo.value", scene.param_get(f"{mid}.albedo.value", 3))
            super().__setitem__(f"{mid}.scale", scene.param_get(f"{mid}.scale", 1))
            super().__setitem__(f"{mid}.phase_function.g", scene.param_get(f"{mid}.phase_function.g", 1))
        for i in range(scene.desc.n_media):                   # `parenchyma` also traverses its absorbers (src/media/parenchyma.cpp:154-160)
            m = scene.desc.media[i]
            if m.type == _lib.MEDIUM["parenchyma"]:
                mid = m.id.decode()
                for k in ("sigma_blood.value", "sigma_bile.value", "sigma_lipid_water.value"):
                    super().__setitem__(f"{mid}.{k}", scene.param_get(f"{mid}.{k}", 3))
                super().__setitem__(f"{mid}.sigma_hepatocity", scene.param_get(f"{mid}.sigma_hepatocity", 1))
        self._dirty = set()

    def __setitem__(self, k, v):
        if k not in self:
            raise KeyError(k)
        super().__setitem__(k, np.atleast_1d(np.asarray(v, dtype=np.float32)))
        self._dirty.add(k)

    def update(self, values=None):
        if values:
            for k, v in values.items():
                self[k] = v
        for k in sorted(self._dirty):
            if k.endswith(".phase_function.g") and float(self[k][0]) == 0.0:
                try:                                   # g = 0 is a value of an hg phase function; an isotropic one has no such key
                    self._scene.param_set(k, self[k])
                except RuntimeError as e:
                    if "isotropic" not in str(e): raise
                continue
            self._scene.param_set(k, self[k])
        self._dirty.clear()


def traverse(scene):
    return SceneParameters(scene)


def write_volume_grid(path, data, bbox_min=(0.0, 0.0, 0.0), bbox_max=(1.0, 1.0, 1.0)):
    """mi.VolumeGrid(array).write(path) (src/render/volumegrid.cpp:94-118): a one-channel float32 grid, array shape
    (res_z, res_y, res_x), as the version-3 ".vol" file `gridvolume` reads."""
    import struct
    a = np.ascontiguousarray(data, dtype=np.float32)
    if a.ndim != 3:
        raise ValueError("write_volume_grid: expected an array of shape (res_z, res_y, res_x)")
    with open(path, "wb") as f:
        f.write(b"VOL" + struct.pack("<B", 3) + struct.pack("<iiiii", 1, a.shape[2], a.shape[1], a.shape[0], 1))
        f.write(struct.pack("<6f", *bbox_min, *bbox_max))
        f.write(a.tobytes())


def _defines(kw):
    arr = (C.c_char_p * len(kw))(*[f"{k}={v}".encode() for k, v in kw.items()])
    return arr, len(kw)


def load_file(path, **defines):
    """mi.load_file(path, **defines): `defines` replace `$name` parameters (like `-Dname=value`)."""
    L = _lib.lib()
    h = C.c_void_p()
    arr, n = _defines(defines)
    _lib.check(L.lrt_scene_load_xml(os.fspath(path).encode(), arr, n, C.byref(h)))
    return Scene(h.value)


def load_string(xml, base_dir=".", **defines):
    L = _lib.lib()
    h = C.c_void_p()
    arr, n = _defines(defines)
    _lib.check(L.lrt_scene_load_xml_string(xml.encode(), os.fspath(base_dir).encode(), arr, n, C.byref(h)))
    return Scene(h.value)


def load_dict(d, base_dir="."):
    return load_string(dict_to_xml(d), base_dir)


def render(scene, spp=0, seed=0, integrator=None, **kw):
    """mi.render(scene, spp=..., seed=...): developed H x W x (3|4) float32 image."""
    return scene.render(spp=spp, seed=seed, integrator=integrator, **kw)


def render_stats(scene):
    return scene.stats()


def scene_from_buffers(positions, faces, normals=None, texcoords=None, reflectance=(0.5, 0.5, 0.5), film=(64, 64),
                       sensor_to_world=None, fov=45.0, spp=4, integrator="path", max_depth=-1, constant_radiance=None):
    """Build a one-mesh scene from raw buffers through `lrt_scene_from_desc` (the "from buffers" entry of the C ABI).
    The mesh gets a diffuse BSDF; an optional constant environment emitter lights it."""
    L = _lib.lib()
    pos = np.ascontiguousarray(positions, dtype=np.float32).reshape(-1, 3)
    fc = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
    nv, nf = pos.shape[0], fc.shape[0]
    nrm = np.zeros((nv, 3), np.float32) if normals is None else np.ascontiguousarray(normals, dtype=np.float32).reshape(-1, 3)
    uv = np.zeros((nv, 2), np.float32) if texcoords is None else np.ascontiguousarray(texcoords, dtype=np.float32).reshape(-1, 2)
    fshape = np.zeros(nf, np.uint32)
    FP, UP = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    shape = _lib.ShapeDesc(kind=0, first_face=0, n_faces=nf, bsdf=0, emitter=-1, interior_medium=-1, exterior_medium=-1,
                           has_normals=int(normals is not None), has_texcoords=int(texcoords is not None), flip_normals=0)
    shape.to_world[:] = list(np.eye(4, dtype=np.float32).reshape(-1))
    tex = _lib.TextureDesc(type=0, width=0, height=0, channels=0)
    tex.color0[:] = list(reflectance); tex.color1[:] = list(reflectance); tex.to_uv[:] = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    bsdf = _lib.BsdfDesc(type=0, reflectance=0, eta=1.0, nested=-1, texture=-1, scale=1.0)
    d = _lib.SceneDesc()
    d.n_vertices, d.n_faces, d.n_shapes, d.n_bsdfs, d.n_textures, d.n_media = nv, nf, 1, 1, 1, 0
    d.positions, d.normals, d.texcoords = pos.ctypes.data_as(FP), nrm.ctypes.data_as(FP), uv.ctypes.data_as(FP)
    d.faces, d.face_shape = fc.ctypes.data_as(UP), fshape.ctypes.data_as(UP)
    d.shapes, d.bsdfs, d.textures = C.pointer(shape), C.pointer(bsdf), C.pointer(tex)
    em = _lib.EmitterDesc(type=2, shape=-1, scale=1.0)
    if constant_radiance is not None:
        em.radiance[:] = list(constant_radiance); em.to_world[:] = list(np.eye(4, dtype=np.float32).reshape(-1))
        d.n_emitters, d.emitters = 1, C.pointer(em)
    tw = ScalarTransform4f() if sensor_to_world is None else sensor_to_world
    d.sensor.to_world[:] = [float(x) for x in np.asarray(tw.matrix, dtype=np.float32).reshape(-1)]
    d.sensor.fov_x, d.sensor.near_clip, d.sensor.far_clip, d.sensor.medium = fov, 1e-2, 1e4, -1
    d.film.width, d.film.height = film; d.film.crop_width, d.film.crop_height = film
    d.film.has_alpha, d.film.rfilter, d.film.rfilter_param = 0, 0, 0.5
    d.integrator.type, d.integrator.max_depth, d.integrator.rr_depth, d.integrator.hide_emitters = _lib.INTEGRATOR[integrator], max_depth, 5, 0
    d.use_spectral_mis = 1
    d.sample_count, d.sampler_seed = spp, 0
    h = C.c_void_p()
    _lib.check(L.lrt_scene_from_desc(C.byref(d), C.byref(h)))
    return Scene(h.value)


def read_image(path):
    """mi.Bitmap(path) as a float32 array (h, w, channels); PNG values are in [0, 1] as stored."""
    L = _lib.lib()
    w, h, c = C.c_int(), C.c_int(), C.c_int(); data = C.POINTER(C.c_float)()
    _lib.check(L.lrt_image_read(os.fspath(path).encode(), C.byref(w), C.byref(h), C.byref(c), C.byref(data)))
    try:
        return np.ctypeslib.as_array(data, (h.value, w.value, c.value)).copy()
    finally:
        L.lrt_image_free(data)


def write_png(path, image):
    """8-bit sRGB PNG of a linear float image (LiverRenderer.py:383-385: Bitmap.convert(RGBA, UInt8, srgb_gamma=True))."""
    img = np.ascontiguousarray(image, dtype=np.float32)
    if img.ndim == 2:
        img = img[..., None]
    _lib.check(_lib.lib().lrt_image_write_png(os.fspath(path).encode(), img.shape[1], img.shape[0], img.shape[2], img.ctypes.data))


def write_exr(path, image):
    img = np.ascontiguousarray(image, dtype=np.float32)
    if img.ndim == 2:
        img = img[..., None]
    _lib.check(_lib.lib().lrt_image_write_exr(os.fspath(path).encode(), img.shape[1], img.shape[0], img.shape[2], img.ctypes.data))


# ---- the handful of image-side names the reference's drivers use (MitsubaRunner.py:166-167, LiverRenderer.py:383-385),
# so that those scripts run with the import swapped: mi.Bitmap(path | array), Bitmap.convert(RGBA, UInt8, srgb_gamma=True),
# mi.util.write_bitmap(path, image).  Arrays stay float32 and linear; the sRGB / 8-bit conversion happens in the writer.
class Struct:
    class Type:
        UInt8, Float16, Float32 = "uint8", "float16", "float32"


class Bitmap:
    class PixelFormat:
        Y, YA, RGB, RGBA = "y", "ya", "rgb", "rgba"

    def __init__(self, src):
        self.data = read_image(src) if isinstance(src, (str, os.PathLike)) else np.asarray(src, dtype=np.float32)
        if self.data.ndim == 2:
            self.data = self.data[..., None]
        self.component_format, self.srgb_gamma = Struct.Type.Float32, False

    def size(self):
        return (self.data.shape[1], self.data.shape[0])

    def channel_count(self):
        return self.data.shape[2]

    def convert(self, pixel_format=None, component_format=None, srgb_gamma=None):
        """src/core/bitmap.cpp Bitmap::convert, for the conversions the drivers request: channel layout now, the transfer
        function and the quantisation when the bitmap is written."""
        d, out = self.data, Bitmap.__new__(Bitmap)
        n = {"y": 1, "ya": 2, "rgb": 3, "rgba": 4}.get(pixel_format, d.shape[2])
        if n != d.shape[2]:
            rgb = d[..., :3] if d.shape[2] >= 3 else np.repeat(d[..., :1], 3, axis=2)
            alpha = d[..., -1:] if d.shape[2] in (2, 4) else np.ones_like(d[..., :1])
            lum = (0.212671 * rgb[..., :1] + 0.715160 * rgb[..., 1:2] + 0.072169 * rgb[..., 2:3])
            d = {1: lum, 2: np.concatenate([lum, alpha], 2), 3: rgb, 4: np.concatenate([rgb, alpha], 2)}[n]
        out.data = np.ascontiguousarray(d, dtype=np.float32)
        out.component_format = component_format or self.component_format
        out.srgb_gamma = self.srgb_gamma if srgb_gamma is None else bool(srgb_gamma)
        return out

    def write(self, path):
        util.write_bitmap(path, self)

    def __array__(self, dtype=None):
        return self.data if dtype is None else self.data.astype(dtype)


class util:
    @staticmethod
    def write_bitmap(path, image, write_async=False):
        """src/python/python/util.py write_bitmap: 8-bit formats get the sRGB transfer function, EXR stays linear float."""
        data = image.data if isinstance(image, Bitmap) else np.asarray(image, dtype=np.float32)
        if os.fspath(path).lower().endswith(".png"):
            write_png(path, data)
        elif os.fspath(path).lower().endswith(".exr"):
            write_exr(path, data)
        else:
            raise RuntimeError(f"write_bitmap: unsupported file format \"{path}\" (supported: .png, .exr)")

    cornell_box = staticmethod(lambda: cornell_box())
