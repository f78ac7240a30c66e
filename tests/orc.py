"""ctypes wrapper of the CPU oracle (oracle/liborc.so).  TEST INFRASTRUCTURE ONLY:
imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

from liverrenderer_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORC_DIR = os.path.join(ROOT, "oracle")
ORC_LIB = os.path.join(ORC_DIR, "liborc.so")


class OrcStats(C.Structure):
    _fields_ = [("n_samples", C.c_uint64), ("n_iter", C.c_uint64), ("n_shadow", C.c_uint64), ("n_shadow_needed", C.c_uint64)]


_orc = None


def build():
    subprocess.run(["make", "-s", "-C", ORC_DIR], check=True)


def use_library(path):
    """Load another build of the oracle from now on (bench.py's CPU baseline: a -O3 -march=native build made on the host it runs on)."""
    global _orc, ORC_LIB
    ORC_LIB = path; _orc = None


def lib():
    global _orc
    if _orc is not None:
        return _orc
    if not os.path.exists(ORC_LIB):
        build()
    L = C.CDLL(ORC_LIB)
    P = C.POINTER
    L.orc_scene_create.argtypes = [P(_lib.SceneDesc)]
    L.orc_scene_create.restype = C.c_void_p
    L.orc_scene_free.argtypes = [C.c_void_p]
    L.orc_scene_free.restype = None
    L.orc_param_set.argtypes = [C.c_void_p, C.c_char_p, P(C.c_float), C.c_int]
    L.orc_render.argtypes = [C.c_void_p, P(_lib.RenderOpts), C.c_int, C.c_void_p, C.c_void_p, P(OrcStats)]
    L.orc_render_scalar.argtypes = [C.c_void_p, P(_lib.RenderOpts), C.c_int, C.c_void_p, C.c_void_p, P(OrcStats)]
    L.orc_render_samples.argtypes = [C.c_void_p, P(_lib.RenderOpts), C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, P(OrcStats)]
    L.orc_render_backward.argtypes = [C.c_void_p, P(_lib.RenderOpts), C.c_int, C.c_void_p, P(_lib.ParamGrads)]
    L.orc_trace.argtypes = [C.c_void_p, P(_lib.RaysSoA), P(_lib.HitsSoA), C.c_uint32, C.c_int, C.c_int]
    L.orc_tea32.argtypes = [C.c_uint32, C.c_uint32, C.c_int, P(C.c_uint32), P(C.c_uint32)]
    L.orc_tea32.restype = None
    L.orc_ld_sample.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, P(C.c_float)]
    L.orc_ld_sample.restype = None
    L.orc_ld_round_sample_count.argtypes = [C.c_uint32]; L.orc_ld_round_sample_count.restype = C.c_uint32
    L.orc_permute.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32]; L.orc_permute.restype = C.c_uint32
    L.orc_tea_float32.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    L.orc_tea_float32.restype = C.c_float
    L.orc_tea_float64.argtypes = [C.c_uint32, C.c_uint32, C.c_int]
    L.orc_tea_float64.restype = C.c_double
    L.orc_pcg32_u32.argtypes = [C.c_uint64, C.c_uint64, C.c_uint32, P(C.c_uint32)]
    L.orc_pcg32_u32.restype = None
    L.orc_lane_stream.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, P(C.c_float)]
    L.orc_lane_stream.restype = None
    L.orc_math_eval.argtypes = [C.c_int, P(C.c_float), P(C.c_float), C.c_uint32, P(C.c_float), P(C.c_float)]
    L.orc_math_eval.restype = None
    L.orc_hg_sample.argtypes = [C.c_float, P(C.c_float), C.c_float, C.c_float, P(C.c_float), P(C.c_float)]
    L.orc_hg_sample.restype = None
    L.orc_hg_eval.argtypes = [C.c_float, C.c_float]
    L.orc_hg_eval.restype = C.c_float
    L.orc_square_to_cosine_hemisphere.argtypes = [C.c_float, C.c_float, P(C.c_float)]
    L.orc_square_to_uniform_sphere.argtypes = [C.c_float, C.c_float, P(C.c_float)]
    L.orc_fresnel.argtypes = [C.c_float, C.c_float, P(C.c_float)]
    L.orc_envmap_sample.argtypes = [C.c_void_p, C.c_float, C.c_float, P(C.c_float), P(C.c_float), P(C.c_float), P(C.c_float)]
    L.orc_envmap_pdf.argtypes = [C.c_void_p, P(C.c_float)]
    L.orc_envmap_pdf.restype = C.c_float
    L.orc_envmap_eval.argtypes = [C.c_void_p, P(C.c_float), P(C.c_float)]
    L.orc_rfilter_eval.argtypes = [C.c_void_p, C.c_float]
    L.orc_rfilter_eval.restype = C.c_float
    L.orc_sample_ray.argtypes = [C.c_void_p, C.c_float, C.c_float, P(C.c_float), P(C.c_float), P(C.c_float)]
    L.orc_scene_set_bio_reading.argtypes = [C.c_void_p, C.c_int]
    L.orc_scene_set_bio_reading.restype = None
    L.orc_bio_sample_interaction.argtypes = [C.c_void_p, C.c_int, P(C.c_float), P(C.c_float), C.c_float, C.c_float, C.c_uint32, C.c_float, C.c_int, P(C.c_float)]
    L.orc_bio_sample_interaction.restype = None
    L.orc_last_error.restype = C.c_char_p
    _orc = L
    return L


FP = C.POINTER(C.c_float)


def _fp(a):
    return a.ctypes.data_as(FP)


class OrcScene:
    """Oracle-side scene built from the POD description of a loaded `liverrenderer_amd.Scene`."""

    def __init__(self, scene):
        self._L = lib()
        self._scene = scene            # keeps the description's arrays alive while we copy
        self._h = C.c_void_p(self._L.orc_scene_create(C.byref(scene.desc)))
        self.film_shape = scene.film_shape()
        self.raw_channels = scene.raw_channels()

    def __del__(self):
        try:
            if self._h and self._h.value:
                self._L.orc_scene_free(self._h); self._h = C.c_void_p(None)
        except Exception:
            pass

    def set_bio_reading(self, scalar):
        """bio transport: False = the JIT variants' lane semantics (default), True = scalar_rgb (oracle/orc_bio.h)"""
        self._L.orc_scene_set_bio_reading(self._h, int(bool(scalar)))

    def bio_sample_interaction(self, medium, o, d, maxt, sample, channel, depth, jit=True):
        o = np.asarray(o, np.float32); d = np.asarray(d, np.float32); out = np.zeros(9, np.float32)
        self._L.orc_bio_sample_interaction(self._h, medium, _fp(o), _fp(d), maxt, sample, channel, depth, int(jit), _fp(out))
        return {"t": out[0], "transmittance": out[1:4].copy(), "p": out[4:7].copy(), "bio_type": int(out[7]), "distance": out[8]}

    def param_set(self, key, value):
        v = np.atleast_1d(np.asarray(value, dtype=np.float32))
        if self._L.orc_param_set(self._h, key.encode(), _fp(v), int(v.size)) != 0:
            raise KeyError(key)

    def render(self, threads=0, return_raw=False, scalar=False, **kw):
        h, w, c = self.film_shape
        img = np.empty((h, w, c), np.float32); raw = np.empty((h, w, self.raw_channels), np.float32)
        o = _lib.make_opts(kw.get("integrator"), kw.get("max_depth"), kw.get("rr_depth"), kw.get("hide_emitters"),
                           kw.get("spp", 0), kw.get("seed", 0))
        st = OrcStats()
        fn = self._L.orc_render_scalar if scalar else self._L.orc_render
        if fn(self._h, C.byref(o), threads, raw.ctypes.data, img.ctypes.data, C.byref(st)) != 0:
            raise RuntimeError(self._L.orc_last_error().decode())
        self.last_stats = {k: getattr(st, k) for k, _ in OrcStats._fields_}
        return (img, raw) if return_raw else img

    def render_samples(self, lane_begin, n, threads=0, **kw):
        out = np.empty((n, 4), np.float32)
        o = _lib.make_opts(kw.get("integrator"), kw.get("max_depth"), kw.get("rr_depth"), kw.get("hide_emitters"),
                           kw.get("spp", 0), kw.get("seed", 0))
        st = OrcStats()
        self._L.orc_render_samples(self._h, C.byref(o), int(lane_begin), int(n), threads, out.ctypes.data, C.byref(st))
        self.last_stats = {k: getattr(st, k) for k, _ in OrcStats._fields_}
        return out

    def render_backward(self, grad_image, threads=0, **kw):
        g = np.ascontiguousarray(grad_image, np.float32)
        o = _lib.make_opts(kw.get("integrator"), kw.get("max_depth"), kw.get("rr_depth"), kw.get("hide_emitters"),
                           kw.get("spp", 0), kw.get("seed", 0), grad_medium=kw.get("medium", -1))
        out = _lib.ParamGrads()
        if self._L.orc_render_backward(self._h, C.byref(o), threads, g.ctypes.data, C.byref(out)) != 0:
            raise RuntimeError("orc_render_backward failed")
        return {"sigma_t": np.array(out.d_sigma_t[:], np.float32), "albedo": np.array(out.d_albedo[:], np.float32), "g": float(out.d_g)}

    def trace(self, o, d, tmax=None, any_hit=False, brute_force=False):
        o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32)
        n = o.shape[0]
        tmax = np.full(n, np.finfo(np.float32).max, np.float32) if tmax is None else np.ascontiguousarray(tmax, np.float32)
        cols = [np.ascontiguousarray(a) for a in (o[:, 0], o[:, 1], o[:, 2], d[:, 0], d[:, 1], d[:, 2], tmax)]
        t = np.empty(n, np.float32); u = np.empty(n, np.float32); v = np.empty(n, np.float32); prim = np.empty(n, np.uint32)
        rays = _lib.RaysSoA(*[_fp(c) for c in cols])
        hits = _lib.HitsSoA(_fp(t), _fp(u), _fp(v), prim.ctypes.data_as(C.POINTER(C.c_uint32)))
        self._L.orc_trace(self._h, C.byref(rays), C.byref(hits), n, int(any_hit), int(brute_force))
        return t, u, v, prim

    def sample_ray(self, px, py):
        o = np.zeros(3, np.float32); d = np.zeros(3, np.float32); mt = C.c_float()
        self._L.orc_sample_ray(self._h, px, py, _fp(o), _fp(d), C.byref(mt))
        return o, d, mt.value

    def rfilter_eval(self, x):
        return self._L.orc_rfilter_eval(self._h, x)

    def envmap_sample(self, u1, u2, ref=(0, 0, 0)):
        r = np.asarray(ref, np.float32); d = np.zeros(3, np.float32); rgb = np.zeros(3, np.float32); pdf = C.c_float()
        self._L.orc_envmap_sample(self._h, u1, u2, _fp(r), _fp(d), C.byref(pdf), _fp(rgb))
        return d, pdf.value, rgb

    def envmap_pdf(self, d):
        d = np.asarray(d, np.float32)
        return self._L.orc_envmap_pdf(self._h, _fp(d))

    def envmap_eval(self, d):
        d = np.asarray(d, np.float32); rgb = np.zeros(3, np.float32)
        self._L.orc_envmap_eval(self._h, _fp(d), _fp(rgb))
        return rgb


def math_eval(fn, x, y=None):
    x = np.ascontiguousarray(x, np.float32); y = x if y is None else np.ascontiguousarray(y, np.float32)
    out = np.empty_like(x); out2 = np.empty_like(x)
    lib().orc_math_eval(fn, _fp(x), _fp(y), x.size, _fp(out), _fp(out2))
    return out, out2


def vae_scatter(blob, in_pos, in_dir, poly, albedo, g, ior, sigma_t, fit_scale, seed=0):
    """oracle/orc_vae.cpp: the network stage of the learned subsurface model on the CPU (same argument meaning as lrt_vae_scatter)"""
    f = lambda a, shape: np.ascontiguousarray(np.asarray(a, np.float32).reshape(shape))
    n = int(np.asarray(in_pos).reshape(-1, 3).shape[0])
    b, ip, idr, pc, al, sg = f(blob, (-1,)), f(in_pos, (n, 3)), f(in_dir, (n, 3)), f(poly, (n, 20)), f(albedo, (3,)), f(sigma_t, (3,))
    out, ab = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
    L = lib()
    L.orc_vae_scatter.argtypes = [C.POINTER(C.c_float), C.c_uint32] + [C.POINTER(C.c_float)] * 4 + [C.c_float, C.c_float, C.POINTER(C.c_float), C.c_float, C.c_uint32,
                                  C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.orc_vae_scatter.restype = None
    L.orc_vae_scatter(_fp(b), n, _fp(ip), _fp(idr), _fp(pc), _fp(al), float(g), float(ior), _fp(sg), float(fit_scale), int(seed), _fp(out), _fp(ab))
    return out, ab
