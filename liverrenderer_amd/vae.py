"""Network stage of the fork's learned subsurface model (docs/SUBSURFACE_NOTES.md): the shape-adaptive
scatter network `ScatterModelSimShared<3, 4, 64, 64>::run` (include/mitsuba/render/scattereigen.h:249-480) on the GPU.

    model = vae.load_scatter_model(model_dir, stats_json)      # the reference's variables/*.bin + data_stats.json
    out_pos, absorbed = model.scatter(in_pos, in_dir, poly_coeffs, albedo, g, ior, sigma_t, fit_scale, seed)

The plugin around the network (src/subsurface/vaescatter.cpp) is not built: see the notes."""
import ctypes as C
import json
import os
import struct

import numpy as np

from . import _lib

# (file stem, rows, cols) in blob order (include/liverrt.h LRT_VAE_*); cols == 0: a vector
_LAYOUT = [("shared_preproc_mlp_2_shapemlp_fcn_0_weights", 64, 23), ("shared_preproc_mlp_2_shapemlp_fcn_0_biases", 64, 0),
           ("shared_preproc_mlp_2_shapemlp_fcn_1_weights", 64, 64), ("shared_preproc_mlp_2_shapemlp_fcn_1_biases", 64, 0),
           ("shared_preproc_mlp_2_shapemlp_fcn_2_weights", 64, 64), ("shared_preproc_mlp_2_shapemlp_fcn_2_biases", 64, 0),
           ("absorption_mlp_fcn_0_weights", 32, 64), ("absorption_mlp_fcn_0_biases", 32, 0),
           ("absorption_dense_kernel", 1, 32), ("absorption_dense_bias", 1, 0),
           ("scatter_decoder_fcn_fcn_0_weights", 64, 68), ("scatter_decoder_fcn_fcn_0_biases", 64, 0),
           ("scatter_decoder_fcn_fcn_1_weights", 64, 64), ("scatter_decoder_fcn_fcn_1_biases", 64, 0),
           ("scatter_decoder_fcn_fcn_2_weights", 64, 64), ("scatter_decoder_fcn_fcn_2_biases", 64, 0),
           ("scatter_dense_2_kernel", 3, 64), ("scatter_dense_2_bias", 3, 0)]
N_FLOATS = 44 + sum(r * max(c, 1) for _, r, c in _LAYOUT)


def read_bin(path):
    """NetworkHelpers::load* (scattereigen.h:44-137): int32 rank, int32 dims[rank], float32 data (row-major)."""
    b = open(path, "rb").read()
    nd = struct.unpack("<i", b[:4])[0]
    dims = struct.unpack(f"<{nd}i", b[4:4 + 4 * nd])
    a = np.frombuffer(b, dtype="<f4", offset=4 + 4 * nd)
    if a.size != int(np.prod(dims)):
        raise ValueError(f"{path}: {a.size} values for dims {dims}")
    return a.reshape(dims).astype(np.float32)


def pack_blob(model_dir, stats_json):
    """The float32 weight blob of include/liverrt.h from the reference's files (statistics: "effAlbedo", "g", "mlsPoly3", the keys
    ScatterModelSimShared's constructor reads, scattereigen.h:279-290)."""
    st = json.load(open(stats_json))
    parts = [np.array([st["effAlbedo_mean"][0], st["effAlbedo_stdinv"][0], st["g_mean"][0], st["g_stdinv"][0]], np.float32),
             np.asarray(st["mlsPoly3_mean"], np.float32), np.asarray(st["mlsPoly3_stdinv"], np.float32)]
    for stem, rows, cols in _LAYOUT:
        a = read_bin(os.path.join(model_dir, "variables", stem + ".bin"))
        want = (rows,) if cols == 0 else (rows, cols)
        if a.shape != want:
            raise ValueError(f"{stem}: shape {a.shape}, expected {want}")
        parts.append(a.reshape(-1))
    blob = np.concatenate(parts).astype(np.float32)
    assert blob.size == N_FLOATS
    return blob


class ScatterModel:
    def __init__(self, blob):
        self.blob = np.ascontiguousarray(blob, np.float32)
        L = _lib.lib()
        L.lrt_vae_model_create.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.POINTER(C.c_void_p)]
        L.lrt_vae_model_free.argtypes = [C.c_void_p]; L.lrt_vae_model_free.restype = None
        L.lrt_vae_scatter.argtypes = [C.c_void_p, C.c_uint32] + [C.POINTER(C.c_float)] * 4 + [C.c_float, C.c_float, C.POINTER(C.c_float), C.c_float,
                                      C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
        h = C.c_void_p()
        _lib.check(L.lrt_vae_model_create(self.blob.ctypes.data_as(C.POINTER(C.c_float)), self.blob.size, C.byref(h)))
        self._h, self._L = h, L

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.lrt_vae_model_free(self._h); self._h = None

    def scatter(self, in_pos, in_dir, poly_coeffs, albedo, g, ior, sigma_t, fit_scale, seed=0, device=0):
        f = lambda a, shape: np.ascontiguousarray(np.asarray(a, np.float32).reshape(shape))
        n = int(np.asarray(in_pos).reshape(-1, 3).shape[0])
        ip, idr, pc, al, sg = f(in_pos, (n, 3)), f(in_dir, (n, 3)), f(poly_coeffs, (n, 20)), f(albedo, (3,)), f(sigma_t, (3,))
        out, ab = np.zeros((n, 3), np.float32), np.zeros(n, np.float32)
        p = lambda a: a.ctypes.data_as(C.POINTER(C.c_float))
        _lib.check(self._L.lrt_vae_scatter(self._h, n, p(ip), p(idr), p(pc), p(al), float(g), float(ior), p(sg), float(fit_scale), int(seed), p(out), p(ab), int(device)))
        return out, ab


def load_scatter_model(model_dir, stats_json):
    return ScatterModel(pack_blob(model_dir, stats_json))
