"""ctypes binding of the CPU oracle (oracle/liboracle.so).

Test infrastructure only: imported by tests/, by __graft_entry__.smoke() and
by the cpu_baseline leg of bench.py, never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.environ.get("ORACLE_LIB", os.path.join(ROOT, "oracle", "liboracle.so"))
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.orc_last_error.restype = C.c_char_p
        for name in ("orc_scene_from_file", "orc_scene_from_text"):
            f = getattr(L, name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_char_p, C.c_double, C.c_uint64, C.c_char_p]
        L.orc_scene_spheres.restype = C.c_void_p
        L.orc_scene_spheres.argtypes = [C.c_double, C.c_uint64]
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_scene_size.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.orc_camera_nudge_half_view.argtypes = [C.c_void_p, C.c_int]
        L.orc_row_seeds.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_counters.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_render_rows.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p]
        L.orc_render_row_list.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_render_pixels.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_colour_at.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.orc_lighting.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_intensity_at.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.orc_matrix_inverse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ray_for_pixel.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
        L.orc_to_u8.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_rng_gen_range.argtypes = [C.c_uint64, C.c_uint32, C.c_double, C.c_double, C.c_void_p]
        L.orc_rng_u64.argtypes = [C.c_uint64, C.c_uint32, C.c_void_p]
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class OracleScene:
    """Scene built the way src/main.rs:27-45 does: rng seeded, scene parsed with it."""

    def __init__(self, handle):
        if not handle:
            raise RuntimeError(lib().orc_last_error().decode())
        self.h = handle
        w, h = C.c_uint32(), C.c_uint32()
        lib().orc_scene_size(self.h, C.byref(w), C.byref(h))
        self.width, self.height = w.value, h.value

    @classmethod
    def from_file(cls, path, scale=1.0, seed=0, base_dir=None):
        base = base_dir if base_dir is not None else ROOT
        return cls(lib().orc_scene_from_file(path.encode(), scale, seed, base.encode()))

    @classmethod
    def from_text(cls, text, scale=1.0, seed=0, base_dir=None):
        base = base_dir if base_dir is not None else ROOT
        return cls(lib().orc_scene_from_text(text.encode(), scale, seed, base.encode()))

    @classmethod
    def spheres(cls, scale=1.0, seed=0):
        return cls(lib().orc_scene_spheres(scale, seed))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_scene_free(self.h)
            self.h = None

    def nudge_half_view(self, ulps):
        """Move tan(fov/2) by whole ulps (platform-libm difference of a golden render)."""
        lib().orc_camera_nudge_half_view(self.h, ulps)

    def row_seeds(self):
        out = np.zeros(self.height, dtype=np.uint64)
        lib().orc_row_seeds(self.h, _ptr(out))
        return out

    def counters(self):
        out = np.zeros(3, dtype=np.uint64)
        lib().orc_counters(self.h, _ptr(out))
        return {"secondary": int(out[0]), "shadow": int(out[1]), "shade": int(out[2])}

    def render_rows(self, depth, seeds=None, y0=0, y1=None, threads=8):
        y1 = self.height if y1 is None else y1
        out = np.zeros((y1 - y0, self.width, 3), dtype=np.float64)
        rc = lib().orc_render_rows(self.h, depth, _ptr(seeds), y0, y1, threads, _ptr(out))
        if rc != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return out

    def render_row_list(self, depth, rows, seeds=None, threads=8):
        rows = np.ascontiguousarray(rows, dtype=np.uint32)
        out = np.zeros((rows.shape[0], self.width, 3), dtype=np.float64)
        rc = lib().orc_render_row_list(self.h, depth, _ptr(seeds), rows.shape[0], _ptr(rows), threads, _ptr(out))
        if rc != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return out

    def render_pixels(self, depth, xy, threads=8):
        xy = np.ascontiguousarray(xy, dtype=np.uint32)
        out = np.zeros((xy.shape[0], 3), dtype=np.float64)
        rc = lib().orc_render_pixels(self.h, depth, xy.shape[0], _ptr(xy), threads, _ptr(out))
        if rc != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return out

    def colour_at(self, rays, depth, seeds=None):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((rays.shape[0], 3), dtype=np.float64)
        rc = lib().orc_colour_at(self.h, rays.shape[0], _ptr(rays), depth, _ptr(seeds), _ptr(out))
        if rc != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return out

    def probe(self, rays):
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((rays.shape[0], 32), dtype=np.float64)
        rc = lib().orc_probe(self.h, rays.shape[0], _ptr(rays), _ptr(out))
        if rc != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return out

    def lighting(self, points, seeds=None):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 12)
        out = np.zeros((pts.shape[0], 3), dtype=np.float64)
        sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        if lib().orc_lighting(self.h, pts.shape[0], _ptr(pts), _ptr(sd), _ptr(out)) != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return out

    def intensity_at(self, light, points, seed=0):
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(pts.shape[0], dtype=np.float64)
        if lib().orc_intensity_at(self.h, light, pts.shape[0], _ptr(pts), seed, _ptr(out)) != 0:
            raise RuntimeError(lib().orc_last_error().decode())
        return out

    def ray_for_pixel(self, x, y):
        out = np.zeros(6, dtype=np.float64)
        lib().orc_ray_for_pixel(self.h, x, y, _ptr(out))
        return out


def matrix_inverse(m):
    a = np.ascontiguousarray(m, dtype=np.float64).reshape(4, 4)
    out = np.zeros((4, 4), dtype=np.float64)
    det = C.c_double()
    rc = lib().orc_matrix_inverse(_ptr(a), _ptr(out), C.byref(det))
    return (out if rc == 0 else None), det.value


def to_u8(rgb):
    rgb = np.ascontiguousarray(rgb, dtype=np.float64)
    out = np.zeros(rgb.shape, dtype=np.uint8)
    lib().orc_to_u8(_ptr(rgb), rgb.size, _ptr(out))
    return out


def rng_gen_range(seed, n, lo=0.0, hi=1.0):
    out = np.zeros(n, dtype=np.float64)
    lib().orc_rng_gen_range(seed, n, lo, hi, _ptr(out))
    return out


def rng_u64(seed, n):
    out = np.zeros(n, dtype=np.uint64)
    lib().orc_rng_u64(seed, n, _ptr(out))
    return out
