"""raytracer_amd - MI355X rendering core for mpycroft/raytracer scenes.

Thin ctypes binding over the two native libraries built from csrc/:

* ``librt_hip.so``  - the HIP kernels behind the C ABI of ``include/rt_api.h``
* ``librt_host.so`` - the C++ host side (scene loading, flattening, writers)

The classes mirror the reference's public surface (src/lib.rs:17-26):
``Scene.from_file`` / ``Scene.generate_random_spheres`` (src/scene/mod.rs:66-85,
114-293), ``Scene.render`` (src/scene/mod.rs:92-100 -> Camera::render,
src/camera.rs:105-191) returning a ``Canvas`` (src/canvas.rs:7-60).

There is no CPU fallback: rendering needs the HIP library and a GPU, and
raises if either is missing.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_HERE)
HIP_LIB = os.path.join(_HERE, "librt_hip.so")
HOST_LIB = os.environ.get("RT_HOST_LIB", os.path.join(_HERE, "librt_host.so"))  # the variable selects a sanitizer build

RT_OK = 0
RT_ERR_INVALID, RT_ERR_HIP, RT_ERR_UNSUPPORTED, RT_ERR_CAPACITY, RT_ERR_NAN = -1, -2, -3, -4, -5

# every symbol include/rt_api.h declares
ABI_SYMBOLS = (
    "rt_init", "rt_scene_create", "rt_scene_destroy", "rt_render", "rt_colour_at", "rt_probe",
    "rt_max_csg_list", "rt_max_container_list", "rt_last_error", "rt_set_option",
    "rt_colour_at_noalloc", "rt_noalloc_max_rays", "rt_lighting", "rt_intensity_at",
    "rt_host_alloc", "rt_host_free", "rt_render_multi",
)
OPT_STREAMS, OPT_CHUNK_PIXELS, OPT_RECS_PER_PIXEL, OPT_GRID_BLOCKS, OPT_TRAVERSAL, OPT_MIN_CHUNKS, OPT_PIN_OUTPUTS = 1, 2, 3, 4, 5, 6, 7
HOST_ABI_SYMBOLS = (
    "rth_last_error", "rth_scene_from_file", "rth_scene_from_text", "rth_scene_spheres", "rth_scene_free",
    "rth_scene_info", "rth_scene_desc", "rth_camera", "rth_camera_nudge_half_view", "rth_row_seeds", "rth_upload",
    "rth_device_scene", "rth_render", "rth_colour_at", "rth_probe", "rth_write_ppm", "rth_write_png",
    "rth_lighting", "rth_intensity_at", "rth_matrix_inverse",
)


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"[{code}] {message}")
        self.code = code


class Partition(C.Structure):
    """rt_partition: rows y with (y // block_rows) % world == rank."""
    _fields_ = [("block_rows", C.c_uint32), ("rank", C.c_uint32), ("world", C.c_uint32)]


class Stats(C.Structure):
    """rt_stats"""
    _fields_ = [
        ("primary_rays", C.c_uint64), ("secondary_rays", C.c_uint64), ("shadow_rays", C.c_uint64),
        ("shade_hits", C.c_uint64), ("levels", C.c_uint32), ("chunks", C.c_uint32),
        ("ms_total", C.c_double), ("ms_trace", C.c_double), ("ms_shadow", C.c_double), ("ms_shade", C.c_double),
        ("trace_launches", C.c_uint64), ("flop_trace", C.c_uint64),
        ("flop_shadow", C.c_uint64), ("flop_light", C.c_uint64), ("fused_shadows", C.c_uint64),
        ("mesh_nodes", C.c_uint64), ("mesh_tris", C.c_uint64), ("big_chunks", C.c_uint64),
    ]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class SceneDesc(C.Structure):
    """rt_scene_desc (include/rt_api.h), field for field."""
    _u32p, _f64p = C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    _fields_ = [
        ("n_nodes", C.c_uint32), ("node_kind", _u32p), ("node_skip", _u32p), ("node_arg", _u32p), ("node_bbox", _f64p),
        ("n_leaves", C.c_uint32), ("leaf_shape", _u32p), ("leaf_inv", _f64p), ("leaf_param", _f64p),
        ("leaf_tri", _u32p), ("leaf_class", _u32p), ("leaf_material", _u32p), ("leaf_casts_shadow", C.POINTER(C.c_uint8)),
        ("n_tris", C.c_uint32), ("tri_data", _f64p),
        ("n_csgs", C.c_uint32), ("csg_op", _u32p), ("n_classes", C.c_uint32), ("csg_left_bits", C.POINTER(C.c_uint64)),
        ("n_materials", C.c_uint32), ("material", _f64p), ("material_pattern", _u32p),
        ("n_patterns", C.c_uint32), ("pat_kind", _u32p), ("pat_a", _u32p), ("pat_b", _u32p), ("pat_inv", _f64p),
        ("pat_colour", _f64p),
        ("n_lights", C.c_uint32), ("light_kind", _u32p), ("light_data", _f64p), ("light_steps", _u32p),
    ]


class CameraDesc(C.Structure):
    """rt_camera"""
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("inv", C.c_double * 12),
                ("half_width", C.c_double), ("half_height", C.c_double), ("pixel_size", C.c_double)]


_hip = None
_host = None


def hip_lib():
    """The device library. Raises when it has not been built."""
    global _hip
    if _hip is None:
        if not os.path.exists(HIP_LIB):
            raise RtError(RT_ERR_HIP, f"{HIP_LIB} is missing: build it with `make -C raytracer_amd/csrc`")
        L = C.CDLL(HIP_LIB, mode=C.RTLD_GLOBAL)
        L.rt_last_error.restype = C.c_char_p
        L.rt_init.argtypes = [C.c_int]
        L.rt_max_csg_list.restype = C.c_uint32
        L.rt_max_container_list.restype = C.c_uint32
        L.rt_render.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                C.c_int, C.c_void_p]
        L.rt_colour_at.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rt_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rt_set_option.argtypes = [C.c_void_p, C.c_int, C.c_uint64]
        L.rt_colour_at_noalloc.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p]
        L.rt_noalloc_max_rays.restype = C.c_uint32
        L.rt_render_multi.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_void_p]
        L.rt_host_alloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        L.rt_host_free.argtypes = [C.c_void_p]
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()
        if not os.path.exists(HOST_LIB):
            raise RtError(RT_ERR_INVALID, f"{HOST_LIB} is missing: build it with `make -C raytracer_amd/csrc`")
        L = C.CDLL(HOST_LIB)
        L.rth_last_error.restype = C.c_char_p
        for name in ("rth_scene_from_file", "rth_scene_from_text"):
            f = getattr(L, name)
            f.restype = C.c_void_p
            f.argtypes = [C.c_char_p, C.c_double, C.c_uint64, C.c_char_p]
        L.rth_scene_spheres.restype = C.c_void_p
        L.rth_scene_spheres.argtypes = [C.c_double, C.c_uint64]
        L.rth_scene_free.argtypes = [C.c_void_p]
        L.rth_scene_info.argtypes = [C.c_void_p, C.c_void_p]
        L.rth_scene_desc.restype = C.c_void_p
        L.rth_scene_desc.argtypes = [C.c_void_p]
        L.rth_camera.restype = C.c_void_p
        L.rth_camera.argtypes = [C.c_void_p]
        L.rth_camera_nudge_half_view.argtypes = [C.c_void_p, C.c_int]
        L.rth_row_seeds.argtypes = [C.c_void_p, C.c_void_p]
        L.rth_upload.argtypes = [C.c_void_p]
        L.rth_device_scene.restype = C.c_void_p
        L.rth_device_scene.argtypes = [C.c_void_p]
        L.rth_render.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                 C.c_void_p]
        L.rth_colour_at.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rth_probe.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
        L.rth_lighting.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.rth_intensity_at.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p]
        L.rth_matrix_inverse.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.rth_write_ppm.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        L.rth_write_png.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32]
        _host = L
    return _host


def matrix_inverse(m):
    """Matrix::invert / determinant of the host library (src/math/transformation/matrix.rs:99-123,227-245).
    Returns (inverse or None for a matrix the reference refuses to invert, determinant)."""
    a = np.ascontiguousarray(m, dtype=np.float64).reshape(4, 4)
    out = np.zeros((4, 4), dtype=np.float64)
    det = C.c_double()
    rc = host_lib().rth_matrix_inverse(_ptr(a), _ptr(out), C.byref(det))
    return (out if rc == RT_OK else None), det.value


class _HostBlock:
    """Owner of one rt_host_alloc allocation; numpy arrays made over it keep it alive."""

    def __init__(self, nbytes):
        self.ptr = C.c_void_p()
        rc = hip_lib().rt_host_alloc(C.byref(self.ptr), max(1, nbytes))
        if rc != RT_OK:
            raise RtError(rc, hip_lib().rt_last_error().decode())
        self.nbytes = nbytes

    def __del__(self):
        if getattr(self, "ptr", None) and _hip is not None:
            _hip.rt_host_free(self.ptr)
            self.ptr = None


def host_empty(shape, dtype):
    """A numpy array in page-locked host memory (rt_host_alloc) for Scene.render_into to fill."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    block = _HostBlock(n)
    buf = (C.c_ubyte * max(1, n)).from_address(block.ptr.value)
    buf._owner = block
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


def render_multi(scenes, depth, row_seeds, block_rows=8, rgb=None, rgb8=None, on_device=False):
    """rt_render_multi: one frame over several uploaded replicas of a scene (normally one per GPU), assembled into
    `rgb` / `rgb8` (numpy arrays of the whole frame, or raw device pointers with on_device).  Returns one stats
    dict per replica."""
    handles = (C.c_void_p * len(scenes))(*[s.upload() for s in scenes])
    stats = (Stats * len(scenes))()
    as_ptr = (lambda a: a) if on_device else _ptr
    rc = hip_lib().rt_render_multi(handles, len(scenes), scenes[0].camera_ptr(), depth, _ptr(row_seeds), block_rows,
                                   as_ptr(rgb), as_ptr(rgb8), 1 if on_device else 0, stats)
    if rc != RT_OK:
        raise RtError(rc, hip_lib().rt_last_error().decode())
    return [st.as_dict() for st in stats]


def init(device=0):
    """Select the HIP device of this process (one process per GPU)."""
    rc = hip_lib().rt_init(device)
    if rc != RT_OK:
        raise RtError(rc, hip_lib().rt_last_error().decode())


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def local_rows(height, part):
    """Rows a partition selects, in increasing y."""
    if part is None or part.world <= 1:
        return np.arange(height)
    y = np.arange(height)
    return y[(y // part.block_rows) % part.world == part.rank]


class Canvas:
    """Canvas { width, height, pixels } (src/canvas.rs:7-11): row-major linear RGB, plus Colour::to_u8 bytes."""

    def __init__(self, width, height, pixels, rgb8=None):
        self.width, self.height = width, height
        self.pixels = pixels  # (height, width, 3) float64
        self.rgb8 = rgb8      # (height, width, 3) uint8 from the device's to_u8

    def get_pixel(self, x, y):
        return self.pixels[y, x]

    def to_ppm(self, path):
        """Canvas::to_ppm (src/canvas.rs:50-60) written to `path`."""
        rgb8 = np.ascontiguousarray(self.rgb8)
        rc = host_lib().rth_write_ppm(path.encode(), _ptr(rgb8), self.width, self.height)
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())

    def save_png(self, path):
        """8-bit RGB PNG (src/main.rs:63-71)."""
        rgb8 = np.ascontiguousarray(self.rgb8)
        rc = host_lib().rth_write_png(path.encode(), _ptr(rgb8), self.width, self.height)
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())


class Scene:
    """Scene { camera, world } (src/scene/mod.rs:53-57), built as src/main.rs:27-45 does:
    the rng is seeded with `seed`, the scene is parsed or generated with it and the
    same rng later supplies the per-row seeds."""

    def __init__(self, handle):
        if not handle:
            raise RtError(RT_ERR_INVALID, host_lib().rth_last_error().decode())
        self.h = handle
        self._refresh()

    def _refresh(self):
        info = np.zeros(10, dtype=np.uint32)
        host_lib().rth_scene_info(self.h, _ptr(info))
        (self.width, self.height, self.n_nodes, self.n_leaves, self.n_classes, self.n_csgs, self.n_lights,
         self.n_tris, self.n_materials, self.n_patterns) = (int(v) for v in info)

    @classmethod
    def from_file(cls, path, scale=1.0, seed=0, base_dir=None):
        base = ROOT if base_dir is None else base_dir
        return cls(host_lib().rth_scene_from_file(os.fspath(path).encode(), scale, seed, base.encode()))

    @classmethod
    def from_text(cls, text, scale=1.0, seed=0, base_dir=None):
        base = ROOT if base_dir is None else base_dir
        return cls(host_lib().rth_scene_from_text(text.encode(), scale, seed, base.encode()))

    @classmethod
    def generate_random_spheres(cls, scale=1.0, seed=0):
        return cls(host_lib().rth_scene_spheres(scale, seed))

    def __del__(self):
        if getattr(self, "h", None) and _host is not None:
            _host.rth_scene_free(self.h)
            self.h = None

    def horizontal_size(self):
        return self.width

    def vertical_size(self):
        return self.height

    def nudge_half_view(self, ulps):
        host_lib().rth_camera_nudge_half_view(self.h, ulps)

    def row_seeds(self):
        """The per-row seeds Camera::render draws (src/camera.rs:159-161); advances the scene rng."""
        out = np.zeros(self.height, dtype=np.uint64)
        host_lib().rth_row_seeds(self.h, _ptr(out))
        return out

    def desc_ptr(self):
        return host_lib().rth_scene_desc(self.h)

    def camera_ptr(self):
        return host_lib().rth_camera(self.h)

    def desc(self):
        """The flattened world as numpy arrays (copies)."""
        d = C.cast(self.desc_ptr(), C.POINTER(SceneDesc)).contents
        words = (d.n_classes + 63) // 64

        def arr(ptr, n, width=1):
            if n * width == 0 or not ptr:
                a = np.zeros(0, dtype=np.ctypeslib.as_ctypes_type(ptr._type_))
            else:
                a = np.ctypeslib.as_array(ptr, shape=(n * width,)).copy()
            return a.reshape(n, width) if width > 1 else a
        return {
            "node_kind": arr(d.node_kind, d.n_nodes), "node_skip": arr(d.node_skip, d.n_nodes),
            "node_arg": arr(d.node_arg, d.n_nodes), "node_bbox": arr(d.node_bbox, d.n_nodes, 6),
            "leaf_shape": arr(d.leaf_shape, d.n_leaves), "leaf_inv": arr(d.leaf_inv, d.n_leaves, 12),
            "leaf_param": arr(d.leaf_param, d.n_leaves, 3), "leaf_tri": arr(d.leaf_tri, d.n_leaves),
            "leaf_class": arr(d.leaf_class, d.n_leaves), "leaf_material": arr(d.leaf_material, d.n_leaves),
            "leaf_casts_shadow": arr(d.leaf_casts_shadow, d.n_leaves), "tri_data": arr(d.tri_data, d.n_tris, 18),
            "csg_op": arr(d.csg_op, d.n_csgs), "csg_left_bits": arr(d.csg_left_bits, d.n_csgs * words),
            "material": arr(d.material, d.n_materials, 7), "material_pattern": arr(d.material_pattern, d.n_materials),
            "pat_kind": arr(d.pat_kind, d.n_patterns), "pat_a": arr(d.pat_a, d.n_patterns),
            "pat_b": arr(d.pat_b, d.n_patterns), "pat_inv": arr(d.pat_inv, d.n_patterns, 12),
            "pat_colour": arr(d.pat_colour, d.n_patterns, 3), "light_kind": arr(d.light_kind, d.n_lights),
            "light_data": arr(d.light_data, d.n_lights, 12), "light_steps": arr(d.light_steps, d.n_lights, 2),
            "n_classes": d.n_classes,
        }

    def camera(self):
        return C.cast(self.camera_ptr(), C.POINTER(CameraDesc)).contents

    def upload(self):
        rc = host_lib().rth_upload(self.h)
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())
        return host_lib().rth_device_scene(self.h)

    def set_option(self, option, value):
        """rt_set_option on the uploaded scene (OPT_STREAMS, OPT_CHUNK_PIXELS, OPT_RECS_PER_PIXEL, OPT_GRID_BLOCKS,
        OPT_TRAVERSAL: 0 auto, 1 wave walk, 2 lane walk, 3 the reference's walk; OPT_MIN_CHUNKS, OPT_PIN_OUTPUTS)."""
        rc = hip_lib().rt_set_option(self.upload(), option, value)
        if rc != RT_OK:
            raise RtError(rc, hip_lib().rt_last_error().decode())

    def render(self, depth=5, row_seeds=None, part=None, want_f64=True, want_u8=True):
        """Scene::render. Returns (Canvas, stats dict); with a partition the canvas holds only its rows."""
        if row_seeds is None:
            row_seeds = self.row_seeds()
        rows = len(local_rows(self.height, part))
        rgb = np.zeros((rows, self.width, 3), dtype=np.float64) if want_f64 else None
        rgb8 = np.zeros((rows, self.width, 3), dtype=np.uint8) if want_u8 else None
        stats = Stats()
        rc = host_lib().rth_render(self.h, depth, _ptr(row_seeds), C.byref(part) if part is not None else None,
                                   _ptr(rgb), _ptr(rgb8), 0, C.byref(stats))
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())
        return Canvas(self.width, rows, rgb, rgb8), stats.as_dict()

    def render_into(self, depth, row_seeds, part, rgb, rgb8):
        """Render into host arrays the caller owns (either may be None); returns the stats dict."""
        stats = Stats()
        rc = host_lib().rth_render(self.h, depth, _ptr(row_seeds), C.byref(part) if part is not None else None,
                                   _ptr(rgb), _ptr(rgb8), 0, C.byref(stats))
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())
        return stats.as_dict()

    def render_device(self, depth, row_seeds, part, out_rgb_ptr, out_rgb8_ptr):
        """Render into device buffers (raw pointers); returns the stats dict."""
        stats = Stats()
        rc = host_lib().rth_render(self.h, depth, _ptr(row_seeds), C.byref(part) if part is not None else None,
                                   out_rgb_ptr, out_rgb8_ptr, 1, C.byref(stats))
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())
        return stats.as_dict()

    def colour_at(self, rays, depth=5, seeds=None):
        """World::colour_at for explicit rays (src/world.rs:33-49)."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((rays.shape[0], 3), dtype=np.float64)
        rc = host_lib().rth_colour_at(self.h, rays.shape[0], _ptr(rays), depth, _ptr(seeds), _ptr(out))
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())
        return out

    def lighting(self, points, seeds=None):
        """Material::lighting (src/material.rs:47-87) for explicit shade points: rows of
        (leaf, light, point xyz, eye xyz, normal xyz, intensity)."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 12)
        out = np.zeros((pts.shape[0], 3), dtype=np.float64)
        sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint64)
        rc = host_lib().rth_lighting(self.h, pts.shape[0], _ptr(pts), _ptr(sd), _ptr(out))
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())
        return out

    def intensity_at(self, light, points, seed=0):
        """Lightable::intensity_at (src/light/area.rs:75-90) at each point in turn, on one generator."""
        pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(pts.shape[0], dtype=np.float64)
        rc = host_lib().rth_intensity_at(self.h, light, pts.shape[0], _ptr(pts), seed, _ptr(out))
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())
        return out

    def probe(self, rays):
        """prepare_computations of each ray's hit; 32 f64 per ray (layout in include/rt_api.h)."""
        rays = np.ascontiguousarray(rays, dtype=np.float64).reshape(-1, 6)
        out = np.zeros((rays.shape[0], 32), dtype=np.float64)
        rc = host_lib().rth_probe(self.h, rays.shape[0], _ptr(rays), _ptr(out))
        if rc != RT_OK:
            raise RtError(rc, host_lib().rth_last_error().decode())
        return out
