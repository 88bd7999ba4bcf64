"""Frame-level parity of the HIP path against the CPU oracle and the reference's own renders."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as ol
import raytracer_amd as ra
from golden_scenes import GOLDEN, SCALE, SEED

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REL_TOL = 1e-9  # tolerance on linear colour (relative): the 8-bit image must be identical, f64 may differ by rounding order only


def pair(scene, scale, seed=0):
    if scene is None:
        return ra.Scene.generate_random_spheres(scale, seed), ol.OracleScene.spheres(scale, seed)
    return ra.Scene.from_file(os.path.join(ROOT, scene), scale, seed), ol.OracleScene.from_file(os.path.join(ROOT, scene), scale, seed)


def compare(canvas, ref):
    ref8 = ol.to_u8(ref)
    bad8 = int((canvas.rgb8 != ref8).any(axis=2).sum())
    err = np.abs(canvas.pixels - ref)
    rel = err / np.maximum(np.abs(ref), 1e-12)
    return bad8, float(rel.max())


# scene, scale, depth: every benchmarked configuration at a size the oracle renders in seconds, plus the other scene files;
# odd scales exercise widths that are not multiples of the 8x8 tile
CASES = [
    ("scenes/chapter6.yaml", 0.5, 5), ("scenes/chapter6.yaml", 2.0, 5), ("scenes/chapter8.yaml", 0.33, 5),
    ("scenes/chapter9.yaml", 0.5, 5), ("scenes/chapter11.yaml", 0.4, 30), ("scenes/chapter11.yaml", 0.25, 5),
    ("scenes/chapter11-water.yaml", 0.5, 5), ("scenes/chapter12.yaml", 0.3, 5), ("scenes/chapter13.yaml", 0.4, 5),
    ("scenes/chapter14.yaml", 0.4, 5), (None, 0.5, 20), ("scenes/chapter15.yaml", 0.2, 5),
    ("scenes/chapter15-dragon.yaml", 0.15, 5), ("scenes/chapter16.yaml", 0.4, 5), ("scenes/bounding-box.yaml", 0.2, 5),
    # the 10x10 jittered area light: 400 rng draws per shade point in the reference's per-row order
    # (light/area.rs:48-91, material.rs:57-64, camera.rs:143-161); f64 must come out bit for bit
    ("scenes/area-light.yaml", 0.5, 5), ("scenes/area-light.yaml", 1.0, 5),
]


@pytest.mark.parametrize("scene,scale,depth", CASES)
def test_frame_matches_oracle(scene, scale, depth):
    g, o = pair(scene, scale)
    seeds = g.row_seeds()
    assert (seeds == o.row_seeds()).all()
    canvas, st = g.render(depth, seeds)
    ref = o.render_rows(depth, seeds, threads=16)
    bad8, rel = compare(canvas, ref)
    assert bad8 == 0, f"{bad8} differing 8-bit pixels"
    assert rel <= REL_TOL, rel
    if scene == "scenes/area-light.yaml":
        # ray paths do not depend on the rng and every draw is taken in the reference's order, so the
        # only differences left are those of device pow() against glibc's
        assert rel <= 1e-13, rel
    c = o.counters()
    assert st["primary_rays"] == g.width * g.height
    assert (st["secondary_rays"], st["shadow_rays"], st["shade_hits"]) == (c["secondary"], c["shadow"], c["shade"])


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_frame_matches_reference_render(name):
    """Rows of the reference's own images at --scale 3 (tests/golden), 8-bit exact; bounding-box
    within its counted z-fighting band (golden_scenes.py)."""
    cfg = GOLDEN[name]
    g = ra.Scene.generate_random_spheres(SCALE, SEED) if cfg["scene"] is None else ra.Scene.from_file(os.path.join(ROOT, cfg["scene"]), SCALE, SEED)
    if cfg.get("half_view_ulps"):
        g.nudge_half_view(cfg["half_view_ulps"])
    gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    canvas, _ = g.render(cfg["depth"], g.row_seeds(), want_f64=False)
    assert (canvas.width, canvas.height) == (int(gold["width"]), int(gold["height"]))
    assert int((canvas.rgb8[gold["rows"]] != gold["rgb8"]).any(axis=2).sum()) <= cfg.get("max_mismatch", 0)


def test_row_partition_renders_the_same_pixels():
    g, _ = pair("scenes/chapter11.yaml", 0.3)
    seeds = g.row_seeds()
    whole, _ = g.render(12, seeds)
    for world in (2, 3):
        frame = np.zeros_like(whole.pixels)
        for rank in range(world):
            part = ra.Partition(8, rank, world)
            c, _ = g.render(12, seeds, part)
            frame[ra.local_rows(g.height, part)] = c.pixels
        assert np.array_equal(frame.view(np.uint64), whole.pixels.view(np.uint64))


@pytest.mark.parametrize("scene,scale,depth", [("scenes/chapter11.yaml", 0.3, 12), ("scenes/area-light.yaml", 0.25, 5),
                                               ("scenes/chapter11.yaml", 0.37, 5)])
def test_render_multi_assembles_the_frame_behind_the_c_abi(scene, scale, depth):
    """rt_render_multi with 2, 3 and 5 replicas (all on this one GPU): the assembled Canvas equals the single
    render bit for bit, in host memory and in device memory, and the replicas' ray counts add up."""
    import torch
    g, _ = pair(scene, scale)
    seeds = g.row_seeds()
    whole, st = g.render(depth, seeds)
    for world, block_rows in ((2, 8), (3, 8), (5, 16)):
        replicas = [pair(scene, scale)[0] for _ in range(world)]
        rgb, rgb8 = np.full(whole.pixels.shape, -1.0), np.full(whole.pixels.shape, 9, dtype=np.uint8)
        stats = ra.render_multi(replicas, depth, seeds, block_rows, rgb, rgb8)
        assert np.array_equal(rgb.view(np.uint64), whole.pixels.view(np.uint64)), world
        assert np.array_equal(rgb8, whole.rgb8), world
        for key in ("primary_rays", "secondary_rays", "shadow_rays", "shade_hits"):
            assert sum(s[key] for s in stats) == st[key], (world, key)
        d8 = torch.full(whole.pixels.shape, 9, dtype=torch.uint8, device="cuda")
        d64 = torch.full(whole.pixels.shape, -1.0, dtype=torch.float64, device="cuda")
        ra.render_multi(replicas, depth, seeds, block_rows, d64.data_ptr(), d8.data_ptr(), on_device=True)
        torch.cuda.synchronize()
        assert np.array_equal(d8.cpu().numpy(), whole.rgb8), world
        assert np.array_equal(d64.cpu().numpy().view(np.uint64), whole.pixels.view(np.uint64)), world
    with pytest.raises(ra.RtError):
        ra.render_multi([g, g], depth, seeds, 8, rgb, rgb8)
    with pytest.raises(ra.RtError):
        ra.render_multi(replicas, depth, seeds, 12, rgb, rgb8)


def test_host_outputs_arrive_whole_whatever_the_buffer():
    """Host outputs are filled chunk by chunk while later chunks render when the buffer is page-locked (by the
    caller, or by rt_render for the call) and by one pageable copy otherwise: the same bytes every way, also
    when chunks overflow their record arrays and are rendered again."""
    g, _ = pair("scenes/chapter11.yaml", 0.5)
    seeds = g.row_seeds()
    base, _ = g.render(10, seeds)
    shape = base.pixels.shape
    for opts in ({}, {ra.OPT_PIN_OUTPUTS: 0}, {ra.OPT_CHUNK_PIXELS: 16 * g.width, ra.OPT_RECS_PER_PIXEL: 1},
                 {ra.OPT_STREAMS: 1, ra.OPT_CHUNK_PIXELS: 8 * g.width}):
        h, _ = pair("scenes/chapter11.yaml", 0.5)
        for k, v in opts.items():
            h.set_option(k, v)
        for locked in (False, True):
            if locked:
                rgb, rgb8 = ra.host_empty(shape, np.float64), ra.host_empty(shape, np.uint8)
            else:
                rgb, rgb8 = np.empty(shape), np.empty(shape, dtype=np.uint8)
            rgb[:] = -1.0
            rgb8[:] = 7
            h.render_into(10, seeds, None, rgb, rgb8)
            assert np.array_equal(rgb.view(np.uint64), base.pixels.view(np.uint64)), (opts, locked)
            assert np.array_equal(rgb8, base.rgb8), (opts, locked)
            rgb8[:] = 7
            h.render_into(10, seeds, None, None, rgb8)
            assert np.array_equal(rgb8, base.rgb8), (opts, locked)


def test_render_is_deterministic_and_depth_zero_has_no_secondary_rays():
    g, o = pair("scenes/chapter11.yaml", 0.3)
    a, _ = g.render(7, None)
    b, st = g.render(7, None)
    assert np.array_equal(a.pixels.view(np.uint64), b.pixels.view(np.uint64))
    c, st0 = g.render(0, None)
    assert st0["secondary_rays"] == 0 and st0["levels"] == 1
    ref = o.render_rows(0, None, threads=16)
    assert compare(c, ref)[0] == 0


@pytest.mark.parametrize("scene,scale,depth", [
    ("scenes/chapter11.yaml", 0.3, 30), ("scenes/chapter16.yaml", 0.3, 5), ("scenes/chapter13.yaml", 0.3, 5),
    ("scenes/chapter15.yaml", 0.2, 5), ("scenes/bounding-box.yaml", 0.15, 5), ("scenes/area-light.yaml", 0.25, 5)])
def test_both_tree_walks_give_the_same_image(scene, scale, depth):
    """The wave-uniform walk and the lane-by-lane walk (meshes) are two schedules of the same
    arithmetic: bit-identical f64 frames and ray counts, whichever the tree size would select.
    The lane-by-lane walk goes through a mesh by its own hierarchy and so counts fewer tests:
    the work counters only have to agree for scenes without a mesh."""
    frames = []
    for mode in (1, 2, 4):
        g, _ = pair(scene, scale)
        g.set_option(ra.OPT_TRAVERSAL, mode)
        seeds = g.row_seeds()
        c, st = g.render(depth, seeds)
        frames.append((c, st))
    (a, sa), (b, sb), (h, sh) = frames
    # the hybrid walk: the wave's walk outside the meshes, the lane walk's inside them
    assert np.array_equal(h.pixels.view(np.uint64), a.pixels.view(np.uint64))
    for key in ("secondary_rays", "shadow_rays", "shade_hits"):
        assert sh[key] == sa[key], key
    assert np.array_equal(a.pixels.view(np.uint64), b.pixels.view(np.uint64))
    assert np.array_equal(a.rgb8, b.rgb8)
    has_mesh = scene in ("scenes/chapter15.yaml", "scenes/bounding-box.yaml")
    for key in ("secondary_rays", "shadow_rays", "shade_hits") + (() if has_mesh else ("flop_trace", "flop_shadow")):
        assert sa[key] == sb[key], key
    if has_mesh:
        assert sb["flop_trace"] < sa["flop_trace"] and sb["flop_shadow"] < sa["flop_shadow"]


@pytest.mark.parametrize("scene,depth", [("scenes/chapter11.yaml", 30), ("scenes/area-light.yaml", 5)])
def test_tuning_options_do_not_change_the_image(scene, depth):
    """Tiny chunks, a record budget that forces chunks to be split and the budget to grow, one or
    two streams, a small grid: every setting must give the default's image bit for bit."""
    g, _ = pair(scene, 0.3)
    seeds = g.row_seeds()
    base, st0 = g.render(depth, seeds)
    for opts in ({ra.OPT_STREAMS: 1}, {ra.OPT_STREAMS: 2, ra.OPT_CHUNK_PIXELS: 8 * g.width},
                 {ra.OPT_CHUNK_PIXELS: 24 * g.width, ra.OPT_RECS_PER_PIXEL: 1, ra.OPT_GRID_BLOCKS: 7},
                 {ra.OPT_STREAMS: 1, ra.OPT_CHUNK_PIXELS: 8 * g.width, ra.OPT_RECS_PER_PIXEL: 1}):
        h, _ = pair(scene, 0.3)
        for k, v in opts.items():
            h.set_option(k, v)
        c, st = h.render(depth, seeds)
        assert np.array_equal(c.pixels.view(np.uint64), base.pixels.view(np.uint64)), opts
        assert np.array_equal(c.rgb8, base.rgb8)
        assert (st["secondary_rays"], st["shadow_rays"], st["shade_hits"]) == (
            st0["secondary_rays"], st0["shadow_rays"], st0["shade_hits"]), opts
        if ra.OPT_CHUNK_PIXELS in opts:
            assert st["chunks"] > 1
    with pytest.raises(ra.RtError):
        g.set_option(ra.OPT_STREAMS, 5)


def test_full_size_headline_frame_properties():
    """chapter11 at 10000x5000, depth 30: far too large for the oracle as a whole, so check
    (a) 24 evenly spaced rows against the oracle, 8-bit exact and within the colour tolerance,
    (b) the 2-rank row split reproduces the single-rank frame bit for bit,
    (c) the reference's walk (traversal 3) gives the same 50 M pixels, f64 for f64."""
    g, o = pair("scenes/chapter11.yaml", 10.0)
    seeds = g.row_seeds()
    canvas, st = g.render(30, seeds, want_f64=True)
    assert st["primary_rays"] == 50_000_000 and st["levels"] == 31
    rows = np.unique(np.linspace(0, g.height - 1, 24).astype(np.uint32))
    ref = o.render_row_list(30, rows, seeds, threads=16)
    assert int((canvas.rgb8[rows] != ol.to_u8(ref)).any(axis=2).sum()) == 0
    rel = np.abs(canvas.pixels[rows] - ref) / np.maximum(np.abs(ref), 1e-12)
    assert rel.max() <= REL_TOL
    part_frames = []
    for rank in range(2):
        part = ra.Partition(8, rank, 2)
        c, _ = g.render(30, seeds, part, want_f64=False)
        part_frames.append((ra.local_rows(g.height, part), c.rgb8))
    for rows_r, img in part_frames:
        assert np.array_equal(img, canvas.rgb8[rows_r])
    # (c) the whole frame again with the reference's walk (no device-only shortcut): every f64 equal
    g.set_option(ra.OPT_TRAVERSAL, 3)
    plain, st3 = g.render(30, seeds, want_f64=True)
    assert np.array_equal(plain.pixels.view(np.uint64), canvas.pixels.view(np.uint64))
    for key in ("secondary_rays", "shadow_rays", "shade_hits"):
        assert st3[key] == st[key], key


def test_error_paths():
    g, _ = pair("scenes/chapter6.yaml", 0.2)
    scene = g.upload()
    lib = ra.hip_lib()
    cam = g.camera()
    bad = ra.Partition(5, 0, 2)  # block_rows not a multiple of 8
    assert lib.rt_render(scene, C.byref(cam), 5, None, C.byref(bad), None, None, 0, None) == ra.RT_ERR_INVALID
    assert b"partition" in lib.rt_last_error()
    assert lib.rt_render(None, C.byref(cam), 5, None, None, None, None, 0, None) == ra.RT_ERR_INVALID
    # NaN intersection distance: a zero-length direction makes the sphere's quadratic 0/0 (the reference panics)
    out = np.zeros(3)
    ray = np.array([0, 0, -5, 0, 0, 0], dtype=np.float64)
    two = ra.Scene.from_text(open(os.path.join(ROOT, "scenes", "chapter8.yaml")).read())
    with pytest.raises(ra.RtError) as e:
        two.colour_at(ray, 5)
    assert e.value.code == ra.RT_ERR_NAN
    with pytest.raises(RuntimeError):
        ol.OracleScene.from_file(os.path.join(ROOT, "scenes", "chapter8.yaml")).colour_at(ray, 5)


def test_a_failing_frame_leaves_the_scene_usable():
    """A camera with a NaN in its matrix makes every distance NaN (the reference panics while sorting them):
    rt_render reports it after all its streams have drained, and the same scene then renders as before."""
    g, _ = pair("scenes/chapter11.yaml", 0.4)
    seeds = g.row_seeds()
    g.set_option(ra.OPT_CHUNK_PIXELS, 16 * g.width)   # several chunks on two streams
    good, _ = g.render(6, seeds)
    lib = ra.hip_lib()
    cam = ra.CameraDesc.from_buffer_copy(g.camera())
    cam.inv[5] = float("nan")
    rgb = np.zeros((g.height, g.width, 3))
    rc = lib.rt_render(g.upload(), C.byref(cam), 6, None, None, rgb.ctypes.data_as(C.c_void_p), None, 0, None)
    assert rc == ra.RT_ERR_NAN, lib.rt_last_error().decode()
    again, _ = g.render(6, seeds)
    assert np.array_equal(again.pixels.view(np.uint64), good.pixels.view(np.uint64))


def _create_from_arrays(d):
    """rt_scene_create straight from (possibly edited) rt_scene_desc arrays; returns (handle, keepalive)."""
    keep = {k: np.ascontiguousarray(v) for k, v in d.items() if isinstance(v, np.ndarray)}
    sd = ra.SceneDesc()

    def p(name, ctype):
        a = keep[name]
        return a.ctypes.data_as(C.POINTER(ctype)) if a.size else C.POINTER(ctype)()
    sd.n_nodes = len(keep["node_kind"])
    sd.node_kind, sd.node_skip, sd.node_arg = p("node_kind", C.c_uint32), p("node_skip", C.c_uint32), p("node_arg", C.c_uint32)
    sd.node_bbox = p("node_bbox", C.c_double)
    sd.n_leaves = len(keep["leaf_shape"])
    sd.leaf_shape, sd.leaf_inv, sd.leaf_param = p("leaf_shape", C.c_uint32), p("leaf_inv", C.c_double), p("leaf_param", C.c_double)
    sd.leaf_tri, sd.leaf_class, sd.leaf_material = p("leaf_tri", C.c_uint32), p("leaf_class", C.c_uint32), p("leaf_material", C.c_uint32)
    sd.leaf_casts_shadow = p("leaf_casts_shadow", C.c_uint8)
    sd.n_tris, sd.tri_data = len(keep["tri_data"]), p("tri_data", C.c_double)
    sd.n_csgs, sd.csg_op = len(keep["csg_op"]), p("csg_op", C.c_uint32)
    sd.n_classes, sd.csg_left_bits = int(d["n_classes"]), p("csg_left_bits", C.c_uint64)
    sd.n_materials, sd.material, sd.material_pattern = len(keep["material"]), p("material", C.c_double), p("material_pattern", C.c_uint32)
    sd.n_patterns = len(keep["pat_kind"])
    sd.pat_kind, sd.pat_a, sd.pat_b = p("pat_kind", C.c_uint32), p("pat_a", C.c_uint32), p("pat_b", C.c_uint32)
    sd.pat_inv, sd.pat_colour = p("pat_inv", C.c_double), p("pat_colour", C.c_double)
    sd.n_lights = len(keep["light_kind"])
    sd.light_kind, sd.light_data, sd.light_steps = p("light_kind", C.c_uint32), p("light_data", C.c_double), p("light_steps", C.c_uint32)
    h = C.c_void_p()
    rc = ra.hip_lib().rt_scene_create(C.byref(sd), C.byref(h))
    assert rc == 0, ra.hip_lib().rt_last_error().decode()
    return h, (keep, sd)


@pytest.mark.parametrize("scene,scale,root_depth", [("scenes/chapter15.yaml", 0.2, 0), ("scenes/bounding-box.yaml", 0.15, 2)])
def test_mesh_hierarchy_respects_the_stored_boxes(scene, scale, root_depth):
    """A group's stored box decides whether its children are tested at all (group/mod.rs:30-32).
    The walk through a mesh's own hierarchy must drop exactly the intersections the reference's
    walk never makes.  With the reference's boxes that only happens on a box's rim; here every
    box inside a mesh is shrunk on purpose, so triangles stick out of their groups' boxes by the
    thousand: the frame must still equal, bit for bit, the one made with the reference's walk
    (traversal 3: no device-only shortcut), and differ from the frame of the untouched scene."""
    g, _ = pair(scene, scale)
    d = g.desc()
    lib = ra.hip_lib()
    cam = g.camera_ptr()
    W, H = g.width, g.height

    def frame(desc, mode):
        h, keep = _create_from_arrays(desc)
        assert lib.rt_set_option(h, ra.OPT_TRAVERSAL, mode) == 0
        rgb = np.zeros((H, W, 3))
        st = ra.Stats()
        rc = lib.rt_render(h, cam, 5, None, None, rgb.ctypes.data_as(C.c_void_p), None, 0, C.byref(st))
        assert rc == 0, lib.rt_last_error().decode()
        lib.rt_scene_destroy(h)
        return rgb, st.as_dict()

    a0, _ = frame(d, 3)
    b0, _ = frame(d, 2)
    assert np.array_equal(a0.view(np.uint64), b0.view(np.uint64))
    assert np.array_equal(a0.view(np.uint64), frame(d, 4)[0].view(np.uint64))
    # shrink the box of every group below a mesh's root
    kind, skip = d["node_kind"], d["node_skip"]
    depth_of = np.zeros(len(kind), dtype=np.int64)
    open_ = []
    for i in range(len(kind)):
        while open_ and skip[open_[-1]] <= i:
            open_.pop()
        depth_of[i] = len(open_)
        if kind[i] != 0:
            open_.append(i)
    bb = d["node_bbox"].copy()
    inner = (kind == 1) & (depth_of > root_depth) & np.isfinite(bb).all(axis=1)
    assert inner.sum() > 10
    with np.errstate(invalid="ignore"):
        c = 0.5 * (bb[:, :3] + bb[:, 3:])
        e = 0.5 * (bb[:, 3:] - bb[:, :3])
    bb[inner, :3] = (c - 0.7 * e)[inner]
    bb[inner, 3:] = (c + 0.7 * e)[inner]
    d2 = dict(d)
    d2["node_bbox"] = bb
    a1, sa = frame(d2, 3)
    b1, sb = frame(d2, 2)
    assert np.array_equal(a1.view(np.uint64), b1.view(np.uint64))
    assert sa["shade_hits"] == sb["shade_hits"] and sa["secondary_rays"] == sb["secondary_rays"]
    assert np.array_equal(a1.view(np.uint64), frame(d2, 4)[0].view(np.uint64))
    assert not np.array_equal(a0, a1)


@pytest.mark.parametrize("scene,scale,depth", [
    ("scenes/chapter11.yaml", 1.0, 30), ("scenes/chapter11-water.yaml", 0.5, 5), ("scenes/chapter13.yaml", 0.5, 5),
    ("scenes/chapter14.yaml", 0.5, 5), (None, 0.5, 20), ("scenes/chapter15-dragon.yaml", 0.3, 5),
    ("scenes/chapter16.yaml", 1.0, 5), ("scenes/bounding-box.yaml", 0.3, 5), ("scenes/area-light.yaml", 0.3, 5)])
def test_shortcuts_do_not_change_the_image(scene, scale, depth):
    """Everything the device adds to the reference's walk (bounding-sphere and distance culling,
    the meshes' own hierarchies, shadow rays that stop at the light or at their first hit or are
    not cast at all) must leave every f64 of the frame as the reference's walk computes it, at
    sizes beyond what the CPU oracle renders in a test."""
    frames = []
    for mode in (3, 0):
        g, _ = pair(scene, scale)
        g.set_option(ra.OPT_TRAVERSAL, mode)
        seeds = g.row_seeds()
        frames.append(g.render(depth, seeds))
    (a, sa), (b, sb) = frames
    assert np.array_equal(a.pixels.view(np.uint64), b.pixels.view(np.uint64))
    assert np.array_equal(a.rgb8, b.rgb8)
    for key in ("secondary_rays", "shadow_rays", "shade_hits"):
        assert sa[key] == sb[key], key


def test_coincident_triangles_resolve_like_the_reference_list():
    """Every triangle of the teapot is doubled: leaf 2k+1 becomes a copy of leaf 2k with another
    material, so each hit is an exact tie in t between two leaves.  The reference's stable sort
    keeps the earlier leaf (world.rs:90-106, intersection/list.rs:24-50); the walk through the
    mesh's own hierarchy meets the two in any order and must pick the same one."""
    g, _ = pair("scenes/chapter15.yaml", 0.2)
    d = g.desc()
    lib = ra.hip_lib()
    tri_leaves = np.nonzero(d["leaf_shape"] == 5)[0]
    pairs = len(tri_leaves) // 2
    even, odd = tri_leaves[0:2 * pairs:2], tri_leaves[1:2 * pairs:2]
    d2 = dict(d)
    tri = d["tri_data"].copy()
    tri[d["leaf_tri"][odd]] = tri[d["leaf_tri"][even]]
    d2["tri_data"] = tri
    # a second material: the first triangle's with a solid green pattern
    m0 = d["leaf_material"][even[0]]
    d2["pat_kind"] = np.append(d["pat_kind"], np.uint32(0))
    d2["pat_a"] = np.append(d["pat_a"], np.uint32(0xFFFFFFFF))
    d2["pat_b"] = np.append(d["pat_b"], np.uint32(0xFFFFFFFF))
    d2["pat_inv"] = np.vstack([d["pat_inv"], np.array([[1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0]], dtype=np.float64)])
    d2["pat_colour"] = np.vstack([d["pat_colour"], np.array([[0.0, 1.0, 0.0]])])
    d2["material"] = np.vstack([d["material"], d["material"][m0:m0 + 1]])
    d2["material_pattern"] = np.append(d["material_pattern"], np.uint32(len(d["pat_kind"])))
    lm = d["leaf_material"].copy()
    lm[odd] = len(d["material"])
    d2["leaf_material"] = lm
    frames = []
    for mode in (3, 2, 1, 4):
        h, keep = _create_from_arrays(d2)
        assert lib.rt_set_option(h, ra.OPT_TRAVERSAL, mode) == 0
        rgb = np.zeros((g.height, g.width, 3))
        rc = lib.rt_render(h, g.camera_ptr(), 5, None, None, rgb.ctypes.data_as(C.c_void_p), None, 0, None)
        assert rc == 0, lib.rt_last_error().decode()
        lib.rt_scene_destroy(h)
        frames.append(rgb)
    assert np.array_equal(frames[0].view(np.uint64), frames[1].view(np.uint64))
    assert np.array_equal(frames[0].view(np.uint64), frames[2].view(np.uint64))
    # which leaf wins shows: with the materials of each pair exchanged the frame is another one
    lm2 = d["leaf_material"].copy()
    lm2[even] = len(d["material"])
    d3 = dict(d2)
    d3["leaf_material"] = lm2
    h, keep = _create_from_arrays(d3)
    rgb = np.zeros((g.height, g.width, 3))
    assert lib.rt_render(h, g.camera_ptr(), 5, None, None, rgb.ctypes.data_as(C.c_void_p), None, 0, None) == 0
    lib.rt_scene_destroy(h)
    assert (rgb != frames[0]).any(axis=2).mean() > 0.01


def test_area_light_rows_keep_their_generators_under_a_partition():
    """Each image row has its own generator (camera.rs:143-161) and every shade point's state is
    reached by jumping from the row's seed: rendering the rows in interleaved blocks on two or
    three ranks must reproduce the whole frame bit for bit."""
    g, _ = pair("scenes/area-light.yaml", 0.25)
    seeds = g.row_seeds()
    whole, _ = g.render(5, seeds)
    for world in (2, 3):
        frame = np.zeros_like(whole.pixels)
        for rank in range(world):
            part = ra.Partition(8, rank, world)
            c, _ = g.render(5, seeds, part)
            frame[ra.local_rows(g.height, part)] = c.pixels
        assert np.array_equal(frame.view(np.uint64), whole.pixels.view(np.uint64))
