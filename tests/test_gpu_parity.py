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

REL_TOL = 1e-9  # linear colour tolerance of BASELINE.json's north_star


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


# scene, scale, depth: every BASELINE config at a size the oracle renders in seconds, plus the other scene files;
# odd scales exercise widths that are not multiples of the 8x8 tile
CASES = [
    ("scenes/chapter6.yaml", 0.5, 5), ("scenes/chapter6.yaml", 2.0, 5), ("scenes/chapter8.yaml", 0.33, 5),
    ("scenes/chapter9.yaml", 0.5, 5), ("scenes/chapter11.yaml", 0.4, 30), ("scenes/chapter11.yaml", 0.25, 5),
    ("scenes/chapter11-water.yaml", 0.5, 5), ("scenes/chapter12.yaml", 0.3, 5), ("scenes/chapter13.yaml", 0.4, 5),
    ("scenes/chapter14.yaml", 0.4, 5), (None, 0.5, 20), ("scenes/chapter15.yaml", 0.2, 5),
    ("scenes/chapter15-dragon.yaml", 0.15, 5), ("scenes/chapter16.yaml", 0.4, 5), ("scenes/bounding-box.yaml", 0.2, 5),
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
    c = o.counters()
    assert st["primary_rays"] == g.width * g.height
    assert (st["secondary_rays"], st["shadow_rays"], st["shade_hits"]) == (c["secondary"], c["shadow"], c["shade"])


@pytest.mark.parametrize("name", ["chapter6", "chapter11", "chapter13", "chapter14-spheres", "chapter16"])
def test_frame_matches_reference_render(name):
    """Rows of the reference's own images at --scale 3 (tests/golden), 8-bit exact."""
    cfg = GOLDEN[name]
    g = ra.Scene.generate_random_spheres(SCALE, SEED) if cfg["scene"] is None else ra.Scene.from_file(os.path.join(ROOT, cfg["scene"]), SCALE, SEED)
    if cfg.get("half_view_ulps"):
        g.nudge_half_view(cfg["half_view_ulps"])
    gold = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    canvas, _ = g.render(cfg["depth"], g.row_seeds(), want_f64=False)
    assert (canvas.width, canvas.height) == (int(gold["width"]), int(gold["height"]))
    assert int((canvas.rgb8[gold["rows"]] != gold["rgb8"]).any(axis=2).sum()) == 0


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
    for mode in (1, 2):
        g, _ = pair(scene, scale)
        g.set_option(ra.OPT_TRAVERSAL, mode)
        seeds = g.row_seeds()
        c, st = g.render(depth, seeds)
        frames.append((c, st))
    (a, sa), (b, sb) = frames
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
    (b) the 2-rank row split reproduces the single-rank frame bit for bit (on a 10000x400 band)."""
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
