"""Random scenes: every device walk against the reference's walk, and against the CPU oracle.

The scenes are generated as scene-file text (the grammar of src/scene/*.rs: shapes, groups with
`divide`, csgs nested through `type:`, materials with patterns, `shadow: false`, point and area
lights), so they also run through the host's loader and the flattening.  What must hold:

* traversal 0 (default), 1 (wave walk) and 2 (lane walk, with the meshes' own hierarchies) give
  the frame of traversal 3 (the reference's walk, no device-only shortcut) bit for bit, with the
  same ray counters;
* every other scene also equals the oracle's frame (8-bit image identical, f64 within 1e-9).
"""
import os

import numpy as np
import pytest

import oracle_lib as ol
import raytracer_amd as ra

pytestmark = pytest.mark.gpu
FUZZ_OBJ = os.environ.get("RT_FUZZ_OBJ", "obj/teapot.obj")  # the mesh the scenes use (obj/dragon.obj: 23 490 triangles)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REL_TOL = 1e-9


def emit(value, indent=0):
    """Block-style text in the shipped scenes' style: mappings and lists of mappings in block form,
    lists of scalars (and the rows of a transform list) in flow form."""
    pad = " " * indent
    out = []
    for key, v in value.items():
        if isinstance(v, dict):
            out.append("%s%s:" % (pad, key))
            out.append(emit(v, indent + 4))
        elif isinstance(v, list) and v and isinstance(v[0], dict):
            out.append("%s%s:" % (pad, key))
            for item in v:
                body = emit(item, indent + 6).split("\n")
                out.append("%s    - %s" % (pad, body[0].lstrip()))
                out.extend(body[1:])
        elif isinstance(v, list) and v and isinstance(v[0], list):
            out.append("%s%s:" % (pad, key))
            for row in v:
                out.append("%s    - [%s]" % (pad, ", ".join(scalar(x) for x in row)))
        elif isinstance(v, list):
            out.append("%s%s: [%s]" % (pad, key, ", ".join(scalar(x) for x in v)))
        else:
            out.append("%s%s: %s" % (pad, key, scalar(v)))
    return "\n".join(out)


def scalar(x):
    if isinstance(x, bool):
        return "true" if x else "false"
    if isinstance(x, float):
        return repr(round(x, 4))
    return str(x)


def scene_text(items):
    lines = []
    for item in items:
        body = emit(item, 2).split("\n")
        lines.append("- " + body[0].lstrip())
        lines.extend(body[1:])
    return "\n".join(lines) + "\n"


class Gen:
    def __init__(self, seed, style=0):
        """style 0: a loose arrangement; 1: crowded glass (long container lists, origins inside
        objects); 2: mostly csgs, nested deeper, with planes among their operands; 3: meshes - glass
        ones, ones that cast no shadow, in groups beside other shapes, as csg operands."""
        self.r = np.random.RandomState(seed)
        self.style = style
        self.p_glass = 0.7 if style == 1 else 0.5 if style == 3 else 0.3
        self.spread = 1.3 if style == 1 else 3.0
        self.csg_depth = 3 if style == 2 else 2

    def u(self, lo, hi):
        return float(self.r.uniform(lo, hi))

    def chance(self, p):
        return bool(self.r.uniform() < p)

    def colour(self):
        return [self.u(0, 1), self.u(0, 1), self.u(0, 1)]

    def transform(self, spread):
        ops = []
        if self.chance(0.7):
            s = self.u(0.3, 1.2)
            ops.append(["scale", s, s * self.u(0.6, 1.4), s * self.u(0.6, 1.4)] if self.chance(0.5) else ["scale", s, s, s])
        if self.chance(0.5):
            ops.append([["rotate-x", "rotate-y", "rotate-z"][self.r.randint(3)], self.u(-3.1, 3.1)])
        ops.append(["translate", self.u(-spread, spread), self.u(0, spread), self.u(-spread, spread)])
        return ops

    def pattern(self, depth):
        """Every kind of pattern/*.rs but the perturbed one; a and b are colours or, nested, patterns
        with a transformation of their own (pattern/mod.rs:83-94)."""
        kind = ["stripe", "checker", "ring", "gradient", "radial-gradient", "blend"][self.r.randint(6)]

        def side():
            return self.pattern(depth + 1) if depth < 2 and self.chance(0.3) else self.colour()
        p = {"kind": kind, "a": side(), "b": side()}
        if self.chance(0.8):
            t = [["scale", self.u(0.2, 0.8), self.u(0.2, 0.8), self.u(0.2, 0.8)]]
            if self.chance(0.4):
                t.append([["rotate-x", "rotate-y", "rotate-z"][self.r.randint(3)], self.u(-3.1, 3.1)])
            if self.chance(0.3):
                t.append(["translate", self.u(-1, 1), self.u(-1, 1), self.u(-1, 1)])
            p["transform"] = t
        return p

    def material(self, allow_glass=True):
        m = {}
        if self.chance(0.35):
            m["pattern"] = self.pattern(0)
        else:
            m["color"] = self.colour()
        if self.chance(0.5):
            m["specular"] = self.u(0, 1)
            m["shininess"] = self.u(5, 200)
        if self.chance(0.35):
            m["reflective"] = self.u(0.1, 0.9)
        if allow_glass and self.chance(self.p_glass):
            m["transparency"] = self.u(0.3, 1.0)
            m["refractive_index"] = self.u(1.0, 2.0)
            m["diffuse"] = self.u(0.05, 0.4)
        return m

    def leaf(self, key, spread):
        kind = ["sphere", "cube", "cylinder", "cone", "sphere", "cube"][self.r.randint(6)]
        if self.style == 2 and key == "type" and self.chance(0.1):
            kind = "plane"
        o = {key: kind}
        if kind in ("cylinder", "cone"):
            if self.chance(0.8):
                lo = self.u(-1.5, 0.0)
                o["min"] = lo
                o["max"] = lo + self.u(0.5, 2.0)
                o["closed"] = self.chance(0.6)
        o["material"] = self.material()
        o["transform"] = self.transform(spread)
        if self.chance(0.15):
            o["shadow"] = False
        return o

    def csg(self, key, spread, depth):
        def side():
            roll = self.r.uniform()
            if depth < self.csg_depth and roll < 0.25:
                return self.csg("type", spread, depth + 1)
            if depth < 2 and roll < 0.4:
                return self.group("type", spread * 0.7, depth + 1, in_csg=True)
            return self.leaf("type", spread)
        # the two sides overlap more often than not
        left, right = side(), side()
        if "transform" in left and "transform" in right and self.chance(0.7):
            t = left["transform"][-1]
            right["transform"][-1] = ["translate", t[1] + self.u(-0.6, 0.6), t[2] + self.u(-0.6, 0.6), t[3] + self.u(-0.6, 0.6)]
        return {key: "csg", "operation": ["union", "intersection", "difference"][self.r.randint(3)], "left": left, "right": right}

    def group(self, key, spread, depth, in_csg=False):
        n = self.r.randint(1, 5)
        children = []
        for _ in range(n):
            roll = self.r.uniform()
            if depth < 2 and roll < 0.2:
                children.append(self.group("add", spread * 0.7, depth + 1, in_csg))
            elif depth < 2 and roll < 0.35 and not in_csg:
                children.append(self.csg("add", spread * 0.7, depth + 1))
            else:
                children.append(self.leaf("add", spread))
        g = {key: "group", "children": children}
        if self.chance(0.6):
            g["transform"] = self.transform(spread)
        if self.chance(0.3):
            g["divide"] = int(self.r.randint(1, 4))
        return g

    def mesh(self, key):
        o = {key: "obj", "file": FUZZ_OBJ, "material": self.material(),
             "transform": [["scale", 0.08, 0.08, 0.08], ["rotate-x", -1.5708], ["rotate-y", self.u(-3, 3)],
                           ["translate", self.u(-2.5, 2.5), self.u(0, 1), self.u(-1, 2.5)]]}
        if self.chance(0.7):
            o["divide"] = int(self.r.randint(2, 80))
        if self.chance(0.25):
            o["shadow"] = False
        return o

    def scene(self, with_mesh, with_area):
        eye = [self.u(-1, 1), self.u(1.5, 3), -7.0]
        if self.style == 1 and self.chance(0.5):  # among (often inside) the glass objects
            eye = [self.u(-1, 1), self.u(0.2, 1.5), self.u(-1.5, 0.5)]
        # frames whose sides are no multiples of the 8x8 pixel tiles, taller than wide now and then
        width, height = int(self.r.randint(40, 200)), int(self.r.randint(30, 130))
        items = [{"add": "camera", "width": width, "height": height, "field-of-view": 1.0 if eye[2] < -6 else 1.6,
                  "from": eye, "to": [0.0, 1.0, 1.0 if eye[2] > -6 else 0.0], "up": [0.0, 1.0, 0.0]}]
        if self.chance(0.8):
            floor = {"add": "plane", "material": self.material(allow_glass=False)}
            items.append(floor)
        p_group, p_csg = (0.15, 0.85) if self.style == 2 else (0.3, 0.55)
        for _ in range(self.r.randint(2, 6)):
            roll = self.r.uniform()
            if roll < p_group:
                items.append(self.group("add", self.spread, 0))
            elif roll < p_csg:
                items.append(self.csg("add", self.spread, 0))
            else:
                items.append(self.leaf("add", self.spread))
        if self.style == 3:
            for _ in range(self.r.randint(1, 3)):
                roll = self.r.uniform()
                if roll < 0.5:
                    items.append(self.mesh("add"))
                elif roll < 0.8:
                    items.append({"add": "group", "children": [self.mesh("add"), self.leaf("add", 2.0)],
                                  "transform": self.transform(1.0)})
                else:
                    items.append({"add": "csg", "operation": ["union", "intersection", "difference"][self.r.randint(3)],
                                  "left": self.mesh("type"), "right": self.leaf("type", 2.0)})
        elif with_mesh:
            items.append({"add": "obj", "file": FUZZ_OBJ, "divide": int(self.r.randint(5, 60)),
                          "material": self.material(),
                          "transform": [["scale", 0.08, 0.08, 0.08], ["rotate-x", -1.5708],
                                        ["translate", self.u(-2, 2), 0.0, self.u(-1, 2)]]})
        for _ in range(self.r.randint(1, 3)):
            items.append({"add": "light", "at": [self.u(-8, 8), self.u(4, 10), self.u(-10, -2)],
                          "intensity": [self.u(0.3, 1.0)] * 3})
        if with_area:
            items.append({"add": "light", "corner": [self.u(-2, 2), 6.0, -4.0], "uvec": [1.5, 0.0, 0.0], "usteps": int(self.r.randint(1, 5)),
                          "vvec": [0.0, 0.0, 1.5], "vsteps": int(self.r.randint(1, 5)), "intensity": [0.5, 0.5, 0.5]})
        return scene_text(items)


def render(text, mode, seed, depth=4):
    g = ra.Scene.from_text(text, 1.0, seed, ROOT)
    seeds = g.row_seeds()
    g.set_option(ra.OPT_TRAVERSAL, mode)
    return g.render(depth, seeds)


# A longer campaign: RT_FUZZ_COUNT=3000 RT_FUZZ_BASE=50000 python -m pytest tests/test_gpu_fuzz.py -m gpu
FUZZ_COUNT = int(os.environ.get("RT_FUZZ_COUNT", "240"))
FUZZ_BASE = int(os.environ.get("RT_FUZZ_BASE", "1000"))


@pytest.mark.parametrize("seed", range(FUZZ_COUNT))
def test_random_scene_every_walk_equals_the_reference_walk(seed):
    gen = Gen(FUZZ_BASE + seed, style=(seed // 12) % 4)
    text = gen.scene(with_mesh=seed % 3 == 0, with_area=seed % 6 == 5)
    try:
        depth = seed % 9  # 0 .. 8 levels of recursion
        ref, sref = render(text, 3, seed, depth)
    except ra.RtError as e:
        # a NaN distance (the reference panics while sorting it); lists past the fast kernels' bounds take the slow path
        assert e.code == ra.RT_ERR_NAN, text
        pytest.skip("reference walk: %s" % e)
    for mode in (0, 1, 2, 4):
        got, st = render(text, mode, seed, depth)
        same = np.array_equal(got.pixels.view(np.uint64), ref.pixels.view(np.uint64))
        assert same, "traversal %d differs in %d pixels\n%s" % (mode, int((got.pixels != ref.pixels).any(axis=2).sum()), text)
        for key in ("secondary_rays", "shadow_rays", "shade_hits"):
            assert st[key] == sref[key], (mode, key)
    if seed % 12 < 6:
        o = ol.OracleScene.from_text(text, 1.0, seed, ROOT)
        want = o.render_rows(depth, o.row_seeds(), threads=16)
        assert int((ref.rgb8 != ol.to_u8(want)).any(axis=2).sum()) == 0, text
        rel = np.abs(ref.pixels - want) / np.maximum(np.abs(want), 1e-12)
        assert float(rel.max()) <= REL_TOL, text


@pytest.mark.parametrize("seed", range(24))
def test_random_scene_options_partitions_and_single_rays(seed):
    """The same random scenes through the other entry points and settings: tiny chunks with a record
    budget that forces splits, a frame put together from the rows of three ranks, and
    World::colour_at for single rays (world.rs:33-49) against the oracle."""
    gen = Gen(5000 + seed, style=seed % 4)
    text = gen.scene(with_mesh=seed % 3 == 0, with_area=seed % 2 == 1)
    g = ra.Scene.from_text(text, 1.0, seed, ROOT)
    seeds = g.row_seeds()
    try:
        base, _ = g.render(4, seeds)
    except ra.RtError as e:
        assert e.code == ra.RT_ERR_NAN, text
        pytest.skip(str(e))
    h = ra.Scene.from_text(text, 1.0, seed, ROOT)
    h.row_seeds()
    for k, v in {ra.OPT_STREAMS: 1 + seed % 3, ra.OPT_CHUNK_PIXELS: 8 * g.width, ra.OPT_RECS_PER_PIXEL: 1,
                 ra.OPT_GRID_BLOCKS: 5 + seed}.items():
        h.set_option(k, v)
    got, _ = h.render(4, seeds)
    assert np.array_equal(got.pixels.view(np.uint64), base.pixels.view(np.uint64)), text
    frame = np.zeros_like(base.pixels)
    for rank in range(3):
        part = ra.Partition(8, rank, 3)
        c, _ = g.render(4, seeds, part)
        frame[ra.local_rows(g.height, part)] = c.pixels
    assert np.array_equal(frame.view(np.uint64), base.pixels.view(np.uint64)), text
    # single rays from points around the scene towards its middle
    r = np.random.RandomState(seed)
    n = 257
    origin = r.uniform(-6, 6, size=(n, 3))
    origin[:, 1] = np.abs(origin[:, 1]) + 0.1
    target = r.uniform(-2, 2, size=(n, 3))
    direction = target - origin
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    rays = np.hstack([origin, direction])
    ray_seeds = r.randint(0, 2**62, size=n).astype(np.uint64)
    o = ol.OracleScene.from_text(text, 1.0, seed, ROOT)
    want = o.colour_at(rays, 4, ray_seeds)
    have = g.colour_at(rays, 4, ray_seeds)
    rel = np.abs(have - want) / np.maximum(np.abs(want), 1e-12)
    assert float(rel.max()) <= REL_TOL, text


def _glass(index):
    return {"color": [0.9, 0.8, 0.7], "ambient": 0.05, "diffuse": 0.2, "reflective": 0.2, "transparency": 0.9,
            "refractive_index": index}


def _ball(key, z, r=0.6, x=0.0):
    return {key: "sphere", "material": _glass(1.0 + 0.02 * (abs(z) % 7) + 0.1), "transform": [["scale", r, r, r], ["translate", x, 1.0, z]]}


def long_list_scenes():
    """Scenes whose rays build lists past the 32 entries and 6 nested csgs of the fast kernels."""
    camera = {"add": "camera", "width": 64, "height": 40, "field-of-view": 0.9, "from": [0.4, 1.3, -9.0], "to": [0.0, 1.0, 0.0],
              "up": [0.0, 1.0, 0.0]}
    light = {"add": "light", "at": [-6.0, 8.0, -8.0], "intensity": [1.0, 1.0, 1.0]}
    floor = {"add": "plane", "material": {"color": [0.4, 0.5, 0.6], "reflective": 0.3}}
    # (a) 24 overlapping glass balls in a row along the view, unioned left-deep: 23 nested csgs, 48 intersections in one list
    chain = _ball("type", -2.0)
    for k in range(1, 24):
        chain = {"type": "csg", "operation": "union", "left": chain, "right": _ball("type", -2.0 + 0.45 * k)}
    chain["add"] = chain.pop("type")
    # (b) the same balls loose, farthest first: every one of a ray's 48 crossings sorts before the running hit
    loose = [_ball("add", -2.0 + 0.45 * k) for k in reversed(range(24))]
    # (c) differences and intersections nested 9 deep around a row of balls, a group among the operands
    deep = _ball("type", 0.0, 1.5)
    for k in range(9):
        other = {"type": "group", "children": [_ball("add", -1.5 + 0.5 * j, 0.5, 0.2 * k - 0.8) for j in range(4)]} if k % 3 == 2 \
            else _ball("type", -1.0 + 0.3 * k, 0.9, 0.15 * k - 0.6)
        deep = {"type": "csg", "operation": ["difference", "union", "intersection"][k % 3], "left": deep, "right": other}
    deep["add"] = deep.pop("type")
    return {"chain": scene_text([camera, floor, chain, light]), "loose": scene_text([camera, floor] + loose + [light]),
            "deep": scene_text([camera, floor, deep, _ball("add", 3.0, 0.8, 2.0), light])}


@pytest.mark.parametrize("name", ["chain", "loose", "deep"])
def test_lists_past_the_fast_kernels_take_the_slow_path(name):
    """The reference's lists are unbounded Vecs (object/csg/mod.rs:88-106, intersection/mod.rs:62-94); the fast
    kernels keep 32 entries and 6 nested csgs per ray.  A chunk that needs more is rendered again with the
    lists in global memory: same image as the oracle, whichever walk was asked for, frames and single rays."""
    text = long_list_scenes()[name]
    o = ol.OracleScene.from_text(text, 1.0, 0, ROOT)
    want = o.render_rows(6, o.row_seeds(), threads=16)
    for mode in (0, 2, 3, 4):
        got, st = render(text, mode, 0, 6)
        if mode == 3:  # the reference's walk records every crossing; the culling walks may stay below the bounds
            assert st["big_chunks"] >= 1, (name, mode, st)
        assert int((got.rgb8 != ol.to_u8(want)).any(axis=2).sum()) == 0, (name, mode)
        rel = np.abs(got.pixels - want) / np.maximum(np.abs(want), 1e-12)
        assert float(rel.max()) <= REL_TOL, (name, mode)
    g = ra.Scene.from_text(text, 1.0, 0, ROOT)
    r = np.random.RandomState(3)
    origin = np.array([0.4, 1.3, -9.0]) + r.uniform(-0.3, 0.3, size=(200, 3))
    target = np.array([0.0, 1.0, 0.0]) + r.uniform(-1.0, 1.0, size=(200, 3))
    direction = target - origin
    direction /= np.linalg.norm(direction, axis=1, keepdims=True)
    rays = np.hstack([origin, direction])
    rel = np.abs(g.colour_at(rays, 6) - o.colour_at(rays, 6)) / np.maximum(np.abs(o.colour_at(rays, 6)), 1e-12)
    assert float(rel.max()) <= REL_TOL, name


def test_patterns_nested_ten_deep():
    """Patterns nest without a bound in the reference (pattern/mod.rs:83-94); the device evaluates them with a
    stack of 16 levels.  Ten levels of every kind inside one another, each with a transformation of its own."""
    kinds = ["blend", "stripe", "checker", "gradient", "ring", "radial-gradient"]
    pat = [0.9, 0.2, 0.1]
    for k in range(10):
        pat = {"kind": kinds[k % 6], "a": pat, "b": [0.1 * k, 1.0 - 0.08 * k, 0.5] if k % 2 else
               {"kind": kinds[(k + 3) % 6], "a": [0.2, 0.3, 0.9], "b": [0.8, 0.9, 0.1 * k], "transform": [["rotate-y", 0.3 * k]]},
               "transform": [["scale", 0.9 - 0.05 * k, 0.7, 0.8], ["rotate-z", 0.2 * k], ["translate", 0.1 * k, 0.0, 0.05]]}
    items = [{"add": "camera", "width": 72, "height": 48, "field-of-view": 0.9, "from": [0.0, 2.0, -6.0], "to": [0.0, 0.8, 0.0],
              "up": [0.0, 1.0, 0.0]},
             {"add": "plane", "material": {"pattern": pat, "specular": 0.0}},
             {"add": "sphere", "material": {"pattern": pat, "diffuse": 0.7}, "transform": [["translate", 0.0, 1.0, 0.0]]},
             {"add": "light", "at": [-5.0, 8.0, -6.0], "intensity": [1.0, 1.0, 1.0]}]
    text = scene_text(items)
    o = ol.OracleScene.from_text(text, 1.0, 0, ROOT)
    want = o.render_rows(3, o.row_seeds(), threads=16)
    got, _ = render(text, 0, 0, 3)
    assert len(np.unique(ol.to_u8(want).reshape(-1, 3), axis=0)) > 50   # the pattern really varies over the frame
    assert int((got.rgb8 != ol.to_u8(want)).any(axis=2).sum()) == 0
    rel = np.abs(got.pixels - want) / np.maximum(np.abs(want), 1e-12)
    assert float(rel.max()) <= REL_TOL
