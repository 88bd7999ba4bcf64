"""The HIP path (through the C ABI) against the reference's known answers and, ray by ray, against
the CPU oracle's prepare_computations."""
import numpy as np
import pytest

import common_cases as cc
import known_answers as ka
import oracle_lib as ol
import raytracer_amd as ra

pytestmark = pytest.mark.gpu


def make(text):
    return ra.Scene.from_text(text)


def test_colour_known_answers():
    cc.check_colours(make)


def test_n1_n2_table():
    cc.check_n1_n2(make)


def test_schlick():
    cc.check_schlick(make)


def test_shape_intersections_and_normals():
    cc.check_shapes(make)


def test_prepare_computations():
    cc.check_prepare(make)


def test_bounding_boxes():
    cc.check_bounding_boxes(make)


def test_csg_truth_tables_and_filter():
    cc.check_csg(make)


def test_groups():
    cc.check_groups(make)


def test_list_hits():
    cc.check_list_hits(make)


def test_triangles():
    cc.check_triangles(make)


def test_lighting():
    cc.check_lighting(make)


def test_intensity_at():
    cc.check_intensity_at(make)


def test_patterns():
    cc.check_patterns(make)


def test_lighting_and_intensity_are_bit_exact_against_the_oracle():
    rng = np.random.default_rng(5)
    for text, lights in ((ka.LIGHTING_WORLD, 3), (ka.AREA_LIGHTING_WORLD, 1), (ka.AREA_INTENSITY_WORLD, 1)):
        n = 500
        leaves = 3 if text is ka.LIGHTING_WORLD else (1 if text is ka.AREA_LIGHTING_WORLD else 2)
        rows = np.zeros((n, 12))
        rows[:, 0] = rng.integers(0, leaves, n)
        rows[:, 1] = rng.integers(0, lights, n)
        rows[:, 2:5] = rng.uniform(-3, 3, (n, 3))
        for c in (5, 8):
            v = rng.normal(size=(n, 3))
            rows[:, c:c + 3] = v / np.sqrt((v * v).sum(axis=1, keepdims=True))
        rows[:, 11] = rng.integers(0, 5, n) / 4.0
        seeds = rng.integers(0, 2 ** 63, n, dtype=np.uint64)
        a = make(text).lighting(rows, seeds)
        b = ol.OracleScene.from_text(text).lighting(rows, seeds)
        assert np.array_equal(a.view(np.uint64), b.view(np.uint64)), text
        pts = rng.uniform(-3, 3, (200, 3))
        a = make(text).intensity_at(0, pts, 11)
        b = ol.OracleScene.from_text(text).intensity_at(0, pts, 11)
        assert np.array_equal(a, b), text


def test_empty_ray_list_and_miss():
    s = make(ka.TEST_WORLD)
    assert s.colour_at(np.zeros((0, 6)), 5).shape == (0, 3)
    assert s.probe(np.zeros((0, 6))).shape == (0, 32)
    assert s.colour_at(np.array([[0, 0, -5, 0, 1, 0.0]]), 5).tolist() == [[0, 0, 0]]


def random_rays(n, seed, centre=(0, 1, 0), radius=12.0):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * radius + np.array(centre)
    target = np.array(centre) + rng.uniform(-3, 3, size=(n, 3))
    d = target - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], axis=1)


@pytest.mark.parametrize("scene,centre,radius", [
    ("scenes/chapter11.yaml", (0, 1, 0), 4.0), ("scenes/chapter13.yaml", (0, 1, 10), 12.0),
    ("scenes/chapter14.yaml", (0, 1, 2), 10.0), ("scenes/chapter16.yaml", (0, 1, 9), 9.0),
    ("scenes/chapter12.yaml", (0, 1, 0), 8.0), ("scenes/chapter15.yaml", (-0.4, 1, 7), 5.0),
])
def test_probe_is_bit_exact_against_the_oracle(root, scene, centre, radius):
    """hit leaf, t, points, normal, reflect vector, n1, n2, schlick, list length and shadow flags: every value is
    produced by IEEE add/mul/div/sqrt in the reference's order, so the device must match the CPU bit for bit."""
    g = ra.Scene.from_file(f"{root}/{scene}")
    o = ol.OracleScene.from_file(f"{root}/{scene}")
    rays = random_rays(20000, 7, centre, radius)
    a, b = g.probe(rays), o.probe(rays)
    hit = b[:, 0] >= 0
    assert hit.sum() > 2000
    assert np.array_equal(a[:, 0], b[:, 0])
    assert np.array_equal(a[:, 24], b[:, 24])                       # intersections in the world list
    refractive = np.zeros(len(rays), dtype=bool)
    mat = g.desc()
    leaf = b[hit, 0].astype(int)
    refractive[np.nonzero(hit)[0]] = mat["material"][mat["leaf_material"][leaf], 5] != 0
    cols = [c for c in range(1, 29) if c not in (21, 22, 23)]
    assert np.array_equal(a[hit][:, cols].view(np.uint64), b[hit][:, cols].view(np.uint64))
    # n1 / n2 / schlick are defined by the device only where the reference uses them (refractive hits)
    assert np.array_equal(a[refractive][:, 21:24].view(np.uint64), b[refractive][:, 21:24].view(np.uint64))
