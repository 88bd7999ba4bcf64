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
