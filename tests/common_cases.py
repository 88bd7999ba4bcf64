"""Backend-independent bodies of the known-answer checks.  `make(text)` returns an
object with colour_at(rays, depth) and probe(rays) - the CPU oracle or the HIP path."""
import numpy as np

import known_answers as ka


def check_colours(make):
    for name, text, ray, depth, expected in ka.COLOUR_CASES:
        got = make(text).colour_at(np.array([ray], dtype=np.float64), depth)[0]
        assert np.allclose(got, expected, atol=1e-5, rtol=0), (name, got, expected)


def check_n1_n2(make):
    s = make(ka.N1N2_SCENE)
    for t, n1, n2 in ka.N1N2_TABLE:
        z0 = -4.0 + t - 0.125  # start just before the intersection at z = -4 + t
        p = s.probe(np.array([[0, 0, z0, 0, 0, 1]], dtype=np.float64))[0]
        assert abs(p[1] - 0.125) < 1e-12, (t, p[1])
        assert p[21] == n1 and p[22] == n2, (t, p[21], p[22], n1, n2)
        assert p[24] == 6


def check_schlick(make):
    s = make(ka.GLASS_BALL)
    for name, ray, expected, tol in ka.SCHLICK_CASES:
        p = s.probe(np.array([ray], dtype=np.float64))[0]
        assert abs(p[23] - expected) <= tol, (name, p[23])


def check_shapes(make):
    cache = {}
    for name, text, ray, t, normal in ka.SHAPE_CASES:
        s = cache.setdefault(text, make(text))
        p = s.probe(np.array([ray], dtype=np.float64))[0]
        if t is None:
            assert p[0] == -1, (name, p[:2])
        else:
            assert p[0] == 0 and abs(p[1] - t) < 1e-5, (name, p[:2], t)
        if normal is not None:
            assert np.allclose(p[14:17], normal, atol=1e-12), (name, p[14:17])
    for text, cases in ((ka.CLOSED_CYL, ka.CAPPED_CYL_CASES), (ka.TRUNCATED_CYL, ka.TRUNCATED_CYL_CASES),
                        (ka.CLOSED_CONE, ka.CAPPED_CONE_CASES)):
        s = make(text)
        for ray, count in cases:
            p = s.probe(np.array([ray], dtype=np.float64))[0]
            assert p[24] == count, (text, ray, p[24], count)


def check_prepare(make):
    # intersection/mod.rs:150-223
    s = make(ka.SINGLE % "sphere")
    p = s.probe(np.array([[0, 0, -5, 0, 0, 1]], dtype=np.float64))[0]
    assert p[1] == 4.0 and list(p[2:5]) == [0, 0, -1] and list(p[11:14]) == [0, 0, -1] and list(p[14:17]) == [0, 0, -1]
    assert p[17] == 0
    assert p[7] < -2.220446049250313e-16 / 2 and p[4] > p[7]          # over_point is offset towards the eye
    assert p[10] > 2.220446049250313e-16 / 2 - 1 and p[4] < p[10]     # under_point below the surface
    p = s.probe(np.array([[0, 0, 0, 0, 0, 1]], dtype=np.float64))[0]
    assert p[1] == 1.0 and list(p[2:5]) == [0, 0, 1] and list(p[14:17]) == [0, 0, -1] and p[17] == 1
    s = make(ka.SINGLE % "plane")
    p = s.probe(np.array([[0, 1, -1, 0, -ka.S22, ka.S22]], dtype=np.float64))[0]
    assert np.allclose(p[18:21], [0, ka.S22, ka.S22], atol=1e-15)
    # smooth triangle normal (intersection/mod.rs:290-315) through an OBJ with vertex normals
    s = make(ka.CAMERA + ka.LIGHT + "- add: obj\n  file: tests/golden/smooth_triangle.obj\n")
    p = s.probe(np.array([[-0.2, 0.3, -2, 0, 0, 1]], dtype=np.float64))[0]
    assert p[0] == 0 and abs(p[1] - 2.0) < 1e-12
    assert np.allclose(p[14:17], ka.SMOOTH_TRIANGLE_NORMAL, atol=1e-5), p[14:17]
