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


def _probe(s, ray):
    return s.probe(np.array([ray], dtype=np.float64))[0]


def check_bounding_boxes(make):
    # object/bounding_box/mod.rs:295-416
    for text, rays in ((ka.BBOX_UNIT, ka.BBOX_UNIT_RAYS), (ka.BBOX_NON_CUBIC, ka.BBOX_NON_CUBIC_RAYS)):
        s = make(text)
        for origin, direction, intersects in rays:
            p = _probe(s, list(origin) + list(direction))
            assert p[24] == (2 if intersects else 0), (origin, direction, p[24])
            assert (p[0] == 0) == intersects, (origin, direction, p[0])


def _csg_walk(op, events):
    """Csg::filter_intersections (object/csg/mod.rs:95-123) over (t, is_left) events, driven by the reference's table."""
    inl = inr = False
    kept = []
    for t, left in events:
        if ka.CSG_RULES[op][(int(left), int(inl), int(inr))]:
            kept.append((t, left))
        if left:
            inl = not inl
        else:
            inr = not inr
    return kept


def check_csg(make):
    # the filter test, csg/mod.rs:230-265: intersections L R L R keep (0,3), (1,2), (0,1)
    left, right = ka.CSG_ARRANGEMENTS[0]
    events = [(1.0, True), (2.0, False), (3.0, True), (4.0, False)]
    for op, keep in ka.CSG_FILTER_EXPECTED.items():
        assert _csg_walk(op, events) == [events[keep[0]], events[keep[1]]]
    rows_seen = {op: set() for op in ka.CSG_RULES}
    for op in ka.CSG_RULES:
        for left, right in ka.CSG_ARRANGEMENTS:
            events = sorted([(left[0] - left[1], True), (left[0] + left[1], True),
                             (right[0] - right[1], False), (right[0] + right[1], False)])
            inl = inr = False
            for t, is_left in events:
                rows_seen[op].add((int(is_left), int(inl), int(inr)))
                inl, inr = (not inl, inr) if is_left else (inl, not inr)
            kept = _csg_walk(op, events)
            s = make(ka.csg_scene(op, left, right))
            for k, (t, _) in enumerate(events + [(5.0, True)]):
                z0 = t - 0.125          # the ray starts just before this event
                p = _probe(s, [0, 0, z0, 0, 0, 1])
                # a csg wholly behind the ray fails its bounding box (bounding_box/mod.rs:84) and delivers nothing
                behind = z0 > events[-1][0]
                assert p[24] == (0 if behind else len(kept)), (op, left, right, p[24], kept)
                ahead = [(tt, ll) for tt, ll in kept if tt > z0]
                if not ahead:
                    assert p[0] == -1, (op, left, right, z0, p[:2])
                else:
                    assert p[0] == (0 if ahead[0][1] else 1), (op, left, right, z0, p[:2], ahead)
                    assert abs(p[1] - (ahead[0][0] - z0)) < 1e-12, (op, left, right, z0, p[:2], ahead)
    # between them the arrangements reach every row of the table but the two a convex pair cannot: a hit on the
    # left while already inside the left is always a leaving hit, covered as (1, 1, x)
    for op, rows in rows_seen.items():
        assert len(rows) >= 6, (op, rows)
    # csg/mod.rs:268-297
    p = _probe(make(ka.CSG_MISS), [0, 2, -5, 0, 0, 1])
    assert p[0] == -1 and p[24] == 0
    s = make(ka.CSG_HIT)
    p = _probe(s, [0, 0, -5, 0, 0, 1])
    assert p[24] == 2 and p[0] == 0 and abs(p[1] - 4.0) < 1e-12
    p = _probe(s, [0, 0, 0.25, 0, 0, 1])        # past the first: the second one is on the right sphere
    assert p[24] == 2 and p[0] == 1 and abs(p[1] - 1.25) < 1e-12


def check_groups(make):
    # object/group/mod.rs:225-289
    s = make(ka.GROUP_OF_THREE)
    p = _probe(s, [0, 0, -5, 0, 0, 1])
    assert p[24] == 4 and p[0] == 1 and abs(p[1] - 1.0) < 1e-12, p[:2]
    p = _probe(s, [0, 0, -3, 0, 0, 1])
    assert p[24] == 4 and p[0] == 1 and abs(p[1] - 1.0) < 1e-12, p[:2]
    p = _probe(s, [0, 0, -1.5, 0, 0, 1])
    assert p[24] == 4 and p[0] == 0 and abs(p[1] - 0.5) < 1e-12, p[:2]
    p = _probe(make(ka.GROUP_ONE), [0, 0, -5, 0, 1, 0])       # misses the group's box
    assert p[0] == -1 and p[24] == 0
    p = _probe(make(ka.GROUP_TRANSFORMED), [10, 0, -10, 0, 0, 1])
    assert p[24] == 2 and p[0] == 0 and abs(p[1] - 8.0) < 1e-12, p[:2]
    p = _probe(make(ka.GROUP_TRANSFORMED), [10, 0, 0, 0, 0, 1])
    assert p[24] == 2 and p[0] == 0 and abs(p[1] - 2.0) < 1e-12, p[:2]


def check_list_hits(make):
    # intersection/list.rs:122-170
    for spans, expected in ka.LIST_HIT_CASES:
        p = _probe(make(ka.spheres_at(*spans)), [0, 0, 0, 0, 0, 1])
        assert p[24] == 2 * len(spans)
        if expected is None:
            assert p[0] == -1
        else:
            assert abs(p[1] - expected) < 1e-12, (spans, p[:2])
            assert p[0] == [i for i, (a, b) in enumerate(spans) if expected in (a, b)][0]


def check_triangles(make):
    # object/shapes/triangle.rs:148-303
    s = make(ka.FLAT_TRIANGLE)
    for ray, t in ka.TRIANGLE_CASES:
        p = _probe(s, ray)
        if t is None:
            assert p[0] == -1 and p[24] == 0, (ray, p[:2])
        else:
            assert p[0] == 0 and p[24] == 1 and abs(p[1] - t) < 1e-12, (ray, p[:2])
    for ray in ka.TRIANGLE_NORMAL_RAYS:
        p = _probe(s, ray)
        assert p[0] == 0 and list(p[14:17]) == [0, 0, -1], (ray, p[14:17])


def _lighting_rows(cases):
    return np.array([[leaf, light] + list(point) + list(eye) + list(normal) + [intensity]
                     for leaf, light, point, eye, normal, intensity in cases], dtype=np.float64)


def check_lighting(make):
    # material.rs:230-487
    s = make(ka.LIGHTING_WORLD)
    got = s.lighting(_lighting_rows([c[1:7] for c in ka.LIGHTING_CASES]))
    for (name, *_, expected, tol), colour in zip(ka.LIGHTING_CASES, got):
        assert np.allclose(colour, expected, atol=max(tol, 1e-15), rtol=0), (name, colour, expected)
    s = make(ka.AREA_LIGHTING_WORLD)
    for point, expected in ka.AREA_LIGHTING_CASES:
        eye = np.array([0.0, 0.0, -5.0]) - np.array(point)
        eye = eye / np.sqrt((eye * eye).sum())
        colour = s.lighting(_lighting_rows([(0, 0, point, eye, point, 1.0)]), np.array([0], dtype=np.uint64))[0]
        assert np.allclose(colour, [expected] * 3, atol=1e-5, rtol=0), (point, colour, expected)


def check_intensity_at(make):
    # light/area.rs:164-198: five calls on one generator seeded 0
    got = make(ka.AREA_INTENSITY_WORLD).intensity_at(0, [p for p, _ in ka.AREA_INTENSITY_SEQUENCE], 0)
    assert list(got) == [e for _, e in ka.AREA_INTENSITY_SEQUENCE], got
    # light/point.rs:59-94
    got = make(ka.TEST_WORLD).intensity_at(0, [p for p, _ in ka.POINT_INTENSITY], 0)
    assert list(got) == [e for _, e in ka.POINT_INTENSITY], got


def check_patterns(make):
    for text, rows, tol in ka.PATTERN_CASES:
        s = make(text)
        got = s.lighting(_lighting_rows([(0, 0, point, ka.NZ, ka.NZ, 1.0) for point, _ in rows]))
        for (point, expected), colour in zip(rows, got):
            if tol == 0:
                assert list(colour) == list(expected), (text, point, colour, expected)
            else:
                assert np.allclose(colour, expected, atol=tol, rtol=0), (text, point, colour, expected)


def check_matrix_inverses(inverse):
    # math/transformation/matrix.rs:717-829
    for m, det, inv in ka.MATRIX_INVERSES:
        got, d = inverse(m)
        assert got is not None
        if det is not None:
            assert abs(d - det) < 1e-9 * abs(det), (d, det)
        if inv is not None:
            assert np.allclose(got, inv, atol=1e-5, rtol=0), (got, inv)
        assert np.allclose(np.array(m, dtype=np.float64) @ got, np.eye(4), atol=1e-12)
    got, d = inverse(ka.MATRIX_NOT_INVERTIBLE)
    assert got is None and d == 0.0
    a, b = (np.array(x, dtype=np.float64) for x in ka.MATRIX_PRODUCT_ROUND_TRIP)
    got, _ = inverse(b)
    assert np.allclose((a @ b) @ got, a, atol=1e-12)
