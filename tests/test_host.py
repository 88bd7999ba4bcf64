"""Host side without a GPU: scene loading, OBJ parsing, flattening, the C ABI's exported
symbols and the output writers."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

import known_answers as ka
import oracle_lib as ol
import raytracer_amd as ra

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "tests", "golden", "obj")


def obj_scene(name, extra=""):
    return ka.CAMERA + ka.LIGHT + "- add: obj\n  file: tests/golden/obj/%s\n%s" % (name, extra)


def test_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rt_api.h")).read()
    declared = set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", header))
    declared -= {"rt_scene_desc", "rt_camera", "rt_partition", "rt_stats", "rt_scene"}
    assert declared == set(ra.ABI_SYMBOLS), declared ^ set(ra.ABI_SYMBOLS)
    lib = ra.hip_lib()
    for sym in declared:
        assert getattr(lib, sym) is not None
    assert lib.rt_max_csg_list() >= 20 and lib.rt_max_container_list() >= 16
    assert isinstance(lib.rt_last_error(), bytes)


def test_host_abi_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "rt_host.h")).read()
    body = header[header.index("#ifndef RT_HOST_H"):]
    declared = set(re.findall(r"\b(rth_[a-z0-9_]+)\s*\(", body))
    assert declared == set(ra.HOST_ABI_SYMBOLS), declared ^ set(ra.HOST_ABI_SYMBOLS)
    lib = ra.host_lib()
    for sym in declared:
        assert getattr(lib, sym) is not None


def test_scene_create_rejects_bad_input_before_touching_the_gpu():
    lib = ra.hip_lib()
    out = C.c_void_p()
    assert lib.rt_scene_create(None, C.byref(out)) == ra.RT_ERR_INVALID
    d = ra.SceneDesc()  # all zero: no objects
    assert lib.rt_scene_create(C.byref(d), C.byref(out)) == ra.RT_ERR_INVALID
    assert b"No objects" in lib.rt_last_error()


def test_struct_layouts_match_the_header():
    assert C.sizeof(ra.Partition) == 12
    assert C.sizeof(ra.CameraDesc) == 8 + 12 * 8 + 3 * 8
    assert C.sizeof(ra.Stats) == 4 * 8 + 8 + 4 * 8 + 8 * 8


def test_loading_the_reference_fixture_scene():  # scene/mod.rs:309-340 (src/scene/tests/simple.yaml)
    s = ra.Scene.from_file(os.path.join(ROOT, "tests", "golden", "simple.yaml"), scale=2.5)
    assert (s.width, s.height) == (500, 500) and s.n_leaves == 1 and s.n_lights == 1
    cam = s.camera()
    assert math.isclose(cam.half_width, math.tan(math.pi / 6)) and math.isclose(cam.pixel_size, cam.half_width * 2 / 500)


def test_matrix_inverses_of_the_host_library():  # matrix.rs:717-829
    import common_cases as cc
    cc.check_matrix_inverses(ra.matrix_inverse)
    rng = np.random.default_rng(3)
    for _ in range(50):     # the host library and the oracle invert alike, to the bit
        m = rng.uniform(-5, 5, (4, 4))
        a, da = ra.matrix_inverse(m)
        b, db = ol.matrix_inverse(m)
        assert da == db and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def test_scale_truncates():  # camera.rs:78-87
    s = ra.Scene.from_text(ka.TEST_WORLD, scale=0.37)
    assert (s.width, s.height) == (4, 4)


def test_scene_errors():  # scene/list.rs:28-34, add.rs:21-23, shapes.rs:227, define.rs, material.rs
    def err(text):
        with pytest.raises(ra.RtError) as e:
            ra.Scene.from_text(text)
        return str(e.value)
    assert "A camera must be defined" in err(ka.LIGHT + "- add: sphere\n")
    assert "No lights were defined" in err(ka.CAMERA + "- add: sphere\n")
    assert "No objects were defined" in err(ka.CAMERA + ka.LIGHT)
    assert "Only one camera can be added" in err(ka.CAMERA + ka.CAMERA + ka.LIGHT + "- add: sphere\n")
    assert "Reference to shape 'foo' that was not defined" in err(ka.CAMERA + ka.LIGHT + "- add: foo\n")
    assert "Reference to material 'm' that was not defined" in err(ka.CAMERA + ka.LIGHT + "- add: sphere\n  material: m\n")
    assert "Unknown operator" in err(ka.CAMERA + ka.LIGHT + "- add: sphere\n  transform:\n      - [spin, 1]\n")
    assert "requires 3 arguments, found 2" in err(ka.CAMERA + ka.LIGHT + "- add: sphere\n  transform:\n      - [scale, 1, 2]\n")
    assert "cannot be inverted" in err(ka.CAMERA + ka.LIGHT + "- add: sphere\n  transform:\n      - [scale, 0, 1, 1]\n")
    assert "Only one of pattern or colour" in err(
        ka.CAMERA + ka.LIGHT + "- add: sphere\n  material:\n      color: [1, 0, 0]\n      pattern:\n"
        "          kind: stripe\n          a: [1, 1, 1]\n          b: [0, 0, 0]\n")


def test_defines_extend_and_named_transforms():  # scene/define.rs:18-59, shapes.rs:192-225, transformations.rs:18-36
    text = ka.CAMERA + ka.LIGHT + """
- define: base-material
  value:
      color: [1, 0, 0]
      diffuse: 0.5
- define: shiny
  extend: base-material
  value:
      diffuse: 0.25
      reflective: 0.75
- define: up-two
  value:
      - [translate, 0, 2, 0]
- define: big-up
  value:
      - [scale, 3, 3, 3]
      - up-two
- define: ball
  value:
      add: sphere
      transform:
          - [scale, 0.5, 0.5, 0.5]
      material: shiny
- add: ball
  transform:
      - big-up
- add: ball
  material: base-material
  shadow: false
"""
    d = ra.Scene.from_text(text).desc()
    assert d["leaf_shape"].tolist() == [0, 0] and d["leaf_casts_shadow"].tolist() == [1, 0]
    m0 = d["material"][d["leaf_material"][0]]
    m1 = d["material"][d["leaf_material"][1]]
    assert m0[1] == 0.25 and m0[4] == 0.75 and m1[1] == 0.5 and m1[4] == 0.0
    # scale .5, then scale 3, then translate y 2: inverse = scale 1/1.5, translate y -2/1.5
    inv = d["leaf_inv"][0].reshape(3, 4)
    assert np.allclose(inv, [[1 / 1.5, 0, 0, 0], [0, 1 / 1.5, 0, -2 / 1.5], [0, 0, 1 / 1.5, 0]])
    assert np.allclose(d["leaf_inv"][1].reshape(3, 4), [[2, 0, 0, 0], [0, 2, 0, 0], [0, 0, 2, 0]])


def test_angle_forms():  # angle.rs:75-106, 225-250
    text = ka.CAMERA + ka.LIGHT + """
- add: cube
  transform:
      - [rotate-y, "PI / 2"]
- add: cube
  transform:
      - [rotate-y, degrees: 90]
- add: cube
  transform:
      - [rotate-y, 1.5707963267948966]
- add: cube
  transform:
      - [rotate-y, "-PI / 2"]
"""
    d = ra.Scene.from_text(text).desc()
    assert np.array_equal(d["leaf_inv"][0], d["leaf_inv"][1]) and np.array_equal(d["leaf_inv"][0], d["leaf_inv"][2])
    assert d["leaf_inv"][3][2] == -d["leaf_inv"][0][2]
    assert d["n_classes"] == 2 and d["leaf_class"].tolist() == [0, 0, 0, 1]


def test_material_with_pattern_consumes_one_rng_value():  # scene/material.rs:26-31
    plain = ra.Scene.from_text(ka.TEST_WORLD, seed=3)
    patterned = ra.Scene.from_text(ka.TEST_WORLD + """
- add: plane
  material:
      pattern:
          kind: checker
          a: [1, 1, 1]
          b: [0, 0, 0]
""", seed=3)
    a, b = plain.row_seeds(), patterned.row_seeds()
    assert a[1] == b[0] and a[0] != b[0]
    assert (ol.rng_u64(3, 12)[:11] == a).all()


def test_obj_parser():  # obj_parser.rs:240-506 with the reference's own fixtures
    def leaves(name):
        return ra.Scene.from_text(obj_scene(name)).desc()
    assert len(leaves("triangles.obj")["leaf_shape"]) == 2       # two triangles in two named groups
    d = leaves("triangulating.obj")                              # a pentagon -> fan of three triangles
    assert d["leaf_shape"].tolist() == [5, 5, 5]
    t = d["tri_data"]
    assert t[0][:3].tolist() == [-1, 1, 0] and t[1][:3].tolist() == [-1, 1, 0] and t[2][:3].tolist() == [-1, 1, 0]
    d = leaves("face_normals.obj")                               # f a/b/c and a//c forms, smooth normals
    assert len(d["leaf_shape"]) == 2 and np.array_equal(d["tri_data"][0][9:], d["tri_data"][1][9:])
    assert d["tri_data"][0][9:].tolist() == [0, 1, 0, -1, 0, 0, 1, 0, 0]    # obj_parser.rs:454-477
    for bad, msg in (("too_many_vertices.obj", "Expected 'v' followed by 3 space separated numbers for a vertex.\nFound 5 items."),
                     ("invalid_vertices.obj", "invalid float literal"),
                     ("too_many_normals.obj", "Expected 'vn' followed by 3 space separated numbers for a normal."),
                     ("invalid_normals.obj", "invalid float literal"),
                     ("not_enough_faces.obj", "Expected 'f' followed by at least 3"),
                     ("invalid_faces.obj", "does not exist"),  # index out of bounds panic in the reference
                     ("invalid_groups.obj", "Group FirstGroup is repeated."),
                     ("inconsistent_face_normals.obj", "If one vertex normal is specified, all faces must also provide vertex normals."),
                     ("invalid_face_normals.obj", "Expected face values to be either 'num' or 'num//num' or 'num/num/num'\nFound 2///3."),
                     ("invalid_index_face_normals.obj", "does not exist")):  # a panic in the reference
        with pytest.raises(ra.RtError) as e:
            ra.Scene.from_text(obj_scene(bad))
        assert msg in str(e.value), (bad, str(e.value))


def test_gibberish_obj_gives_an_empty_group():
    # obj_parser.rs:240-262: every line ignored; the resulting group has no children and an inverted-infinite box
    s = ra.Scene.from_text(obj_scene("gibberish.obj") + "- add: sphere\n")
    d = s.desc()
    assert d["node_kind"].tolist() == [1, 0] and d["node_skip"].tolist() == [1, 2]
    assert d["node_bbox"][0].tolist() == [math.inf] * 3 + [-math.inf] * 3


def test_flattened_tree_structure():
    s = ra.Scene.from_file(os.path.join(ROOT, "scenes", "chapter14.yaml"))
    d = s.desc()
    kinds, skip = d["node_kind"], d["node_skip"]
    assert kinds[0] == 1 and skip[0] == len(kinds) and (kinds == 0).sum() == 12
    assert d["node_arg"][kinds == 0].tolist() == list(range(12))          # leaves numbered in node order
    for i in range(len(kinds)):
        assert i < skip[i] <= len(kinds)
        if kinds[i] == 0:
            assert skip[i] == i + 1
    # group box = union of its children's world boxes (group/mod.rs:130-140, bounding_box/mod.rs:182-201)
    o = ol.OracleScene.from_file(os.path.join(ROOT, "scenes", "chapter14.yaml"))
    assert o.width == s.width
    box = d["node_bbox"][0]
    assert np.all(box[:3] < box[3:])


def test_divide_matches_the_reference_examples():  # group/mod.rs:578-686
    text = ka.CAMERA + ka.LIGHT + """
- add: group
  divide: 1
  children:
      - add: sphere
        transform:
            - [translate, -2, 0, 0]
      - add: sphere
        transform:
            - [translate, 2, 0, 0]
      - add: sphere
"""
    d = ra.Scene.from_text(text).desc()
    # s3 (straddles the split plane) stays; [s1] and [s2] become sub-groups appended after it
    assert d["node_kind"].tolist() == [1, 0, 1, 0, 1, 0]
    inv = d["leaf_inv"].reshape(-1, 3, 4)
    assert [inv[k][0][3] for k in range(3)] == [0.0, 2.0, -2.0]     # leaf order: s3, s1, s2
    text2 = text.replace("divide: 1", "divide: 4")
    assert ra.Scene.from_text(text2).desc()["node_kind"].tolist() == [1, 0, 0, 0]  # below the threshold: untouched
    # nested: group(divide 3) [ subgroup[s1, s2, s3], s4 ] -> the subgroup is divided, the parent is not
    text3 = ka.CAMERA + ka.LIGHT + """
- add: group
  divide: 3
  children:
      - add: group
        children:
            - add: sphere
              transform:
                  - [translate, -2, 0, 0]
            - add: sphere
              transform:
                  - [translate, 2, 1, 0]
            - add: sphere
              transform:
                  - [translate, 2, -1, 0]
      - add: sphere
"""
    # group/mod.rs:643-686: [ [ [s1], [s2, s3] ], s4 ]
    d = ra.Scene.from_text(text3).desc()
    assert d["node_kind"].tolist() == [1, 1, 1, 0, 1, 0, 0, 0]
    inv = d["leaf_inv"].reshape(-1, 3, 4)
    assert [(inv[k][0][3], inv[k][1][3]) for k in range(4)] == [(2.0, 0.0), (-2.0, -1.0), (-2.0, 1.0), (0.0, 0.0)]


def test_csg_left_class_sets():
    d = ra.Scene.from_file(os.path.join(ROOT, "scenes", "chapter16.yaml")).desc()
    assert d["csg_op"].tolist() == [0, 1, 2, 0, 1, 2, 2]
    # csg k's left set: first csg's left child is leaf 1 (class 1) -> bit 1
    assert d["csg_left_bits"].tolist() == [1 << 1, 1 << 3, 1 << 5, (1 << 7) | (1 << 8), 1 << 7, (1 << 9) | (1 << 10), 1 << 9]


def test_structural_classes_merge_equal_leaves():  # shape.rs:108: casts_shadow is not part of equality
    text = ka.CAMERA + ka.LIGHT + "- add: sphere\n- add: sphere\n  shadow: false\n- add: cube\n- add: sphere\n  material:\n      ambient: 0.2\n"
    assert ra.Scene.from_text(text).desc()["leaf_class"].tolist() == [0, 0, 1, 2]


def test_sphere_scene_generator_matches_oracle_seeding():  # scene/mod.rs:114-293
    s = ra.Scene.generate_random_spheres(1.0, 0)
    assert (s.width, s.height) == (1000, 800) and s.n_leaves == 121 and s.n_nodes == 129 and s.n_lights == 2
    assert (s.row_seeds() == ol.OracleScene.spheres(1.0, 0).row_seeds()).all()


def test_perturbed_pattern_is_refused():
    s = ra.Scene.from_file(os.path.join(ROOT, "scenes", "chapter10.yaml"))
    with pytest.raises(ra.RtError) as e:
        s.upload()
    assert e.value.code == ra.RT_ERR_UNSUPPORTED


def test_writers(tmp_path):  # canvas.rs:148-219
    rgb8 = np.array([[[255, 0, 0], [0, 128, 0], [0, 0, 255]], [[1, 2, 3], [4, 5, 6], [7, 8, 9]]], dtype=np.uint8)
    c = ra.Canvas(3, 2, None, rgb8)
    c.to_ppm(str(tmp_path / "a.ppm"))
    assert open(tmp_path / "a.ppm").read() == "P3\n3 2\n255\n255 0 0\n0 128 0\n0 0 255\n1 2 3\n4 5 6\n7 8 9\n"
    c.save_png(str(tmp_path / "a.png"))
    from PIL import Image
    assert np.array_equal(np.asarray(Image.open(tmp_path / "a.png")), rgb8)
