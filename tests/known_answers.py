"""Known-answer cases taken from the reference's own unit tests (values from
*The Ray Tracer Challenge*, 5 decimals, epsilon 1e-5).  Each case is a tiny
YAML scene plus rays; the same table is run through the CPU oracle
(tests/test_oracle_known_answers.py) and through the HIP path
(tests/test_gpu_known_answers.py).

`probe` indices follow include/rt_api.h (rt_probe).
"""
import math

import numpy as np

S2 = math.sqrt(2.0)
S22 = S2 / 2.0

CAMERA = """
- add: camera
  width: 11
  height: 11
  field-of-view: "PI / 2"
  from: [0, 0, -5]
  to: [0, 0, 0]
  up: [0, 1, 0]
"""
LIGHT = """
- add: light
  at: [-10, 10, -10]
  intensity: [1, 1, 1]
"""
OUTER = """
- add: sphere
  material:
      color: [0.8, 1.0, 0.6]
      diffuse: 0.7
      specular: 0.2
"""
INNER = """
- add: sphere
  transform:
      - [scale, 0.5, 0.5, 0.5]
"""
# world::test_world() (reference src/world.rs:188-216)
TEST_WORLD = CAMERA + LIGHT + OUTER + INNER


def plane_at_minus_one(material):
    return """
- add: plane
  transform:
      - [translate, 0, -1, 0]
  material:
""" + "".join("      %s\n" % line for line in material)


RED_BALL = """
- add: sphere
  transform:
      - [translate, 0, -3.5, -0.5]
  material:
      color: [1, 0, 0]
      ambient: 0.5
"""

DOWN_RAY = [0.0, 0.0, -3.0, 0.0, -S22, S22]

# (name, scene text, ray, depth, expected colour)  -- reference file:line in the name
COLOUR_CASES = [
    ("world.rs:275 ray misses", TEST_WORLD, [0, 0, -5, 0, 1, 0], 5, (0.0, 0.0, 0.0)),
    ("world.rs:284 ray hits", TEST_WORLD, [0, 0, -5, 0, 0, 1], 2, (0.38066, 0.47583, 0.2855)),
    ("world.rs:298 intersection behind the ray",
     CAMERA + LIGHT + """
- add: sphere
  material:
      color: [0.8, 1.0, 0.6]
      ambient: 1.0
      diffuse: 0.7
      specular: 0.2
- add: sphere
  transform:
      - [scale, 0.5, 0.5, 0.5]
  material:
      ambient: 1.0
""", [0, 0, 0.75, 0, 0, -1], 1, (1.0, 1.0, 1.0)),
    ("world.rs:361 shading from the inside",
     CAMERA + """
- add: light
  at: [0, 0.25, 0]
  intensity: [1, 1, 1]
""" + OUTER + INNER, [0, 0, 0, 0, 0, 1], 5, (0.90498, 0.90498, 0.90498)),
    ("world.rs:385 intersection in shadow",
     CAMERA + """
- add: light
  at: [0, 0, -10]
  intensity: [1, 1, 1]
- add: sphere
- add: sphere
  transform:
      - [translate, 0, 0, 10]
""", [0, 0, 5, 0, 0, 1], 3, (0.1, 0.1, 0.1)),
    ("world.rs:411 reflective material", TEST_WORLD + plane_at_minus_one(["reflective: 0.5"]),
     DOWN_RAY, 5, (0.87676, 0.92434, 0.82917)),
    ("world.rs:440 transparent material",
     TEST_WORLD + plane_at_minus_one(["transparency: 0.5", "refractive_index: 1.5"]) + RED_BALL,
     DOWN_RAY, 5, (0.93643, 0.68643, 0.68643)),
    ("world.rs:489 reflective transparent material (schlick)",
     TEST_WORLD + plane_at_minus_one(["reflective: 0.5", "transparency: 0.5", "refractive_index: 1.5"]) + RED_BALL,
     DOWN_RAY, 5, (0.93392, 0.69643, 0.69243)),
    ("world.rs:703 reflection at depth 0 adds nothing", TEST_WORLD + plane_at_minus_one(["reflective: 0.5"]),
     DOWN_RAY, 0, (0.68643, 0.68643, 0.68643)),
    ("material.rs:232 eye between light and surface",
     CAMERA + """
- add: light
  at: [0, 0, -10]
  intensity: [1, 1, 1]
- add: sphere
""", [0, 0, -5, 0, 0, 1], 5, (1.9, 1.9, 1.9)),
]

GLASS = ["ambient: 0.01", "diffuse: 0.01", "transparency: 1.0"]


def glass_sphere(transform, index):
    return "- add: sphere\n  transform:\n      - %s\n  material:\n" % transform + "".join(
        "      %s\n" % line for line in GLASS + ["refractive_index: %s" % index])


# intersection/mod.rs:224-270: A (scale 2, n 1.5), B (z -0.25, n 2.0), C (z +0.25, n 2.5)
N1N2_SCENE = CAMERA + LIGHT + glass_sphere("[scale, 2, 2, 2]", 1.5) + glass_sphere("[translate, 0, 0, -0.25]", 2.0) + \
    glass_sphere("[translate, 0, 0, 0.25]", 2.5)
# the six intersections of the ray (0,0,-4)->z are at t = 2, 2.75, 3.25, 4.75, 5.25, 6; starting the ray just
# before each one makes it the hit while the earlier ones stay in the list with negative t
N1N2_TABLE = [(2.0, 1.0, 1.5), (2.75, 1.5, 2.0), (3.25, 2.0, 2.5), (4.75, 2.5, 2.5), (5.25, 2.5, 1.5), (6.0, 1.5, 1.0)]

GLASS_BALL = CAMERA + LIGHT + "- add: sphere\n  material:\n" + "".join(
    "      %s\n" % line for line in GLASS + ["refractive_index: 1.5"])
# computations.rs:60-104
SCHLICK_CASES = [
    ("total internal reflection", [0, 0, S22, 0, 1, 0], 1.0, 1e-12),
    ("perpendicular viewing angle", [0, 0, 0, 0, 1, 0], 0.04, 1e-9),
    # the reference feeds a hand-rounded t = 1.8589 here; the exact hit moves the value by 8e-5
    ("small angle, n2 > n1", [0, 0.99, -2, 0, 0, 1], 0.48873, 2e-4),
]

SINGLE = CAMERA + LIGHT + "- add: %s\n"

# (name, scene, ray, expected hit t or None, expected normal or None)
# sphere.rs:58-163, plane.rs:50-99, cube.rs:58-151, cylinder.rs:130-306, cone.rs:144-251
SHAPE_CASES = [
    ("sphere two points", SINGLE % "sphere", [0, 0, -5, 0, 0, 1], 4.0, (0, 0, -1)),
    ("sphere tangent", SINGLE % "sphere", [0, 1, -5, 0, 0, 1], 5.0, None),
    ("sphere miss", SINGLE % "sphere", [0, 2, -5, 0, 0, 1], None, None),
    ("sphere from inside", SINGLE % "sphere", [0, 0, 0, 0, 0, 1], 1.0, None),
    ("sphere behind", SINGLE % "sphere", [0, 0, 5, 0, 0, 1], None, None),
    ("plane parallel", SINGLE % "plane", [0, 10, 0, 0, 0, 1], None, None),
    ("plane coplanar", SINGLE % "plane", [0, 0, 0, 0, 0, 1], None, None),
    ("plane from above", SINGLE % "plane", [0, 1, 0, 0, -1, 0], 1.0, (0, 1, 0)),
    ("plane from below", SINGLE % "plane", [0, -1, 0, 0, 1, 0], 1.0, None),
    ("cube +x", SINGLE % "cube", [5, 0.5, 0, -1, 0, 0], 4.0, (1, 0, 0)),
    ("cube -x", SINGLE % "cube", [-5, 0.5, 0, 1, 0, 0], 4.0, (-1, 0, 0)),
    ("cube +y", SINGLE % "cube", [0.5, 5, 0, 0, -1, 0], 4.0, (0, 1, 0)),
    ("cube -y", SINGLE % "cube", [0.5, -5, 0, 0, 1, 0], 4.0, (0, -1, 0)),
    ("cube +z", SINGLE % "cube", [0.5, 0, 5, 0, 0, -1], 4.0, (0, 0, 1)),
    ("cube -z", SINGLE % "cube", [0.5, 0, -5, 0, 0, 1], 4.0, (0, 0, -1)),
    ("cube inside", SINGLE % "cube", [0, 0.5, 0, 0, 0, 1], 1.0, None),
    ("cube miss 1", SINGLE % "cube", [-2, 0, 0, 0.2673, 0.5345, 0.8018], None, None),
    ("cube miss 2", SINGLE % "cube", [0, -2, 0, 0.8018, 0.2673, 0.5345], None, None),
    ("cube miss 3", SINGLE % "cube", [2, 0, 2, 0, 0, -1], None, None),
    ("cylinder miss 1", SINGLE % "cylinder", [1, 0, 0, 0, 1, 0], None, None),
    ("cylinder miss 2", SINGLE % "cylinder", [0, 0, -5, 1 / math.sqrt(3), 1 / math.sqrt(3), 1 / math.sqrt(3)], None, None),
    ("cylinder tangent", SINGLE % "cylinder", [1, 0, -5, 0, 0, 1], 5.0, (1, 0, 0)),
    ("cylinder centre", SINGLE % "cylinder", [0, 0, -5, 0, 0, 1], 4.0, (0, 0, -1)),
    ("cylinder angled", SINGLE % "cylinder", [0.5, 0, -5] + list(np.array([0.1, 1, 1]) / math.sqrt(2.01)), 6.80798, None),
    ("cone 1", SINGLE % "cone", [0, 0, -5, 0, 0, 1], 5.0, None),
    ("cone 2", SINGLE % "cone", [0, 0, -5] + list(np.array([1, 1, 1]) / math.sqrt(3)), 8.66025, None),
    ("cone 3", SINGLE % "cone", [1, 1, -5] + list(np.array([-0.5, -1, 1]) / 1.5), 4.55006, None),
    ("cone parallel to one half", SINGLE % "cone", [0, 0, -1] + list(np.array([0, 1, 1]) / S2), 0.35355, None),
]

CLOSED_CYL = CAMERA + LIGHT + """
- add: cylinder
  min: 1
  max: 2
  closed: true
"""
# cylinder.rs: capped cylinder intersections (count of intersections in the list) and end-cap normals
CAPPED_CYL_CASES = [  # cylinder.rs:215-245 (the reference negates the book's directions; the list still has 2 entries)
    ([0, 3, 0, 0, -1, 0], 2), ([0, 3, -2] + list(-np.array([0, -1, 2]) / math.sqrt(5)), 2),
    ([0, 4, -2] + list(-np.array([0, -1, 1]) / S2), 2), ([0, 0, -2] + list(-np.array([0, 1, 2]) / math.sqrt(5)), 2),
    ([0, -1, -2] + list(-np.array([0, 1, 1]) / S2), 2),
]
TRUNCATED_CYL = CAMERA + LIGHT + """
- add: cylinder
  min: 1
  max: 2
"""
TRUNCATED_CYL_CASES = [
    ([0, 1.5, 0] + list(np.array([0.1, 1, 0]) / math.sqrt(1.01)), 0), ([0, 3, -5, 0, 0, 1], 0), ([0, 0, -5, 0, 0, 1], 0),
    ([0, 2, -5, 0, 0, 1], 0), ([0, 1, -5, 0, 0, 1], 0), ([0, 1.5, -2, 0, 0, 1], 2),
]
CLOSED_CONE = CAMERA + LIGHT + """
- add: cone
  min: -0.5
  max: 0.5
  closed: true
"""
CAPPED_CONE_CASES = [
    ([0, 0, -5, 0, 1, 0], 0), ([0, 0, -0.25] + list(np.array([0, 1, 1]) / S2), 2), ([0, 0, -0.25, 0, 1, 0], 4),
]

SMOOTH_TRIANGLE_NORMAL = (-0.5547, 0.83205, 0.0)  # intersection/mod.rs:290-315

# camera.rs:330-366 (201x101, fov PI/2)
CAMERA_CASES = [
    ("from: [0, 0, 0]\n  to: [0, 0, -1]\n  up: [0, 1, 0]", (100, 50), (0, 0, 0), (0, 0, -1)),
    ("from: [0, 0, 0]\n  to: [0, 0, -1]\n  up: [0, 1, 0]", (0, 0), (0, 0, 0), (0.66519, 0.33259, -0.66851)),
]


def camera_scene(view):
    return """
- add: camera
  width: 201
  height: 101
  field-of-view: "PI / 2"
  %s
""" % view + LIGHT + "- add: sphere\n"


# light/area.rs:136-160: Area::new(origin, (2,0,0), 4, z_axis, 2) -> u cell (0.5,0,0), v cell (0,0,0.5);
# point_on_light(u, v) = corner + ucell*(u + r1) + vcell*(v + r2) with the rng seeded 0
AREA_POINTS = [((0, 0), (0.16229, 0.19112)), ((1, 0), (0.67981, 0.00573)), ((0, 1), (0.24764, 0.51028)),
               ((2, 0), (1.42862, 0.42275)), ((3, 1), (1.64744, 0.53712))]
