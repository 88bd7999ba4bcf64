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


# ---------------------------------------------------------------------------------------------
# bounding boxes (object/bounding_box/mod.rs:295-416): a group's cull test.  The group holds one
# cube whose box is exactly the reference's box, so a ray passes iff BoundingBox::intersect says so
# (twice: the group's stored box in world space and the cube's own +-1 box in object space).
def boxed_group(scale, translate):
    return CAMERA + LIGHT + """
- add: group
  children:
      - add: cube
        transform:
            - [scale, %s, %s, %s]
            - [translate, %s, %s, %s]
""" % (tuple(scale) + tuple(translate))


BBOX_UNIT = boxed_group((1, 1, 1), (0, 0, 0))                # (-1,-1,-1) .. (1,1,1)
BBOX_NON_CUBIC = boxed_group((3, 3, 3.5), (8, 1, 3.5))       # (5,-2,0) .. (11,4,7)
BBOX_UNIT_RAYS = [  # (origin, direction, intersects)
    ((5, 0.5, 0), (-1, 0, 0), True), ((-5, 0.5, 0), (1, 0, 0), True), ((0.5, 5, 0), (0, -1, 0), True),
    ((0.5, -5, 0), (0, 1, 0), True), ((0.5, 0, 5), (0, 0, -1), True), ((0.5, 0, -5), (0, 0, 1), True),
    ((0, 0.5, 0), (0, 0, 1), True),
    ((-2, 0, 0), (2, 4, 6), False), ((0, -2, 0), (6, 2, 4), False), ((0, 0, -2), (4, 6, 2), False),
    ((2, 0, 2), (0, 0, -1), False), ((0, 2, 2), (0, -1, 0), False), ((2, 2, 0), (-1, 0, 0), False),
]
BBOX_NON_CUBIC_RAYS = [
    ((15, 1, 2), (-1, 0, 0), True), ((-5, -1, 4), (1, 0, 0), True), ((7, 6, 5), (0, -1, 0), True),
    ((9, -5, 6), (0, 1, 0), True), ((8, 2, 12), (0, 0, -1), True), ((6, 0, -5), (0, 0, 1), True),
    ((8, 1, 3.5), (0, 0, 1), True),
    ((9, -1, -8), (2, 4, 6), False), ((8, 3, -4), (6, 2, 4), False), ((9, -1, -2), (4, 6, 2), False),
    ((4, 0, 9), (0, 0, -1), False), ((8, 6, -1), (0, -1, 0), False), ((12, 5, 4), (-1, 0, 0), False),
]

# ---------------------------------------------------------------------------------------------
# csg (object/csg/mod.rs:177-297).  CSG_RULES is the reference's truth table, (left hit, inside
# left, inside right) -> allowed, line for line.
CSG_RULES = {
    "union": {(1, 1, 1): 0, (1, 1, 0): 1, (1, 0, 1): 0, (1, 0, 0): 1, (0, 1, 1): 0, (0, 1, 0): 0, (0, 0, 1): 1, (0, 0, 0): 1},
    "intersection": {(1, 1, 1): 1, (1, 1, 0): 0, (1, 0, 1): 1, (1, 0, 0): 0, (0, 1, 1): 1, (0, 1, 0): 1, (0, 0, 1): 0, (0, 0, 0): 0},
    "difference": {(1, 1, 1): 0, (1, 1, 0): 1, (1, 0, 1): 0, (1, 0, 0): 1, (0, 1, 1): 1, (0, 1, 0): 1, (0, 0, 1): 0, (0, 0, 0): 0},
}
# Two spheres on the z axis, given as (centre, radius); a ray along +z crosses them at centre -+ radius.
# Together the three arrangements put every row of the table to work.
CSG_ARRANGEMENTS = [  # (left, right)
    ((2.0, 1.0), (3.0, 1.0)),    # left [1,3], right [2,4]: the reference's filter test (l[0..3] = L R L R)
    ((2.5, 1.5), (2.5, 0.5)),    # right inside left
    ((2.5, 0.5), (2.5, 1.5)),    # left inside right
]
CSG_FILTER_EXPECTED = {"union": (0, 3), "intersection": (1, 2), "difference": (0, 1)}  # csg/mod.rs:260-262


def csg_scene(op, left, right):
    def ball(c, r):
        return "      type: sphere\n      transform:\n          - [scale, %r, %r, %r]\n          - [translate, 0, 0, %r]\n" % (r, r, r, c)
    return CAMERA + LIGHT + "- add: csg\n  operation: %s\n  left:\n%s  right:\n%s" % (op, ball(*left), ball(*right))


CSG_MISS = CAMERA + LIGHT + "- add: csg\n  operation: union\n  left:\n      type: sphere\n  right:\n      type: cube\n"
CSG_HIT = CAMERA + LIGHT + """- add: csg
  operation: union
  left:
      type: sphere
  right:
      type: sphere
      transform:
          - [translate, 0, 0, 0.5]
"""

# ---------------------------------------------------------------------------------------------
# groups (object/group/mod.rs:225-289) and List::hit (intersection/list.rs:122-170)
GROUP_OF_THREE = CAMERA + LIGHT + """
- add: group
  children:
      - add: sphere
      - add: sphere
        transform:
            - [translate, 0, 0, -3]
      - add: sphere
        transform:
            - [translate, 5, 0, 0]
"""
GROUP_ONE = CAMERA + LIGHT + "- add: group\n  children:\n      - add: sphere\n"
GROUP_TRANSFORMED = CAMERA + LIGHT + """
- add: group
  transform:
      - [scale, 2, 2, 2]
  children:
      - add: sphere
        transform:
            - [translate, 5, 0, 0]
"""


def spheres_at(*spans):
    """Spheres on the z axis whose intersections with the ray from the origin along +z are the given (t1, t2)."""
    return CAMERA + LIGHT + "".join(
        "- add: sphere\n  transform:\n      - [scale, %r, %r, %r]\n      - [translate, 0, 0, %r]\n" % (((b - a) / 2,) * 3 + ((a + b) / 2,))
        for a, b in spans)


# (intersections of the list, expected hit t or None)
LIST_HIT_CASES = [([(1.0, 2.0)], 1.0), ([(-1.0, 1.0)], 1.0), ([(-2.0, -1.0)], None), ([(5.0, 7.0), (-3.0, 2.0)], 2.0)]

# ---------------------------------------------------------------------------------------------
# triangles (object/shapes/triangle.rs:148-303): p1 (0,1,0), p2 (-1,0,0), p3 (1,0,0)
FLAT_TRIANGLE = CAMERA + LIGHT + "- add: obj\n  file: tests/golden/flat_triangle.obj\n"
TRIANGLE_CASES = [  # (ray, t or None)
    ([0, -1, -2, 0, 1, 0], None), ([1, 1, -2, 0, 0, 1], None), ([-1, 1, -2, 0, 0, 1], None), ([0, -1, -2, 0, 0, 1], None),
    ([0, 0.5, -2, 0, 0, 1], 2.0),
]
TRIANGLE_NORMAL_RAYS = [[0, 0.5, -2, 0, 0, 1], [-0.5, 0.25, -2, 0, 0, 1], [0.5, 0.25, -2, 0, 0, 1]]  # the normal is -z wherever a ray strikes

# ---------------------------------------------------------------------------------------------
# Material::lighting called directly (material.rs:230-487).  One world with the three light
# positions the reference uses; rows are (leaf, light, point, eye, normal, intensity) -> colour.
LIGHTING_WORLD = CAMERA + """
- add: light
  at: [0, 0, -10]
  intensity: [1, 1, 1]
- add: light
  at: [0, 10, -10]
  intensity: [1, 1, 1]
- add: light
  at: [0, 0, 10]
  intensity: [1, 1, 1]
- add: sphere
- add: sphere
  material:
      pattern:
          kind: stripe
          a: [1, 1, 1]
          b: [0, 0, 0]
      ambient: 1.0
      diffuse: 0.0
      specular: 0.0
- add: sphere
  material:
      ambient: 0.1
      diffuse: 0.9
      specular: 0.0
"""
O = (0.0, 0.0, 0.0)
NZ = (0.0, 0.0, -1.0)
LIGHTING_CASES = [  # (name, leaf, light, point, eye, normal, intensity, expected, tolerance)
    ("material.rs:232 surface in shadow", 0, 0, O, NZ, NZ, 0.0, (0.1, 0.1, 0.1), 1e-12),
    ("material.rs:250 eye between light and surface", 0, 0, O, NZ, NZ, 1.0, (1.9, 1.9, 1.9), 1e-12),
    ("material.rs:268 eye offset 45 degrees", 0, 0, O, (0.0, S22, -S22), NZ, 1.0, (1.0, 1.0, 1.0), 1e-12),
    ("material.rs:287 light offset 45 degrees", 0, 1, O, NZ, NZ, 1.0, (0.7364, 0.7364, 0.7364), 1e-4),
    ("material.rs:306 eye in the path of the reflection", 0, 1, O, (0.0, -S22, -S22), NZ, 1.0, (1.6364, 1.6364, 1.6364), 1e-4),
    ("material.rs:326 light behind the surface", 0, 2, O, NZ, NZ, 1.0, (0.1, 0.1, 0.1), 1e-12),
    ("material.rs:344 reflect dot eye below zero", 0, 0, O, (0.0, 0.0, 0.3), NZ, 1.0, (1.0, 1.0, 1.0), 1e-12),
    ("material.rs:362 stripe pattern, white side", 1, 0, (0.9, 0.0, 0.0), NZ, NZ, 1.0, (1.0, 1.0, 1.0), 1e-12),
    ("material.rs:362 stripe pattern, black side", 1, 0, (1.1, 0.0, 0.0), NZ, NZ, 1.0, (0.0, 0.0, 0.0), 1e-12),
    ("material.rs:411 intensity 1.0", 2, 0, NZ, NZ, NZ, 1.0, (1.0, 1.0, 1.0), 1e-12),
    ("material.rs:411 intensity 0.5", 2, 0, NZ, NZ, NZ, 0.5, (0.55, 0.55, 0.55), 1e-12),
    ("material.rs:411 intensity 0.0", 2, 0, NZ, NZ, NZ, 0.0, (0.1, 0.1, 0.1), 1e-12),
]

AREA_2X2 = """
- add: light
  corner: [-0.5, -0.5, -5]
  uvec: [1, 0, 0]
  usteps: 2
  vvec: [0, 1, 0]
  vsteps: 2
  intensity: [1, 1, 1]
"""
# material.rs:447-487: lighting samples the 2x2 area light, rng seeded 0; eye at (0,0,-5), normal = the point
AREA_LIGHTING_WORLD = CAMERA + AREA_2X2 + """
- add: sphere
  material:
      ambient: 0.1
      diffuse: 0.9
      specular: 0.0
"""
AREA_LIGHTING_CASES = [((0.0, 0.0, -1.0), 0.99367), ((0.0, S2, -S2), 0.78005)]
# light/area.rs:164-198: intensity_at over world::test_world(), five calls on one rng seeded 0
AREA_INTENSITY_WORLD = CAMERA + AREA_2X2 + OUTER + INNER
AREA_INTENSITY_SEQUENCE = [((0, 0, 2), 0.0), ((1, -1, 2), 0.5), ((1.5, 0, 2), 0.5), ((1.25, 1.25, 3), 0.75), ((0, 0, -2), 1.0)]
# light/point.rs:59-94: the point light of world::test_world()
POINT_INTENSITY = [((0, 1.00001, 0), 1.0), ((-1.00001, 0, 0), 1.0), ((0, 0, -1.00001), 1.0), ((0, 0, 1.00001), 0.0),
                   ((1.00001, 0, 0), 0.0), ((0, -1.00001, 0), 0.0), ((0, 0, 0), 0.0)]

# ---------------------------------------------------------------------------------------------
# patterns (pattern/stripe.rs, ring.rs, checker.rs, gradient.rs, radial_gradient.rs, blend.rs and
# pattern/mod.rs:321-373), read back as Material::lighting with ambient 1, diffuse 0, specular 0
# under a white light: the colour is pattern_at(object, point).
def pattern_world(kind, a=(1, 1, 1), b=(0, 0, 0), object_transform="", pattern_transform=""):
    return CAMERA + LIGHT + """- add: sphere
%s  material:
      pattern:
          kind: %s
          a: [%r, %r, %r]
          b: [%r, %r, %r]
%s      ambient: 1.0
      diffuse: 0.0
      specular: 0.0
""" % ((object_transform, kind) + tuple(a) + tuple(b) + (pattern_transform,))


W3, B3 = (1.0, 1.0, 1.0), (0.0, 0.0, 0.0)


def grey(v):
    return (v, v, v)


PATTERN_CASES = [  # (world, [(point, expected colour)], tolerance)
    (pattern_world("stripe"), [((0, 0, 0), W3), ((0, 1, 0), W3), ((0, 2, 0), W3), ((0, 0, 1), W3), ((0, 0, 2), W3),
                               ((0.9, 0, 0), W3), ((1, 0, 0), B3), ((-0.1, 0, 0), B3), ((-1, 0, 0), B3), ((-1.1, 0, 0), W3)], 0),
    (pattern_world("ring"), [((0, 0, 0), W3), ((1, 0, 0), B3), ((0, 0, 1), B3), ((0.708, 0, 0.708), B3),
                             ((0, 1, 0), W3), ((0, 4.6, 0), W3), ((0, -1.5, 0), W3)], 0),
    (pattern_world("checker"), [((0, 0, 0), W3), ((0.99, 0, 0), W3), ((1.01, 0, 0), B3), ((-1.99, 0, 0), W3), ((-2.01, 0, 0), B3),
                                ((0, 0.99, 0), W3), ((0, 1.01, 0), B3), ((0, -1.99, 0), W3), ((0, -2.01, 0), B3),
                                ((0, 0, 0.99), W3), ((0, 0, 1.01), B3), ((0, 0, -1.99), W3), ((0, 0, -2.01), B3)], 0),
    (pattern_world("gradient"), [((0, 0, 0), W3), ((0.25, 0, 0), grey(0.75)), ((0.5, 0, 0), grey(0.5)), ((0.75, 0, 0), grey(0.25)),
                                 ((-0.25, 0, 0), grey(0.25)), ((-0.5, 0, 0), grey(0.5)), ((-0.75, 0, 0), grey(0.75)),
                                 ((0, 3, 0), W3), ((0, -2.1, 0), W3), ((0, 0, 1.5), W3), ((0, 0, -0.1), W3), ((0, -4.2, 2.4), W3)], 1e-12),
    (pattern_world("radial-gradient"), [((0, 0, 0), W3), ((0.25, 0, 0.25), grey(0.646447)), ((1.5, 0, 1.5), grey(0.87868)),
                                        ((-0.25, 0, -0.25), grey(0.646447)), ((-1.5, 0, -1.5), grey(0.87868)),
                                        ((0, 1, 0), W3), ((0, 4.6, 0), W3), ((0, -1.5, 0), W3)], 1e-5),
    (pattern_world("blend", (1, 0, 0), (0, 1, 0)), [((0, 0, 0), (0.5, 0.5, 0.0)), ((1.5, 2.9, 0.3), (0.5, 0.5, 0.0)),
                                                    ((-4.0, 0.21, -1.1), (0.5, 0.5, 0.0))], 1e-12),
    # pattern/mod.rs:321-373: object and pattern transformations
    (pattern_world("stripe", object_transform="  transform:\n      - [scale, 2, 2, 2]\n"), [((1.5, 0, 0), W3)], 0),
    (pattern_world("stripe", pattern_transform="          transform:\n              - [scale, 2, 2, 2]\n"), [((1.5, 0, 0), W3)], 0),
    (pattern_world("stripe", object_transform="  transform:\n      - [scale, 2, 2, 2]\n",
                   pattern_transform="          transform:\n              - [translate, 0.5, 0, 0]\n"), [((2.5, 0, 0), W3)], 0),
]

# ---------------------------------------------------------------------------------------------
# matrix inverses (math/transformation/matrix.rs:717-829)
MATRIX_INVERSES = [  # (matrix, determinant or None, inverse to 5 decimals or None)
    ([[6, 4, 4, 4], [5, 5, 7, 6], [4, -9, 3, -7], [9, 1, 7, -6]], -2120.0, None),
    ([[-5, 2, 6, -8], [1, -5, 1, 8], [7, 7, -6, -7], [1, -3, 7, 4]], 532.0,
     [[0.21805, 0.45113, 0.2406, -0.04511], [-0.80827, -1.45677, -0.44361, 0.52068],
      [-0.07895, -0.22368, -0.05263, 0.19737], [-0.52256, -0.81391, -0.30075, 0.30639]]),
    ([[8, -5, 9, 2], [7, 5, 6, 1], [-6, 0, 9, 6], [-3, 0, -9, -4]], None,
     [[-0.15385, -0.15385, -0.28205, -0.53846], [-0.07692, 0.12308, 0.02564, 0.03077],
      [0.35897, 0.35897, 0.4359, 0.92308], [-0.69231, -0.69231, -0.76923, -1.92308]]),
    ([[9, 3, 0, 9], [-5, -2, -6, -3], [-4, 9, 6, 4], [-7, 6, 6, 2]], None,
     [[-0.04074, -0.07778, 0.14444, -0.22222], [-0.07778, 0.03333, 0.36667, -0.33333],
      [-0.02901, -0.1463, -0.10926, 0.12963], [0.17778, 0.06667, -0.26667, 0.33333]]),
]
MATRIX_NOT_INVERTIBLE = [[-4, 2, -2, -3], [9, 6, 2, 6], [0, -5, 1, -5], [0, 0, 0, 0]]
MATRIX_PRODUCT_ROUND_TRIP = ([[3, -9, 7, 3], [3, -8, 2, -9], [-4, 4, 4, 1], [-6, 5, -1, 1]],
                             [[8, 2, 2, 2], [3, -1, 7, 0], [7, 0, 5, 4], [6, -2, 0, 5]])
