"""How each golden fixture (tests/golden/<name>.npz, rows sampled from the
reference's own images/<name>.png) is reproduced: scene source, recursion
depth, rng seed and the notes the comparison needs.

The reference does not record its render flags.  They were found by matching:
`--scale 3` everywhere; `--depth 5` (the CLI default, src/arguments.rs:33-35)
except chapter11 and the sphere scene, which match only with `--depth 20`;
`--seed 0` for the two rng-dependent images.

half_view_ulps = -1: the reference's images were rendered with a libm whose
tan(pi/6) is one ulp below glibc's (correctly rounded) value; with that single
input substituted chapter13 and chapter15 reproduce with zero differing pixels
(385 and 1 otherwise, all on knife-edge pixels).  The other "PI / 3" cameras
reproduce with either value.

max_mismatch: bounding-box.png keeps 402 differing pixels of 3.6 M in the
full image, 397 of them in one z-fighting band where a transparent cube's
bottom face coincides with a pedestal cap; the sampled rows are allowed the
share of that band they cross.
"""
GOLDEN = {
    # name: dict(scene=yaml path or None for the generated sphere scene, depth, seed, half_view_ulps, max_mismatch)
    "chapter6": dict(scene="scenes/chapter6.yaml", depth=5),
    "chapter8": dict(scene="scenes/chapter8.yaml", depth=5),
    "chapter9": dict(scene="scenes/chapter9.yaml", depth=5),
    "chapter11": dict(scene="scenes/chapter11.yaml", depth=20),
    "chapter11-water": dict(scene="scenes/chapter11-water.yaml", depth=5),
    "chapter12": dict(scene="scenes/chapter12.yaml", depth=5),
    "chapter13": dict(scene="scenes/chapter13.yaml", depth=5, half_view_ulps=-1),
    "chapter14": dict(scene="scenes/chapter14.yaml", depth=5),
    "chapter14-spheres": dict(scene=None, depth=20),
    "chapter15": dict(scene="scenes/chapter15.yaml", depth=5, half_view_ulps=-1),
    "chapter16": dict(scene="scenes/chapter16.yaml", depth=5),
    "bounding-box": dict(scene="scenes/bounding-box.yaml", depth=5, max_mismatch=40),
    "area-light": dict(scene="scenes/area-light.yaml", depth=5),
}
SCALE = 3.0
SEED = 0
