"""The CPU oracle against rows sampled from the reference's own renders (tests/golden/*.npz)."""
import os

import numpy as np
import pytest

import oracle_lib as ol
from golden_scenes import GOLDEN, SCALE, SEED

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_oracle(name):
    cfg = GOLDEN[name]
    if cfg["scene"] is None:
        s = ol.OracleScene.spheres(SCALE, SEED)
    else:
        s = ol.OracleScene.from_file(os.path.join(ROOT, cfg["scene"]), SCALE, SEED)
    if cfg.get("half_view_ulps"):
        s.nudge_half_view(cfg["half_view_ulps"])
    return s, cfg


@pytest.mark.parametrize("name", sorted(GOLDEN))
def test_oracle_reproduces_reference_render(name):
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    s, cfg = load_oracle(name)
    assert (s.width, s.height) == (int(g["width"]), int(g["height"]))
    seeds = s.row_seeds()
    rows = g["rows"]
    if name == "bounding-box":
        rows = rows[::2]  # 141k triangles: keep the CPU suite quick
    got = ol.to_u8(s.render_row_list(cfg["depth"], rows, seeds, threads=8))
    want = g["rgb8"][::2] if name == "bounding-box" else g["rgb8"]
    mismatch = int((got != want).any(axis=2).sum())
    assert mismatch <= cfg.get("max_mismatch", 0), f"{name}: {mismatch} differing pixels of {want.shape[0] * want.shape[1]}"
