"""Samples the reference's own renders (reference images/*.png, made with
`--scale 3`) into small fixtures: for every scene a set of evenly spaced whole
rows of the 8-bit image.  Whole rows are kept because the reference threads one
rng stream through each row (src/camera.rs:143-157).

Run here (needs /root/reference and PIL); the .npz files are committed and the
tests read only those.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
from PIL import Image

REF = "/root/reference/images"
OUT = os.path.dirname(os.path.abspath(__file__))
ROWS = 32

# scene name -> png; render flags that reproduce the png are recorded in tests/golden_scenes.py
SCENES = ["chapter6", "chapter8", "chapter9", "chapter11", "chapter11-water", "chapter12", "chapter13",
          "chapter14", "chapter14-spheres", "chapter15", "chapter16", "bounding-box", "area-light"]

for name in SCENES:
    img = np.asarray(Image.open(os.path.join(REF, name + ".png")).convert("RGB"))
    h, w = img.shape[:2]
    rows = np.unique(np.linspace(0, h - 1, ROWS).astype(np.uint32))
    np.savez_compressed(os.path.join(OUT, name + ".npz"), width=np.uint32(w), height=np.uint32(h), rows=rows,
                        rgb8=img[rows])
    print(name, w, h, len(rows), "rows")
