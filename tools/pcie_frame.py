#!/usr/bin/env python3
"""Times the headline frame through host buffers (rt_render with out_on_device = 0): the frame is staged
on the device and copied out over PCIe.  Prints wall time per frame for the 8-bit image alone and with the
f64 canvas as well."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import raytracer_amd as ra  # noqa: E402

ra.init(0)
scene = ra.Scene.from_file(os.path.join(ROOT, "scenes/chapter11.yaml"), 10.0, 0)
seeds = scene.row_seeds()
import ctypes as C  # noqa: E402

import numpy as np  # noqa: E402

W, H = scene.width, scene.height
rgb = np.zeros((H, W, 3), dtype=np.float64)
rgb8 = np.zeros((H, W, 3), dtype=np.uint8)
rgb[:] = 1.0  # touch the pages once: a caller's canvas is reused from frame to frame
lib = ra.host_lib()
for want_f64 in (False, True):
    stats = ra.Stats()

    def frame():
        rc = lib.rth_render(scene.h, 30, seeds.ctypes.data_as(C.c_void_p), None,
                            rgb.ctypes.data_as(C.c_void_p) if want_f64 else None, rgb8.ctypes.data_as(C.c_void_p), 0,
                            C.byref(stats))
        assert rc == 0
    frame()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        frame()
    dt = (time.perf_counter() - t0) / n
    rays = stats.primary_rays + stats.secondary_rays
    print("f64 canvas %-5s  %.1f ms per frame  %.0f Mrays/s" % (want_f64, dt * 1e3, rays / dt / 1e6))
