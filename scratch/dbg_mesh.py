import sys, os, ctypes as C
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import raytracer_amd as ra
import test_gpu_parity as T
ra.init(0)
g, _ = T.pair("scenes/chapter15.yaml", 0.2)
d = g.desc(); lib = ra.hip_lib(); cam = g.camera_ptr(); W, H = g.width, g.height
kind, skip = d["node_kind"], d["node_skip"]
depth_of = np.zeros(len(kind), dtype=np.int64); open_ = []
for i in range(len(kind)):
    while open_ and skip[open_[-1]] <= i: open_.pop()
    depth_of[i] = len(open_)
    if kind[i] != 0: open_.append(i)
print("nodes", len(kind), "groups", (kind==1).sum(), "max depth", depth_of.max(), "top-level", (depth_of==0).sum(), kind[depth_of==0])
bb = d["node_bbox"].copy()
inner = (kind == 1) & (depth_of >= 1) & np.isfinite(bb).all(axis=1)
c = 0.5*(bb[:, :3]+bb[:, 3:]); e = 0.5*(bb[:, 3:]-bb[:, :3])
bb[inner, :3] = (c-0.7*e)[inner]; bb[inner, 3:] = (c+0.7*e)[inner]
d2 = dict(d); d2["node_bbox"] = bb
for depth in (0, 5):
    fr = []
    for mode in (3, 2):
        h, keep = T._create_from_arrays(d2)
        lib.rt_set_option(h, ra.OPT_TRAVERSAL, mode)
        rgb = np.zeros((H, W, 3)); st = ra.Stats()
        rc = lib.rt_render(h, cam, depth, None, None, rgb.ctypes.data_as(C.c_void_p), None, 0, C.byref(st))
        assert rc == 0
        lib.rt_scene_destroy(h)
        fr.append((rgb, st.as_dict()))
    a, b = fr[0][0], fr[1][0]
    diff = (a != b).any(axis=2)
    print("depth", depth, "differing px", diff.sum(), {k: (fr[0][1][k], fr[1][1][k]) for k in ("shade_hits", "shadow_rays", "secondary_rays")})
    ys, xs = np.nonzero(diff)
    for y, x in list(zip(ys, xs))[:6]:
        print(" ", x, y, a[y, x], b[y, x])
