"""rt_colour_at_noalloc (every device buffer from the library's static __device__ array) against
rt_colour_at (hipMalloc'ed workspace): the same kernels, so every f64 must be identical."""
import ctypes as C
import os

import numpy as np
import pytest

import raytracer_amd as ra

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rays_towards(n, seed, centre, radius):
    rng = np.random.default_rng(seed)
    o = rng.normal(size=(n, 3))
    o = o / np.linalg.norm(o, axis=1, keepdims=True) * radius + np.array(centre)
    o[:, 1] = np.abs(o[:, 1]) + 0.2
    d = np.array(centre) + rng.uniform(-2, 2, size=(n, 3)) - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.ascontiguousarray(np.concatenate([o, d], axis=1))


def test_noalloc_entry_point_matches_colour_at_twice_in_a_row():
    import torch

    lib = ra.hip_lib()
    g = ra.Scene.from_file(os.path.join(ROOT, "scenes", "chapter11.yaml"))
    scene = g.upload()
    dev = torch.device("cuda", 0)
    # two calls in a row on the same static array, the second with a smaller input (and a larger depth)
    for n, depth, seed in ((30000, 4, 1), (7000, 6, 2)):
        rays = rays_towards(n, seed, (0.0, 1.0, 0.0), 6.0)
        want = g.colour_at(rays, depth)
        assert (want != 0).any(axis=1).mean() > 0.5
        d_rays = torch.from_numpy(rays).to(dev)
        d_out = torch.full((n, 3), -1.0, dtype=torch.float64, device=dev)
        torch.cuda.synchronize()
        rc = lib.rt_colour_at_noalloc(scene, n, d_rays.data_ptr(), depth, d_out.data_ptr())
        assert rc == 0, lib.rt_last_error().decode()
        torch.cuda.synchronize()
        got = d_out.cpu().numpy()
        assert np.array_equal(got.view(np.uint64), want.view(np.uint64)), (n, depth)
    # limits are refused with a clear message, and refusing leaves the entry point usable
    too_many = lib.rt_noalloc_max_rays() + 1
    assert lib.rt_colour_at_noalloc(scene, too_many, d_rays.data_ptr(), 1, d_out.data_ptr()) == ra.RT_ERR_CAPACITY
    assert b"at most" in lib.rt_last_error()
    big = rays_towards(60000, 3, (0.0, 1.0, 0.0), 6.0)
    d_big = torch.from_numpy(big).to(dev)
    d_big_out = torch.empty((60000, 3), dtype=torch.float64, device=dev)
    rc = lib.rt_colour_at_noalloc(scene, 60000, d_big.data_ptr(), 200, d_big_out.data_ptr())
    assert rc in (0, ra.RT_ERR_CAPACITY)
    if rc != 0:
        assert b"ray records" in lib.rt_last_error()
    area = ra.Scene.from_file(os.path.join(ROOT, "scenes", "area-light.yaml"), 0.1)
    assert lib.rt_colour_at_noalloc(area.upload(), 10, d_rays.data_ptr(), 1, d_out.data_ptr()) == ra.RT_ERR_UNSUPPORTED
    rc = lib.rt_colour_at_noalloc(scene, n, d_rays.data_ptr(), depth, d_out.data_ptr())
    assert rc == 0, lib.rt_last_error().decode()
    torch.cuda.synchronize()
    assert np.array_equal(d_out.cpu().numpy().view(np.uint64), want.view(np.uint64))
