"""Row-block split + gather on two CPU processes (gloo).  The ranks 'render' their rows with the
CPU oracle here; on GPUs the same partition / gather code runs over RCCL (bench.py)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import known_answers as ka
    import oracle_lib as ol
    import raytracer_amd as ra
    from raytracer_amd import multi

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    text = ka.TEST_WORLD.replace("width: 11", "width: 37").replace("height: 11", "height: 45")
    o = ol.OracleScene.from_text(text)
    part = multi.partition_for(rank, world)
    rows = ra.local_rows(o.height, part)
    local = torch.from_numpy(ol.to_u8(o.render_row_list(5, rows.astype(np.uint32), None, threads=1)))
    frame = multi.gather_rows(local, o.height, rank, world)
    if rank == 0:
        full = ol.to_u8(o.render_rows(5, None, threads=1))
        ok = frame is not None and np.array_equal(frame.numpy(), full)
        open(os.path.join(tmp, "result"), "w").write("ok" if ok else "mismatch")
    else:
        assert frame is None
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_row_split_reassembles_the_frame(tmp_path):
    port = 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "result").read() == "ok"


def test_partition_rows_cover_the_image_once():
    import raytracer_amd as ra
    from raytracer_amd import multi
    for height in (1, 7, 8, 45, 320, 5000):
        for world in (1, 2, 4, 8):
            seen = np.concatenate([ra.local_rows(height, multi.partition_for(r, world)) for r in range(world)])
            assert sorted(seen.tolist()) == list(range(height))
            assert multi.max_local_rows(height, world) >= height / world
