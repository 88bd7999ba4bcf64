"""Row-block split of one frame over the GPUs of a node.

Pixels are independent except that one rng stream spans an image row
(reference src/camera.rs:143-157), so the frame is split by whole rows:
blocks of `block_rows` rows are dealt round-robin to the ranks (reflective
regions are spatially clustered, interleaving balances them).  The scene is
replicated per rank; nothing is exchanged while tracing.  The only collective
is the gather of the per-rank row blocks to rank 0, which then restores row
order.  Works with any torch.distributed backend ("nccl" = RCCL on GPUs,
"gloo" on CPU for tests).
"""
import torch
import torch.distributed as dist

from . import Partition, local_rows

BLOCK_ROWS = 8


def partition_for(rank, world, block_rows=BLOCK_ROWS):
    return Partition(block_rows, rank, world)


def max_local_rows(height, world, block_rows=BLOCK_ROWS):
    return max(len(local_rows(height, partition_for(r, world, block_rows))) for r in range(world))


_row_index_cache = {}


def _row_indices(height, world, block_rows, device):
    """Per rank, the frame rows it renders, as an index tensor on `device` (built once per frame shape)."""
    key = (height, world, block_rows, str(device))
    if key not in _row_index_cache:
        _row_index_cache[key] = [torch.as_tensor(local_rows(height, partition_for(r, world, block_rows)), device=device)
                                 for r in range(world)]
    return _row_index_cache[key]


def gather_rows(local, height, rank, world, block_rows=BLOCK_ROWS, dst=0):
    """Gather per-rank row tensors (rows_local, width, C) to `dst` and restore row order.

    Returns the (height, width, C) frame on `dst`, None elsewhere.
    """
    if world == 1:
        return local
    # gloo only moves host memory reliably: a rehearsal of the multi-rank path on GPUs without RCCL (ranks that
    # share one GPU) stages through the host, with synchronous copies
    device = local.device
    via_host = dist.get_backend() == "gloo" and local.is_cuda
    if via_host:
        local = local.cpu()
    pad_rows = max_local_rows(height, world, block_rows)
    padded = local
    if local.shape[0] != pad_rows:
        padded = torch.zeros((pad_rows,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        padded[: local.shape[0]] = local
    padded = padded.contiguous()
    parts = None
    if rank == dst:
        parts = [torch.empty_like(padded) for _ in range(world)]
    dist.gather(padded, parts, dst=dst)
    if rank != dst:
        return None
    frame = torch.empty((height,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    for r, rows in enumerate(_row_indices(height, world, block_rows, local.device)):
        frame.index_copy_(0, rows, parts[r][: rows.shape[0]])
    return frame.to(device) if via_host else frame
