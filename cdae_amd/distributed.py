"""Host helpers around the library's multi-GPU layouts (cdae_hip_multi_*, cdae_hip_comm_*, cdae_hip_exchange_*: cdae_multi.hip).

The exchange itself — staging, the RCCL all-reduce, merging, the item-rows phases — runs inside the library; the reference is
single-process and strictly sequential (cdae.hpp:136-146), so none of this has a reference counterpart.  What remains here:
`shard_bounds` (the contiguous, interaction-balanced user ranges a one-process-per-GPU host cuts its data by: bench.py) and a
zero-copy torch view of library-owned device buffers for tests and tools.
"""
from __future__ import annotations

import numpy as np

RULE_SUM = 0
RULE_TOUCH_MEAN = 1


def shard_bounds(num_users: int, world: int, rank: int, row_ptr=None):
    """Contiguous user range of `rank`.  With row_ptr, ranges are balanced by interactions (nnz), not by user count
    (SURVEY.md §8(e)): range r starts at the first user whose prefix reaches r/world of the interactions and keeps at least
    one user — cdae_exchange_algebra.h's balanced_cuts, the rule the library's own shards are cut by."""
    if row_ptr is None:
        per = (num_users + world - 1) // world
        return min(num_users, rank * per), min(num_users, (rank + 1) * per)
    total = int(row_ptr[num_users])
    cuts = [0]
    for s in range(1, world):
        want = (total * s + world - 1) // world
        u = int(np.searchsorted(row_ptr[:num_users + 1], want, side="left"))
        cuts.append(min(max(u, cuts[-1] + 1), num_users - (world - s)))
    cuts.append(num_users)
    return cuts[rank], cuts[rank + 1]


class _DeviceBuffer:
    """Zero-copy view of library-owned device memory for torch.as_tensor (tests, tools)."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


def wrap_device_floats(ptr: int, count: int, device_index: int = 0):
    """torch tensor over `count` floats the library owns at `ptr` (no copy)."""
    import torch
    t = torch.as_tensor(_DeviceBuffer(ptr, count), device=torch.device("cuda", device_index))
    assert t.data_ptr() == ptr, "torch copied the buffer instead of wrapping it"
    return t
