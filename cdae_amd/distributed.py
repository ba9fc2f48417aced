"""Data-parallel exchange for the CDAE hot path: one process per GPU, torch.distributed (RCCL over xGMI).

The reference is single-process and strictly sequential (cdae.hpp:136-146), so this layer has no
reference counterpart.  Users are sharded across ranks (each rank holds only its own rows of the
interaction matrix and its own Wu / Wu_ag rows — the north star's "V_u stays local"); the item-side
parameters W, W_ag, (V, V_ag), b', b'_ag, b, b_ag are replicated.  One exchange step:

    begin()   snapshot the shared block                                  (cdae_hip_delta_begin)
    ...       every rank trains its batch of users from that snapshot    (cdae_hip_train_users)
    finish()  delta = current - snapshot                                 (cdae_hip_delta_compute)
              all-reduce(sum) of ONE contiguous fp32 buffer [delta | touch]   <- the only collective
              current = snapshot + combine(sum)                          (cdae_hip_delta_apply)

`PipelinedDeltaExchange` is the overlapped form used by bench.py for N > 1: every `period` batches the rank stages
its delta (cdae_hip_delta_stage), starts ONE asynchronous all-reduce of it and keeps training; the other ranks' part of
the sum is merged (cdae_hip_delta_merge) at the next period boundary, i.e. one period late.  All ranks hold
initial + sum of all staged deltas once flush() has run.  The all-reduce of the ~22 MB shared block (ML-10M, K=200)
takes about as long over xGMI as a 512-user batch computes, so the synchronous form cannot scale past ~50 %.

The delta of W is the accumulated -lr * AdaGrad-preconditioned gradient of the rank's examples and the
delta of W_ag the accumulated squared gradient, so summing them is the data-parallel "all-reduce of the
shared gradients"; with world_size == 1 the step is the identity.  `combine_reference` restates the
device kernel in torch ops: the gloo CPU tests run it, the GPU test checks the HIP kernel against it.
"""
from __future__ import annotations

import numpy as np

RULE_SUM = 0
RULE_TOUCH_MEAN = 1


def shard_bounds(num_users: int, world: int, rank: int, row_ptr=None):
    """Contiguous user range of `rank`.  With row_ptr, ranges are balanced by interactions (nnz), not by
    user count."""
    if row_ptr is None:
        per = (num_users + world - 1) // world
        return min(num_users, rank * per), min(num_users, (rank + 1) * per)
    nnz = int(row_ptr[num_users])
    cuts = [int(np.searchsorted(row_ptr, nnz * r / world, side="left")) for r in range(world + 1)]
    cuts[0], cuts[-1] = 0, num_users
    return cuts[rank], cuts[rank + 1]


def combine_reference(base, summed, touch_sum, n_matrix: int, Kp: int, num_items: int, world: int, rule: int):
    """torch restatement of apply_delta_kernel (cdae_amd/csrc/cdae_kernels.hpp)."""
    import torch
    if rule == RULE_SUM:
        return base + summed
    w = torch.ones_like(summed)
    t = torch.clamp(touch_sum, min=1.0)
    n_total = base.numel()
    n_mats = n_matrix // (num_items * Kp)
    w[:n_matrix] = (1.0 / t).repeat_interleave(Kp).repeat(n_mats)
    w[n_matrix:n_total - 2 * Kp] = (1.0 / t).repeat(2)
    w[n_total - 2 * Kp:] = 1.0 / world
    return base + summed * w


class _DeviceBuffer:
    """Zero-copy view of library-owned device memory for torch.as_tensor."""

    def __init__(self, ptr: int, count: int):
        self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 2}


class DeltaExchange:
    """GPU path: the delta buffer lives in the library; torch only wraps it (zero copy) for the RCCL all-reduce,
    which is enqueued on the library's own HIP stream (torch.cuda.ExternalStream) — the whole step
    snapshot -> train -> delta -> all-reduce -> apply is stream-ordered and never blocks the host."""

    def __init__(self, model, dist, world: int, rule: int = RULE_SUM):
        import torch
        self.model, self.dist, self.world, self.rule = model, dist, world, rule
        model.delta_begin()
        ptr, count = model.delta_device_ptr()
        dev = torch.device("cuda", torch.cuda.current_device())
        self.buf = torch.as_tensor(_DeviceBuffer(ptr, count), device=dev)
        assert self.buf.data_ptr() == ptr, "torch copied the delta buffer instead of wrapping it"
        self.stream = torch.cuda.ExternalStream(model.stream_handle(), device=dev)
        self.torch = torch

    def begin(self):
        self.model.delta_begin()

    def finish(self):
        self.model.delta_compute()
        if self.world > 1:
            with self.torch.cuda.stream(self.stream):
                self.dist.all_reduce(self.buf, op=self.dist.ReduceOp.SUM)
        self.model.delta_apply(self.world, self.rule)


class PipelinedDeltaExchange:
    """GPU path, overlapped: stage -> async all-reduce (RCCL's own stream) -> train `period` more batches -> merge."""

    def __init__(self, model, dist, world: int, period: int = 2):
        import torch
        self.model, self.dist, self.world, self.period = model, dist, world, max(1, int(period))
        self.torch = torch
        model.delta_begin()
        model.delta_stage()                      # allocates the receive buffer (stages a zero delta)
        ptr, count = model.delta_recv_device_ptr()
        dev = torch.device("cuda", torch.cuda.current_device())
        self.recv = torch.as_tensor(_DeviceBuffer(ptr, count), device=dev)
        assert self.recv.data_ptr() == ptr, "torch copied the receive buffer instead of wrapping it"
        self.stream = torch.cuda.ExternalStream(model.stream_handle(), device=dev)
        self.work = None
        self.batches = 0

    def after_batch(self):
        """Call once after every enqueued batch."""
        self.batches += 1
        if self.batches % self.period == 0:
            self._boundary(start_next=True)

    def _boundary(self, start_next: bool):
        with self.torch.cuda.stream(self.stream):          # "current stream" = the library's stream
            pending = self.work is not None
            if pending:
                self.work.wait()                           # stream-level wait, the host does not block
                self.work = None
            if pending and start_next:
                self.model.delta_merge_stage()             # one pass over the block instead of two
            elif pending:
                self.model.delta_merge()
            elif start_next:
                self.model.delta_stage()
            if start_next:
                if self.dist is not None:                  # also with one rank: same stream semantics, RCCL no-op
                    self.work = self.dist.all_reduce(self.recv, op=self.dist.ReduceOp.SUM, async_op=True)
                else:
                    self.work = _Done()

    def flush(self):
        """Stage what is left, reduce it and merge: afterwards every rank holds the same shared parameters."""
        self._boundary(start_next=True)
        self._boundary(start_next=False)

    def time_all_reduce(self, repeats: int = 5) -> float:
        """Seconds per all-reduce of the exchange buffer, measured with the device otherwise idle (call between
        boundaries: the buffer's contents are summed `repeats` + 2 times, so only use it before the first stage that
        matters, e.g. during warm-up)."""
        torch = self.torch
        if self.dist is None:
            return 0.0
        with torch.cuda.stream(self.stream):
            if self.work is not None:
                self.work.wait()
            for _ in range(2):
                self.dist.all_reduce(self.recv, op=self.dist.ReduceOp.SUM)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(self.stream)
            for _ in range(repeats):
                self.dist.all_reduce(self.recv, op=self.dist.ReduceOp.SUM)
            b.record(self.stream)
            b.synchronize()
            return a.elapsed_time(b) * 1e-3 / repeats

    def choose_period(self, step_seconds: float, lo: int = 1, hi: int = 8, slack: float = 1.5):
        """Smallest period whose training time covers one all-reduce (x slack: the collective shares the chip with the
        kernels it overlaps); agreed across ranks (MAX).  Call it right after flush(): it restages a zero delta.  Returns
        (period, seconds per all-reduce)."""
        torch = self.torch
        t_ar = self.time_all_reduce()
        want = int(min(hi, max(lo, -(-slack * t_ar // max(step_seconds, 1e-9)))))
        if self.dist is not None and self.world > 1:
            t = torch.tensor([want], dtype=torch.int32, device=self.recv.device)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            want = int(t.item())
        self.period = want
        self.batches = 0
        self.work = None
        with torch.cuda.stream(self.stream):
            self.model.delta_begin()                       # fresh base; the timing runs left garbage in the receive buffer
            self.model.delta_stage()
        return want, t_ar


class _Done:
    def wait(self):
        return True


class HostPipelinedDeltaExchange:
    """PipelinedDeltaExchange's protocol on host tensors (gloo): the CPU tests' stand-in for the device kernels."""

    def __init__(self, get_shared, set_shared, dist, world: int, period: int = 2):
        self.get_shared, self.set_shared, self.dist, self.world = get_shared, set_shared, dist, world
        self.period = max(1, int(period))
        self.base = get_shared().clone()                        # A: the state every replica agrees on, bit for bit
        self.snap = None                                        # parameters when the last delta was staged
        self.recv = None
        self.work = None
        self.batches = 0

    def after_batch(self):
        self.batches += 1
        if self.batches % self.period == 0:
            self._boundary(True)

    def _boundary(self, start_next: bool):
        if self.recv is not None:                               # delta_pipe_kernel<MERGE>: A += recv ; cur = A + (cur - snap)
            if self.work is not None:
                self.work.wait()
            self.base = self.base + self.recv
            self.set_shared(self.base + (self.get_shared() - self.snap))
            self.recv = self.snap = self.work = None
        if start_next:                                          # delta_pipe_kernel<STAGE>: send = recv = cur - A ; snap = cur
            cur = self.get_shared()
            self.recv = cur - self.base
            self.snap = cur.clone()
            self.work = self.dist.all_reduce(self.recv, op=self.dist.ReduceOp.SUM, async_op=True) if self.world > 1 else None

    def flush(self):
        self._boundary(True)
        self._boundary(False)


class HostDeltaExchange:
    """The same exchange protocol over host tensors (any torch.distributed backend, e.g. gloo).

    `get_shared()` / `set_shared(t)` move the flat shared block [matrices | bp | bp_ag | b | b_ag]; `touched()`
    returns the per-item 0/1 indicator of this step.  Used by the multi-process CPU tests to exercise the
    sharding + all-reduce + combine logic without a GPU; the GPU path (DeltaExchange) runs the identical
    protocol with the buffers and the combine kernel inside libcdae_hip.so.
    """

    def __init__(self, get_shared, set_shared, touched, dist, world: int, n_matrix: int, Kp: int, num_items: int,
                 rule: int = RULE_SUM):
        self.get_shared, self.set_shared, self.touched = get_shared, set_shared, touched
        self.dist, self.world, self.rule = dist, world, rule
        self.n_matrix, self.Kp, self.I = n_matrix, Kp, num_items
        self.base = None

    def begin(self):
        self.base = self.get_shared().clone()

    def finish(self):
        import torch
        cur = self.get_shared()
        buf = torch.cat([cur - self.base, self.touched().to(cur.dtype)])
        if self.world > 1:
            self.dist.all_reduce(buf, op=self.dist.ReduceOp.SUM)
        n = cur.numel()
        self.set_shared(combine_reference(self.base, buf[:n], buf[n:], self.n_matrix, self.Kp, self.I, self.world, self.rule))
