"""CPU, world_size 2 over gloo: the data-parallel exchange protocol of cdae_amd/distributed.py.

Each rank trains its own user shard with the oracle (test infrastructure standing in for the GPU
kernels), exchanges shared-parameter deltas through HostDeltaExchange, and the result must equal a
single-process emulation of the same protocol.  Also checks shard_bounds and combine_reference.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from cdae_amd import synth  # noqa: E402
from cdae_amd.distributed import (HostDeltaExchange, HostPipelinedDeltaExchange, combine_reference,  # noqa: E402
                                  shard_bounds, RULE_SUM, RULE_TOUCH_MEAN)

SHARED = [0, 1, 8, 9, 6, 7]     # W, W_ag, bp, bp_ag, b, b_ag — the library's shared-block order (tied mode)
K, B, STEPS = 8, 16, 3


def _make_oracle(data):
    import oracle as orc
    from oracle import binding as ob
    o = orc.Oracle(orc.OracleConfig(num_dim=K, loss_type=ob.LOSS_CE, beta=1.0), data.num_users, data.num_items,
                   data.train_ptr, data.train_col)
    o.init_params(5)
    return o


def _get_shared(o):
    return torch.from_numpy(np.concatenate([o.get(w) for w in SHARED]))


def _set_shared(o, t):
    a, off = t.numpy(), 0
    for w in SHARED:
        n = o.get(w).size
        o.set(w, a[off:off + n])
        off += n


def _run_rank(rank, world, port, rule, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = synth.generate_shape("tiny", seed=5)
    u0, u1 = shard_bounds(data.num_users, world, rank, data.train_ptr)
    o = _make_oracle(data)
    I = data.num_items
    state = {"before": None}

    def touched():
        d = (_get_shared(o) - state["before"]).numpy()
        return torch.from_numpy(((np.abs(d[:I * K].reshape(I, K)).sum(1) + np.abs(d[2 * I * K:2 * I * K + I])) > 0).astype(np.float64))

    ex = HostDeltaExchange(lambda: _get_shared(o), lambda t: _set_shared(o, t), touched, dist, world,
                           n_matrix=2 * I * K, Kp=K, num_items=I, rule=rule)
    for step in range(STEPS):
        ex.begin()
        state["before"] = _get_shared(o)
        s0 = u0 + step * B
        o.train_batched(9, 0, B, s0, min(u1, s0 + B))
        ex.finish()
    np.save(os.path.join(out_dir, f"shared_{rank}.npy"), _get_shared(o).numpy())
    np.save(os.path.join(out_dir, f"wu_{rank}.npy"), o.get(4))
    dist.destroy_process_group()


def _emulate(world, rule):
    data = synth.generate_shape("tiny", seed=5)
    I = data.num_items
    reps = [_make_oracle(data) for _ in range(world)]
    bounds = [shard_bounds(data.num_users, world, r, data.train_ptr) for r in range(world)]
    for step in range(STEPS):
        base = _get_shared(reps[0])
        deltas, touches = [], []
        for r, o in enumerate(reps):
            u0, u1 = bounds[r]
            s0 = u0 + step * B
            o.train_batched(9, 0, B, s0, min(u1, s0 + B))
            d = _get_shared(o) - base
            deltas.append(d)
            dn = d.numpy()
            touches.append(torch.from_numpy(((np.abs(dn[:I * K].reshape(I, K)).sum(1) + np.abs(dn[2 * I * K:2 * I * K + I])) > 0).astype(np.float64)))
        new = combine_reference(base, sum(deltas), sum(touches), 2 * I * K, K, I, world, rule)
        for o in reps:
            _set_shared(o, new)
    return _get_shared(reps[0]).numpy(), [o.get(4) for o in reps], bounds


def _run_rank_pipelined(rank, world, port, period, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    data = synth.generate_shape("tiny", seed=5)
    u0, u1 = shard_bounds(data.num_users, world, rank, data.train_ptr)
    o = _make_oracle(data)
    ex = HostPipelinedDeltaExchange(lambda: _get_shared(o), lambda t: _set_shared(o, t), dist, world, period=period)
    for step in range(PIPE_STEPS):
        s0 = u0 + step * B
        o.train_batched(9, 0, B, s0, min(u1, s0 + B))
        ex.after_batch()
    ex.flush()
    np.save(os.path.join(out_dir, f"shared_{rank}.npy"), _get_shared(o).numpy())
    dist.destroy_process_group()


PIPE_STEPS = 5


def _emulate_pipelined(world, period):
    """Single-process restatement: rank r trains from its own replica; at every boundary the previous period's peer
    deltas are merged, then this period's own deltas are staged (they reach the peers one period later)."""
    data = synth.generate_shape("tiny", seed=5)
    reps = [_make_oracle(data) for _ in range(world)]
    bounds = [shard_bounds(data.num_users, world, r, data.train_ptr) for r in range(world)]
    base = [_get_shared(o) for o in reps]
    in_flight = None                                  # per-rank deltas staged at the previous boundary

    def boundary(start_next):
        nonlocal in_flight
        if in_flight is not None:
            total = sum(in_flight)
            for r, o in enumerate(reps):
                peers = total - in_flight[r]
                _set_shared(o, _get_shared(o) + peers)
                base[r] = base[r] + peers
            in_flight = None
        if start_next:
            in_flight = []
            for r, o in enumerate(reps):
                cur = _get_shared(o)
                in_flight.append(cur - base[r])
                base[r] = cur.clone()

    for step in range(PIPE_STEPS):
        for r, o in enumerate(reps):
            u0, u1 = bounds[r]
            s0 = u0 + step * B
            o.train_batched(9, 0, B, s0, min(u1, s0 + B))
        if (step + 1) % period == 0:
            boundary(True)
    boundary(True)
    boundary(False)
    return [_get_shared(o).numpy() for o in reps]


@pytest.mark.parametrize("period", [1, 2])
def test_two_rank_gloo_pipelined_exchange(built, tmp_path, period):
    world = 2
    mp.spawn(_run_rank_pipelined, args=(world, _free_port(), period, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"shared_{r}.npy") for r in range(world)]
    ref = _emulate_pipelined(world, period)
    np.testing.assert_allclose(got[0], got[1], rtol=1e-12, atol=1e-14)     # replicas converge after flush()
    for r in range(world):
        np.testing.assert_allclose(got[r], ref[r], rtol=1e-12, atol=1e-14)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("rule", [RULE_SUM, RULE_TOUCH_MEAN])
def test_two_rank_gloo_exchange_matches_emulation(built, tmp_path, rule):
    world = 2
    mp.spawn(_run_rank, args=(world, _free_port(), rule, str(tmp_path)), nprocs=world, join=True)
    ref_shared, ref_wu, bounds = _emulate(world, rule)
    got = [np.load(tmp_path / f"shared_{r}.npy") for r in range(world)]
    np.testing.assert_array_equal(got[0], got[1])                 # replicas stay identical
    np.testing.assert_allclose(got[0], ref_shared, rtol=1e-13, atol=1e-15)
    for r in range(world):                                        # the user node never leaves its rank
        u0, u1 = bounds[r]
        wu = np.load(tmp_path / f"wu_{r}.npy").reshape(-1, K)
        np.testing.assert_allclose(wu[u0:u1], ref_wu[r].reshape(-1, K)[u0:u1], rtol=1e-13)
        other = np.ones(wu.shape[0], bool); other[u0:u1] = False
        init = _make_oracle(synth.generate_shape("tiny", seed=5)).get(4).reshape(-1, K)
        np.testing.assert_array_equal(wu[other], init[other])


def test_shard_bounds_cover_and_balance():
    d = synth.generate_shape("tiny", seed=5)
    for world in (1, 2, 3, 8):
        cuts = [shard_bounds(d.num_users, world, r, d.train_ptr) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == d.num_users
        assert all(cuts[r][1] == cuts[r + 1][0] for r in range(world - 1))
        nnz = [d.train_ptr[b] - d.train_ptr[a] for a, b in cuts]
        assert max(nnz) - min(nnz) <= np.diff(d.train_ptr).max() * 2
        even = [shard_bounds(d.num_users, world, r) for r in range(world)]
        assert even[0][0] == 0 and even[-1][1] == d.num_users


def test_combine_reference_rules():
    I, Kp = 5, 4
    n_matrix = 2 * I * Kp
    n = n_matrix + 2 * I + 2 * Kp
    base = torch.arange(n, dtype=torch.float64)
    summed = torch.ones(n, dtype=torch.float64) * 6
    touch = torch.tensor([0., 1., 2., 3., 6.], dtype=torch.float64)
    out = combine_reference(base, summed, touch, n_matrix, Kp, I, 3, RULE_SUM)
    assert torch.equal(out, base + 6)
    out = combine_reference(base, summed, touch, n_matrix, Kp, I, 3, RULE_TOUCH_MEAN) - base
    w = 6 / torch.clamp(touch, min=1)
    assert torch.allclose(out[:I * Kp].reshape(I, Kp), w[:, None].expand(I, Kp))
    assert torch.allclose(out[I * Kp:n_matrix].reshape(I, Kp), w[:, None].expand(I, Kp))
    assert torch.allclose(out[n_matrix:n_matrix + I], w) and torch.allclose(out[n_matrix + I:n_matrix + 2 * I], w)
    assert torch.allclose(out[-2 * Kp:], torch.full((2 * Kp,), 2.0, dtype=torch.float64))
