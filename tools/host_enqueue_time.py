#!/usr/bin/env python
"""Developer aid: is the training loop bound by the HOST's launch rate?  Enqueues N batches (cdae_hip_enqueue_users returns as
soon as the launches are queued) and reports host enqueue time vs time to completion, per batch.

    python tools/host_enqueue_time.py [batch_users] [batches]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import cdae_amd  # noqa: E402
from cdae_amd import synth  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 200
d = synth.generate_shape("ml10m")
m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=B))
m.set_interactions(d.num_users, d.num_items, d.train_ptr, d.train_col)
m.init_params(1)
m.enqueue_users(1, 0, 0, 40 * B)
m.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    m.enqueue_users(1, 0, 40 * B, (40 + N) * B)
    t1 = time.perf_counter()
    m.synchronize()
    t2 = time.perf_counter()
    print(f"B={B} batches={N}: host enqueue {1e6 * (t1 - t0) / N:.1f} us/batch, until done {1e6 * (t2 - t0) / N:.1f} us/batch")
