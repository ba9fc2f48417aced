#!/usr/bin/env python
"""Wall-clock of the per-epoch evaluation path next to training: what one iteration of the
reference's solver loop (train_one_iteration + current_loss + TOPN evaluation, solver-inl.hpp:53-69) costs on the GPU.

    python tools/eval_bench.py            # ML-10M shape, K=200
"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cdae_amd
from cdae_amd import synth
d = synth.generate_shape("ml10m")
m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=512))
m.set_interactions(d.num_users, d.num_items, d.train_ptr, d.train_col)
m.init_params(1)
m.train_one_iteration(1, 0)
for name, fn in (("recommend_all(top10)", lambda: m.recommend_all(10)), ("data_loss", lambda: m.data_loss(1, 1)), ("penalty_loss", lambda: m.penalty_loss()),
                 ("train_epoch", lambda: m.train_one_iteration(1, 1))):
    fn()
    t0 = time.perf_counter(); fn(); dt = time.perf_counter() - t0
    print(f"{name:24s} {dt*1e3:9.2f} ms  ({d.num_users/dt/1e6:.2f} M users/s)")
