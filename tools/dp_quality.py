"""Oracle simulation: Recall@10 of data-parallel schedules (sum rule) on the 'small' shape."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from cdae_amd import synth
import oracle as orc
from oracle import binding as ob

SHARED = [0, 1, 8, 9, 6, 7]
K, B, EPOCHS = 32, 64, 6
data = synth.generate_shape("small", seed=5)

def make():
    o = orc.Oracle(orc.OracleConfig(num_dim=K, loss_type=ob.LOSS_CE, beta=1.0), data.num_users, data.num_items, data.train_ptr, data.train_col)
    o.init_params(5)
    return o
def get(o): return np.concatenate([o.get(w).ravel() for w in SHARED])
def put(o, a):
    off = 0
    for w in SHARED:
        n = o.get(w).size; o.set(w, a[off:off + n]); off += n
def recall(o):
    rec = o.recommend(10)
    return float(orc.eval_topn(rec, data.test_ptr, data.test_col)[0])

def run(world, mode, period=1):
    reps = [make() for _ in range(world)]
    per = (data.num_users + world - 1) // world
    bounds = [(r * per, min(data.num_users, (r + 1) * per)) for r in range(world)]
    out = []
    base = [get(o) for o in reps]
    inflight = None
    def boundary(start):
        nonlocal inflight
        if inflight is not None:
            tot = sum(inflight)
            for r, o in enumerate(reps):
                p = tot - inflight[r]; put(o, get(o) + p); base[r] = base[r] + p
            inflight = None
        if start:
            inflight = []
            for r, o in enumerate(reps):
                c = get(o); inflight.append(c - base[r]); base[r] = c.copy()
    for ep in range(EPOCHS):
        steps = (per + B - 1) // B
        for s in range(steps):
            if mode == "sync":
                b0 = get(reps[0]); ds = []
                for r, o in enumerate(reps):
                    u0, u1 = bounds[r]; a = u0 + s * B
                    if a < u1: o.train_batched(9, ep, B, a, min(u1, a + B))
                    ds.append(get(o) - b0)
                new = b0 + sum(ds)
                for o in reps: put(o, new)
            else:
                for r, o in enumerate(reps):
                    u0, u1 = bounds[r]; a = u0 + s * B
                    if a < u1: o.train_batched(9, ep, B, a, min(u1, a + B))
                if (s + 1) % period == 0: boundary(True)
        if mode != "sync":
            boundary(True); boundary(False)
        # Wu is private: evaluate with rank r's Wu rows for its users -> copy rows into replica 0
        for r in range(1, world):
            u0, u1 = bounds[r]
            for w in (4, 5):
                a = reps[0].get(w).reshape(data.num_users, -1); b = reps[r].get(w).reshape(data.num_users, -1)
                a[u0:u1] = b[u0:u1]; reps[0].set(w, a.ravel())
        out.append(round(recall(reps[0]), 4))
    return out

for world, mode, period in [(1, "sync", 1), (2, "sync", 1), (2, "pipe", 2), (8, "sync", 1), (8, "pipe", 1), (8, "pipe", 2), (8, "pipe", 4)]:
    print(world, mode, period, run(world, mode, period), flush=True)
