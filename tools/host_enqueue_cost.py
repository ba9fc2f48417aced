import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cdae_amd
from cdae_amd import synth
d = synth.generate_shape("ml10m")
B = 512
m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=B))
m.set_interactions(d.num_users, d.num_items, d.train_ptr, d.train_col)
m.init_params(1)
nbat = d.num_users // B
for rep in range(2):
    m.synchronize()
    t0 = time.perf_counter(); cpu = 0.0
    for i in range(nbat):
        a = time.perf_counter()
        m.enqueue_users(1, rep, i * B, (i + 1) * B)
        m.prefetch_users(1, rep, ((i + 1) % nbat) * B, ((i + 1) % nbat + 1) * B)
        cpu += time.perf_counter() - a
    t1 = time.perf_counter()
    m.synchronize()
    t2 = time.perf_counter()
    print(f"rep {rep}: host enqueue {cpu / nbat * 1e6:.1f} us/step, loop wall {(t1 - t0) / nbat * 1e6:.1f} us/step, total {(t2 - t0) / nbat * 1e6:.1f} us/step")
