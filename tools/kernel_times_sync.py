"""Per-kernel-family HIP-event times with NO prep/compute overlap: one batch per synchronous train_users call.

    python tools/kernel_times_sync.py cdae_amd/lib/libcdae_hip.so [batch_users]
"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cdae_amd import binding, synth
import cdae_amd
binding.load_library(sys.argv[1])
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
d = synth.generate_shape("ml10m")
m = cdae_amd.CDAE(cdae_amd.CDAEConfig(num_dim=200, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=B))
m.set_interactions(d.num_users, d.num_items, d.train_ptr, d.train_col)
m.init_params(1)
m.set_profiling(True)
acc = {}
n = 0
for i in range(60):
    st = m.train_users(1, 0, i * B, (i + 1) * B).as_dict()
    if i < 10:
        continue
    n += 1
    for k, v in st.items():
        if k.startswith("ms_"):
            acc[k] = acc.get(k, 0.0) + v
print(os.path.basename(sys.argv[1]), os.environ.get("CDAE_DECODE_ONE_ROW_PER_WAVE", "-"), os.environ.get("CDAE_DECODE_HOT_POS", "-"),
      {k: round(v / n, 4) for k, v in acc.items()})
