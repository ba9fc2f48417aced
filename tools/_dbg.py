import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cdae_amd
from cdae_amd import synth
d = synth.generate(600, 33_000, 36_000, seed=6, min_items=20)
def run(env, B=256, epochs=2):
    for k in ("CDAE_FULL_ROWS_KH", "CDAE_FULL_ROWS_SEPARATE", "CDAE_FULL_ROWS_DT"): os.environ.pop(k, None)
    os.environ.update(env)
    cfg = cdae_amd.CDAEConfig(num_dim=300, lt=cdae_amd.CROSS_ENTROPY, beta=1.0, batch_users=B, full_output=True)
    m = cdae_amd.CDAE(cfg); m.reset(d, seed=4)
    for e in range(epochs): m.train_one_iteration(4, e)
    out = {w: m.get(w) for w in (0, 1, 8, 9)}
    m.close(); return out
seq = [("f640", {}, 640), ("f256", {}, 256), ("dt256", {"CDAE_FULL_ROWS_DT": "1"}, 256), ("kh1", {"CDAE_FULL_ROWS_KH": "1"}, 256), ("f256b", {}, 256), ("sep", {"CDAE_FULL_ROWS_SEPARATE": "1"}, 256)]
res = {n: run(e, B) for n, e, B in seq}
ref = res["f256"]
for n in ("dt256", "kh1", "f256b", "sep"):
    x = res[n]
    print(n, {w: float(np.abs(x[w] - ref[w]).max()) for w in x}, {w: int((x[w] != ref[w]).sum()) for w in x})
    bad = np.nonzero(x[8] != ref[8])[0]
    print("   b' differs at", bad[:10], "n", bad.size, "of", x[8].size)
