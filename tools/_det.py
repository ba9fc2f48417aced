import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import cdae_amd
from cdae_amd import synth
from test_gpu_multi import emulate, cfg_of, SHARED
d = synth.generate(1200, 500, 60_000, seed=9)
cfg = cfg_of()
def multi(period, shards):
    mm = cdae_amd.MultiCDAE(cfg, devices=[0]*shards, exchange_every=period); mm.reset(d, seed=11)
    for ep in range(2): mm.train_one_iteration(3, ep)
    return mm
for period in (0, 2):
    a = multi(period, 2); b = multi(period, 2)
    print('multi vs multi', period, [int((a.get(w) != b.get(w)).sum()) for w in SHARED])
    cuts = a.shards()
    e1 = emulate(d, cfg, cuts, 11, 3, 2, period); e2 = emulate(d, cfg, cuts, 11, 3, 2, period)
    print('emu vs emu', period, [int((e1[0].get(w) != e2[0].get(w)).sum()) for w in SHARED])
    print('multi vs emu', period, [int((a.get(w) != e1[0].get(w)).sum()) for w in SHARED], 'Wu', int((a.get(4) != np.concatenate([r.get(4) for r in e1])).sum()))
# after ONE step only
for steps in (1, 2):
    import ctypes as C
    mm = cdae_amd.MultiCDAE(cfg, devices=[0,0], exchange_every=0); mm.reset(d, seed=11)
