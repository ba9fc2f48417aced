#!/usr/bin/env python
"""Fixture of the oracle's REFERENCE-SEQUENCED mode (tests/golden/ref_sequenced_tiny.npz).

    python tests/golden/make_ref_sequenced.py

One epoch of `train_one_iteration` (/root/reference/src/model/recsys/cdae.hpp:136-146) on the synthetic "tiny" data set, driven by
the reference's own generators in the reference's own order instead of the counter streams of include/cdae_rng.h:
  * `rand()` after `srand(1)` (the reference never seeds it) — first for Eigen's `DMatrix::Random` in `reset()`
    (cdae.hpp:112-120: W then Wu, row-major, `-1 + 2 rand()/RAND_MAX` per coefficient, times 4 sqrt(6 / (I + K))), then for
    `sample_negative_item` (`rand() % num_items_` until unrated, recsys_model_base.hpp:46-57, called n_u * num_neg times per user at
    cdae.hpp:217-220);
  * `std::mt19937_64` seeded with `Random::seed(MT_SEED)` + `uniform_real_distribution<>` for the dropout mask, one draw per train
    item in the visiting order of the user's `std::unordered_map<size_t,double>` (random.hpp:14,34-37; cdae.hpp:361-371), the map
    built by inserting the user's items in data order (data-inl.hpp:414-429; here: ascending item id);
  * per user-corruption the mask is drawn first (cdae.hpp:142), then the negatives (:217-220).
The file holds the initial parameters, every user's draws (visiting order of the positives, kept inputs, negatives) and the
parameters after the epoch (fp64, the oracle's literal step).  tests/test_gpu_parity.py feeds exactly these draws through
cdae_hip_train_one_user_corruption and must land on the same parameters; a reader who can build the reference (Eigen, Boost, glog,
gflags) reproduces the file's inputs with `Random::seed(MT_SEED)` and the same text file and compares
`W` / `Wu` / `b` / `b_prime` after one iteration.  The reference itself cannot be built in this image: "parity unpinned" applies.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from cdae_amd import synth  # noqa: E402
import oracle as orc  # noqa: E402
from oracle import binding as ob  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_sequenced_tiny.npz")
MT_SEED, RAND_SEED, DATA_SEED = 20141119, 1, 5
CFG = dict(num_dim=8, loss_type=ob.LOSS_CE, beta=1.0, corruption_ratio=0.5, num_neg=5, scaled=True, learn_rate=0.1, lambda_=0.01)


def main():
    d = synth.generate_shape("tiny", seed=DATA_SEED)
    mk = lambda: orc.Oracle(orc.OracleConfig(**CFG), d.num_users, d.num_items, d.train_ptr, d.train_col)  # noqa: E731
    a, b = mk(), mk()
    for o in (a, b):
        o.ref_seed(MT_SEED, RAND_SEED)
        o.ref_init_params()
    init = {f"init_{n}": a.get(w) for n, w in (("W", ob.P_W), ("W_ag", ob.P_W_AG), ("Wu", ob.P_WU), ("Wu_ag", ob.P_WU_AG), ("b", ob.P_B),
                                               ("b_ag", ob.P_B_AG), ("bp", ob.P_BP), ("bp_ag", ob.P_BP_AG))}
    a.train_reference_sequenced()
    pos, inp, neg, in_ptr = [], [], [], [0]
    for u in range(d.num_users):
        p, i, n = b.ref_draw_user(u)
        pos.append(p); inp.append(i); neg.append(n); in_ptr.append(in_ptr[-1] + i.size)
    final = {f"final_{n}": a.get(w) for n, w in (("W", ob.P_W), ("W_ag", ob.P_W_AG), ("Wu", ob.P_WU), ("Wu_ag", ob.P_WU_AG), ("b", ob.P_B),
                                                 ("b_ag", ob.P_B_AG), ("bp", ob.P_BP), ("bp_ag", ob.P_BP_AG))}
    np.savez_compressed(OUT, mt_seed=MT_SEED, rand_seed=RAND_SEED, data_seed=DATA_SEED, shape="tiny",
                        cfg=np.array(sorted(CFG.items()), dtype=object).astype(str),
                        num_users=d.num_users, num_items=d.num_items, train_ptr=d.train_ptr, train_col=d.train_col,
                        pos_order=np.concatenate(pos), inputs=np.concatenate(inp), in_ptr=np.array(in_ptr, dtype=np.int64),
                        negatives=np.concatenate(neg), **init, **final)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
