"""Shared helpers for the parity tests (test infrastructure)."""
import numpy as np

import cdae_amd
import oracle as orc
from oracle import binding as ob

PARAMS = range(12)


def make_pair(data, *, K=16, B=1, loss=cdae_amd.CROSS_ENTROPY, seed=11, full_output=False, **kw):
    """A HIP model and an oracle that start from the same fp32 parameters."""
    flags = dict(using_adagrad=True, asymmetric=False, user_factor=True, linear=False, scaled=True, tanh=False,
                 linear_function=False)
    hyper = dict(lambda_=0.01, learn_rate=0.1, corruption_ratio=0.5, beta=1.0, num_neg=5, num_corruptions=1)
    for k, v in kw.items():
        (flags if k in flags else hyper)[k] = v
    cfg = cdae_amd.CDAEConfig(num_dim=K, lt=loss, batch_users=B, full_output=full_output, **flags, **hyper)
    model = cdae_amd.CDAE(cfg)
    model.reset(data, seed=seed)
    ocfg = orc.OracleConfig(num_dim=K, loss_type=loss, **flags, **hyper)
    o = orc.Oracle(ocfg, data.num_users, data.num_items, data.train_ptr, data.train_col)
    o.init_params(seed)
    sync_oracle_from_gpu(model, o)
    return model, o


def sync_oracle_from_gpu(model, o):
    for which in PARAMS:
        if o.get(which).size:
            o.set(which, model.get(which).astype(np.float64))


def max_param_err(model, o):
    """max over parameters of max|gpu - oracle| / (1e-3 + max|oracle|)"""
    worst, name = 0.0, None
    for which in PARAMS:
        ref = o.get(which)
        if not ref.size:
            continue
        got = model.get(which).astype(np.float64).ravel()
        err = np.abs(got - ref).max() / (1e-3 + np.abs(ref).max())
        if err > worst:
            worst, name = err, which
    return worst, name
