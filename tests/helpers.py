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


def record_measured(name, **values):
    """Developer aid for the bf16 guards: with CDAE_RECORD_MEASURED=<file> every guarded quantity is appended to that file as it is measured
    (one line per call), so that a bound can be set at <= 1.3 x what a full run of the suite actually sees."""
    import os
    path = os.environ.get("CDAE_RECORD_MEASURED")
    if path:
        with open(path, "a") as f:
            f.write(name + " " + " ".join(f"{k}={float(v):.5g}" for k, v in values.items()) + "\n")


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


# ---- the host layer's own binary cache (src/base/data.hpp Data::write; src/base/io/file.hpp save) -------------------------------
def read_data_cache(path):
    """-> dict(user_names, item_names, users, items, labels) from a cache written by libcf::save(Data) of this repository's
    host layer: magic, columnar tag, the two first-seen dictionaries, then the columns."""
    import struct
    buf = open(path, "rb").read()
    off = 0

    def u64():
        nonlocal off
        v = struct.unpack_from("<Q", buf, off)[0]
        off += 8
        return v
    assert u64() == 0x3145414443464C43, "not a cache of this build"          # "CLFCDAE1"
    assert u64() == 0x324C4F4345414443, "not the columnar format"            # "CDAECOL2"
    assert u64() == 2
    names = []
    for _ in range(2):
        u64(); u64()                         # feature type, dense length
        n = u64()
        grp = []
        for _ in range(n):
            ln = u64()
            grp.append(buf[off:off + ln].decode())
            off += ln
        names.append(grp)
    n, uniform = u64(), u64()
    uniform_label = struct.unpack_from("<d", buf, off)[0]
    off += 8
    users = np.frombuffer(buf, np.uint32, n, off); off += 4 * n
    items = np.frombuffer(buf, np.uint32, n, off); off += 4 * n
    labels = np.full(n, uniform_label) if uniform else np.frombuffer(buf, np.float64, n, off)
    off += 0 if uniform else 8 * n
    assert off == len(buf)
    return dict(user_names=names[0], item_names=names[1], users=users, items=items, labels=labels)


def csr_of(users, items, num_users):
    """uid -> sorted unique item ids (what Data::to_csr / the reference's uid -> {iid -> label} table hold)"""
    key = np.unique(users.astype(np.int64) << 32 | items.astype(np.int64))
    ptr = np.zeros(num_users + 1, np.int64)
    np.add.at(ptr, (key >> 32) + 1, 1)
    return np.cumsum(ptr), (key & 0xFFFFFFFF).astype(np.uint32)


def fnv1a64(*arrays):
    """FNV-1a over the little-endian bytes of the arrays, as src/model/recsys/cdae.hpp logs for the rows it hands to the device"""
    h = 0xcbf29ce484222325
    for a in arrays:
        for b in np.ascontiguousarray(a).tobytes():
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
    return h
