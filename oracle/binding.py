"""ctypes binding of oracle/libcdae_oracle.so (fp64 CPU restatement of the reference; test infra)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcdae_oracle.so")

LOSS_SQUARE = 0
LOSS_CE = 5
P_W, P_W_AG, P_V, P_V_AG, P_WU, P_WU_AG, P_B, P_B_AG, P_BP, P_BP_AG, P_UU, P_UU_AG = range(12)
P_COUNT = 12


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "cdae_oracle.cpp")
    src2 = os.path.join(_HERE, "mf_oracle.cpp")
    hdr = os.path.join(_HERE, "..", "include", "cdae_rng.h")
    stale = (not os.path.exists(_SO)) or any(
        os.path.exists(p) and os.path.getmtime(p) > os.path.getmtime(_SO) for p in (src, src2, hdr))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "libcdae_oracle.so"])
    return _SO


class _Cfg(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "num_dim", "num_neg", "num_corruptions", "loss_type", "using_adagrad", "asymmetric",
        "user_factor", "linear", "scaled", "tanh_act", "linear_function")] + [(n, C.c_double) for n in (
            "lambda_", "learn_rate", "corruption_ratio", "beta")]


@dataclass
class OracleConfig:
    """Field names follow libcf::CDAEConfig (/root/reference/src/model/recsys/cdae.hpp:13-31)."""
    num_dim: int = 10
    num_neg: int = 5
    num_corruptions: int = 1
    loss_type: int = LOSS_SQUARE
    using_adagrad: bool = True
    asymmetric: bool = False
    user_factor: bool = True
    linear: bool = False
    scaled: bool = True
    tanh: bool = False
    linear_function: bool = False
    lambda_: float = 0.01
    learn_rate: float = 0.1
    corruption_ratio: float = 0.5
    beta: float = 1.0

    def _c(self) -> _Cfg:
        return _Cfg(self.num_dim, self.num_neg, self.num_corruptions, self.loss_type,
                    int(self.using_adagrad), int(self.asymmetric), int(self.user_factor),
                    int(self.linear), int(self.scaled), int(self.tanh), int(self.linear_function), self.lambda_,
                    self.learn_rate, self.corruption_ratio, self.beta)


class _MfCfg(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("num_dim", "num_neg", "loss_type", "using_adagrad", "using_bias_term", "pairwise")] + [
        (n, C.c_double) for n in ("lambda_", "learn_rate", "beta")]


LOSS_LOGISTIC, LOSS_LOG, LOSS_HINGE = 1, 2, 3
MF_UV, MF_UV_AG, MF_IV, MF_IV_AG, MF_UB, MF_UB_AG, MF_IB, MF_IB_AG = range(8)


@dataclass
class MfConfig:
    """libcf::IMFConfig / BPRConfig (/root/reference/src/model/recsys/imf.hpp:12-23, bpr.hpp:12-24); pairwise = BPR."""
    num_dim: int = 10
    num_neg: int = 5
    loss_type: int = LOSS_SQUARE
    using_adagrad: bool = True
    using_bias_term: bool = True
    pairwise: bool = False
    lambda_: float = 0.01
    learn_rate: float = 0.1
    beta: float = 1.0


class MfOracle:
    """fp64 restatement of IMF / BPR (oracle/mf_oracle.cpp): literal and block schedules."""

    def __init__(self, cfg: MfConfig, num_users, num_items, row_ptr, col_idx):
        self.lib = _load()
        self.cfg = cfg
        self.U, self.I, self.K = int(num_users), int(num_items), int(cfg.num_dim)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col_idx, dtype=np.uint32)
        c = _MfCfg(cfg.num_dim, cfg.num_neg, cfg.loss_type, int(cfg.using_adagrad), int(cfg.using_bias_term), int(cfg.pairwise),
                   cfg.lambda_, cfg.learn_rate, cfg.beta)
        self.h = self.lib.mf_oracle_create(C.byref(c), self.U, self.I, _p(self.row_ptr), _p(self.col))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.mf_oracle_destroy(self.h)
            self.h = None

    def init_params(self, seed):
        self.lib.mf_oracle_init_params(self.h, seed)

    def get(self, which):
        n = self.lib.mf_oracle_param_size(self.h, which)
        out = np.empty(n)
        assert self.lib.mf_oracle_get_param(self.h, which, _p(out), n) == 0
        return out

    def set(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        assert self.lib.mf_oracle_set_param(self.h, which, _p(a), a.size) == 0

    def train_literal(self, seed, epoch, u0=0, u1=None):
        self.lib.mf_oracle_train_literal(self.h, seed, epoch, u0, self.U if u1 is None else u1)

    def train_batched(self, seed, epoch, batch_users, u0=0, u1=None):
        self.lib.mf_oracle_train_batched(self.h, seed, epoch, u0, self.U if u1 is None else u1, batch_users)

    def predict(self, u, i):
        return self.lib.mf_oracle_predict(self.h, u, i)

    def loss_grad(self, pred, truth):
        return self.lib.mf_oracle_loss_grad(self.h, pred, truth)

    def recommend(self, topk=10, u0=0, u1=None, with_scores=False):
        u1 = self.U if u1 is None else u1
        out = np.empty((u1 - u0, topk), dtype=np.uint32)
        sc = np.empty((u1 - u0, topk)) if with_scores else None
        self.lib.mf_oracle_recommend(self.h, u0, u1, topk, _p(out), _p(sc) if with_scores else None)
        return (out, sc) if with_scores else out


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    lib = C.CDLL(build())
    vp, u64, u32, dbl = C.c_void_p, C.c_uint64, C.c_uint32, C.c_double
    lib.oracle_create.restype = vp
    lib.oracle_create.argtypes = [C.POINTER(_Cfg), u64, u64, vp, vp]
    lib.oracle_destroy.argtypes = [vp]
    lib.oracle_init_params.argtypes = [vp, u64]
    lib.oracle_param_size.restype = C.c_size_t
    lib.oracle_param_size.argtypes = [vp, u32]
    lib.oracle_get_param.argtypes = [vp, u32, vp, C.c_size_t]
    lib.oracle_set_param.argtypes = [vp, u32, vp, C.c_size_t]
    lib.oracle_train_users_literal.argtypes = [vp, u64, u32, u64, u64]
    lib.oracle_train_users_batched.argtypes = [vp, u64, u32, u64, u64, u64]
    lib.oracle_train_users_full.argtypes = [vp, u64, u32, u64, u64, u64]
    lib.oracle_step_user.argtypes = [vp, u64, vp, u64, vp, u64, vp, vp, vp, vp]
    lib.oracle_ref_seed.argtypes = [vp, u64, u32]
    lib.oracle_ref_set_insertion_order.argtypes = [vp, vp]
    lib.oracle_ref_init_params.argtypes = [vp]
    lib.oracle_train_users_reference_sequenced.argtypes = [vp, u64, u64]
    lib.oracle_ref_draw_user.argtypes = [vp, u64, vp, vp, vp, vp]
    lib.oracle_step_user_seq.argtypes = [vp, u64, vp, u64, vp, u64, vp, u64]
    lib.oracle_ref_rand.restype = u32
    lib.oracle_ref_rand.argtypes = [vp]
    lib.oracle_ref_mt.restype = u64
    lib.oracle_ref_mt.argtypes = [vp]
    lib.oracle_ref_uniform.restype = dbl
    lib.oracle_ref_uniform.argtypes = [vp]
    lib.oracle_draw_inputs.argtypes = [vp, u64, u32, u64, u32, u32, vp, vp]
    lib.oracle_draw_negatives.argtypes = [vp, u64, u32, u64, u32, vp]
    lib.oracle_encode.argtypes = [vp, u64, u32, C.c_int, vp, u64, vp]
    lib.oracle_data_loss.restype = dbl
    lib.oracle_data_loss.argtypes = [vp, u64, u32]
    lib.oracle_penalty_loss.restype = dbl
    lib.oracle_penalty_loss.argtypes = [vp]
    lib.oracle_loss_eval.restype = dbl
    lib.oracle_loss_eval.argtypes = [vp, dbl, dbl]
    lib.oracle_loss_grad.restype = dbl
    lib.oracle_loss_grad.argtypes = [vp, dbl, dbl]
    lib.oracle_recommend.argtypes = [vp, u64, u64, u32, vp, vp]
    lib.oracle_eval_topn.argtypes = [vp, u32, u64, vp, vp, vp]
    lib.oracle_eval_rec_list.argtypes = [vp, u64, vp, u64, vp]
    lib.mf_oracle_create.restype = vp
    lib.mf_oracle_create.argtypes = [C.POINTER(_MfCfg), u64, u64, vp, vp]
    lib.mf_oracle_destroy.argtypes = [vp]
    lib.mf_oracle_init_params.argtypes = [vp, u64]
    lib.mf_oracle_param_size.restype = C.c_size_t
    lib.mf_oracle_param_size.argtypes = [vp, u32]
    lib.mf_oracle_get_param.argtypes = [vp, u32, vp, C.c_size_t]
    lib.mf_oracle_set_param.argtypes = [vp, u32, vp, C.c_size_t]
    lib.mf_oracle_train_literal.argtypes = [vp, u64, u32, u64, u64]
    lib.mf_oracle_train_batched.argtypes = [vp, u64, u32, u64, u64, u64]
    lib.mf_oracle_predict.restype = dbl
    lib.mf_oracle_predict.argtypes = [vp, u64, u64]
    lib.mf_oracle_loss_grad.restype = dbl
    lib.mf_oracle_loss_grad.argtypes = [vp, dbl, dbl]
    lib.mf_oracle_recommend.argtypes = [vp, u64, u64, u32, vp, vp]
    lib.oracle_heap_create.restype = vp
    lib.oracle_heap_destroy.argtypes = [vp]
    lib.oracle_heap_push.argtypes = [vp, u64, dbl]
    lib.oracle_heap_push_and_pop.argtypes = [vp, u64, dbl]
    lib.oracle_heap_size.restype = u64
    lib.oracle_heap_size.argtypes = [vp]
    lib.oracle_heap_sorted.argtypes = [vp, vp, vp]
    lib.oracle_heap_front.restype = u64
    lib.oracle_heap_front.argtypes = [vp]
    lib.oracle_heap_pop.restype = u64
    lib.oracle_heap_pop.argtypes = [vp]
    _lib = lib
    return lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    def __init__(self, cfg: OracleConfig, num_users: int, num_items: int, row_ptr, col_idx):
        self.lib = _load()
        self.cfg = cfg
        self.U, self.I, self.K = int(num_users), int(num_items), int(cfg.num_dim)
        self.row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int64)
        self.col = np.ascontiguousarray(col_idx, dtype=np.uint32)
        c = cfg._c()
        self.h = self.lib.oracle_create(C.byref(c), self.U, self.I, _p(self.row_ptr), _p(self.col))

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_destroy(self.h)
            self.h = None

    def init_params(self, seed: int):
        self.lib.oracle_init_params(self.h, seed)

    def get(self, which: int) -> np.ndarray:
        n = self.lib.oracle_param_size(self.h, which)
        out = np.empty(n, dtype=np.float64)
        if n:
            assert self.lib.oracle_get_param(self.h, which, _p(out), n) == 0
        return out

    def set(self, which: int, arr):
        a = np.ascontiguousarray(arr, dtype=np.float64).ravel()
        assert self.lib.oracle_set_param(self.h, which, _p(a), a.size) == 0

    def train_literal(self, seed: int, epoch: int, u0: int = 0, u1: int | None = None):
        self.lib.oracle_train_users_literal(self.h, seed, epoch, u0, self.U if u1 is None else u1)

    def train_batched(self, seed: int, epoch: int, batch_users: int, u0: int = 0, u1: int | None = None):
        self.lib.oracle_train_users_batched(self.h, seed, epoch, u0, self.U if u1 is None else u1, batch_users)

    def train_full(self, seed: int, epoch: int, batch_users: int, u0: int = 0, u1: int | None = None):
        """Full-output decode (every unrated item is a negative once), block-summed gradients."""
        self.lib.oracle_train_users_full(self.h, seed, epoch, u0, self.U if u1 is None else u1, batch_users)

    # ---- reference-sequenced mode: the reference's own generators in the reference's own order (cdae_oracle.cpp GlibcRand) ----
    def ref_seed(self, mt_seed: int, rand_seed: int = 1):
        """Random::seed(mt_seed) (random.hpp:29-31) and srand(rand_seed) — the reference never calls srand, i.e. 1"""
        self.lib.oracle_ref_seed(self.h, mt_seed, rand_seed)

    def ref_set_insertion_order(self, items=None):
        a = None if items is None else np.ascontiguousarray(items, dtype=np.uint32)
        self.lib.oracle_ref_set_insertion_order(self.h, None if a is None else _p(a))

    def ref_init_params(self):
        self.lib.oracle_ref_init_params(self.h)

    def train_reference_sequenced(self, u0: int = 0, u1: int | None = None):
        self.lib.oracle_train_users_reference_sequenced(self.h, u0, self.U if u1 is None else u1)

    def ref_draw_user(self, uid: int):
        """(train items in the reference's visiting order, kept inputs in theirs, negatives) of the next user-corruption"""
        n = int(self.row_ptr[uid + 1] - self.row_ptr[uid])
        pos, inp, neg = np.empty(n, np.uint32), np.empty(n, np.uint32), np.empty(n * self.cfg.num_neg, np.uint32)
        n_in = C.c_uint64()
        self.lib.oracle_ref_draw_user(self.h, uid, _p(pos), _p(inp), C.byref(n_in), _p(neg))
        return pos, inp[:n_in.value].copy(), neg

    def step_user_seq(self, uid: int, pos_seq, in_seq, neg_items):
        p, i, n = (np.ascontiguousarray(a, dtype=np.uint32) for a in (pos_seq, in_seq, neg_items))
        self.lib.oracle_step_user_seq(self.h, uid, _p(p), p.size, _p(i), i.size, _p(n), n.size)

    def ref_rand(self) -> int:
        return int(self.lib.oracle_ref_rand(self.h))

    def ref_mt(self) -> int:
        return int(self.lib.oracle_ref_mt(self.h))

    def ref_uniform(self) -> float:
        return float(self.lib.oracle_ref_uniform(self.h))

    def step_user(self, uid: int, in_items, neg_items):
        i = np.ascontiguousarray(in_items, dtype=np.uint32)
        n = np.ascontiguousarray(neg_items, dtype=np.uint32)
        n_pos = int(self.row_ptr[uid + 1] - self.row_ptr[uid])
        z = np.empty(self.K); hg = np.empty(self.K)
        y = np.empty(n_pos + n.size); g = np.empty(n_pos + n.size)
        self.lib.oracle_step_user(self.h, uid, _p(i), i.size, _p(n), n.size, _p(z), _p(y), _p(g), _p(hg))
        return z, y, g, hg

    def draw_inputs(self, seed, epoch, uid, cidx=0, stream=0) -> np.ndarray:
        n_pos = int(self.row_ptr[uid + 1] - self.row_ptr[uid])
        out = np.empty(max(n_pos, 1), dtype=np.uint32)
        n = C.c_uint64(0)
        self.lib.oracle_draw_inputs(self.h, seed, epoch, uid, cidx, stream, _p(out), C.byref(n))
        return out[:n.value].copy()

    def draw_negatives(self, seed, epoch, uid, cidx=0) -> np.ndarray:
        n_pos = int(self.row_ptr[uid + 1] - self.row_ptr[uid])
        out = np.empty(max(n_pos * self.cfg.num_neg, 1), dtype=np.uint32)
        self.lib.oracle_draw_negatives(self.h, seed, epoch, uid, cidx, _p(out))
        return out[:n_pos * self.cfg.num_neg].copy()

    def encode(self, seed, epoch, mode, uids) -> np.ndarray:
        u = np.ascontiguousarray(uids, dtype=np.uint32)
        Z = np.empty((u.size, self.K))
        self.lib.oracle_encode(self.h, seed, epoch, mode, _p(u), u.size, _p(Z))
        return Z

    def data_loss(self, seed, epoch) -> float:
        return self.lib.oracle_data_loss(self.h, seed, epoch)

    def penalty_loss(self) -> float:
        return self.lib.oracle_penalty_loss(self.h)

    def loss_eval(self, pred, truth) -> float:
        return self.lib.oracle_loss_eval(self.h, pred, truth)

    def loss_grad(self, pred, truth) -> float:
        return self.lib.oracle_loss_grad(self.h, pred, truth)

    def recommend(self, topk=10, u0=0, u1=None, with_scores=False):
        u1 = self.U if u1 is None else u1
        out = np.empty((u1 - u0, topk), dtype=np.uint32)
        sc = np.empty((u1 - u0, topk)) if with_scores else None
        self.lib.oracle_recommend(self.h, u0, u1, topk, _p(out), _p(sc) if with_scores else None)
        return (out, sc) if with_scores else out


def eval_topn(rec: np.ndarray, test_ptr, test_col) -> np.ndarray:
    """[P@1, P@5, P@10, R@1, R@5, R@10, MAP@5, MAP@10] (evaluation.hpp:97-111, 113-219)."""
    lib = _load()
    rec = np.ascontiguousarray(rec, dtype=np.uint32)
    tp = np.ascontiguousarray(test_ptr, dtype=np.int64)
    tc = np.ascontiguousarray(test_col, dtype=np.uint32)
    out = np.empty(8)
    lib.oracle_eval_topn(_p(rec), rec.shape[1], rec.shape[0], _p(tp), _p(tc), _p(out))
    return out


def eval_rec_list(lst, truth) -> np.ndarray:
    lib = _load()
    l = np.ascontiguousarray(lst, dtype=np.uint32)
    t = np.ascontiguousarray(sorted(truth), dtype=np.uint32)
    out = np.empty(8)
    lib.oracle_eval_rec_list(_p(l), l.size, _p(t), t.size, _p(out))
    return out


class OracleHeap:
    """Heap<pair<size_t,double>>(sort_by_second_desc) — /root/reference/src/base/heap.hpp."""

    def __init__(self):
        self.lib = _load()
        self.h = self.lib.oracle_heap_create()

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.oracle_heap_destroy(self.h)
            self.h = None

    def push(self, i, v):
        self.lib.oracle_heap_push(self.h, i, v)

    def push_and_pop(self, i, v):
        self.lib.oracle_heap_push_and_pop(self.h, i, v)

    def size(self):
        return self.lib.oracle_heap_size(self.h)

    def front(self):
        return self.lib.oracle_heap_front(self.h)

    def pop(self):
        return self.lib.oracle_heap_pop(self.h)

    def sorted(self):
        n = self.size()
        ids = np.empty(n, dtype=np.uint64); vals = np.empty(n)
        self.lib.oracle_heap_sorted(self.h, _p(ids), _p(vals))
        return ids, vals
