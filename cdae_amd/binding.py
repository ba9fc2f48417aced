"""ctypes binding of libcdae_hip.so (include/cdae_hip.h) and a host-side mirror of libcf::CDAE.

`CDAEConfig` / `CDAE` keep the field and method names of the reference's model class
(/root/reference/src/model/recsys/cdae.hpp:13-31, 36-196) so that parity tests read like the reference's
call sites (apps/yelp/yelp.cpp:168-199, src/solver/solver-inl.hpp:19,53,55).  There is NO CPU fallback:
constructing a CDAE without the HIP library or without a GPU raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CDAE_HIP_LIBRARY") or os.path.join(_HERE, "lib", "libcdae_hip.so")   # (developer builds: another .so of the same ABI)
# the -DCDAE_DEVELOPER build of the same sources (made by __graft_entry__.build()): the only library that reads the developer
# environment switches; loaded by the tests that flip one (developer_library() below), never by the product path
DEV_LIB_PATH = os.path.join(os.path.dirname(_HERE), "build", "libcdae_hip_dev.so")

# libcf::LossType values (/root/reference/src/model/loss.hpp:10-18)
SQUARE, LOGISTIC, LOG, HINGE, SQUARED_HINGE, CROSS_ENTROPY, LOGM = range(7)

P_W, P_W_AG, P_V, P_V_AG, P_WU, P_WU_AG, P_B, P_B_AG, P_BP, P_BP_AG, P_UU, P_UU_AG = range(12)
P_UB, P_UB_AG = 12, 13
P_COUNT = 14

PLAN_FUSED_DECODE, PLAN_GEMM2_TN, PLAN_ROWS_FUSED = 1, 2, 4     # cdae_hip_full_output_plan bits (include/cdae_hip.h)
IMF_DEFAULT_BATCH_USERS = 16   # CDAE_IMF_DEFAULT_BATCH_USERS / CDAE_BPR_DEFAULT_BATCH_USERS (include/cdae_hip.h): what an IMF / BPR handle created
BPR_DEFAULT_BATCH_USERS = 8    # with batch_users = 0 trains on a BASELINE-sized data set (cdae_hip_mf_default_batch_users); 1 on smaller ones
DEFAULT_BATCH_USERS = 0        # 0 = the library's default (cdae_hip_default_batch_users: num_users / 160, within [32, 256])


class _Config(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "struct_size", "num_dim", "num_neg", "num_corruptions", "loss_type", "using_adagrad",
        "asymmetric", "user_factor", "linear", "scaled", "tanh_act", "batch_users", "full_output", "linear_function")] + [
            (n, C.c_double) for n in ("lambda_", "learn_rate", "corruption_ratio", "beta")]


class _MfConfig(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("struct_size", "num_dim", "num_neg", "loss_type", "using_adagrad", "using_bias_term",
                                          "pairwise", "batch_users")] + [(n, C.c_double) for n in ("lambda_", "learn_rate", "beta")]


class Stats(C.Structure):
    _fields_ = [("wall_seconds", C.c_double), ("users", C.c_uint64), ("examples", C.c_uint64),
                ("batches", C.c_uint64), ("ms_sample", C.c_double), ("ms_sort", C.c_double),
                ("ms_encode", C.c_double), ("ms_decode", C.c_double), ("ms_hidden", C.c_double),
                ("ms_input", C.c_double), ("launches_decode", C.c_uint64)]

    def as_dict(self):
        return {n: getattr(self, n) for n, _ in self._fields_}


EXPORTS = {
    # name: (restype, argtypes) — every symbol include/cdae_hip.h declares
    "cdae_hip_last_error": (C.c_char_p, []),
    "cdae_hip_abi_version": (C.c_int, []),
    "cdae_hip_create": (C.c_int, [C.POINTER(_Config), C.c_int, C.POINTER(C.c_void_p)]),
    "cdae_hip_create_mf": (C.c_int, [C.POINTER(_MfConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "cdae_hip_destroy": (C.c_int, [C.c_void_p]),
    "cdae_hip_set_interactions": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "cdae_hip_row_stride": (C.c_uint32, [C.c_void_p]),
    "cdae_hip_default_batch_users": (C.c_uint32, [C.c_uint64]),
    "cdae_hip_mf_default_batch_users": (C.c_uint32, [C.c_uint64, C.c_uint32]),
    "cdae_hip_batch_users": (C.c_uint32, [C.c_void_p]),
    "cdae_hip_full_output_plan": (C.c_uint32, [C.c_void_p]),
    "cdae_hip_set_decode_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "cdae_hip_decode_plan": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "cdae_hip_user_order": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "cdae_hip_set_user_id_offset": (C.c_int, [C.c_void_p, C.c_uint64]),
    "cdae_hip_init_params": (C.c_int, [C.c_void_p, C.c_uint64]),
    "cdae_hip_set_param": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "cdae_hip_get_param": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "cdae_hip_param_device_ptr": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "cdae_hip_train_epoch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(Stats)]),
    "cdae_hip_train_users": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(Stats)]),
    "cdae_hip_enqueue_users": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64]),
    "cdae_hip_prefetch_users": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64]),
    "cdae_hip_collect_stats": (C.c_int, [C.c_void_p, C.POINTER(Stats)]),
    "cdae_hip_train_one_user_corruption": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]),
    "cdae_hip_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "cdae_hip_set_profiling_families": (C.c_int, [C.c_void_p, C.c_uint32]),
    "cdae_hip_synchronize": (C.c_int, [C.c_void_p]),
    "cdae_hip_debug_sample_batch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32] + [C.c_void_p] * 8
                                    + [C.POINTER(C.c_uint64)]),
    "cdae_hip_encode": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p]),
    "cdae_hip_data_loss": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    "cdae_hip_penalty_loss": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "cdae_hip_recommend_all": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "cdae_hip_recommend_user": (C.c_int, [C.c_void_p, C.c_uint64, C.c_void_p, C.c_size_t, C.c_uint32, C.c_void_p]),
    "cdae_hip_set_test_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "cdae_hip_eval_topn": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cdae_hip_stream": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "cdae_hip_delta_begin": (C.c_int, [C.c_void_p]),
    "cdae_hip_delta_compute": (C.c_int, [C.c_void_p]),
    "cdae_hip_delta_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "cdae_hip_delta_apply": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "cdae_hip_delta_stage": (C.c_int, [C.c_void_p]),
    "cdae_hip_delta_set_combine": (C.c_int, [C.c_void_p, C.c_uint32]),
    "cdae_hip_delta_recv_device_ptr": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "cdae_hip_delta_merge": (C.c_int, [C.c_void_p]),
    "cdae_hip_delta_merge_stage": (C.c_int, [C.c_void_p]),
    "cdae_hip_comm_unique_id": (C.c_int, [C.c_void_p, C.c_size_t]),
    "cdae_hip_comm_init_rank": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]),
    "cdae_hip_exchange_configure": (C.c_int, [C.c_void_p, C.c_int]),
    "cdae_hip_exchange_step": (C.c_int, [C.c_void_p]),
    "cdae_hip_exchange_flush": (C.c_int, [C.c_void_p]),
    "cdae_hip_exchange_time_all_reduce": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double)]),
    "cdae_hip_multi_create": (C.c_int, [C.POINTER(_Config), C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_void_p)]),
    "cdae_hip_multi_destroy": (C.c_int, [C.c_void_p]),
    "cdae_hip_multi_set_layout": (C.c_int, [C.c_void_p, C.c_uint32]),
    "cdae_hip_multi_num_shards": (C.c_int, [C.c_void_p]),
    "cdae_hip_multi_shard": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "cdae_hip_multi_set_interactions": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]),
    "cdae_hip_multi_init_params": (C.c_int, [C.c_void_p, C.c_uint64]),
    "cdae_hip_multi_set_exchange": (C.c_int, [C.c_void_p, C.c_int]),
    "cdae_hip_multi_set_schedule": (C.c_int, [C.c_void_p, C.c_void_p]),
    "cdae_hip_multi_train_epoch": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(Stats)]),
    "cdae_hip_multi_steps_per_epoch": (C.c_uint64, [C.c_void_p]),
    "cdae_hip_multi_train_steps": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(Stats)]),
    "cdae_hip_multi_train_users": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(Stats)]),
    "cdae_hip_multi_data_loss": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    "cdae_hip_multi_penalty_loss": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "cdae_hip_multi_recommend_all": (C.c_int, [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_void_p]),
    "cdae_hip_multi_eval_topn": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "cdae_hip_multi_get_param": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
    "cdae_hip_multi_set_param": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_size_t]),
}
COMM_ID_BYTES = 128

_libs = {}                 # path -> loaded library
_default_path = LIB_PATH   # what load_library() without a path returns (developer_library swaps it for the length of a test)


def load_library(path: str | None = None):
    """dlopen the C-ABI library and bind every declared export.  Raises if it is missing."""
    path = path or _default_path
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the CDAE hot path.")
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64 (ROCm 7.0) next
    # to the system's (7.2).  If this library pulled the system copies in first, a later `import torch` (needed
    # only for torch.distributed) finds a foreign runtime already loaded and reports "No HIP GPUs are available".
    # Loading torch first makes both share torch's copy, which the kernels run on unchanged.
    # three library streams + the caller's (torch, RCCL) need more than HIP's default four hardware queues to stay
    # concurrent (bench.py has the measurement); only effective if HIP has not been initialised yet
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)       # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _libs[path] = lib
    return lib



@contextlib.contextmanager
def developer_library():
    """Inside the block every new CDAE / MultiCDAE / MF object is created on the DEVELOPER build (environment switches compiled in).
    Handles of the two libraries are not interchangeable; a comparison of a switched path with the default one runs both sides here."""
    global _default_path
    load_library(DEV_LIB_PATH)
    prev, _default_path = _default_path, DEV_LIB_PATH
    try:
        yield _libs[DEV_LIB_PATH]
    finally:
        _default_path = prev


class CDAEError(RuntimeError):
    pass


def _chk(lib, rc):
    if rc != 0:
        raise CDAEError(lib.cdae_hip_last_error().decode())


@dataclass
class CDAEConfig:
    """libcf::CDAEConfig (/root/reference/src/model/recsys/cdae.hpp:13-31); defaults are the struct's."""
    lambda_: float = 0.01
    learn_rate: float = 0.1
    lt: int = LOGISTIC
    num_dim: int = 10
    using_adagrad: bool = True
    corruption_ratio: float = 0.5
    num_corruptions: int = 1
    asymmetric: bool = False
    user_factor: bool = True
    linear: bool = False
    num_neg: int = 5
    scaled: bool = True
    beta: float = 0.0
    linear_function: bool = False
    tanh: bool = False
    # not in the reference: users per parameter snapshot (1 == the reference's sequential schedule)
    batch_users: int = DEFAULT_BATCH_USERS
    # not in the reference: full-output decode — every unrated item is a negative once (MFMA path)
    full_output: bool = False


class CDAE:
    """Host mirror of libcf::CDAE over the C ABI.

    reset(train) ~ cdae.hpp:109-134, train_one_iteration ~ :136-146, current_loss ~ model_base.hpp:29-32,
    pre_recommend/recommend ~ :162-196 (all users at once on the GPU, then table lookups).
    """

    def __init__(self, mcfg: CDAEConfig, device: int = 0):
        self.lib = load_library()
        self.cfg = mcfg
        c = _Config(C.sizeof(_Config), mcfg.num_dim, mcfg.num_neg, mcfg.num_corruptions, mcfg.lt,
                    int(mcfg.using_adagrad), int(mcfg.asymmetric), int(mcfg.user_factor), int(mcfg.linear),
                    int(mcfg.scaled), int(mcfg.tanh), mcfg.batch_users, int(mcfg.full_output), int(mcfg.linear_function),
                    mcfg.lambda_, mcfg.learn_rate,
                    mcfg.corruption_ratio, mcfg.beta)
        self.h = C.c_void_p()
        _chk(self.lib, self.lib.cdae_hip_create(C.byref(c), device, C.byref(self.h)))
        self.num_users = self.num_items = 0
        self._rec = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.cdae_hip_destroy(self.h)
            self.h = None

    __del__ = close

    # ---- reset -------------------------------------------------------------------------------------
    def set_interactions(self, num_users, num_items, row_ptr, col_idx, user_id_offset: int = 0):
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        ci = np.ascontiguousarray(col_idx, dtype=np.uint32)
        if rp.size != num_users + 1 or ci.size != rp[-1]:
            raise CDAEError("row_ptr / col_idx sizes do not match num_users")
        _chk(self.lib, self.lib.cdae_hip_set_interactions(self.h, num_users, num_items, rp.ctypes.data, ci.ctypes.data))
        _chk(self.lib, self.lib.cdae_hip_set_user_id_offset(self.h, user_id_offset))
        self.num_users, self.num_items = int(num_users), int(num_items)
        self._row_ptr = rp

    def reset(self, train, seed: int = 0):
        """train: object with num_users, num_items, train_ptr, train_col (cdae_amd.synth.Interactions)."""
        self.set_interactions(train.num_users, train.num_items, train.train_ptr, train.train_col)
        self.init_params(seed)

    def init_params(self, seed: int):
        _chk(self.lib, self.lib.cdae_hip_init_params(self.h, seed))

    def user_order(self) -> np.ndarray:
        """out[position] = user id of the handle's training order (the identity except for IMF / BPR block schedules)"""
        out = np.empty(self.num_users, dtype=np.uint32)
        _chk(self.lib, self.lib.cdae_hip_user_order(self.h, out.ctypes.data, out.size))
        return out

    @property
    def batch_users(self) -> int:
        """users per parameter snapshot the handle is using (the library's choice when the config asked for 0)"""
        return int(self.lib.cdae_hip_batch_users(self.h))

    def set_decode_fused(self, allow: bool):
        """cdae_hip_set_decode_fused: off for handles trained side by side with another handle on the same device"""
        _chk(self.lib, self.lib.cdae_hip_set_decode_fused(self.h, int(bool(allow))))

    @property
    def decode_plan(self) -> dict:
        """{hot_rows, late_rows, fused} of the sampled step (include/cdae_hip.h cdae_hip_decode_plan)"""
        a, b, c = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
        _chk(self.lib, self.lib.cdae_hip_decode_plan(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(hot_rows=int(a.value), late_rows=int(b.value), fused=bool(c.value))

    @property
    def full_output_plan(self) -> int:
        """CDAE_PLAN_* bits (include/cdae_hip.h): which launches this handle's full-output decode is made of"""
        return int(self.lib.cdae_hip_full_output_plan(self.h))

    # ---- parameters --------------------------------------------------------------------------------
    def _shape(self, which):
        K = self.cfg.num_dim
        if which in (P_BP, P_BP_AG):
            return (self.num_items,)
        if which in (P_B, P_B_AG):
            return (K,)
        if which in (P_WU, P_WU_AG, P_UU, P_UU_AG):
            return (self.num_users, K)
        return (self.num_items, K)

    def get(self, which) -> np.ndarray:
        out = np.empty(self._shape(which), dtype=np.float32)
        _chk(self.lib, self.lib.cdae_hip_get_param(self.h, which, out.ctypes.data, out.size))
        return out

    def set(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32).reshape(self._shape(which))
        _chk(self.lib, self.lib.cdae_hip_set_param(self.h, which, a.ctypes.data, a.size))

    def param_device_ptr(self, which):
        p, n = C.c_void_p(), C.c_size_t()
        _chk(self.lib, self.lib.cdae_hip_param_device_ptr(self.h, which, C.byref(p), C.byref(n)))
        return p.value, n.value

    # ---- training ----------------------------------------------------------------------------------
    def set_profiling(self, period, families=None):
        """0/False: off; k >= 1 (True = 1): HIP-event kernel timing on every k-th batch, of the families named (e.g. ("decode",))
        or of all of them."""
        names = ("sample", "sort", "encode", "decode", "hidden", "input")
        mask = 0xFFFFFFFF if families is None else sum(1 << names.index(f) for f in families)
        _chk(self.lib, self.lib.cdae_hip_set_profiling_families(self.h, mask))
        _chk(self.lib, self.lib.cdae_hip_set_profiling(self.h, int(period)))

    def train_one_iteration(self, seed: int, epoch: int) -> Stats:
        st = Stats()
        _chk(self.lib, self.lib.cdae_hip_train_epoch(self.h, seed, epoch, C.byref(st)))
        return st

    def train_users(self, seed: int, epoch: int, u_begin: int, u_end: int) -> Stats:
        st = Stats()
        _chk(self.lib, self.lib.cdae_hip_train_users(self.h, seed, epoch, u_begin, u_end, C.byref(st)))
        return st

    def enqueue_users(self, seed: int, epoch: int, u_begin: int, u_end: int):
        _chk(self.lib, self.lib.cdae_hip_enqueue_users(self.h, seed, epoch, u_begin, u_end))

    def prefetch_users(self, seed: int, epoch: int, u_begin: int, u_end: int):
        _chk(self.lib, self.lib.cdae_hip_prefetch_users(self.h, seed, epoch, u_begin, u_end))

    def collect_stats(self) -> Stats:
        st = Stats()
        _chk(self.lib, self.lib.cdae_hip_collect_stats(self.h, C.byref(st)))
        return st

    def synchronize(self):
        _chk(self.lib, self.lib.cdae_hip_synchronize(self.h))

    def debug_sample_batch(self, seed: int, epoch: int, u_begin: int, n_users: int, cidx: int = 0) -> dict:
        """Integer work of one batch (masks, negatives, item sort, segments, duplicate numbering) copied back from the
        device (cdae_hip_debug_sample_batch) — the bit-exact parity tests compare it with the oracle's draws."""
        nnz = int(self._row_ptr[u_begin + n_users] - self._row_ptr[u_begin])
        E = nnz * (1 if self.cfg.full_output else 1 + self.cfg.num_neg)
        out = {"ex_item": np.empty(E, np.uint32), "ex_val": np.empty(E, np.uint64), "sorted_item": np.empty(E, np.uint32),
               "sorted_val": np.empty(E, np.uint64), "seg_begin": np.empty(self.num_items, np.uint32),
               "seg_end": np.empty(self.num_items, np.uint32), "dup_of_pos": np.empty(E, np.uint32),
               "dup_of_ex": np.empty(E, np.uint32)}
        n = C.c_uint64(E)
        _chk(self.lib, self.lib.cdae_hip_debug_sample_batch(
            self.h, seed, epoch, u_begin, n_users, cidx, *[a.ctypes.data for a in out.values()], C.byref(n)))
        assert n.value == E
        return out

    def train_one_user_corruption(self, uid: int, input_items, negative_items):
        """cdae.hpp:198-200 with explicit input set; negatives as the reference would have drawn them."""
        i = np.ascontiguousarray(input_items, dtype=np.uint32)
        n = np.ascontiguousarray(negative_items, dtype=np.uint32)
        _chk(self.lib, self.lib.cdae_hip_train_one_user_corruption(self.h, uid, i.ctypes.data, i.size, n.ctypes.data, n.size))

    def get_hidden_values(self, uids, seed: int = 0, epoch: int = 0, mode: int = 0) -> np.ndarray:
        u = np.ascontiguousarray(uids, dtype=np.uint32)
        Z = np.empty((u.size, self.cfg.num_dim), dtype=np.float32)
        _chk(self.lib, self.lib.cdae_hip_encode(self.h, seed, epoch, mode, u.ctypes.data, u.size, Z.ctypes.data))
        return Z

    def data_loss(self, seed: int, epoch: int) -> float:
        v = C.c_double()
        _chk(self.lib, self.lib.cdae_hip_data_loss(self.h, seed, epoch, C.byref(v)))
        return v.value

    def penalty_loss(self) -> float:
        v = C.c_double()
        _chk(self.lib, self.lib.cdae_hip_penalty_loss(self.h, C.byref(v)))
        return v.value

    def current_loss(self, seed: int, epoch: int) -> float:
        return self.data_loss(seed, epoch) + self.penalty_loss()      # model_base.hpp:29-32

    # ---- evaluation --------------------------------------------------------------------------------
    def recommend_all(self, topk: int = 10, u_begin: int = 0, u_end: int | None = None) -> np.ndarray:
        u_end = self.num_users if u_end is None else u_end
        out = np.empty((u_end - u_begin, topk), dtype=np.uint32)
        _chk(self.lib, self.lib.cdae_hip_recommend_all(self.h, u_begin, u_end, topk, out.ctypes.data))
        return out

    def recommend_user(self, uid: int, rated_items, topk: int = 10) -> np.ndarray:
        """recommend(uid, topk, rated_item_set) for a set that is not the train row (cdae.hpp:162-196)."""
        r = np.ascontiguousarray(rated_items, dtype=np.uint32)
        out = np.empty(topk, dtype=np.uint32)
        _chk(self.lib, self.lib.cdae_hip_recommend_user(self.h, uid, r.ctypes.data, r.size, topk, out.ctypes.data))
        return out

    def set_test_rows(self, test_ptr, test_col):
        """the validation rows TOPN_Evaluation scores against (evaluation.hpp:118-120), CSR over this handle's users"""
        tp = np.ascontiguousarray(test_ptr, dtype=np.int64)
        tc = np.ascontiguousarray(test_col, dtype=np.uint32)
        _chk(self.lib, self.lib.cdae_hip_set_test_rows(self.h, tp.ctypes.data, tc.ctypes.data))

    def eval_topn(self, topk: int = 10, with_ids: bool = False):
        """TOPN_Evaluation::evaluate on the device (evaluation.hpp:113-219): (rets[8], hits[3]) or (rets, hits, ids)."""
        rets, hits = np.empty(8, dtype=np.float64), np.empty(3, dtype=np.uint64)
        ids = np.empty((self.num_users, topk), dtype=np.uint32) if with_ids else None
        _chk(self.lib, self.lib.cdae_hip_eval_topn(self.h, topk, rets.ctypes.data, hits.ctypes.data, ids.ctypes.data if with_ids else None))
        return (rets, hits, ids) if with_ids else (rets, hits)

    def pre_recommend(self, topk: int = 10):
        self._rec = self.recommend_all(topk)

    def recommend(self, uid: int, topk: int = 10):
        if self._rec is None or self._rec.shape[1] != topk:
            self.pre_recommend(topk)
        return self._rec[uid]

    # ---- data-parallel exchange --------------------------------------------------------------------
    def stream_handle(self) -> int:
        p = C.c_void_p()
        _chk(self.lib, self.lib.cdae_hip_stream(self.h, C.byref(p)))
        return p.value

    def delta_begin(self):
        _chk(self.lib, self.lib.cdae_hip_delta_begin(self.h))

    def delta_compute(self):
        _chk(self.lib, self.lib.cdae_hip_delta_compute(self.h))

    def delta_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        _chk(self.lib, self.lib.cdae_hip_delta_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def delta_apply(self, world_size: int, rule: int = 0):
        _chk(self.lib, self.lib.cdae_hip_delta_apply(self.h, world_size, rule))

    def delta_set_combine(self, combine: int):
        _chk(self.lib, self.lib.cdae_hip_delta_set_combine(self.h, combine))

    def delta_stage(self):
        _chk(self.lib, self.lib.cdae_hip_delta_stage(self.h))

    def delta_recv_device_ptr(self):
        p, n = C.c_void_p(), C.c_size_t()
        _chk(self.lib, self.lib.cdae_hip_delta_recv_device_ptr(self.h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def delta_merge(self):
        _chk(self.lib, self.lib.cdae_hip_delta_merge(self.h))

    def delta_merge_stage(self):
        _chk(self.lib, self.lib.cdae_hip_delta_merge_stage(self.h))

    # ---- library-owned communicator + exchange schedule (one process per GPU) ---------------------------------------
    def comm_init_rank(self, world_size: int, rank: int, unique_id: bytes):
        buf = C.create_string_buffer(bytes(unique_id), COMM_ID_BYTES)
        _chk(self.lib, self.lib.cdae_hip_comm_init_rank(self.h, world_size, rank, buf, COMM_ID_BYTES))

    def exchange_configure(self, period: int):
        _chk(self.lib, self.lib.cdae_hip_exchange_configure(self.h, period))

    def exchange_step(self):
        _chk(self.lib, self.lib.cdae_hip_exchange_step(self.h))

    def exchange_flush(self):
        _chk(self.lib, self.lib.cdae_hip_exchange_flush(self.h))

    def exchange_time_all_reduce(self, repeats: int = 5) -> float:
        v = C.c_double()
        _chk(self.lib, self.lib.cdae_hip_exchange_time_all_reduce(self.h, repeats, C.byref(v)))
        return v.value


@dataclass
class MFConfig:
    """libcf::IMFConfig / BPRConfig (/root/reference/src/model/recsys/imf.hpp:12-23, bpr.hpp:12-24); pairwise=True is BPR."""
    learn_rate: float = 0.1
    beta: float = 1.0
    lambda_: float = 0.01
    lt: int = SQUARE
    num_dim: int = 10
    num_neg: int = 5
    using_bias_term: bool = True
    using_adagrad: bool = True
    pairwise: bool = False
    batch_users: int = DEFAULT_BATCH_USERS


class MF(CDAE):
    """Host mirror of libcf::IMF / libcf::BPR over the same C ABI handle (cdae_hip_create_mf).  get / set use P_WU = uv_,
    P_W = iv_, P_UB = ub_, P_BP = ib_ (and their *_AG accumulators)."""

    def __init__(self, mcfg: MFConfig, device: int = 0):
        self.lib = load_library()
        self.cfg = mcfg
        c = _MfConfig(C.sizeof(_MfConfig), mcfg.num_dim, mcfg.num_neg, mcfg.lt, int(mcfg.using_adagrad), int(mcfg.using_bias_term),
                      int(mcfg.pairwise), mcfg.batch_users, mcfg.lambda_, mcfg.learn_rate, mcfg.beta)
        self.h = C.c_void_p()
        _chk(self.lib, self.lib.cdae_hip_create_mf(C.byref(c), device, C.byref(self.h)))
        self.num_users = self.num_items = 0
        self._rec = None

    def _shape(self, which):
        if which in (P_UB, P_UB_AG):
            return (self.num_users,)
        return CDAE._shape(self, which)


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the library: rank 0 makes it, every rank passes it to CDAE.comm_init_rank."""
    lib = load_library()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _chk(lib, lib.cdae_hip_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


COMBINE_SUM, COMBINE_GLOBAL_ACC = 0, 1


class _MultiSchedule(C.Structure):
    _fields_ = [("period", C.c_int32), ("combine", C.c_uint32), ("sync_batch_users", C.c_uint32), ("reserved", C.c_uint32),
                ("relay_epochs", C.c_double)]


class MultiCDAE:
    """Several user shards behind one handle (cdae_hip_multi_*): the data-parallel form of libcf::CDAE in one process.

    devices = [0, 1, ..., N-1]: one shard per GPU (RCCL); devices = [0] * N: N logical shards of GPU 0 (tests, accuracy
    envelope).  exchange_every: 0 = synchronous exchange of the shared-parameter deltas at every step, k >= 1 = pipelined."""

    def __init__(self, mcfg: CDAEConfig, devices, exchange_every: int = 0, item_rows: bool = False):
        """item_rows=True: CDAE_LAYOUT_ITEM_ROWS — the shards cut the item rows (exact single-GPU schedule, sampled or full-output decode; the user node sharded by user)."""
        self.lib = load_library()
        self.cfg = mcfg
        c = _Config(C.sizeof(_Config), mcfg.num_dim, mcfg.num_neg, mcfg.num_corruptions, mcfg.lt,
                    int(mcfg.using_adagrad), int(mcfg.asymmetric), int(mcfg.user_factor), int(mcfg.linear),
                    int(mcfg.scaled), int(mcfg.tanh), mcfg.batch_users, int(mcfg.full_output), int(mcfg.linear_function),
                    mcfg.lambda_, mcfg.learn_rate, mcfg.corruption_ratio, mcfg.beta)
        devs = (C.c_int * len(devices))(*devices)
        self.h = C.c_void_p()
        _chk(self.lib, self.lib.cdae_hip_multi_create(C.byref(c), len(devices), devs, C.byref(self.h)))
        _chk(self.lib, self.lib.cdae_hip_multi_set_exchange(self.h, exchange_every))
        if item_rows:
            _chk(self.lib, self.lib.cdae_hip_multi_set_layout(self.h, 1))
        self.num_users = self.num_items = 0

    def close(self):
        if getattr(self, "h", None):
            self.lib.cdae_hip_multi_destroy(self.h)
            self.h = None

    __del__ = close

    def set_exchange(self, period: int):
        _chk(self.lib, self.lib.cdae_hip_multi_set_exchange(self.h, period))

    def set_schedule(self, period: int = 0, combine: int = COMBINE_SUM, sync_batch_users: int = 0, relay_epochs: float = 0.0):
        """cdae_hip_multi_set_schedule (user-sharded layout): the first `relay_epochs` epochs (fractions allowed) on the single-GPU
        schedule handed from shard to shard, the rest as exchanged steps of `sync_batch_users` users per shard folded in by `combine`"""
        sc = _MultiSchedule(period, combine, sync_batch_users, 0, relay_epochs)
        _chk(self.lib, self.lib.cdae_hip_multi_set_schedule(self.h, C.byref(sc)))

    def set_interactions(self, num_users, num_items, row_ptr, col_idx):
        rp = np.ascontiguousarray(row_ptr, dtype=np.int64)
        ci = np.ascontiguousarray(col_idx, dtype=np.uint32)
        _chk(self.lib, self.lib.cdae_hip_multi_set_interactions(self.h, num_users, num_items, rp.ctypes.data, ci.ctypes.data))
        self.num_users, self.num_items = int(num_users), int(num_items)

    def reset(self, train, seed: int = 0):
        self.set_interactions(train.num_users, train.num_items, train.train_ptr, train.train_col)
        self.init_params(seed)

    def init_params(self, seed: int):
        _chk(self.lib, self.lib.cdae_hip_multi_init_params(self.h, seed))

    def shard_profiling(self, shard: int, period: int, families=None):
        """cdae_hip_set_profiling (+ _families) on ONE shard's handle: HIP events around that shard's decode / row launches"""
        h = C.c_void_p()
        _chk(self.lib, self.lib.cdae_hip_multi_shard(self.h, shard, C.byref(h), None, None))
        _chk(self.lib, self.lib.cdae_hip_set_profiling(h, period))
        if families is not None:
            _chk(self.lib, self.lib.cdae_hip_set_profiling_families(h, sum(1 << f for f in families)))

    def shard_stats(self, shard: int) -> Stats:
        """the counters and HIP-event kernel times ONE shard's handle accumulated since they were last collected"""
        h = C.c_void_p()
        _chk(self.lib, self.lib.cdae_hip_multi_shard(self.h, shard, C.byref(h), None, None))
        st = Stats()
        _chk(self.lib, self.lib.cdae_hip_collect_stats(h, C.byref(st)))
        return st

    def shards(self):
        """[(u_begin, u_end)] of every shard"""
        out = []
        for s in range(self.lib.cdae_hip_multi_num_shards(self.h)):
            a, b = C.c_uint64(), C.c_uint64()
            _chk(self.lib, self.lib.cdae_hip_multi_shard(self.h, s, None, C.byref(a), C.byref(b)))
            out.append((a.value, b.value))
        return out

    def train_one_iteration(self, seed: int, epoch: int) -> Stats:
        st = Stats()
        _chk(self.lib, self.lib.cdae_hip_multi_train_epoch(self.h, seed, epoch, C.byref(st)))
        return st

    def train_users(self, seed: int, epoch: int, u_begin: int, u_end: int) -> Stats:
        st = Stats()
        _chk(self.lib, self.lib.cdae_hip_multi_train_users(self.h, seed, epoch, u_begin, u_end, C.byref(st)))
        return st

    @property
    def steps_per_epoch(self) -> int:
        """user-sharded layout: exchanged steps of one epoch (cdae_hip_multi_steps_per_epoch)"""
        return int(self.lib.cdae_hip_multi_steps_per_epoch(self.h))

    def train_steps(self, seed: int, epoch: int, step_begin: int, step_end: int) -> Stats:
        """steps [step_begin, step_end) of the epoch's exchanged part, no relay, flushed at the end (measurement hook)"""
        st = Stats()
        _chk(self.lib, self.lib.cdae_hip_multi_train_steps(self.h, seed, epoch, step_begin, step_end, C.byref(st)))
        return st

    def current_loss(self, seed: int, epoch: int) -> float:
        a, b = C.c_double(), C.c_double()
        _chk(self.lib, self.lib.cdae_hip_multi_data_loss(self.h, seed, epoch, C.byref(a)))
        _chk(self.lib, self.lib.cdae_hip_multi_penalty_loss(self.h, C.byref(b)))
        return a.value + b.value

    def recommend_all(self, topk: int = 10, u_begin: int = 0, u_end=None) -> np.ndarray:
        u_end = self.num_users if u_end is None else u_end
        out = np.empty((u_end - u_begin, topk), dtype=np.uint32)
        _chk(self.lib, self.lib.cdae_hip_multi_recommend_all(self.h, u_begin, u_end, topk, out.ctypes.data))
        return out

    def eval_topn(self, test_ptr, test_col, topk: int = 10):
        tp = np.ascontiguousarray(test_ptr, dtype=np.int64)
        tc = np.ascontiguousarray(test_col, dtype=np.uint32)
        rets, hits = np.empty(8, dtype=np.float64), np.empty(3, dtype=np.uint64)
        _chk(self.lib, self.lib.cdae_hip_multi_eval_topn(self.h, tp.ctypes.data, tc.ctypes.data, topk, rets.ctypes.data, hits.ctypes.data, None))
        return rets, hits

    _shape = CDAE._shape

    def shard_get(self, shard: int, which) -> np.ndarray:
        """a SHARED parameter (item-side matrices, biases) as ONE shard's replica holds it (user-sharded layout: the replicas must agree)"""
        h = C.c_void_p()
        _chk(self.lib, self.lib.cdae_hip_multi_shard(self.h, shard, C.byref(h), None, None))
        out = np.empty(self._shape(which), dtype=np.float32)
        _chk(self.lib, self.lib.cdae_hip_get_param(h, which, out.ctypes.data, out.size))
        return out

    def get(self, which) -> np.ndarray:
        out = np.empty(self._shape(which), dtype=np.float32)
        _chk(self.lib, self.lib.cdae_hip_multi_get_param(self.h, which, out.ctypes.data, out.size))
        return out

    def set(self, which, arr):
        a = np.ascontiguousarray(arr, dtype=np.float32).reshape(self._shape(which))
        _chk(self.lib, self.lib.cdae_hip_multi_set_param(self.h, which, a.ctypes.data, a.size))
