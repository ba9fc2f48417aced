"""cdae_amd — MI355X-native CDAE training hot path (HIP kernels behind a C ABI) and its host mirror.

Layout: csrc/ (gfx950 kernels + C ABI, built into lib/libcdae_hip.so), binding.py (ctypes + the
libcf::CDAE-shaped host classes CDAE / MultiCDAE / MF), synth.py (BASELINE-shaped synthetic data), distributed.py (shard bounds
and device-buffer views for hosts of the library's multi-GPU layouts; the exchange itself lives in csrc/cdae_multi.hip).
"""
from .binding import (CDAE, MultiCDAE, MF, MFConfig, comm_unique_id, CDAEConfig, HINGE, LOG, P_UB, P_UB_AG, CDAEError, CROSS_ENTROPY, SQUARE, LOGISTIC, Stats,  # noqa: F401
                      load_library, developer_library, LIB_PATH, DEV_LIB_PATH, EXPORTS, P_W, P_W_AG, P_V, P_V_AG, P_WU, P_WU_AG, P_B, P_B_AG, P_BP, P_BP_AG,
                      P_UU, P_UU_AG, P_COUNT, COMBINE_SUM, COMBINE_GLOBAL_ACC)
from . import synth  # noqa: F401
