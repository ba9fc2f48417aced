"""CPU: the C-ABI library cross-compiles, loads, and exports exactly what include/cdae_hip.h declares.
No compute is attempted here (there is no GPU in the build container)."""
import ctypes
import os
import re

import pytest

import cdae_amd
from cdae_amd import binding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "cdae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(cdae_hip_[a-z_0-9]+)\s*\(", text)))


def test_header_symbols_are_all_exported_and_bound(built):
    syms = declared_symbols()
    assert len(syms) >= 20
    lib = cdae_amd.load_library()      # (loads torch first: one HIP runtime per process, see binding.py)
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/cdae_hip.h but not exported"
    assert sorted(binding.EXPORTS) == syms, "binding.EXPORTS and the header disagree"


def test_abi_version_and_config_layout(built):
    lib = cdae_amd.load_library()
    assert lib.cdae_hip_abi_version() == 12
    hdr = open(os.path.join(ROOT, "include", "cdae_hip.h")).read()
    assert "#define CDAE_HIP_ABI_VERSION 12" in hdr
    # 14 uint32 + 4 double, naturally aligned
    assert ctypes.sizeof(binding._Config) == 14 * 4 + 4 * 8
    assert ctypes.sizeof(binding.Stats) == 8 * 11


def test_default_batch_users_is_the_certified_one(built):
    """The drop-in default (batch_users = 0 in cdae_hip_config, what src/model/recsys/cdae.hpp passes without CDAE_BATCH_USERS)
    must be the value the accuracy tests certify: tests/test_gpu_accuracy.py trains at bench.DEFAULT_BATCH_USERS."""
    import bench
    lib = cdae_amd.load_library()
    hdr = open(os.path.join(ROOT, "include", "cdae_hip.h")).read()
    cap = int(re.search(r"#define CDAE_DEFAULT_BATCH_USERS_MAX (\d+)u", hdr).group(1))
    assert cap == bench.DEFAULT_BATCH_USERS == 256
    assert lib.cdae_hip_default_batch_users(70_000) == bench.DEFAULT_BATCH_USERS          # ML-10M shape (BASELINE configs[2])
    assert lib.cdae_hip_default_batch_users(480_000) == bench.DEFAULT_BATCH_USERS         # Netflix shape (configs[3])
    assert lib.cdae_hip_default_batch_users(10_000) == 32 and lib.cdae_hip_default_batch_users(17) == 32
    # the sibling models (ABI 10): the certified block of each on BASELINE-sized data sets, the reference loop on smaller ones
    mf = lib.cdae_hip_mf_default_batch_users
    assert (mf(70_000, 0), mf(480_000, 0), mf(10_000, 0), mf(8_191, 0), mf(1, 0)) == (16, 16, 16, 1, 1)
    assert (mf(70_000, 1), mf(480_000, 1), mf(65_535, 1), mf(10_000, 1), mf(1, 1)) == (8, 8, 1, 1, 1)
    assert max(lib.cdae_hip_default_batch_users(u) for u in (1, 5_000, 40_000, 41_000, 10**7, 2**30 - 1)) <= cap


def test_no_cpu_fallback_without_a_device(built):
    """Without a HIP device the product path must fail loudly, not compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(cdae_amd.CDAEError):
        cdae_amd.CDAE(cdae_amd.CDAEConfig(lt=cdae_amd.SQUARE))


def test_library_is_gfx950_code_object(built):
    blob = open(cdae_amd.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    assert b"decode_rows_kernel" in blob and b"hidden_gather_kernel" in blob


def test_the_shipped_library_names_no_developer_switch(built):
    """The developer environment switches (A/B kernel paths, tuning knobs, CDAE_DEBUG_SKIP_* timing experiments that give WRONG results)
    exist in the -DCDAE_DEVELOPER build only: the shipped library calls getenv nowhere, so their names are not in its string table; the
    developer build, made from the same sources, has them."""
    switches = [b"CDAE_DEBUG_SKIP_ROLES", b"CDAE_DEBUG_SKIP_PREP", b"CDAE_SORT_TILE", b"CDAE_GEMM1_TILED", b"CDAE_FULL_UNFUSED", b"CDAE_WAVE_TRACE",
                b"CDAE_PREP2", b"CDAE_DUP_CAP", b"CDAE_XCHG_STREAM", b"CDAE_FULL_B_SUMMED", b"CDAE_DECODE_UNFUSED", b"CDAE_NO_LATE_ROWS", b"CDAE_XCHG_COLLECTIVE_STREAM", b"CDAE_XCHG_FULL_PASSES", b"CDAE_FUSED_BLOCK_ROUNDS", b"CDAE_HOST_PACE_US"]
    blob = open(cdae_amd.LIB_PATH, "rb").read()
    assert not [s for s in switches if s in blob]
    names = set(re.findall(rb"CDAE_[A-Z0-9_]{3,}", blob))
    assert len(names) <= 5, sorted(names)         # (error-message text such as a layout's name; no environment variable)
    dev = open(cdae_amd.DEV_LIB_PATH, "rb").read()
    assert b"CDAE_SORT_TILE" in dev and b"CDAE_GEMM1_TILED" in dev and b"CDAE_DECODE_UNFUSED" in dev and b"CDAE_FULL_B_SUMMED" not in dev


def test_product_sources_do_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing shipped may include, import or link it."""
    bad = []
    for base in ("cdae_amd", "include", "src"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".hpp", ".h", ".cpp", ".cc")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    if re.search(r"(^|\W)(import|from)\s+oracle\b|#include\s*[\"<][^\">]*oracle", text):
                        bad.append(os.path.join(dirpath, f))
    assert not bad, bad
