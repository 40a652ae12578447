"""No-GPU checks of the drop-in boundary: libspacap_hip.so loads, exports every symbol include/spacap_hip.h
declares, argument validation works without touching a device, and the host shim fails loudly on CPU tensors.
No compute call is made here."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "spacap_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(spacap_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from spacap3d_amd import _native
    syms = _declared_symbols()
    assert len(syms) >= 17
    raw = ctypes.CDLL(_native.LIB_PATH)
    for s in syms:
        assert hasattr(raw, s), f"{s} declared in include/spacap_hip.h but not exported"
        assert s in _native.SIGNATURES, f"{s} has no ctypes signature in spacap3d_amd/_native.py"
    assert _native.lib.spacap_abi_version() == _native.ABI_VERSION == 4


def test_opt_n_threads_restated_exactly(oracle_ext):
    from spacap3d_amd import ext
    for w in list(range(1, 70)) + [511, 512, 513, 1024, 2048, 40000, 80000]:
        assert ext.opt_n_threads(w) == oracle_ext.opt_n_threads(w)


def test_argument_validation_needs_no_device():
    from spacap3d_amd._native import lib
    assert lib.spacap_fps_f32(None, 1, 0, 4, None, None, None) == -1      # N < 1
    assert b"bad sizes" in lib.spacap_last_error()
    assert lib.spacap_fps_f32(None, 2, 100, 4, None, None, None) == -1    # null pointers
    assert b"null" in lib.spacap_last_error()
    assert lib.spacap_fps_f32(None, 0, 100, 4, None, None, None) == 0     # empty batch is a no-op
    assert lib.spacap_ball_query_f32(None, None, 0, 10, 10, 0.1, 4, None, None) == 0
    assert lib.spacap_fps_workspace_bytes(8, 40000) >= 8 * 40000 * 4
    assert lib.spacap_mha_bwd_workspace_bytes(8, 8, 256) == 8 * 8 * 256 * 4
    z = [0] * 9
    assert lib.spacap_mha_fwd_f32(None, None, None, *z, None, 0, 0, None, 0, 0, 0, 1, 8, 16, 16, 24, 0.25, 0.0, 0, None,
                                  None, None, None, None) == -1
    assert b"d_k=24" in lib.spacap_last_error()


def test_fps_workspace_covers_the_dispatched_bucket_template():
    """The bucketed FPS kernel indexes its planes with NPAD = NB*4096 where NB is the template instance that gets
    launched (3,4,5,6,8,10,12,16 with the index map in LDS; 16,20 beyond 65 535 points), not ceil(N/4096)."""
    from spacap3d_amd._native import lib
    inst = (3, 4, 5, 6, 8, 10, 12, 16)
    for N in (8193, 24576, 24577, 28672, 33000, 36864, 40000, 40960, 40961, 45056, 50000, 61440, 65535, 65536, 70000,
              77824, 81920):
        nb = (N + 4095) // 4096
        NB = next(v for v in inst if nb <= v) if N <= 65535 else (16 if nb <= 16 else 20)
        planes = 3 if N <= 65535 else 4
        for B in (1, 8):
            assert lib.spacap_fps_workspace_bytes(B, N) >= B * planes * NB * 4096 * 4, N


def test_host_shim_fails_loudly_on_cpu_tensors():
    from spacap3d_amd import attention, ext
    xyz = torch.rand(1, 64, 3)
    for call in (lambda: ext.furthest_point_sampling(xyz, 4),
                 lambda: ext.ball_query(xyz[:, :4].contiguous(), xyz, 0.1, 4),
                 lambda: ext.three_nn(xyz, xyz),
                 lambda: ext.group_points(torch.rand(1, 3, 64), torch.zeros(1, 4, 2, dtype=torch.int32)),
                 lambda: ext.gather_points(torch.rand(1, 3, 64), torch.zeros(1, 4, dtype=torch.int32)),
                 lambda: attention.attention(torch.rand(1, 2, 8, 16), torch.rand(1, 2, 8, 16), torch.rand(1, 2, 8, 16))):
        with pytest.raises(RuntimeError, match="CPU not supported"):
            call()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "spacap3d_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liboracle" not in src, f
