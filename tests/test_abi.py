"""CPU checks of the C-ABI boundary: the shared library builds, loads, and exports every symbol include/lafs_hip.h declares
(no compute calls: there is no GPU in the build container)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "lafs_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lafs_[a-z0-9_]+)\s*\(", txt)))


def test_header_declares_the_documented_surface():
    syms = header_symbols()
    for must in ("lafs_gemm_nt", "lafs_gemm_tn_acc", "lafs_layernorm_fwd", "lafs_layernorm_bwd", "lafs_attention_fwd", "lafs_attention_bwd",
                 "lafs_dino_loss_fwd_bwd", "lafs_clip_adamw_ema", "lafs_trunk_forward", "lafs_trunk_backward", "lafs_patchify",
                 "lafs_margin_softmax_ce", "lafs_patch_gather_fwd", "lafs_mixup_normalize", "lafs_center_ema", "lafs_last_error"):
        assert must in syms


def test_library_builds_loads_and_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from lafs_cvpr2024_amd import _lib
    h = _lib.lib()
    assert h.lafs_version() == 1
    declared = header_symbols()
    missing = [s for s in declared if not hasattr(h, s)]
    assert not missing, f"declared in lafs_hip.h but not exported: {missing}"
    unbound = [s for s in declared if s not in _lib.EXPORTED]
    assert not unbound, f"declared but without a ctypes prototype in _lib.py: {unbound}"
    stale = [s for s in _lib.EXPORTED if s not in declared]
    assert not stale, f"bound in _lib.py but not declared in the header: {stale}"


def test_product_path_fails_loudly_without_a_gpu():
    """No CPU fallback: calling an op with host tensors must raise, not compute."""
    import torch
    from lafs_cvpr2024_amd import _lib, ops
    with pytest.raises(_lib.LafsHipError):
        ops.gemm_nt(torch.zeros(128, 64, dtype=torch.bfloat16), torch.zeros(128, 64, dtype=torch.bfloat16))


def test_bad_arguments_are_rejected_before_any_launch():
    import ctypes as C
    from lafs_cvpr2024_amd import _lib
    h = _lib.lib()
    a = _lib.GemmNTArgs()
    a.M, a.N, a.K = 128, 128, 48            # K not a multiple of 32, null pointers
    rc = h.lafs_gemm_nt(C.byref(a), None)
    assert rc < 0 and h.lafs_last_error()
