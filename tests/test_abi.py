"""CPU-only: libttmi.so loads and exports every symbol include/ttmi.h declares; argument validation works without a GPU;
the product package never imports the oracle."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import PKG, ROOT


def _lib():
    so = os.path.join(PKG, "ttmi", "libttmi.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-j4"])
    return ctypes.CDLL(so)


def test_every_declared_symbol_is_exported():
    hdr = open(os.path.join(ROOT, "include", "ttmi.h")).read()
    names = sorted(set(re.findall(r"\b(ttmi_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25
    lib = _lib()
    for n in names:
        assert hasattr(lib, n), n


def test_every_exported_symbol_is_declared():
    """the other direction: an entry point the Python side binds must be in the published header"""
    hdr = open(os.path.join(ROOT, "include", "ttmi.h")).read()
    declared = set(re.findall(r"\b(ttmi_[a-z0-9_]+)\s*\(", hdr))
    _lib()
    so = os.path.join(PKG, "ttmi", "libttmi.so")
    out = subprocess.check_output(["nm", "-D", "--defined-only", so], text=True)
    exported = {ln.split()[-1] for ln in out.splitlines() if ln.split() and ln.split()[-1].startswith("ttmi_")}
    assert exported, "no ttmi_* exports found"
    assert exported - declared == set(), "exported but not declared in include/ttmi.h: %s" % sorted(exported - declared)
    assert declared - exported == set(), "declared but not exported: %s" % sorted(declared - exported)
    # and everything ttmi/ops.py / ttmi/train.py / ttmi/frontend.py call through ctypes is a declared name
    used = set()
    for d, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                used |= set(re.findall(r"\.(ttmi_[a-z0-9_]+)\b", open(os.path.join(d, f)).read()))
    assert used - declared == set(), "bound in Python but not declared: %s" % sorted(used - declared)


def test_argument_validation_without_gpu():
    lib = _lib()
    lib.ttmi_last_error.restype = ctypes.c_char_p
    assert lib.ttmi_version() >= 100
    rc = lib.ttmi_rnnt_loss_fwd(None, 0, ctypes.c_long(5), None, None, None, 1, 1, 1, 5, 0, None, None, None)
    assert rc < 0 and b"null pointer" in lib.ttmi_last_error()
    rc = lib.ttmi_gemm(None, None, None, None, None, 0, 0, 0, 4, 4, 4, ctypes.c_long(4), ctypes.c_long(4), ctypes.c_long(4),
                       1, 1, *([ctypes.c_long(0)] * 6), ctypes.c_float(1), ctypes.c_float(0), 48, 1, None)
    assert rc < 0
    lib.ttmi_rnnt_workspace_bytes.restype = ctypes.c_size_t
    assert lib.ttmi_rnnt_workspace_bytes(2, 10, 4) > 0
    lib.ttmi_attn_ctx_floats.restype = ctypes.c_size_t
    f32 = lib.ttmi_attn_ctx_floats(2, 50, 64, 2, 32, 0)
    bf = lib.ttmi_attn_ctx_floats(2, 50, 64, 2, 32, 1)
    assert 0 < bf < f32


def test_product_never_imports_oracle():
    for d, _, files in os.walk(PKG):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(d, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), os.path.join(d, f)


def test_ops_fail_loudly_without_device():
    import torch
    from ttmi import ops
    with pytest.raises(ValueError):
        ops.rnnt_loss_fwd(torch.zeros(1, 2, 2, 4), torch.zeros(1, 1, dtype=torch.int32), torch.ones(1, dtype=torch.int32),
                          torch.ones(1, dtype=torch.int32), 0, torch.zeros(64))
    from warprnnt_pytorch import RNNTLoss
    with pytest.raises(ValueError):
        RNNTLoss()(torch.zeros(1, 2, 2, 4), torch.zeros(1, 1, dtype=torch.int32), torch.tensor([2], dtype=torch.int32),
                   torch.tensor([1], dtype=torch.int32))
