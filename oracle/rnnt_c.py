"""ctypes binding of oracle/rnnt_lattice.c (checker only; see that file's header)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "librnnt_oracle.so")
        if not os.path.exists(so):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _LIB = ctypes.CDLL(so)
        _LIB.rnnt_oracle_f32.restype = ctypes.c_int
    return _LIB


def rnnt_loss_c(logits, labels, act_lens, label_lens, blank=0, reduction="mean", want_grad=True):
    logits = np.ascontiguousarray(logits, dtype=np.float32)
    labels = np.ascontiguousarray(labels, dtype=np.int32)
    act_lens = np.ascontiguousarray(act_lens, dtype=np.int32)
    label_lens = np.ascontiguousarray(label_lens, dtype=np.int32)
    B, T, U1, V = logits.shape
    costs = np.zeros(B, dtype=np.float32)
    grad = np.empty_like(logits) if want_grad else None
    scale = 1.0 / B if reduction == "mean" else 1.0
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p) if a is not None else None
    rc = lib().rnnt_oracle_f32(p(logits), p(labels), p(act_lens), p(label_lens), B, T, U1, V, blank,
                               ctypes.c_float(scale), p(costs), p(grad))
    if rc:
        raise ValueError("rnnt_oracle_f32 failed: %d" % rc)
    loss = costs.sum() * scale if reduction in ("mean", "sum") else costs
    return loss, costs, grad
