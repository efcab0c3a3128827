"""ttmi - loader for libttmi.so (hand-written gfx950 HIP kernels behind a C ABI).

There is NO fallback: if the shared library is missing or a symbol is absent the
import of the ops fails loudly.  Build with `python __graft_entry__.py` or
`make -C transformer-transducer_amd/csrc`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TTMI_LIB") or os.path.join(_HERE, "libttmi.so")      # (TTMI_LIB: another build of the same library, for same-box A/B runs)
_lib = None


class TTMIError(RuntimeError):
    pass


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TTMIError(
                "libttmi.so not found at %s - the HIP extension is required (no CPU/PyTorch fallback). "
                "Build it: make -C transformer-transducer_amd/csrc" % LIB_PATH)
        # torch ships its own libamdhip64; it must be the HIP runtime of the process BEFORE libttmi.so resolves that soname - loaded the
        # other way round the process holds two runtimes and the library's launches see "no ROCm-capable device" (streams and device
        # pointers belong to torch's)
        import torch  # noqa: F401
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.ttmi_last_error.restype = ctypes.c_char_p
        _lib.ttmi_rnnt_workspace_bytes.restype = ctypes.c_size_t
        for kv in filter(None, os.environ.get("TTMI_OPTIONS", "").split(",")):       # measurement switches (ttmi_set_option), e.g. "11=1,12=256"
            k, v = kv.replace(":", "=").split("=")
            _lib.ttmi_set_option(int(k), int(v))
    return _lib


def last_error():
    return lib().ttmi_last_error().decode("utf-8", "replace")


def check(rc, what):
    if rc == 0:
        return
    if rc < 0:
        raise ValueError("%s: %s" % (what, last_error()))
    raise TTMIError("%s: HIP error %d: %s" % (what, rc, last_error()))
