"""ctypes view of libe2e_ctc.so: the raw C ABI of include/e2e_ctc.h, symbol by symbol.

This is what the parity tests (tests/gpu_util.py), bench.py's C-ABI leg and tools/diag call, so that what they
exercise is the boundary itself -- plain pointers and sizes, return codes, e2e_last_error().  The product's engines
(end2end_amd/engines.py) reach the same entry points through the pybind11 layer end2end_amd._C instead.
There is no CPU fallback.  Nothing here touches oracle/.
"""
import ctypes as C
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("E2E_CTC_LIB") or os.path.join(_HERE, "csrc", "libe2e_ctc.so")      # (E2E_CTC_LIB: tools/diag A/B builds)

F32, F64, F16, BF16 = 0, 1, 2, 3
ALGO_AUTO, ALGO_EXACT, ALGO_FAST = 0, 1, 2
ABI_VERSION = 3

_lib = None
_lock = threading.Lock()


REDUCE_NONE, REDUCE_SUM, REDUCE_MEAN = 0, 1, 2
CHAINS_F64, CHAINS_F32 = 0, 1


class LossOpts(C.Structure):
    """e2e_ctc_loss_opts (include/e2e_ctc.h)."""
    _fields_ = [("grad_scale", C.c_double), ("reduced", C.c_void_p), ("reduction", C.c_int), ("chains", C.c_int)]


class E2EError(RuntimeError):
    """An error reported by the native library (message from e2e_last_error())."""


def load():
    """Load libe2e_ctc.so once; raise loudly if it is absent or has the wrong ABI."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "end2end_amd: native library %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C end2end_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        vp, i64, i64p = C.c_void_p, C.c_int64, C.c_void_p
        L.e2e_ctc_abi_version.restype = C.c_int
        # (first of all: a library of another ABI may lack symbols bound below, and would fail with an AttributeError instead)
        if L.e2e_ctc_abi_version() != ABI_VERSION:
            raise ImportError("end2end_amd: %s has ABI %d, expected %d" % (LIB_PATH, L.e2e_ctc_abi_version(), ABI_VERSION))
        L.e2e_last_error.restype = C.c_char_p
        L.e2e_ctc_loss_workspace_bytes.restype = C.c_size_t
        L.e2e_ctc_loss_workspace_bytes.argtypes = [C.c_int] * 6
        L.e2e_ctc_loss_takes_dtype.restype = C.c_int
        L.e2e_ctc_loss_takes_dtype.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i64, i64, i64, vp, vp]
        L.e2e_ctc_loss_fwd_bwd.restype = C.c_int
        L.e2e_ctc_loss_fwd_bwd.argtypes = [vp, C.c_int, C.c_int, i64, i64, i64, i64p, i64, i64p, i64p,
                                           C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                           vp, vp, vp, C.c_size_t, C.c_int, vp]
        L.e2e_ctc_loss_fwd_bwd_opt.restype = C.c_int
        L.e2e_ctc_loss_fwd_bwd_opt.argtypes = L.e2e_ctc_loss_fwd_bwd.argtypes + [C.POINTER(LossOpts)]
        L.e2e_ctc_scale_grads.restype = C.c_int
        L.e2e_ctc_scale_grads.argtypes = [vp, C.c_int, vp, C.c_int, i64, vp]
        L.e2e_ctc_greedy.restype = C.c_int
        L.e2e_ctc_greedy.argtypes = [vp, C.c_int, i64, i64, i64, i64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                     i64p, i64p, vp]
        L.e2e_lm_load_arpa.restype = C.c_int
        L.e2e_lm_load_arpa.argtypes = [C.c_char_p, C.POINTER(C.c_char_p), C.c_int, C.c_int, C.POINTER(vp)]
        L.e2e_lm_free.argtypes = [vp]
        L.e2e_lm_order.restype = C.c_int
        L.e2e_lm_order.argtypes = [vp]
        L.e2e_lm_device.restype = C.c_int
        L.e2e_lm_device.argtypes = [vp]
        L.e2e_lm_word_index.restype = C.c_uint32
        L.e2e_lm_word_index.argtypes = [vp, C.c_char_p]
        L.e2e_lm_score.restype = C.c_double
        L.e2e_lm_score.argtypes = [vp, C.POINTER(C.c_uint32), C.c_int, C.c_uint32]
        L.e2e_ctc_beam_workspace_bytes.restype = C.c_size_t
        L.e2e_ctc_beam_workspace_bytes.argtypes = [C.c_int] * 4
        L.e2e_ctc_beam_workspace_bytes_lm.restype = C.c_size_t
        L.e2e_ctc_beam_workspace_bytes_lm.argtypes = [C.c_int] * 5
        L.e2e_ctc_beam_max_width.restype = C.c_int
        L.e2e_ctc_beam_max_width.argtypes = [C.c_int, C.c_int]
        L.e2e_ctc_beam.restype = C.c_int
        L.e2e_ctc_beam.argtypes = [vp, C.c_int, i64, i64, i64, i64p, C.c_int, C.c_int, C.c_int, C.c_int,
                                   C.c_int, C.c_int, vp, C.c_double, C.c_double, C.c_double,
                                   i64p, i64, i64p, vp, C.c_size_t, vp]
        L.e2e_ctc_align_workspace_bytes.restype = C.c_size_t
        L.e2e_ctc_align_workspace_bytes.argtypes = [C.c_int] * 5
        L.e2e_ctc_align.restype = C.c_int
        L.e2e_ctc_align.argtypes = [vp, C.c_int, i64, i64, i64, i64p, i64, i64p, i64p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i64p, i64, vp, C.c_size_t, vp]
        L.e2e_debug_stream_copy.restype = C.c_int
        L.e2e_debug_stream_copy.argtypes = [vp, vp, C.c_size_t, vp]
        # (the dtype codes above are include/e2e_ctc.h's: the pybind layer, which is compiled against the header, carries them)
        try:
            from . import _C as _ext
            assert (F32, F64, F16, BF16) == (_ext.F32, _ext.F64, _ext.F16, _ext.BF16), "dtype codes of _lib.py and include/e2e_ctc.h differ"
        except ImportError:
            pass
        _lib = L
    return _lib


def check(rc):
    if rc != 0:
        raise E2EError("libe2e_ctc: %s (code %d)" % (load().e2e_last_error().decode(errors="replace"), rc))


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("end2end_amd needs an AMD GPU (MI355X / gfx950) visible to PyTorch-ROCm; "
                           "there is no CPU fallback in this package")


def dtype_code(dt):
    if dt == torch.float32:
        return F32
    if dt == torch.float64:
        return F64
    if dt == torch.float16:
        return F16
    if dt == torch.bfloat16:
        return BF16
    raise TypeError("unsupported dtype %s" % dt)


def stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
