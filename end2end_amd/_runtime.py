"""Host-side plumbing shared by the engines: the pybind11 layer (`end2end_amd._C`, a thin wrapper of the C ABI in
include/e2e_ctc.h), device / dtype / stream helpers and the cached workspaces.

There is no CPU fallback: if the extension or libe2e_ctc.so is missing, or no MI355X is visible, the package raises.
Nothing here touches oracle/.
"""
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libe2e_ctc.so")

try:
    from . import _C
except ImportError as e:                                   # pragma: no cover - exercised by tests/test_host_cpu.py
    raise ImportError(
        "end2end_amd: the native extension is missing or does not load (%s) -- run "
        "`python -c 'import __graft_entry__ as g; g.build()'` or `make -C end2end_amd/csrc` "
        "(hipcc --offload-arch=gfx950 + g++/pybind11). There is no CPU fallback." % e) from e

F32, F64 = _C.F32, _C.F64
F16, BF16 = _C.F16, _C.BF16    # (read and written natively by the fast and wide loss paths)
ALGO_AUTO, ALGO_EXACT, ALGO_FAST = _C.ALGO_AUTO, _C.ALGO_EXACT, _C.ALGO_FAST
ABI_VERSION = 3
E2EError = _C.E2EError

if _C.abi_version() != ABI_VERSION or _C.ABI_VERSION != ABI_VERSION:
    raise ImportError("end2end_amd: libe2e_ctc.so has ABI %d, the extension was built for %d, the package expects %d"
                      % (_C.abi_version(), _C.ABI_VERSION, ABI_VERSION))


def require_gpu():
    if not torch.cuda.is_available():
        raise RuntimeError("end2end_amd needs an AMD GPU (MI355X / gfx950) visible to PyTorch-ROCm; "
                           "there is no CPU fallback in this package")


def compute_device(t):
    """Device the kernels run on for tensor t: its own if it is on a GPU, else the current GPU."""
    if t.is_cuda:
        return t.device
    require_gpu()
    return torch.device("cuda", torch.cuda.current_device())


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


_workspaces = {}


def workspace(device, nbytes):
    """A cached per-(device, stream) scratch buffer of at least nbytes (grown geometrically)."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _workspaces.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 256, dtype=torch.uint8, device=device)
        _workspaces[key] = buf
    return buf


def stream_handle(device):
    """The current HIP stream of `device` as an integer (hipStream_t)."""
    return torch.cuda.current_stream(device).cuda_stream
