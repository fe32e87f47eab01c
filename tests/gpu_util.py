"""Helpers for the -m gpu parity tests: raw C-ABI calls on device tensors."""
import numpy as np
import torch

from end2end_amd import _lib


def dev():
    return torch.device("cuda", 0)


def c_abi_loss(x, targets, x_len, t_len, blank=0, logprobs=True, algo=_lib.ALGO_AUTO, keep=None, opts=None, chains=None):
    """x: torch tensor (B,T,V) on any device with any strides (moved to the GPU keeping its layout).
    opts = (grad_scale, reduction): call e2e_ctc_loss_fwd_bwd_opt and return (losses, grads, reduced).
    chains = _lib.CHAINS_F32: the same entry point with e2e_ctc_loss_opts.chains set (returns (losses, grads))."""
    L = _lib.load()
    d = dev()
    if not x.is_cuda:
        # keep the stride pattern: move the underlying storage then re-view
        base = x
        x = torch.empty_strided(base.shape, base.stride(), dtype=base.dtype, device=d)
        x.copy_(base)
    B, T, V = x.shape
    targets = torch.as_tensor(np.asarray(targets)).to(d, torch.long).reshape(B, -1).contiguous()
    if targets.shape[1] == 0:
        targets = torch.zeros((B, 1), dtype=torch.long, device=d)
    xl = torch.as_tensor(np.asarray(x_len)).to(d, torch.long)
    tl = torch.as_tensor(np.asarray(t_len)).to(d, torch.long)
    Smax = targets.shape[1]
    loss_dtype = torch.float32 if x.dtype in (torch.float16, torch.bfloat16) else x.dtype     # (16-bit calls keep their losses in f32)
    losses = torch.full((B,), 7.0, dtype=loss_dtype, device=d)
    grads = torch.full((B, T, V), 7.0, dtype=x.dtype, device=d)
    code = _lib.dtype_code(x.dtype)
    n = L.e2e_ctc_loss_workspace_bytes(B, T, V, Smax, code, algo)
    ws = torch.empty(n, dtype=torch.uint8, device=d)
    sB, sT, sV = x.stride()
    args = (x.data_ptr(), code, 1 if logprobs else 0, sB, sT, sV,
            targets.data_ptr(), targets.stride(0), xl.data_ptr(), tl.data_ptr(),
            B, T, V, Smax, blank, losses.data_ptr(), grads.data_ptr(),
            ws.data_ptr(), ws.numel(), algo, _lib.stream_ptr(d))
    if opts is None and chains is None:
        _lib.check(L.e2e_ctc_loss_fwd_bwd(*args))
    elif opts is None:
        import ctypes
        o = _lib.LossOpts(1.0, None, _lib.REDUCE_NONE, int(chains))
        _lib.check(L.e2e_ctc_loss_fwd_bwd_opt(*args, ctypes.byref(o)))
    else:
        reduced = torch.full((1,), 7.0, dtype=loss_dtype, device=d)
        o = _lib.LossOpts(float(opts[0]), reduced.data_ptr() if opts[1] else None, int(opts[1]))
        import ctypes
        _lib.check(L.e2e_ctc_loss_fwd_bwd_opt(*args, ctypes.byref(o)))
    torch.cuda.synchronize()
    if keep is not None:
        keep["workspace"] = ws              # (diagnostics read the fast path's flag words out of it)
    if grads.dtype in (torch.float16, torch.bfloat16):
        grads = grads.float()                                # (numpy has no bf16)
    if opts is not None:
        return losses.cpu().numpy(), grads.cpu().numpy(), reduced.cpu().numpy()[0]
    return losses.cpu().numpy(), grads.cpu().numpy()


def c_abi_greedy(x, x_len, blank=0):
    L = _lib.load()
    d = dev()
    if not x.is_cuda:
        base = x
        x = torch.empty_strided(base.shape, base.stride(), dtype=base.dtype, device=d)
        x.copy_(base)
    B, T, V = x.shape
    xl = torch.as_tensor(np.asarray(x_len)).to(d, torch.long)
    out = torch.full((B, T), -7, dtype=torch.long, device=d)
    out_len = torch.full((B,), -7, dtype=torch.long, device=d)
    sB, sT, sV = x.stride()
    _lib.check(L.e2e_ctc_greedy(x.data_ptr(), _lib.dtype_code(x.dtype), sB, sT, sV, xl.data_ptr(),
                                B, T, V, blank, out.data_ptr(), out_len.data_ptr(), _lib.stream_ptr(d)))
    torch.cuda.synchronize()
    return out.cpu().numpy(), out_len.cpu().numpy()


def assert_same(got, want, rtol, atol, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    assert np.array_equal(np.isnan(got), np.isnan(want)), what + ": NaN pattern differs"
    inf = np.isinf(want)
    assert np.array_equal(np.isinf(got), inf), what + ": inf pattern differs"
    assert np.array_equal(got[inf], want[inf]), what
    ok = ~(np.isnan(want) | inf)
    np.testing.assert_allclose(got[ok], want[ok], rtol=rtol, atol=atol, err_msg=what)


def c_abi_beam(lp, x_len=None, blank=0, beam_width=100, labels=None, lm=None, lmwt=1.0, wip=0.0, oov_penalty=-1000.0):
    """Raw C-ABI prefix beam search on log-probs -> (ids [B,maxlen], lengths [B])."""
    L = _lib.load()
    d = dev()
    if not lp.is_cuda:
        base = lp
        lp = torch.empty_strided(base.shape, base.stride(), dtype=base.dtype, device=d)
        lp.copy_(base)
    B, T, V = lp.shape
    if x_len is None:
        x_len = [T] * B
    xl = torch.as_tensor(np.asarray(x_len)).to(d, torch.long)
    labels = list(labels or [])
    space_id = labels.index(" ") if " " in labels else -1
    max_out = T + 1
    out = torch.full((B, max_out), -7, dtype=torch.long, device=d)
    out_len = torch.full((B,), -7, dtype=torch.long, device=d)
    n = L.e2e_ctc_beam_workspace_bytes(B, T, V, beam_width)
    ws = torch.empty(n, dtype=torch.uint8, device=d)
    sB, sT, sV = lp.stride()
    _lib.check(L.e2e_ctc_beam(lp.data_ptr(), _lib.dtype_code(lp.dtype), sB, sT, sV, xl.data_ptr(), B, T, V, blank,
                              beam_width, space_id, lm.on(d).handle if lm is not None else None, lmwt, wip, oov_penalty,
                              out.data_ptr(), max_out, out_len.data_ptr(), ws.data_ptr(), ws.numel(), _lib.stream_ptr(d)))
    lens = out_len.cpu().numpy()
    # in-band status (include/e2e_ctc.h): -1 = node pool exhausted, > max_out = truncated
    assert ((lens >= 0) & (lens <= max_out)).all(), lens
    width = int(lens.max()) if B else 0
    return out[:, :width].cpu().numpy(), lens
