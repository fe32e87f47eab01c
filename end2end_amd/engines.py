"""The two engines of the reference's pybind modules, computed on the MI355X.

CTCLossEngine    <-> cpp_ctc_loss.CTCLossEngine     (src/losses/ctc_loss_py.cpp:8-16)
CTCDecoderEngine <-> cpp_ctc_decoder.CTCDecoder     (src/decoders/ctc_decoder_py.cpp:8-38)

Same constructor arguments, keyword names, defaults and results.  The top-level modules `cpp_ctc_loss` and
`cpp_ctc_decoder` of this repository export them under the reference's names, so that the reference's own callers
(`import_module("cpp_ctc_loss")`, pytorch_end2end/modules/ctc_loss.py:74; `import cpp_ctc_decoder`,
pytorch_end2end/decoders/ctc_decoder.py:13) find them.

Unlike the reference (which copies GPU tensors to the host and computes in C++ threads,
src/losses/forward_backward.cpp:12-19) tensors stay on the GPU: the engines hand device addresses, strides and sizes
to the C ABI (include/e2e_ctc.h) through the pybind11 layer `end2end_amd._C`.  CPU tensors are moved to the current GPU
and the loss results moved back to the source device and dtype (forward_backward.cpp:55-56); decode results are CPU
tensors as upstream (ctc_decoder.cpp:157,449) unless `keep_on_device` is set.
"""
import numpy as np
import torch

from . import _runtime as R
from ._runtime import _C


def _as_long(t, device):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    elif t.dtype == torch.long and t.device == device and t.is_contiguous():
        return t                         # (the usual case on a training step: nothing to convert, no dispatcher round trips)
    return t.to(device=device, dtype=torch.long).contiguous()


_REDUCTIONS = {None: _C.REDUCE_NONE, "sum": _C.REDUCE_SUM, "mean": _C.REDUCE_MEAN}
_ws_bytes = {}                           # (B, T, V, Smax, dtype code, algo) -> e2e_ctc_loss_workspace_bytes


def _on_device(dev):
    """A context that makes `dev` the current GPU -- none at all when it already is (the context manager costs ~10 us of host time
    per entry, which on a 130 us step is what decides whether the host stays ahead of the GPU)."""
    if torch.cuda.current_device() == dev.index:
        return _NULL_CTX
    return torch.cuda.device(dev)


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL_CTX = _NullCtx()


class CTCLossEngine:
    """blank_idx -> .compute(logits, targets, logits_lengths, targets_lengths) -> (losses[B], grads[B,T,V])."""

    def __init__(self, blank_idx, algo=R.ALGO_AUTO, f32_chains=False):
        """`f32_chains` (extension, e2e_ctc_loss_opts.chains): let the lattice chains run in packed f32 where that is
        faster (long targets, small alphabets): gradient elements within 2e-5 absolute of the reference instead of 2e-6."""
        self.blank_idx = int(blank_idx)
        self.algo = algo
        self.f32_chains = bool(f32_chains)

    def compute(self, logits, targets, logits_lengths, targets_lengths, input_is_logprobs=True,
                grad_scale=1.0, reduction=None):
        """`logits` is batch-major (B,T,V) (any strides).  With input_is_logprobs=True this is the
        reference engine: log-probabilities in, grads = exp(lp) - posterior.  With False the
        log-softmax is fused in and grads are d loss / d logits.

        Extensions (e2e_ctc_loss_opts): `grad_scale` multiplies every gradient element as the kernel writes it;
        `reduction` = "sum" / "mean" makes the call also return the reduced loss, written by the tail of its last
        kernel: the result is then (losses, grads, reduced)."""
        if logits.dim() != 3:
            raise ValueError("logits must be (batch, time, alphabet)")
        src_device, src_dtype = logits.device, logits.dtype
        dev = R.compute_device(logits)
        x = logits.detach().to(dev)
        if x.dtype in (torch.float16, torch.bfloat16) and self.algo != R.ALGO_EXACT:
            # 16-bit logits (raw autocast outputs): the fast and wide paths read them as they are and write the gradient in
            # the same dtype -- no f32 copy of the (B,T,V) tensors (the reference converts once to double,
            # src/losses/forward_backward.cpp:15,55-56).  Shapes those paths do not take (the library says which:
            # e2e_ctc_loss_takes_dtype) are up-cast below.  The gradient of a fresh allocation is 256-byte aligned.
            B, T, V = x.shape
            targets = _as_long(targets, dev)
            Smax = targets.shape[1] if targets.dim() == 2 else 0
            sB, sT, sV = x.stride()
            if _C.ctc_loss_takes_dtype(R.dtype_code(x.dtype), self.algo, T, V, Smax, sB, sT, sV, x.data_ptr(), 0):
                return self._compute_on_device(x, src_device, src_dtype, dev, targets, logits_lengths, targets_lengths,
                                               input_is_logprobs, grad_scale, reduction)
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        return self._compute_on_device(x, src_device, src_dtype, dev, targets, logits_lengths, targets_lengths,
                                       input_is_logprobs, grad_scale, reduction)

    def _compute_on_device(self, x, src_device, src_dtype, dev, targets, logits_lengths, targets_lengths,
                           input_is_logprobs, grad_scale, reduction):
        B, T, V = x.shape
        half = x.dtype in (torch.float16, torch.bfloat16)
        loss_dtype = torch.float32 if half else x.dtype        # (16-bit I/O: the library keeps losses in f32)
        targets = _as_long(targets, dev)
        if targets.dim() != 2 or targets.shape[0] != B:
            raise ValueError("targets must be (batch, max_target_length)")
        xl = _as_long(logits_lengths, dev)
        tl = _as_long(targets_lengths, dev)
        if xl.numel() != B or tl.numel() != B:
            raise ValueError("lengths must have one entry per utterance")
        Smax = targets.shape[1]
        losses = torch.empty(B, dtype=loss_dtype, device=dev)
        grads = torch.empty((B, T, V), dtype=x.dtype, device=dev)
        if reduction not in (None, "sum", "mean"):
            raise ValueError("reduction must be None, 'sum' or 'mean'")
        if B == 0:
            out = (losses.to(src_device, src_dtype), grads.to(src_device, src_dtype))
            return out if reduction is None else out + (getattr(out[0], reduction)(),)
        reduced = torch.empty((), dtype=loss_dtype, device=dev) if reduction else None
        code = R.dtype_code(x.dtype)
        with _on_device(dev):
            key = (B, T, V, Smax, code, self.algo)
            nbytes = _ws_bytes.get(key)
            if nbytes is None:
                nbytes = _ws_bytes[key] = _C.ctc_loss_workspace_bytes(B, T, V, Smax, code, self.algo)
            ws = R.workspace(dev, nbytes)
            sB, sT, sV = x.stride()
            _C.ctc_loss_fwd_bwd(x.data_ptr(), code, bool(input_is_logprobs), sB, sT, sV,
                                targets.data_ptr(), targets.stride(0), xl.data_ptr(), tl.data_ptr(),
                                B, T, V, Smax, self.blank_idx,
                                losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(),
                                self.algo, R.stream_handle(dev), float(grad_scale),
                                reduced.data_ptr() if reduction else 0,
                                _REDUCTIONS[reduction],
                                _C.CHAINS_F32 if self.f32_chains else _C.CHAINS_F64)
        if src_device != dev or src_dtype != losses.dtype:
            losses = losses.to(src_device, src_dtype)
            if reduction:
                reduced = reduced.to(src_device, src_dtype)
        if src_device != dev or src_dtype != grads.dtype:
            grads = grads.to(src_device, src_dtype)
        return (losses, grads) if reduction is None else (losses, grads, reduced)

    @staticmethod
    def scale_grads_(grads, scale):
        """grads[b] *= scale[b] in place on the GPU (the multiply of functions/forward_backward.py:33 upstream,
        without a second (B,T,V) tensor; rows whose factor is exactly 1 are not touched).  `grads` must be a contiguous
        CUDA tensor, `scale` a (B,) tensor -- or a single element, which then scales the whole tensor."""
        if not (grads.is_cuda and grads.is_contiguous()):
            raise ValueError("scale_grads_ needs a contiguous GPU tensor")
        if scale.device != grads.device or scale.dtype != grads.dtype or not scale.is_contiguous():
            scale = scale.detach().to(device=grads.device, dtype=grads.dtype).contiguous()
        B = grads.shape[0] if scale.numel() != 1 else 1
        if scale.numel() != B:
            raise ValueError("scale must have one entry per utterance (or a single one)")
        if grads.numel():
            with _on_device(grads.device):
                _C.ctc_scale_grads(grads.data_ptr(), R.dtype_code(grads.dtype), scale.data_ptr(), B,
                                   grads.numel() // B, R.stream_handle(grads.device))
        return grads


class LanguageModel:
    """n-gram model read from an ARPA file (plain or .gz); stands where KenLM stands upstream
    (ctc_decoder.cpp:60-71).  The device tables belong to one GPU: `on(device)` returns the copy for that device,
    loading it on first use."""

    def __init__(self, path, labels, case_sensitive):
        self.path, self.labels, self.case_sensitive = path, list(labels), bool(case_sensitive)
        self._per_device = {}
        self._first = self._load()

    def _load(self):
        lm = _C.LanguageModel(self.path, self.labels, self.case_sensitive)
        self._per_device[lm.device()] = lm
        return lm

    def on(self, device):
        lm = self._per_device.get(device.index)
        if lm is None:
            with torch.cuda.device(device):
                lm = self._load()
            if lm.device() != device.index:
                raise R.E2EError("the language model could not be loaded on %s" % device)
        return lm

    def order(self):
        return self._first.order()

    def word_index(self, word):
        return self._first.word_index(word)

    def score(self, ctx, word):
        return self._first.score(list(ctx), word)


class CTCDecoderEngine:
    """Same constructor arguments and defaults as the pybind class
    (src/decoders/ctc_decoder_py.cpp:8-24); methods keep the reference's keyword names."""

    def __init__(self, blank_idx, beam_width_=100, labels=None, lm_path="", lmwt_=1.0, wip_=0.0,
                 oov_penalty_=-1000.0, case_sensitive=False, keep_on_device=False):
        self.blank_idx = int(blank_idx)
        self.beam_width = int(beam_width_)
        self.labels = list(labels or [])
        self.lmwt = float(lmwt_)
        self.wip = float(wip_)
        self.oov_penalty = float(oov_penalty_)
        self.case_sensitive = bool(case_sensitive)
        self.keep_on_device = bool(keep_on_device)
        # index of " " among the labels, else -1 (src/decoders/ctc_decoder.cpp:55-59)
        self.space_id = self.labels.index(" ") if " " in self.labels else -1
        # (a lone surrogate is a legal one-character label, but not legal UTF-32: such alphabets take the per-id join)
        self._codes = (np.array([ord(c) for c in self.labels], dtype="<u4")
                       if self.labels and all(len(c) == 1 and not 0xD800 <= ord(c) <= 0xDFFF for c in self.labels) else None)
        self.lm = None
        if self.labels and self.beam_width > 1:
            # the beam lives in one workgroup's LDS: a width / alphabet it cannot hold is reported now, not at the first
            # decode (upstream has no such limit, ctc_decoder.cpp:353-441; see include/e2e_ctc.h)
            cap = _C.ctc_beam_max_width(len(self.labels), bool(lm_path))
            if self.beam_width > cap:
                raise ValueError("beam_width %d is not supported for an alphabet of %d labels%s: at most %d"
                                 % (self.beam_width, len(self.labels), " with a language model" if lm_path else "", cap))
        if lm_path:
            R.require_gpu()
            self.lm = LanguageModel(lm_path, self.labels, self.case_sensitive)
        else:
            self.lmwt = 0.0   # ctc_decoder.cpp:72-74

    def _strings(self, rows, lens):
        # indices2str, ctc_decoder.cpp:203-220: "" when there are no labels.  The empty prefix wins as [-1] (quirk Q6);
        # the reference then reads labels[-1] out of bounds (undefined behaviour) -- here that id spells nothing.
        if not self.labels:
            return ["" for _ in lens]
        if self._codes is not None and len(lens) and isinstance(rows, torch.Tensor):
            # one-character labels (the usual alphabet): one table lookup and one decode for the whole batch instead of a
            # Python-level join per id (4 ms of a 17 ms beam-search call at B=64, T=1500)
            ids = rows.numpy()
            if ids.size == 0 or int(ids.min()) >= 0:
                width = ids.shape[1]
                text = self._codes[ids].tobytes().decode("utf-32-le")
                return [text[b * width: b * width + n] for b, n in enumerate(lens)]
            rows = ids.tolist()
        elif isinstance(rows, torch.Tensor):
            rows = rows.tolist()
        return ["".join(self.labels[k] for k in row[:n] if k >= 0) for row, n in zip(rows, lens)]

    def _prep(self, logits_, logits_lengths_, native16=True):
        if logits_.dim() != 3:
            raise ValueError("logits must be (batch, time, alphabet)")
        dev = R.compute_device(logits_)
        x = logits_.detach()
        # (16-bit inputs: the greedy kernels compare them as they are, the beam search reads them as they are -- each is an f32
        #  number, so the search is the f32 one's; upstream converts whatever arrives once, ctc_decoder.cpp:157-160)
        if x.dtype not in ((torch.float32, torch.float64, torch.float16, torch.bfloat16) if native16 else (torch.float32, torch.float64)):
            x = x.to(torch.float32)
        x = x.to(dev)
        xl = _as_long(logits_lengths_, dev)
        if xl.numel() != x.shape[0]:
            raise ValueError("logits_lengths_ must have one entry per utterance")
        V = x.shape[2]
        if not 0 <= self.blank_idx < V:
            raise ValueError("blank_idx %d outside the alphabet of %d columns" % (self.blank_idx, V))
        if self.labels and len(self.labels) != V:
            # upstream indexes labels[id] unchecked (ctc_decoder.cpp:203-220); a mismatch would spell garbage
            raise ValueError("the decoder has %d labels but the logits have %d columns" % (len(self.labels), V))
        return x, xl, dev

    def _result(self, t):
        return t if self.keep_on_device else t.cpu()

    def decode_greedy(self, logits_, logits_lengths_):
        """argmax + blank/repeat collapse -> (targets (B,Tmax) int64 zero padded, lengths (B), sentences)."""
        x, xl, dev = self._prep(logits_, logits_lengths_)
        B, T, V = x.shape
        out = torch.empty((B, T), dtype=torch.long, device=dev)
        out_len = torch.empty(B, dtype=torch.long, device=dev)
        if B:
            with torch.cuda.device(dev):
                sB, sT, sV = x.stride()
                _C.ctc_greedy(x.data_ptr(), R.dtype_code(x.dtype), sB, sT, sV, xl.data_ptr(), B, T, V,
                              self.blank_idx, out.data_ptr(), out_len.data_ptr(), R.stream_handle(dev))
        out, out_len = self._result(out), self._result(out_len)
        sentences = self._strings(out.cpu(), out_len.tolist()) if self.labels else ["" for _ in range(B)]
        return out, out_len, sentences

    def decode(self, logits_, logits_lengths_):
        """Prefix beam search on LOG-PROBABILITIES -> (indices (B,maxlen) int64, lengths (B), sentences)."""
        x, xl, dev = self._prep(logits_, logits_lengths_)
        B, T, V = x.shape
        max_out = T + 1
        out = torch.empty((B, max_out), dtype=torch.long, device=dev)
        out_len = torch.empty(B, dtype=torch.long, device=dev)
        if B:
            with torch.cuda.device(dev):
                lm = self.lm.on(dev).handle if self.lm is not None else 0
                nbytes = _C.ctc_beam_workspace_bytes_lm(B, T, V, self.beam_width, self.lm is not None)
                ws = R.workspace(dev, nbytes)
                sB, sT, sV = x.stride()
                _C.ctc_beam(x.data_ptr(), R.dtype_code(x.dtype), sB, sT, sV, xl.data_ptr(),
                            B, T, V, self.blank_idx, self.beam_width, self.space_id, lm,
                            self.lmwt, self.wip, self.oov_penalty,
                            out.data_ptr(), max_out, out_len.data_ptr(),
                            ws.data_ptr(), ws.numel(), R.stream_handle(dev))
        lens = out_len.tolist()
        # per-utterance status rides on the lengths (include/e2e_ctc.h): no extra call, no extra synchronisation
        for b, n in enumerate(lens):
            if n < 0 or n > max_out:
                raise R.E2EError("beam search: utterance %d %s" % (
                    b, "ran out of prefix-tree nodes" if n < 0 else "needs %d output ids, %d provided" % (n, max_out)))
        width = max(lens) if lens else 0
        ids = out[:, :width].contiguous()    # packed to the longest result (ctc_decoder.cpp:192-200)
        ids, out_len = self._result(ids), self._result(out_len)
        return ids, out_len, self._strings(ids.cpu(), lens)

    def print_scores_for_sentence(self, words):
        """src/decoders/ctc_decoder.cpp:141-151: word, decoder index, vocabulary index, log10 score."""
        if self.lm is None:
            return
        ctx = [self.lm.word_index("<s>")]
        order = self.lm.order()
        for w in words:
            key = w if self.case_sensitive else w.lower()
            idx = self.lm.word_index(key)
            print(w, idx, self.lm.word_index(w), self.lm.score(ctx, idx))
            ctx = ([idx] + ctx)[: max(order - 1, 0)]
