"""Python mirrors of the reference's two pybind engines, backed by the HIP C ABI.

CTCLossEngine  <-> cpp_ctc_loss.CTCLossEngine      (src/losses/ctc_loss_py.cpp:8-16)
CTCDecoderEngine <-> cpp_ctc_decoder.CTCDecoder    (src/decoders/ctc_decoder_py.cpp:8-38)

Unlike the reference (which copies GPU tensors to the host and computes in C++ threads,
src/losses/forward_backward.cpp:12-19), tensors stay on the GPU; CPU tensors are moved to
the current GPU and the results moved back to the source device, mirroring the reference's
"results on the source device and dtype" contract (forward_backward.cpp:55-56).
"""
import ctypes as C
import os

import torch

from . import _lib


def _as_long(t, device):
    if not torch.is_tensor(t):
        t = torch.as_tensor(t)
    return t.to(device=device, dtype=torch.long).contiguous()


class CTCLossEngine:
    """blank_idx -> .compute(logits, targets, logits_lengths, targets_lengths) -> (losses[B], grads[B,T,V])."""

    def __init__(self, blank_idx, algo=_lib.ALGO_AUTO):
        self.blank_idx = int(blank_idx)
        self.algo = algo
        _lib.load()

    def compute(self, logits, targets, logits_lengths, targets_lengths, input_is_logprobs=True):
        """`logits` is batch-major (B,T,V) (any strides).  With input_is_logprobs=True this is the
        reference engine: log-probabilities in, grads = exp(lp) - posterior.  With False the
        log-softmax is fused in and grads are d loss / d logits."""
        L = _lib.load()
        if logits.dim() != 3:
            raise ValueError("logits must be (batch, time, alphabet)")
        src_device, src_dtype = logits.device, logits.dtype
        dev = _lib.compute_device(logits)
        x = logits.detach()
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        x = x.to(dev)
        B, T, V = x.shape
        targets = _as_long(targets, dev)
        if targets.dim() != 2 or targets.shape[0] != B:
            raise ValueError("targets must be (batch, max_target_length)")
        xl = _as_long(logits_lengths, dev)
        tl = _as_long(targets_lengths, dev)
        if xl.numel() != B or tl.numel() != B:
            raise ValueError("lengths must have one entry per utterance")
        Smax = targets.shape[1]
        losses = torch.empty(B, dtype=x.dtype, device=dev)
        grads = torch.empty((B, T, V), dtype=x.dtype, device=dev)
        if B == 0:
            return losses.to(src_device, src_dtype), grads.to(src_device, src_dtype)
        code = _lib.dtype_code(x.dtype)
        with torch.cuda.device(dev):
            nbytes = L.e2e_ctc_loss_workspace_bytes(B, T, V, Smax, code, self.algo)
            ws = _lib.workspace(dev, nbytes)
            sB, sT, sV = x.stride()
            _lib.check(L.e2e_ctc_loss_fwd_bwd(
                x.data_ptr(), code, 1 if input_is_logprobs else 0, sB, sT, sV,
                targets.data_ptr(), targets.stride(0), xl.data_ptr(), tl.data_ptr(),
                B, T, V, Smax, self.blank_idx,
                losses.data_ptr(), grads.data_ptr(), ws.data_ptr(), ws.numel(),
                self.algo, _lib.stream_ptr(dev)))
        if src_device != dev or src_dtype != x.dtype:
            losses = losses.to(src_device, src_dtype)
            grads = grads.to(src_device, src_dtype)
        return losses, grads


class LanguageModel:
    """Device-resident n-gram table built from an ARPA file (stands where KenLM stands)."""

    def __init__(self, path, labels, case_sensitive):
        L = _lib.load()
        self._h = C.c_void_p()
        arr = (C.c_char_p * len(labels))(*[s.encode("utf-8") for s in labels])
        _lib.check(L.e2e_lm_load_arpa(os.fsencode(path), arr, len(labels), 1 if case_sensitive else 0,
                                      C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def order(self):
        return _lib.load().e2e_lm_order(self._h)

    def word_index(self, word):
        return _lib.load().e2e_lm_word_index(self._h, word.encode("utf-8"))

    def score(self, ctx, word):
        a = (C.c_uint32 * max(len(ctx), 1))(*ctx)
        return _lib.load().e2e_lm_score(self._h, a, len(ctx), word)

    def __del__(self):
        try:
            if self._h:
                _lib.load().e2e_lm_free(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass


class CTCDecoderEngine:
    """Same constructor arguments and defaults as the pybind class
    (src/decoders/ctc_decoder_py.cpp:8-24); methods keep the reference's keyword names."""

    def __init__(self, blank_idx, beam_width_=100, labels=None, lm_path="", lmwt_=1.0, wip_=0.0,
                 oov_penalty_=-1000.0, case_sensitive=False):
        _lib.load()
        self.blank_idx = int(blank_idx)
        self.beam_width = int(beam_width_)
        self.labels = list(labels or [])
        self.lmwt = float(lmwt_)
        self.wip = float(wip_)
        self.oov_penalty = float(oov_penalty_)
        self.case_sensitive = bool(case_sensitive)
        # index of " " among the labels, else -1 (src/decoders/ctc_decoder.cpp:55-59)
        self.space_id = self.labels.index(" ") if " " in self.labels else -1
        self.lm = None
        if lm_path:
            _lib.require_gpu()
            self.lm = LanguageModel(lm_path, self.labels, self.case_sensitive)
        else:
            self.lmwt = 0.0   # ctc_decoder.cpp:72-74

    def _strings(self, ids, lens):
        # indices2str, ctc_decoder.cpp:203-220: "" when there are no labels
        if not self.labels:
            return ["" for _ in lens]
        out = []
        for row, n in zip(ids, lens):
            out.append("".join(self.labels[k] for k in row[:n]))
        return out

    def _prep(self, logits_, logits_lengths_):
        if logits_.dim() != 3:
            raise ValueError("logits must be (batch, time, alphabet)")
        dev = _lib.compute_device(logits_)
        x = logits_.detach()
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float32)
        x = x.to(dev)
        xl = _as_long(logits_lengths_, dev)
        if xl.numel() != x.shape[0]:
            raise ValueError("logits_lengths_ must have one entry per utterance")
        return x, xl, dev

    def decode_greedy(self, logits_, logits_lengths_):
        """argmax + blank/repeat collapse -> (targets (B,Tmax) int64 zero padded, lengths (B), sentences)."""
        L = _lib.load()
        src_device = logits_.device
        x, xl, dev = self._prep(logits_, logits_lengths_)
        B, T, V = x.shape
        out = torch.empty((B, T), dtype=torch.long, device=dev)
        out_len = torch.empty(B, dtype=torch.long, device=dev)
        if B:
            with torch.cuda.device(dev):
                sB, sT, sV = x.stride()
                _lib.check(L.e2e_ctc_greedy(x.data_ptr(), _lib.dtype_code(x.dtype), sB, sT, sV, xl.data_ptr(),
                                            B, T, V, self.blank_idx, out.data_ptr(), out_len.data_ptr(),
                                            _lib.stream_ptr(dev)))
        sentences = self._strings(out.tolist(), out_len.tolist()) if self.labels else ["" for _ in range(B)]
        return out.to(src_device), out_len.to(src_device), sentences

    def decode(self, logits_, logits_lengths_):
        """Prefix beam search on LOG-PROBABILITIES -> (indices (B,maxlen) int64, lengths (B), sentences)."""
        L = _lib.load()
        src_device = logits_.device
        x, xl, dev = self._prep(logits_, logits_lengths_)
        B, T, V = x.shape
        max_out = T + 1
        out = torch.empty((B, max_out), dtype=torch.long, device=dev)
        out_len = torch.empty(B, dtype=torch.long, device=dev)
        if B:
            with torch.cuda.device(dev):
                nbytes = L.e2e_ctc_beam_workspace_bytes(B, T, V, self.beam_width)
                ws = _lib.workspace(dev, nbytes)
                sB, sT, sV = x.stride()
                _lib.check(L.e2e_ctc_beam(x.data_ptr(), _lib.dtype_code(x.dtype), sB, sT, sV, xl.data_ptr(),
                                          B, T, V, self.blank_idx, self.beam_width, self.space_id,
                                          self.lm.handle if self.lm is not None else None,
                                          self.lmwt, self.wip, self.oov_penalty,
                                          out.data_ptr(), max_out, out_len.data_ptr(),
                                          ws.data_ptr(), ws.numel(), _lib.stream_ptr(dev)))
                # pool exhaustion / truncated output are reported per utterance (synchronises, like .tolist() below)
                _lib.check(L.e2e_ctc_beam_status(ws.data_ptr(), B, T, V, self.beam_width))
        lens = out_len.tolist()
        width = max(lens) if lens else 0
        ids = out[:, :width].contiguous()    # packed to the longest result (ctc_decoder.cpp:192-200)
        rows = ids.tolist()
        if self.labels:
            sentences = []
            for row, n in zip(rows, lens):
                # the empty prefix wins as [-1]; the reference then reads labels[-1] out of bounds
                # (undefined behaviour, quirk Q6) -- here that id spells nothing
                sentences.append("".join(self.labels[k] for k in row[:n] if k >= 0))
        else:
            sentences = ["" for _ in range(B)]
        return ids.to(src_device), out_len.to(src_device), sentences

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
