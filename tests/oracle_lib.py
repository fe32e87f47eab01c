"""ctypes front-end for oracle/liboracle_ctc.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  The product package never does.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "liboracle_ctc.so")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")

_lib = None


def build_oracle(force=False):
    src = os.path.join(ORACLE_DIR, "ctc_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle_ctc.so"], stdout=subprocess.DEVNULL)
    return LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build_oracle()
        L = C.CDLL(LIB_PATH)
        i64p = C.POINTER(C.c_int64)
        dp = C.POINTER(C.c_double)
        L.oracle_log_sum_exp.restype = C.c_double
        L.oracle_log_sum_exp.argtypes = [C.c_double, C.c_double]
        L.oracle_ctc_loss.restype = C.c_int
        L.oracle_ctc_loss.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, i64p, C.c_int64, i64p, i64p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, dp, dp, C.c_int]
        L.oracle_ctc_greedy.restype = C.c_int
        L.oracle_ctc_greedy.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, i64p,
                                        C.c_int, C.c_int, C.c_int, C.c_int, i64p, i64p, C.c_int]
        L.oracle_lm_load_arpa.restype = C.c_void_p
        L.oracle_lm_load_arpa.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.oracle_lm_free.argtypes = [C.c_void_p]
        L.oracle_lm_order.restype = C.c_int
        L.oracle_lm_order.argtypes = [C.c_void_p]
        L.oracle_lm_word_index.restype = C.c_uint32
        L.oracle_lm_word_index.argtypes = [C.c_void_p, C.c_char_p]
        L.oracle_lm_base_score.restype = C.c_double
        L.oracle_lm_base_score.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.c_uint32,
                                           C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        L.oracle_ctc_beam.restype = C.c_int
        L.oracle_ctc_beam.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, i64p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.POINTER(C.c_char_p), C.c_int, C.c_void_p, C.c_int,
                                      C.c_double, C.c_double, C.c_double,
                                      i64p, C.c_int64, i64p, C.c_int]
        L.oracle_ctc_align.restype = C.c_int
        L.oracle_ctc_align.argtypes = [dp, C.c_int64, C.c_int64, C.c_int64, i64p, C.c_int64, i64p, i64p,
                                       C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, i64p, C.c_int]
        _lib = L
    return _lib


def _dptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def ctc_loss(lp, targets, x_len, t_len, blank=0, n_threads=8):
    """lp: (B,T,V) array-like of log-probs (any strides) -> (losses f64 [B], grads f64 [B,T,V])."""
    lp = np.asarray(lp, dtype=np.float64)
    B, T, V = lp.shape
    targets = np.ascontiguousarray(np.asarray(targets, dtype=np.int64).reshape(B, -1))
    if targets.shape[1] == 0:
        targets = np.zeros((B, 1), dtype=np.int64)
    x_len = np.ascontiguousarray(np.asarray(x_len, dtype=np.int64))
    t_len = np.ascontiguousarray(np.asarray(t_len, dtype=np.int64))
    losses = np.zeros(B, dtype=np.float64)
    grads = np.zeros((B, T, V), dtype=np.float64)
    es = lp.itemsize
    rc = lib().oracle_ctc_loss(_dptr(lp), lp.strides[0] // es, lp.strides[1] // es, lp.strides[2] // es,
                               _iptr(targets), targets.shape[1], _iptr(x_len), _iptr(t_len),
                               B, T, V, blank, _dptr(losses), _dptr(grads), n_threads)
    if rc != 0:
        raise ValueError("oracle_ctc_loss rc=%d" % rc)
    return losses, grads


def ctc_greedy(x, x_len=None, blank=0, n_threads=8):
    x = np.asarray(x, dtype=np.float64)
    B, T, V = x.shape
    if x_len is None:
        x_len = np.full(B, T)
    x_len = np.ascontiguousarray(np.asarray(x_len, dtype=np.int64))
    out = np.zeros((B, T), dtype=np.int64)
    out_len = np.zeros(B, dtype=np.int64)
    es = x.itemsize
    rc = lib().oracle_ctc_greedy(_dptr(x), x.strides[0] // es, x.strides[1] // es, x.strides[2] // es,
                                 _iptr(x_len), B, T, V, blank, _iptr(out), _iptr(out_len), n_threads)
    if rc != 0:
        raise ValueError("oracle_ctc_greedy rc=%d" % rc)
    return out, out_len


def ctc_align(lp, targets, x_len, t_len, blank=0, is_ctc=True, pad=-100, n_threads=8):
    """Viterbi forced alignment -> (B,T) int64, rows beyond x_len keep `pad` (get_alignment_3d upstream)."""
    lp = np.asarray(lp, dtype=np.float64)
    B, T, V = lp.shape
    targets = np.ascontiguousarray(np.asarray(targets, dtype=np.int64).reshape(B, -1))
    if targets.shape[1] == 0:
        targets = np.zeros((B, 1), dtype=np.int64)
    x_len = np.ascontiguousarray(np.asarray(x_len, dtype=np.int64))
    t_len = np.ascontiguousarray(np.asarray(t_len, dtype=np.int64))
    out = np.full((B, T), pad, dtype=np.int64)
    es = lp.itemsize
    rc = lib().oracle_ctc_align(_dptr(lp), lp.strides[0] // es, lp.strides[1] // es, lp.strides[2] // es,
                                _iptr(targets), targets.shape[1], _iptr(x_len), _iptr(t_len), B, T, V, blank,
                                int(is_ctc), _iptr(out), n_threads)
    if rc != 0:
        raise ValueError("oracle_ctc_align rc=%d" % rc)
    return out


class OracleLM:
    def __init__(self, path):
        err = C.create_string_buffer(256)
        self.h = lib().oracle_lm_load_arpa(path.encode(), err, 256)
        if not self.h:
            raise ValueError(err.value.decode())

    def order(self):
        return lib().oracle_lm_order(self.h)

    def word_index(self, w):
        return lib().oracle_lm_word_index(self.h, w.encode())

    def base_score(self, ctx, word):
        """ctx: most-recent-first word ids -> (log10 p, new ctx)."""
        n = len(ctx)
        a = (C.c_uint32 * max(n, 1))(*ctx)
        o = (C.c_uint32 * 8)()
        on = C.c_int(0)
        s = lib().oracle_lm_base_score(self.h, a, n, word, o, C.byref(on))
        return s, list(o[: on.value])

    def __del__(self):
        try:
            if self.h:
                lib().oracle_lm_free(self.h)
                self.h = None
        except Exception:
            pass


def ctc_beam(lp, x_len=None, blank=0, beam_width=100, labels=None, lm=None, case_sensitive=True,
             lmwt=1.0, wip=0.0, oov_penalty=-1000.0, n_threads=8):
    """Prefix beam search on log-probs. Returns (out [B,maxlen] i64, lengths [B], sentences)."""
    lp = np.asarray(lp, dtype=np.float64)
    B, T, V = lp.shape
    if x_len is None:
        x_len = np.full(B, T)
    x_len = np.ascontiguousarray(np.asarray(x_len, dtype=np.int64))
    labels = list(labels or [])
    space_id = labels.index(" ") if " " in labels else -1
    lab_arr = None
    if labels:
        lab_arr = (C.c_char_p * len(labels))(*[s.encode() for s in labels])
    max_out = T + 1
    out = np.zeros((B, max_out), dtype=np.int64)
    out_len = np.zeros(B, dtype=np.int64)
    es = lp.itemsize
    rc = lib().oracle_ctc_beam(_dptr(lp), lp.strides[0] // es, lp.strides[1] // es, lp.strides[2] // es,
                               _iptr(x_len), B, T, V, blank, beam_width, lab_arr, space_id,
                               lm.h if lm is not None else None, int(case_sensitive),
                               lmwt, wip, oov_penalty, _iptr(out), max_out, _iptr(out_len), n_threads)
    if rc != 0:
        raise ValueError("oracle_ctc_beam rc=%d" % rc)
    width = int(out_len.max()) if B else 0
    sentences = []
    for b in range(B):
        ids = out[b, : out_len[b]]
        sentences.append("".join(labels[i] for i in ids) if labels else "")
    return out[:, :width].copy(), out_len, sentences


def load_reference_engine():
    """The reference's own compiled loss engine (oracle/_ref/cpp_ctc_loss.so), or None."""
    so = os.path.join(REF_DIR, "cpp_ctc_loss.so")
    if not os.path.exists(so):
        return None
    import torch  # noqa: F401  (libtorch must be loaded first)
    # loaded by file path and NOT entered into sys.modules: the repository's own top-level `cpp_ctc_loss` module (the
    # product under the reference's name) must keep that name
    import importlib.machinery
    import importlib.util
    try:
        loader = importlib.machinery.ExtensionFileLoader("cpp_ctc_loss", so)
        spec = importlib.util.spec_from_file_location("cpp_ctc_loss", so, loader=loader)
        mod = importlib.util.module_from_spec(spec)
        loader.exec_module(mod)
        return mod
    except Exception:
        return None
