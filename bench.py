#!/usr/bin/env python3
"""Headline benchmark: CTC fwd+bwd frames/s at B=256 T=1000 V=29 S<=200 fp32 per GPU
(BASELINE.json configs[1], the LibriSpeech-char shape), synthetic logits.

A step = one pass of the hot path over one batch already resident in HBM: the fused
log-softmax + alpha/beta lattice + gradient launch of libe2e_ctc.so (C ABI
e2e_ctc_loss_fwd_bwd, raw logits in, per-utterance losses and d loss/d logits out), plus the
batch-mean reduction; with N>1 ranks each GPU gets its own 256 utterances (weak scaling, no
data-path collective) and the scalar loss is all-reduced over RCCL.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Started plainly with --gpus N > 1 (no RANK in the environment) it launches the N ranks itself,
as children, before anything touches a GPU, and relays rank 0's line.

Rank 0 prints ONE JSON line:
  value / ms_per_step       the C-ABI step above (the drop-in boundary), whole job
  module_ms_per_step        the metric as SURVEY.md 8(d) words it, through the Python surface:
                            loss = CTCLoss(reduce=True, size_average=True)(logits, ...); loss.backward()
  roofline                  the call's kernels against HBM peak with ALGORITHMIC bytes (2*V*4 B per frame: logits read
                            once, gradient written once); peak = 8 TB/s spec, peak_measured = an on-box device copy
  wide_alphabet             one GPU's share of BASELINE configs[4] (B=512 T=256 V=8000), at every N
  cpu_baseline (N=1)        the reference's own C++ engine (oracle/_ref, kind "reference") or the C restatement
                            (kind "port") on this box's host cores; decode.*.cpu_baseline likewise (port)
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

VALU_PEAK_LANE_OPS = 256 * 4 * 32 * 2.4e9   # f32 lane-operations per second (157.3 TFLOP/s / 2 flops per fma)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 measured copy)

WORKLOAD = dict(name="ctc_fwd_bwd_B256_T1000_V29_S200_f32", B=256, T=1000, V=29, S=200)
WIDE = dict(name="ctc_fwd_bwd_B512_T256_V8000_S64_f32", B=512, T=256, V=8000, S=64)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-decode", action="store_true", help="skip the secondary decode-throughput numbers")
    ap.add_argument("--no-wide", action="store_true", help="skip the V=8000 share")
    return ap.parse_args()


def launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run, one process
    per GPU, rendezvous on 127.0.0.1) BEFORE this process touches a GPU, and relay their output."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def make_batch(seed, B, T, V, S, device):
    """logits randn, targets in [1,V), target lengths in [S/2,S], full-length inputs."""
    import torch
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, T, V, generator=g)
    targets = torch.randint(1, V, (B, S), generator=g)
    t_len = torch.randint(S // 2, S + 1, (B,), generator=g)
    x_len = torch.full((B,), T, dtype=torch.long)
    host = (logits, targets, x_len, t_len)
    return host, tuple(t.to(device) for t in host)


class HotPath:
    """Pre-bound C-ABI call (no per-step allocation, graph-capturable)."""

    def __init__(self, dev_batch, blank=0, chains=0):
        import torch
        from end2end_amd import _lib
        self.torch = torch
        self.lib = _lib
        self.L = _lib.load()
        self.x, self.targets, self.x_len, self.t_len = dev_batch
        self.dev = self.x.device
        B, T, V = self.x.shape
        self.B, self.T, self.V, self.S = B, T, V, self.targets.shape[1]
        self.code = _lib.dtype_code(self.x.dtype)          # (16-bit logits: gradient in the same dtype, losses f32)
        self.losses = torch.empty(B, dtype=torch.float32, device=self.dev)
        self.grads = torch.empty((B, T, V), dtype=self.x.dtype, device=self.dev)
        n = self.L.e2e_ctc_loss_workspace_bytes(B, T, V, self.S, self.code, _lib.ALGO_AUTO)
        self.ws = torch.empty(n, dtype=torch.uint8, device=self.dev)
        self.blank = blank
        self.chains = chains                               # e2e_ctc_loss_opts.chains
        self.bucket = 8                                    # steps per loss all-reduce (N>1): 32 B instead of 8 x 4 B
        self.means = torch.zeros((2, self.bucket), dtype=torch.float32, device=self.dev)   # double-buffered buckets
        self.k = 0

    def call(self, mean_out=None):
        """One e2e_ctc_loss_fwd_bwd_opt launch; mean_out: a 1-element device tensor that receives the batch-mean loss
        (written by the tail of the call's last kernel) and makes the gradient that of the mean (grad_scale = 1/B)."""
        import ctypes
        sB, sT, sV = self.x.stride()
        o = self.lib.LossOpts(1.0 / self.B if mean_out is not None else 1.0,
                              mean_out.data_ptr() if mean_out is not None else None,
                              self.lib.REDUCE_MEAN if mean_out is not None else self.lib.REDUCE_NONE, self.chains)
        self.lib.check(self.L.e2e_ctc_loss_fwd_bwd_opt(
            self.x.data_ptr(), self.code, 0, sB, sT, sV,
            self.targets.data_ptr(), self.targets.stride(0), self.x_len.data_ptr(), self.t_len.data_ptr(),
            self.B, self.T, self.V, self.S, self.blank,
            self.losses.data_ptr(), self.grads.data_ptr(), self.ws.data_ptr(), self.ws.numel(),
            self.lib.ALGO_AUTO, self.lib.stream_ptr(self.dev), ctypes.byref(o)))

    def step(self):
        m = self.means[(self.k // self.bucket) & 1, self.k % self.bucket: self.k % self.bucket + 1]
        self.call(m)
        self.k += 1
        return m


def time_events(torch, fn, reps):
    """Mean duration of fn() in ms, one HIP event pair per call on the launch stream."""
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        fn()
        b.record()
    torch.cuda.synchronize()
    return sum(a.elapsed_time(b) for a, b in ev) / len(ev)


def measured_copy_gbs(torch, dev):
    """On-box HBM rate of a plain device copy (read + write bytes per second), the achievable ceiling the guide quotes
    beside the 8 TB/s spec."""
    from end2end_amd import _lib
    L = _lib.load()
    n = 1 << 28                                             # 1 GiB of f32 each way
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)

    def copy():
        _lib.check(L.e2e_debug_stream_copy(dst.data_ptr(), src.data_ptr(), n * 4, _lib.stream_ptr(dev)))
    for _ in range(3):
        copy()
    ms = min(time_events(torch, copy, 10), time_events(torch, lambda: dst.copy_(src), 10))
    del src, dst
    return 2.0 * n * 4 / (ms * 1e-3) / 1e9


def cpu_baseline_loss(host_batch, frames):
    """The reference path on host cores: log_softmax + engine.compute + backward, as
    pytorch_end2end/modules/ctc_loss.py:25-57 and functions/forward_backward.py:18-35 do it."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    logits, targets, x_len, t_len = host_batch
    ref = O.load_reference_engine()
    kind = "reference" if ref is not None else "port"

    if ref is not None:
        eng = ref.CTCLossEngine(0)

        def engine(lp, tg, xl, tl):
            return eng.compute(lp, tg, xl, tl)
    else:
        def engine(lp, tg, xl, tl):
            l, g = O.ctc_loss(lp.double().numpy(), tg.numpy(), xl.numpy(), tl.numpy(), 0, n_threads=0)
            return torch.from_numpy(l).float(), torch.from_numpy(g).float()

    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, lp, tg, xl, tl):
            loss, grads = engine(lp.detach(), tg, xl, tl)
            ctx.grads = grads
            return loss

        @staticmethod
        def backward(ctx, go):
            return ctx.grads * go.view(-1, 1, 1), None, None, None

    def one(n):
        x = logits[:n].clone().requires_grad_()
        t0 = time.perf_counter()
        loss = Fn.apply(torch.log_softmax(x, 2), targets[:n], x_len[:n], t_len[:n]).mean()
        loss.backward()
        return time.perf_counter() - t0

    B = logits.shape[0]
    one(min(B, 16))                       # warm-up (thread pool, allocator)
    n = B
    times = [one(n) for _ in range(3)]
    best = min(times)
    per_frame = frames * n / B
    return {"value": per_frame / best, "unit": "frames/s", "cores": os.cpu_count(), "kind": kind,
            "sample": "full %d-utterance batch of the same workload, best of 3 after a 16-utterance warm-up, "
                      "one thread per utterance as the reference does (%.2f s per pass)" % (n, best)}


def synthetic_arpa(path, labels, n_words=10000, seed=0):
    """A seeded synthetic 3-gram ARPA (the LibriSpeech LM of the reference's tests is not available offline)."""
    import random
    rng = random.Random(seed)
    letters = [c for c in labels if len(c) == 1 and c.isalpha()]
    words = set()
    while len(words) < n_words:
        words.add("".join(rng.choice(letters) for _ in range(rng.randint(1, 7))))
    words = sorted(words)
    uni = [(-rng.uniform(2.0, 5.0), w, -rng.uniform(0.1, 0.6)) for w in words]
    bi = {(rng.choice(words), rng.choice(words)) for _ in range(3 * n_words)}
    bi |= {("<s>", rng.choice(words)) for _ in range(n_words // 10)}
    bi = sorted(bi)
    tri = sorted({(a, b, rng.choice(words)) for a, b in rng.sample(bi, n_words)})
    with open(path, "w") as f:
        f.write("\\data\\\nngram 1=%d\nngram 2=%d\nngram 3=%d\n\n\\1-grams:\n" % (len(uni) + 3, len(bi), len(tri)))
        f.write("-2.0\t<unk>\n-99\t<s>\t-0.3\n-1.5\t</s>\n")
        for p_, w, b_ in uni:
            f.write("%.4f\t%s\t%.4f\n" % (p_, w, b_))
        f.write("\n\\2-grams:\n")
        for a, b_ in bi:
            f.write("%.4f\t%s %s\t%.4f\n" % (-rng.uniform(0.5, 3.0), a, b_, -rng.uniform(0.05, 0.4)))
        f.write("\n\\3-grams:\n")
        for a, b_, c in tri:
            f.write("%.4f\t%s %s %s\n" % (-rng.uniform(0.2, 2.0), a, b_, c))
        f.write("\n\\end\\\n")


def decode_numbers(dev, with_cpu):
    """Secondary metric of BASELINE.json: decode utterances/s.  Greedy at B=1024 T=1500 V=29 (configs[2]); beam=100 at
    B=64 T=1500 V=29 without and with a 3-gram LM (configs[3], synthetic ARPA).  Inputs resident in HBM.  The CPU
    baselines are the C restatement of the reference's decoder (oracle/, kind "port": the reference decoder itself does
    not compile without KenLM), one thread per utterance as upstream (ctc_decoder.cpp:172-189,465-486)."""
    import tempfile
    import torch
    from end2end_amd import CTCDecoder
    labels = ["_"] + [chr(97 + i) for i in range(26)] + [" ", "'"]
    out = {}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    def cpu_timed(fn, n_utt, what):
        fn()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - t0)
        return {"value": n_utt / min(ts), "unit": "utterances/s", "cores": os.cpu_count(), "kind": "port",
                "sample": "%s, one thread per utterance, best of 3 (%.3f s per pass)" % (what, min(ts))}

    if with_cpu:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O

    g = torch.Generator().manual_seed(2)
    xh = torch.randn(1024, 1500, 29, generator=g) * 3
    x = xh.to(dev)
    xl = torch.full((1024,), 1500, dtype=torch.long, device=dev)
    eng = CTCDecoder(beam_width=1, blank_idx=0, keep_on_device=True)._decoder    # no labels: ids only; results stay in HBM
    dt = timed(lambda: eng.decode_greedy(x, xl), 10)
    kms = time_events(torch, lambda: eng.decode_greedy(x, xl), 20)          # HIP events on the launch stream
    bytes_alg = 1024 * 1500 * (29 * 4 + 8)
    out["greedy"] = {"workload": "B=1024 T=1500 V=29", "utterances_per_s": 1024 / dt, "frames_per_s": 1024 * 1500 / dt,
                     "ms": dt * 1e3, "kernel_ms": kms,
                     "roofline": {"bound": "hbm", "achieved": bytes_alg / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS,
                                  "unit": "GB/s", "frac": bytes_alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                  "algorithmic_bytes_per_launch": bytes_alg,
                                  "note": "HIP events around each engine call on its stream (mean of 20)"},
                     "note": "ms = wall time of the Python engine call, results left on the device"}
    if with_cpu:
        xd = xh.double().numpy()
        out["greedy"]["cpu_baseline"] = cpu_timed(lambda: O.ctc_greedy(xd, None, 0, n_threads=0), 1024,
                                                  "the full 1024-utterance batch (f64 copy of the logits not timed)")
        del xd
    xb = torch.log_softmax(x[:64], -1)
    xlb = xl[:64]
    phases = recorded_beam_phases()

    def beam_leg(eng, workload, lm_key):
        """One beam-search leg: wall time of the engine call over 5 repetitions (what a caller sees: kernel + the copy of the
        ids to the host + the sentences), HIP events around each call on its stream (the device side alone), the HBM
        fraction from the ALGORITHMIC bytes (the V log-probabilities of every frame are read once, 4 B each;
        the decoded ids leave) and the serial bound: the search is one workgroup per utterance and serial in t, so
        cycles per frame -- not bytes -- is what the number is made of."""
        dt = timed(lambda: eng.decode(xb, xlb), 5)
        kms = time_events(torch, lambda: eng.decode(xb, xlb), 5)
        bytes_alg = 64 * 1500 * 29 * 4 + 64 * 1501 * 8
        leg = {"workload": workload, "utterances_per_s": 64 / dt, "ms": dt * 1e3, "kernel_ms": kms, "repetitions": 5,
               "roofline": {"bound": "hbm", "achieved": bytes_alg / (kms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": bytes_alg / (kms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": bytes_alg,
                            "note": "HIP events around each engine call on its stream (mean of 5); the bound that matters "
                                    "is serial_bound below: 64 workgroups, T dependent frames each"},
               "serial_bound": {"frames": 1500, "us_per_frame": kms * 1e3 / 1500,
                                "workgroups": 64, "compute_units": 256}}
        if phases and lm_key in phases:
            leg["serial_bound"]["cycles_per_step"] = phases[lm_key].get("cycles_per_step")
            leg["serial_bound"]["phase_cycles"] = phases[lm_key].get("phase_cycles")
            leg["serial_bound"]["source"] = "profiles/r06_beam_phases.json (tools/diag/beam_phase_profile.py, s_memtime stamps of workgroup 0)"
        return leg

    eng = CTCDecoder(beam_width=100, blank_idx=0, after_logsoftmax=True, labels=labels, wip=1.0)._decoder
    out["beam100"] = beam_leg(eng, "B=64 T=1500 V=29 beam=100, no LM", "no_lm")
    xbh = xb.double().cpu().numpy() if with_cpu else None
    if with_cpu:
        out["beam100"]["cpu_baseline"] = cpu_timed(
            lambda: O.ctc_beam(xbh, None, 0, 100, labels, None, wip=1.0, n_threads=0), 64, "the full 64-utterance batch")
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "synthetic_3gram.arpa")
        synthetic_arpa(path, labels)
        eng = CTCDecoder(beam_width=100, blank_idx=0, after_logsoftmax=True, labels=labels, lm_path=path, lmwt=1.0,
                         wip=1.0, oov_penalty=-10.0)._decoder
        out["beam100_lm"] = beam_leg(eng, "B=64 T=1500 V=29 beam=100 + synthetic 3-gram ARPA (10k words)", "lm")
        if with_cpu:
            olm = O.OracleLM(path)
            out["beam100_lm"]["cpu_baseline"] = cpu_timed(
                lambda: O.ctc_beam(xbh, None, 0, 100, labels, olm, lmwt=1.0, wip=1.0, oov_penalty=-10.0, n_threads=0),
                64, "the full 64-utterance batch, same ARPA")
    # a word-piece-sized alphabet: the general beam kernel (candidate keys in HBM); the reference's default width
    gw = torch.Generator().manual_seed(3)
    xw = torch.log_softmax(torch.randn(16, 256, 8000, generator=gw) * 3, -1).to(dev)
    xlw = torch.full((16,), 256, dtype=torch.long, device=dev)
    eng = CTCDecoder(beam_width=100, blank_idx=0, after_logsoftmax=True)._decoder
    dt = timed(lambda: eng.decode(xw, xlw), 1)
    out["beam100_wide_alphabet"] = {"workload": "B=16 T=256 V=8000 beam=100, no LM (general kernel)",
                                    "utterances_per_s": 16 / dt, "ms": dt * 1e3}
    if with_cpu:
        # a bounded sample: 4 of the 16 utterances, first 64 frames (the restatement scores all V children of every prefix
        # per frame as upstream does: ~0.2 s per utterance and 64 frames on one core)
        xs = xw[:4, :64].double().cpu().numpy()
        cb = cpu_timed(lambda: O.ctc_beam(xs, None, 0, 100, None, None, n_threads=0), 4,
                       "4 of the 16 utterances, frames 0..63 of 256")
        cb["value"] = cb["value"] * 64.0 / 256.0            # utterances/s at the full 256 frames (cost is linear in T)
        cb["sample"] += "; value scaled by 64/256 to full-length utterances"
        out["beam100_wide_alphabet"]["cpu_baseline"] = cb
    return out


def fallback_regime_numbers(dev):
    """The headline shape with emissions that CONTRADICT the targets sharply (logits of scale 3 against random targets): the
    regime in which utterances leave the fast path (f32 segment rows run out of range) and are redone in f64 -- a tracked
    number for the step-time cliff that unit-variance logits never show."""
    import ctypes
    import torch
    from end2end_amd import _lib
    L = _lib.load()
    w = WORKLOAD
    g = torch.Generator().manual_seed(77)
    x = (torch.randn(w["B"], w["T"], w["V"], generator=g) * 3.0).to(dev)
    tg = torch.randint(1, w["V"], (w["B"], w["S"]), generator=g).to(dev)
    tl = torch.randint(w["S"] // 2, w["S"] + 1, (w["B"],), generator=g).to(dev)
    xl = torch.full((w["B"],), w["T"], dtype=torch.long, device=dev)
    hp = HotPath((x, tg, xl, tl))
    for _ in range(2):
        hp.call()
    ms = time_events(torch, hp.call, 5)
    fl = (ctypes.c_int * w["B"])()
    lz = (ctypes.c_double * (2 * w["B"]))()
    L.e2e_debug_fast_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
    L.e2e_debug_fast_state(hp.ws.data_ptr(), w["B"], w["T"], w["V"], w["S"], fl, lz)
    n = sum(1 for v in fl if v)
    return {"workload": "B=256 T=1000 V=29 S<=200, logits x3 against unrelated random targets", "ms": ms,
            "flagged_utterances": n, "frames_per_s": w["B"] * w["T"] / (ms * 1e-3),
            "note": "flagged utterances are redone by the f64 segment redo / the exact kernel inside the same call"}


def aligned_batch(seed, B, T, V, S, boost, blank=0):
    """Emissions CONSISTENT with the targets (what a trained acoustic model produces; the generator of
    tools/diag/peaky_flag_rate.py): unit-variance noise plus `boost` on one random monotone alignment of each utterance's own
    targets -- every label gets one frame, blanks fill the rest (a label that would sit directly behind its equal
    neighbour without a blank is dropped from the path; the targets keep it)."""
    import numpy as np
    import torch
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((B, T, V)).astype(np.float32)
    tg = rng.integers(1, V, size=(B, S))
    tl = rng.integers(max(S // 2, 1), S + 1, size=B)
    for b in range(B):
        n = int(tl[b])
        slots = np.sort(rng.choice(T, size=n, replace=False))
        path = np.full(T, blank)
        path[slots] = tg[b, :n]
        clash = np.nonzero((tg[b, 1:n] == tg[b, :n - 1]) & (slots[1:] == slots[:-1] + 1))[0] + 1
        path[slots[clash]] = blank
        x[b, np.arange(T), path] += boost
    return (torch.from_numpy(x), torch.from_numpy(tg), torch.full((B,), T, dtype=torch.long), torch.from_numpy(tl))


def emission_regime_numbers(dev, reps=5):
    """The headline shape on emissions other than unit-variance noise: what the step costs where users
    run it.  `trained_regime`: peaky emissions consistent with the targets (boost 6 / 10 / 14 on an alignment);
    `label_noise`: the same with 8 of the 256 utterances given another utterance's targets (mislabelled data);
    `sharp_unrelated`: logits x 8 against unrelated random targets (the worst case for a scaled lattice).  One C-ABI call
    each, HIP events, inputs resident; `flagged_utterances` = utterances the f32 lattice handed to the f64 code inside
    the same call, `unsettled` = those of them the segment redo could not settle (recomputed in full)."""
    import ctypes
    import torch
    from end2end_amd import _lib
    L = _lib.load()
    w = WORKLOAD
    B, T, V, S = w["B"], w["T"], w["V"], w["S"]
    L.e2e_debug_fast_state.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p] * 2
    L.e2e_debug_fast_redo_failures.argtypes = [ctypes.c_void_p] + [ctypes.c_int] * 4 + [ctypes.c_void_p]

    def leg(host, what):
        db = tuple(t.to(dev) for t in host)
        hp = HotPath(db)
        for _ in range(2):
            hp.call(hp.means[0, :1])
        ms = time_events(torch, lambda: hp.call(hp.means[0, :1]), reps)
        fl = (ctypes.c_int * B)()
        lz = (ctypes.c_double * (2 * B))()
        L.e2e_debug_fast_state(hp.ws.data_ptr(), B, T, V, S, fl, lz)
        un = ctypes.c_int(0)
        L.e2e_debug_fast_redo_failures(hp.ws.data_ptr(), B, T, V, S, ctypes.byref(un))
        finite = bool(torch.isfinite(hp.losses).all().item()) and bool(torch.isfinite(hp.grads).all().item())
        reasons = {}
        for v in fl:
            for bit in (1, 2, 4, 8, 16, 32, 64):
                if v & bit:
                    reasons[str(bit)] = reasons.get(str(bit), 0) + 1
        return {"workload": what, "ms": ms, "flagged_utterances": sum(1 for v in fl if v), "unsettled": int(un.value),
                "flag_reasons": reasons,
                "frames_per_s": B * T / (ms * 1e-3), "mean_loss": float(hp.losses.mean().item()), "finite": finite}

    out = {"trained_regime": {}}
    for boost in (6.0, 10.0, 14.0):
        host = aligned_batch(int(boost), B, T, V, S, boost)
        out["trained_regime"]["boost_%d" % boost] = leg(
            host, "B=256 T=1000 V=29 S<=200, unit noise + %g on an alignment of the utterance's own targets" % boost)
        if boost == 10.0:
            x, tg, xl, tl = host
            tg2, tl2 = tg.clone(), tl.clone()
            for k in range(8):                                   # utterance 32k gets the targets of utterance 32k+1
                tg2[32 * k], tl2[32 * k] = tg[32 * k + 1], tl[32 * k + 1]
            out["label_noise"] = leg((x, tg2, xl, tl2), "the boost-10 batch with 8 of 256 utterances given another "
                                                        "utterance's targets")
    out["trained_regime"]["ms"] = max(v["ms"] for v in out["trained_regime"].values())
    out["trained_regime"]["flagged_utterances"] = sum(v["flagged_utterances"] for k, v in out["trained_regime"].items()
                                                      if k.startswith("boost_"))
    g = torch.Generator().manual_seed(78)
    x = torch.randn(B, T, V, generator=g) * 8.0
    tg = torch.randint(1, V, (B, S), generator=g)
    tl = torch.randint(S // 2, S + 1, (B,), generator=g)
    out["sharp_unrelated"] = leg((x, tg, torch.full((B,), T, dtype=torch.long), tl),
                                 "B=256 T=1000 V=29 S<=200, logits x8 against unrelated random targets")
    return out


def shape_cliff_numbers(dev):
    """Loss shapes beside the headline one: what a call costs where the fast paths end.  One C-ABI call each,
    HIP events, inputs resident."""
    import torch
    out = {}
    for name, (B, T, V, S) in (("long_targets_T2000_S400", (256, 2000, 29, 400)),
                               ("long_targets_T1000_S300", (256, 1000, 29, 300)),
                               ("wordpiece_V8000_S200", (64, 256, 8000, 200)),
                               ("wordpiece_V32000_S120", (16, 150, 32000, 120)),
                               ("wordpiece_V8000_S400", (32, 700, 8000, 400)),
                               ("mid_alphabet_V200_S200", (256, 1000, 200, 200))):
        _, db = make_batch(7000, B, T, V, S, dev)
        hp = HotPath(db)
        for _ in range(2):
            hp.call()
        ms = time_events(torch, hp.call, 3)
        out[name] = {"workload": "B=%d T=%d V=%d S in [%d,%d] f32" % (B, T, V, S // 2, S), "ms": ms,
                     "frames_per_s": B * T / (ms * 1e-3),
                     "path": "fast path (eight pairs per lane)" if V <= 96 else
                             "fast path, wide-row form (97..448 columns: probability table + halo chains over an f32 ring)" if V <= 448 else
                             "wide path: streaming rows + the fast lattice's wide-row form on the compact columns (more than 95 distinct labels)"}
        del hp, db
        torch.cuda.empty_cache()
    return out


def recorded_beam_phases():
    """Per-phase s_memtime cycles of the fast beam kernel from the committed record (tools/diag/beam_phase_profile.py), or None."""
    try:
        with open(os.path.join(ROOT, "profiles", "r06_beam_phases.json")) as f:
            return json.load(f)
    except Exception:
        return None


def recorded_traffic(workload):
    """HBM bytes per launch from the committed rocprofv3 --pmc passes (profiles/), or None."""
    for name in ("r06_traffic.json", "r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                rec = json.load(f)
            if rec.get("workload") == workload:
                return rec["traffic_bytes_per_launch"]
        except Exception:
            pass
    return None


def main():
    args = parse_args()
    if args.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(args))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the product path)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    # under torchrun (RANK / WORLD_SIZE in the environment) the process group is always initialised, also for one
    # rank, so that the RCCL path is the same code at every N
    distributed = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group("nccl", device_id=dev)
    n_gpus = world if distributed else 1
    if args.gpus != n_gpus and rank == 0:
        print("bench.py: --gpus %d but %d rank(s) were launched; reporting n_gpus=%d" % (args.gpus, n_gpus, n_gpus),
              file=sys.stderr)

    def max_over_ranks(v):
        if not distributed:
            return v
        t = torch.tensor([v], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def fence():
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    w = WORKLOAD
    host_batch, dev_batch = make_batch(1000 + rank, w["B"], w["T"], w["V"], w["S"], dev)
    frames = int(host_batch[2].sum().item())
    hp = HotPath(dev_batch)

    pending = []

    def flush(bucket_index, count):
        # the one exchange of the sharded path: the per-step mean losses of a bucket, summed over ranks (nothing on
        # the device consumes the global loss, so it may arrive a few steps late; every step's loss IS reduced)
        while pending:
            pending.pop(0).wait()
        pending.append(dist.all_reduce(hp.means[bucket_index & 1, :count], op=dist.ReduceOp.SUM, async_op=True))

    def step():
        m = hp.step()
        if distributed and hp.k % hp.bucket == 0:
            flush(hp.k // hp.bucket - 1, hp.bucket)
        return m

    def drain():
        if distributed and hp.k % hp.bucket:
            flush(hp.k // hp.bucket, hp.k % hp.bucket)
            hp.k += hp.bucket - hp.k % hp.bucket           # start the next bucket fresh
        while pending:
            pending.pop(0).wait()

    # ---- leg 1: the C-ABI step (value) ------------------------------------------------------------------------------
    for _ in range(args.warmup):
        step()
    drain()
    fence()
    ev0 = torch.cuda.Event(enable_timing=True)
    ev1 = torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    drain()
    ev1.record()
    fence()
    wall = max_over_ranks(time.perf_counter() - t0)

    # dominant kernels' duration: HIP events on the launch stream around each of K launches (second pass,
    # kernels only, so that the mean/all-reduce tail is not attributed to them)
    kernel_ms = time_events(torch, lambda: hp.call(hp.means[0, :1]), args.steps)

    # ---- SURVEY.md 8(d)'s ragged-length variant of the same shape: x_len = randint(T//2, T+1) ------------------------------
    gr = torch.Generator().manual_seed(2000 + rank)
    xl_r = torch.randint(w["T"] // 2, w["T"] + 1, (w["B"],), generator=gr)
    hpr = HotPath((dev_batch[0], dev_batch[1], xl_r.to(dev), dev_batch[3]))
    for _ in range(max(2, args.warmup)):
        hpr.call(hpr.means[0, :1])
    ragged_ms = time_events(torch, lambda: hpr.call(hpr.means[0, :1]), args.steps)
    ragged_frames = int(xl_r.sum().item())
    del hpr

    # ---- the caller's option e2e_ctc_loss_opts.chains = E2E_CHAINS_F32 (not the headline: looser gradient tolerance) ---
    from end2end_amd import _lib as _lib_mod
    hp32 = HotPath(dev_batch, chains=_lib_mod.CHAINS_F32)
    for _ in range(max(2, args.warmup)):
        hp32.call(hp32.means[0, :1])
    f32_ms = time_events(torch, lambda: hp32.call(hp32.means[0, :1]), args.steps)
    hp.call()
    hp32.call()                       # (unscaled gradients for the comparison)
    f32_grad_dev = float((hp32.grads - hp.grads).abs().max().item())
    f32_loss_dev = float(((hp32.losses - hp.losses).abs() / hp.losses.abs().clamp_min(1.0)).max().item())
    del hp32

    # ---- leg 2: the metric through the Python surface (module_ms_per_step) ------------------------------------------
    from end2end_amd import CTCLoss
    crit = CTCLoss(reduce=True, size_average=True, blank_idx=0)
    xm = dev_batch[0].clone().requires_grad_()
    tg, xl, tl = dev_batch[1], dev_batch[2], dev_batch[3]

    def module_step():
        xm.grad = None
        loss = crit(xm, tg, xl, tl)
        loss.backward()
        return loss

    for _ in range(args.warmup):
        module_step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        lm_ = module_step()
    fence()
    module_wall = max_over_ranks(time.perf_counter() - t0)
    module_loss = float(lm_.item())
    c_abi_loss = float(hp.means[0, 0].item())

    # ---- leg 2b (under the launcher): what the package ships for a batch spread over ranks -- end2end_amd.parallel.ShardedCTCLoss
    #      (per-utterance losses, ONE synchronous 16-byte all-reduce of [sum, count] per call, gradient of the GLOBAL mean) +
    #      backward().  The C-ABI leg above reduces its scalar losses in buckets of 8 steps, asynchronously; this is the module. ----
    sharded_wall = None
    if distributed:
        from end2end_amd.parallel import ShardedCTCLoss
        scrit = ShardedCTCLoss(size_average=True, blank_idx=0)
        xs = dev_batch[0].clone().requires_grad_()

        def sharded_step():
            xs.grad = None
            loss = scrit(xs, tg, xl, tl)
            loss.backward()
            return loss

        for _ in range(args.warmup):
            sharded_step()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ls_ = sharded_step()
        fence()
        sharded_wall = max_over_ranks(time.perf_counter() - t0)
        sharded_loss = float(ls_.item())

    # ---- leg 3: one GPU's share of configs[4] (V=8000), the HBM-bound shape, at every N ------------------------------
    wide = wide_bf16 = None
    if not args.no_wide:
        ww = WIDE
        _, wb = make_batch(5000 + rank, ww["B"], ww["T"], ww["V"], ww["S"], dev)
        wp = HotPath(wb)
        for _ in range(2):
            wp.call()
        fence()
        reps = 5
        t0 = time.perf_counter()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            wp.call()
        e1.record()
        fence()
        wwall = max_over_ranks(time.perf_counter() - t0)
        wms = max_over_ranks(e0.elapsed_time(e1) / reps)
        walgo = 2.0 * ww["V"] * 4 * ww["B"] * ww["T"]
        wide = {"workload": "B=512 per GPU, T=256 V=8000 S<=64 f32 (BASELINE configs[4] sharded by utterance)",
                "n_gpus": n_gpus, "ms": wms, "frames_per_s": n_gpus * ww["B"] * ww["T"] * reps / wwall,
                "algorithmic_bytes_per_gpu": walgo,
                "roofline": {"bound": "hbm", "achieved": walgo / (wms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": walgo / (wms * 1e-3) / 1e9 / HBM_PEAK_GBS},
                "note": "same C-ABI call as the headline; ms = device time per call (max over ranks), frames_per_s = "
                        "whole job over the wall clock of %d back-to-back calls" % reps}
        # the same share with bf16 logits, read and written natively: 2 * V * 2 algorithmic bytes per frame
        wb16 = (wb[0].to(torch.bfloat16),) + tuple(wb[1:])
        del wp
        wp = HotPath(wb16)
        for _ in range(2):
            wp.call()
        fence()
        wms16 = max_over_ranks(time_events(torch, wp.call, reps))
        walgo16 = 2.0 * ww["V"] * 2 * ww["B"] * ww["T"]
        wide_bf16 = {"workload": "B=512 per GPU, T=256 V=8000 S<=64 bf16 logits in, bf16 gradient out (f32 lattice on the compact columns)",
                     "n_gpus": n_gpus, "ms": wms16, "frames_per_s": n_gpus * ww["B"] * ww["T"] / (wms16 * 1e-3),
                     "algorithmic_bytes_per_gpu": walgo16,
                     "roofline": {"bound": "hbm", "achieved": walgo16 / (wms16 * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": walgo16 / (wms16 * 1e-3) / 1e9 / HBM_PEAK_GBS},
                     "note": "HIP events around each call (mean of %d); no f32 copy of the logits or of the gradient exists" % reps}
        del wp, wb, wb16
        torch.cuda.empty_cache()

    if rank == 0:
        total_frames = frames * n_gpus
        ms_per_step = wall * 1e3 / args.steps
        algo_bytes = 2.0 * w["V"] * 4 * frames          # per launch (one GPU's batch)
        achieved = algo_bytes / (kernel_ms * 1e-3) / 1e9
        copy_gbs = measured_copy_gbs(torch, dev)
        # the VALU bound of the recurrence: every lattice cell costs >= 4 lane-operations per direction in
        # the chains and the same again in the segment kernel's recompute
        cells = float((host_batch[2].double() * (2.0 * host_batch[3].double() + 1.0)).sum().item())
        lane_ops = cells * 2 * 4 * 2
        out = {
            "metric": "ctc_fwd_bwd_frames_per_sec", "value": total_frames * args.steps / wall, "unit": "frames/s",
            "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": w["name"], "B_per_gpu": w["B"], "T": w["T"], "V": w["V"], "S_max": w["S"],
                       "input": "raw logits (log-softmax fused)",
                       "api": "C ABI e2e_ctc_loss_fwd_bwd_opt (losses, d mean-loss / d logits, mean loss: one call)"
                       + (" + RCCL all_reduce of the per-step losses, bucketed by 8 steps" if distributed else ""),
                       "sharding": "utterances, %d per GPU" % w["B"]},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": recorded_traffic(w["name"]),
                         "peak_measured": copy_gbs, "frac_of_measured": achieved / copy_gbs,
                         "kernel": "e2e_ctc_loss_fwd_bwd: every kernel of the call counted",
                         "kernel_ms": kernel_ms, "algorithmic_bytes_per_launch": algo_bytes,
                         "valu_frac": lane_ops / (kernel_ms * 1e-3) / VALU_PEAK_LANE_OPS,
                         "valu_note": "minimal lane-operations (lattice cells x 2 directions x 4 x 2 for the recompute = %.3g) "
                                      "/ kernel time / %.3g lane-ops/s (256 CU x 4 SIMD x 32 lanes x 2.4 GHz, the f32 "
                                      "vector peak of MI355X_MICROARCH.md; f64 runs at half of it)" % (lane_ops, VALU_PEAK_LANE_OPS)},
            "event_ms_per_step": ev0.elapsed_time(ev1) / args.steps,
            "module_ms_per_step": module_wall * 1e3 / args.steps,
            "module_frames_per_s": total_frames * args.steps / module_wall,
            "module_api": "loss = end2end_amd.CTCLoss(reduce=True, size_average=True)(logits, targets, lengths...); "
                          "loss.backward()  (same batch; loss %.6f vs C-ABI %.6f)" % (module_loss, c_abi_loss),
        }
        if sharded_wall is not None:
            out["sharded_module_ms_per_step"] = sharded_wall * 1e3 / args.steps
            out["sharded_module_frames_per_s"] = total_frames * args.steps / sharded_wall
            out["sharded_module_api"] = ("loss = end2end_amd.parallel.ShardedCTCLoss(size_average=True)(logits, ...); loss.backward(): every rank "
                                         "its own %d utterances, one synchronous all-reduce of [sum of losses, count] per call "
                                         "(global mean loss %.6f)" % (w["B"], sharded_loss))
        out["ragged_lengths"] = {
            "workload": "the headline batch with x_len = randint(T//2, T+1) (SURVEY.md 8d's ragged variant)",
            "kernel_ms": ragged_ms, "frames": ragged_frames, "frames_per_s_per_gpu": ragged_frames / (ragged_ms * 1e-3)}
        out["f32_chains_option"] = {
            "what": "the same call with e2e_ctc_loss_opts.chains = E2E_CHAINS_F32 (lattice chains in packed f32); an option, "
                    "not the headline: gradient elements are promised to 2e-5 absolute instead of 2e-6",
            "kernel_ms": f32_ms, "frames_per_s_per_gpu": frames / (f32_ms * 1e-3),
            "max_grad_abs_dev_from_default": f32_grad_dev, "max_loss_rel_dev_from_default": f32_loss_dev}
        if wide is not None:
            out["wide_alphabet"] = wide
            out["wide_alphabet_bf16"] = wide_bf16
        if n_gpus == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_loss(host_batch, frames)
        if n_gpus == 1 and not args.no_decode:
            out["decode"] = decode_numbers(dev, not args.no_cpu_baseline)
            out["fallback_regime"] = fallback_regime_numbers(dev)
            out["shape_cliffs"] = shape_cliff_numbers(dev)
            out["emission_regimes"] = emission_regime_numbers(dev)
        print(json.dumps(out), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
