"""`cpp_ctc_loss` -- the name under which the reference's callers find the CTC loss engine
(`import_module("cpp_ctc_loss").CTCLossEngine(blank_idx)`, pytorch_end2end/modules/ctc_loss.py:74-75; the pybind
module of src/losses/ctc_loss_py.cpp:5-17).  Here the class is the MI355X engine: same constructor, same
`compute(logits, targets, logits_lengths, targets_lengths) -> (losses, grads)`, results on the source device and
dtype, computed by libe2e_ctc.so through the pybind11 layer end2end_amd._C.
"""
from end2end_amd.engines import CTCLossEngine

__all__ = ["CTCLossEngine"]
