"""`pytorch_end2end` -- the reference's package name, served by the MI355X implementation in `end2end_amd`.

    from pytorch_end2end import CTCLoss, CTCDecoder, CTCEncoder

works unchanged (pytorch_end2end/__init__.py:1-6 upstream), and so do the sub-module paths the reference's users and
tests import (`pytorch_end2end.modules.ctc_loss`, `.decoders.ctc_decoder`, `.encoders.text_encoders`,
`.functions.forward_backward`).  Only the CTC hot path exists here; the numba back-ends, Gram-CTC, CTC-without-blank
and the alignment losses of the upstream package are out of scope.
"""
from end2end_amd import CTCDecoder, CTCDecoderError, CTCEncoder, CTCLoss, DecoderResults

__all__ = ["CTCLoss", "CTCDecoder", "CTCEncoder"]
