"""end2end_amd -- MI355X-native CTC loss and decoder with the pytorch_end2end API.

    from end2end_amd import CTCLoss, CTCDecoder, CTCEncoder

is a drop-in for `from pytorch_end2end import CTCLoss, CTCDecoder, CTCEncoder`
(pytorch_end2end/__init__.py:1-6): same constructors, call signatures and results, computed
by hand-written HIP kernels (end2end_amd/csrc, C ABI in include/e2e_ctc.h).
"""
from .decoders.ctc_decoder import CTCDecoder, CTCDecoderError, DecoderResults
from .encoders.text_encoders import CTCEncoder
from .modules.ctc_loss import CTCLoss

__all__ = ["CTCLoss", "CTCDecoder", "CTCEncoder", "CTCDecoderError", "DecoderResults"]
