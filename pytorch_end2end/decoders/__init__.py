from .ctc_decoder import CTCDecoder  # noqa: F401
