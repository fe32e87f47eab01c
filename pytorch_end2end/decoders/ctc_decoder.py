from end2end_amd.decoders.ctc_decoder import CTCDecoder, CTCDecoderError, DecoderResults  # noqa: F401
