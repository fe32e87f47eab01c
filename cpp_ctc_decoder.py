"""`cpp_ctc_decoder` -- the name under which the reference's wrapper finds the decoder engine
(`import cpp_ctc_decoder`, pytorch_end2end/decoders/ctc_decoder.py:13,61-64; the pybind module of
src/decoders/ctc_decoder_py.cpp:5-39).  `CTCDecoder(blank_idx, beam_width_=100, labels=[], lm_path="", lmwt_=1.0,
wip_=0.0, oov_penalty_=-1000.0, case_sensitive=False)` with `decode_greedy(logits_, logits_lengths_)`,
`decode(logits_, logits_lengths_)` and `print_scores_for_sentence(words)`: the MI355X engine under the reference's
class name, keyword names and defaults.
"""
from end2end_amd.engines import CTCDecoderEngine as CTCDecoder

__all__ = ["CTCDecoder"]
