"""CTCDecoder wrapper with the reference's constructor, defaults and return type
(pytorch_end2end/decoders/ctc_decoder.py:25-149).  The inputs stay on the GPU instead of being copied to
the host (`.cpu()` at :100,:137 upstream); the results are CPU tensors as upstream unless
``keep_on_device=True`` (an extension: ids and lengths then stay on the GPU that decoded them).
"""
import os
from collections import namedtuple

import torch

from ..engines import CTCDecoderEngine


class CTCDecoderError(Exception):
    pass


DecoderResults = namedtuple("DecoderResults", ["decoded_targets", "decoded_targets_lengths", "decoded_sentences"])


class CTCDecoder:
    """
    :param beam_width: number of hypotheses kept; ``1`` means greedy (argmax) decoding
    :param after_logsoftmax: inputs are log-probabilities (greedy ignores this)
    :param blank_idx: index of the blank label
    :param time_major: inputs are ``(time, batch, alphabet)``
    :param labels: label strings including the blank, e.g. ``["_", "a", "b", " "]``
    :param lm_path: ARPA (optionally gzipped) language model
    :param lmwt: language-model weight
    :param wip: word insertion penalty
    :param oov_penalty: penalty per out-of-vocabulary word
    :param case_sensitive: look words up in the language model with their case
    :param keep_on_device: (extension) leave the decoded ids and lengths on the GPU
    """

    def __init__(self, beam_width=100, after_logsoftmax=False, blank_idx=0, time_major=False, labels=None,
                 lm_path=None, lmwt=1.0, wip=1.0, oov_penalty=-10, case_sensitive=True, keep_on_device=False):
        self._beam_width = beam_width
        self._blank_idx = blank_idx
        self._after_logsoftmax = after_logsoftmax
        self._labels = labels or []
        self._lm_path = os.path.abspath(lm_path) if lm_path else ""
        self._lmwt = lmwt
        self._wip = wip
        self._oov_penalty = oov_penalty
        self._time_major = time_major
        self._case_sensitive = case_sensitive
        self._check_params()
        self._decoder = CTCDecoderEngine(self._blank_idx, self._beam_width, self._labels, self._lm_path,
                                         self._lmwt, self._wip, self._oov_penalty, self._case_sensitive,
                                         keep_on_device=keep_on_device)

    def _check_params(self):
        if self._lm_path:
            # (upstream also tests `self._labels is None` here, which cannot hold after `labels or []`,
            # pytorch_end2end/decoders/ctc_decoder.py:53,69: a model without labels is accepted at construction;
            # decode() then fails because the alphabet cannot spell words)
            if not os.path.isfile(self._lm_path):
                raise CTCDecoderError("Can't find a model: {}".format(self._lm_path))

    def _batch_major(self, logits, logits_lengths):
        if self._time_major:
            logits = logits.transpose(1, 0)
        logits = logits.detach()
        if logits_lengths is None:
            logits_lengths = torch.full((logits.size(0),), logits.size(1), dtype=torch.int32, device=logits.device)
        return logits, logits_lengths

    def decode(self, logits, logits_lengths=None):
        """Prefix beam search (Hannun et al., 2014).  ``beam_width == 1`` routes to greedy decoding.

        :return: ``DecoderResults(decoded_targets (batch, longest), decoded_targets_lengths, decoded_sentences)``
        """
        if self._beam_width == 1:
            return self.decode_greedy(logits, logits_lengths)
        with torch.no_grad():
            if not self._after_logsoftmax:
                logits = torch.log_softmax(logits, -1)
        logits, logits_lengths = self._batch_major(logits, logits_lengths)
        return DecoderResults(*self._decoder.decode(logits_=logits, logits_lengths_=logits_lengths))

    def _print_scores_for_sentence(self, words):
        self._decoder.print_scores_for_sentence(words)

    def decode_greedy(self, logits, logits_lengths=None):
        """Greedy (argmax) decoding: works on raw logits or log-probabilities.

        :return: ``DecoderResults(decoded_targets (batch, time) zero padded, decoded_targets_lengths, decoded_sentences)``
        """
        logits, logits_lengths = self._batch_major(logits, logits_lengths)
        return DecoderResults(*self._decoder.decode_greedy(logits_=logits, logits_lengths_=logits_lengths))
