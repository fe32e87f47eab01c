"""Text <-> label-id helper for CTC (host-side; same behaviour as the reference's
CTCEncoder, pytorch_end2end/encoders/text_encoders.py:8-41)."""
import numpy as np


class CTCEncoder:
    """Maps characters to consecutive ids, leaving `blank_id` free for the CTC blank."""

    def __init__(self, characters, blank_id=0, transform_fn=str.upper):
        self.blank_id = blank_id
        self.transform_fn = transform_fn
        self.char2id = {}
        next_id = 0
        for ch in characters:
            if next_id == blank_id:
                next_id += 1            # skip over the blank's slot
            self.char2id[ch] = next_id
            next_id += 1
        self.id2char = {i: ch for ch, i in self.char2id.items()}
        self.id2char[blank_id] = ""
        self.num_symbols = len(self.id2char)

    def _known(self, text):
        return [ch for ch in self.transform_fn(text) if ch in self.char2id]

    def clean(self, text):
        return "".join(self._known(text))

    def encode(self, text):
        return np.array([self.char2id[ch] for ch in self._known(text)])

    def decode(self, ids_list):
        """Collapse repeats, drop blanks."""
        out, prev = [], object()
        for i in ids_list:
            if i != prev and i != self.blank_id:
                out.append(self.id2char[i])
            prev = i
        return "".join(out)

    def decode_pure(self, ids_list):
        return "".join(self.id2char[i] for i in ids_list)
