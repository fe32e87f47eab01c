import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """Case counts of the oracle-checked fuzz slices (tests/test_gpu_fuzz.py), so that they show in the run's tail."""
    mod = sys.modules.get("test_gpu_fuzz")
    counts = getattr(mod, "COUNTS", None) if mod is not None else None
    if counts:
        loss = sum(v[0] for k, v in counts.items() if k != "align")
        terminalreporter.write_line("fuzz slices against the oracle: %d loss cases (%s), %d alignment cases; beam: 220 cases x 2 kernels in test_gpu_beam.py"
                                    % (loss, ", ".join("%s %d cases / %d utterances" % (k, v[0], v[1]) for k, v in sorted(counts.items()) if k != "align"),
                                       counts.get("align", [0, 0])[0]))
