from .ctc_loss import CTCLoss  # noqa: F401
