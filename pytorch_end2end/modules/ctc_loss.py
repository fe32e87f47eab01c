from end2end_amd.modules.ctc_loss import CTCLoss, ForwardBackwardLossBase, GramCTCLoss  # noqa: F401
