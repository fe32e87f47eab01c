from end2end_amd.functions.forward_backward import ForwardBackwardLossFunction  # noqa: F401
