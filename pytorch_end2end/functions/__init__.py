from .forward_backward import ForwardBackwardLossFunction  # noqa: F401
