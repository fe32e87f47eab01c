from .text_encoders import CTCEncoder  # noqa: F401
