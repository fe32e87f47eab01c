from .text_encoders import CTCEncoder

__all__ = ["CTCEncoder"]
