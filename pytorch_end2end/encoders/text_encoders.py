from end2end_amd.encoders.text_encoders import CTCEncoder  # noqa: F401
