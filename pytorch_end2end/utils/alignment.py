from end2end_amd.utils.alignment import get_alignment_3d  # noqa: F401
