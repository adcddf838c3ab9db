from .smpl import SMPL  # noqa: F401
