from bodyfitting_amd.smpl import SMPL, ModelOutput  # noqa: F401
