from bodyfitting_amd.body_fitting import BodyFitting  # noqa: F401
