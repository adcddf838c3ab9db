from bodyfitting_amd.smplify import SMPLify  # noqa: F401
