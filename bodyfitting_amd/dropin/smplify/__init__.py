"""Put `bodyfitting_amd/dropin` first on sys.path and the reference's own import lines
(`from smplify.body_fitting import BodyFitting`, apps/genebody_fitting.py:9) resolve to the HIP path."""
