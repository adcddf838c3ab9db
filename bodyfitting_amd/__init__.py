"""bodyfitting_amd: MI355X-native multi-view SMPLify inner loop behind the reference's entry points.

    from bodyfitting_amd.smplify import SMPLify            # reference smplify/smplify.py
    from bodyfitting_amd.body_fitting import BodyFitting   # reference smplify/body_fitting.py
    from bodyfitting_amd.smpl import SMPL                   # reference models/smpl.py

Importing the package never touches the GPU; the HIP library is loaded on first use and its absence
is an error (there is no CPU implementation of the product path).
"""
# (`bodyfitting_amd.synthetic` - the generator of test / benchmark inputs - is imported by the tests and bench.py only)
__version__ = "0.1"
