"""`utils.mesh_grid_searcher` of the reference resolves here (smplify/smplify.py:15); every other `utils.*` module
(`utils.io_utils`, apps/genebody_fitting.py:14) still resolves to a `utils` package further down sys.path - the reference's own."""
import os
import sys

_here = os.path.abspath(os.path.dirname(__file__))
for _d in list(sys.path):
    _p = os.path.abspath(os.path.join(_d or ".", "utils"))
    if _p != _here and _p not in __path__ and os.path.isfile(os.path.join(_p, "__init__.py")):
        __path__.append(_p)
