"""Import alias: the package directory is `pytorch-ardae-vae_amd` (not a Python identifier).

    import ardae_amd as net            # same object as importlib.import_module("pytorch-ardae-vae_amd")
"""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("pytorch-ardae-vae_amd")
sys.modules[__name__] = _pkg
