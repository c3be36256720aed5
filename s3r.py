"""Importable alias of the package directory `stereo-3d-reconstruction_amd/` (a hyphenated name
cannot appear in an `import` statement): `import s3r` gives that package."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
sys.modules[__name__] = importlib.import_module("stereo-3d-reconstruction_amd")
