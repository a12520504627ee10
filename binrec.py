"""Import shim: `import binrec` -> the package in ./binary-recommendation_amd/ (hyphenated
directory names cannot be written in an import statement)."""
import importlib
import os
import sys

_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)
_pkg = importlib.import_module("binary-recommendation_amd")
sys.modules[__name__] = _pkg
