"""Makes the `mmx` engine package importable from the drop-in module tree (speech/ and dac-vae/ are what a user
puts on sys.path in place of the reference's directories of the same name)."""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
