"""stainlib_amd -- MI355X-native drop-in for the hot path of sebastianffx/stainlib.

Mirrors the export list of stainlib/__init__.py:19-30 for the classes on the path.
"""
from . import _ffi  # noqa: F401

__version__ = "0.1.0"
