"""Drop-in for the reference's utils_encoding.py (imported by main.py:19)."""
from gsn_amd.encoding import encode, one_hot_unique, one_hot_max  # noqa: F401
