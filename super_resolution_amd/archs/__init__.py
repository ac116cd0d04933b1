"""Importing this package registers `HAT` under the name 'HAT' (hat/archs/__init__.py:8-11 does the
same by scanning `*_arch.py`)."""
from .hat_arch import HAT  # noqa: F401
