"""MI355X-native HAT super-resolution forward pass.

Importing the package is cheap and does not touch the GPU; the HIP library is loaded on first
use by `super_resolution_amd._lib` and its absence is a hard error (there is no CPU fallback).
"""
__version__ = "0.1.0"
