from .hat_model import HATModel  # noqa: F401
