"""Shim: re-exports convexadam_amd.convex_adam_utils under the upstream module name."""
from convexadam_amd.convex_adam_utils import *  # noqa: F401,F403
from convexadam_amd import convex_adam_utils as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
