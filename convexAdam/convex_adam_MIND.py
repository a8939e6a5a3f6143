"""Shim: re-exports convexadam_amd.convex_adam_MIND under the upstream module name."""
from convexadam_amd.convex_adam_MIND import *  # noqa: F401,F403
from convexadam_amd import convex_adam_MIND as _impl

__all__ = [n for n in dir(_impl) if not n.startswith("_")]
