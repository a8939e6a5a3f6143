"""Import shim: `from convexAdam.apply_convex import apply_convex` (reference module name) -> HIP implementation."""
from convexadam_amd.apply_convex import apply_convex  # noqa: F401
