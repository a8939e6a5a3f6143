"""Import shim: `from convexAdam.apply_convex import apply_convex, apply_convex_original_moving` (reference module name) -> HIP implementation."""
from convexadam_amd.apply_convex import apply_convex, apply_convex_original_moving  # noqa: F401
