"""Import shim: `from convexAdam.apply_convex import apply_convex, apply_convex_original_moving` (reference module name) -> HIP
implementation; `python -m convexAdam.apply_convex --input_field ... --input_moving ... --output_warped ...` like the reference."""
from convexadam_amd.apply_convex import apply_convex, apply_convex_original_moving, main  # noqa: F401

if __name__ == "__main__":
    import sys
    sys.exit(main())
