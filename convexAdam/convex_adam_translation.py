"""Import shim: `from convexAdam.convex_adam_translation import ...` (reference module name) -> convexadam_amd implementation;
`python -m convexAdam.convex_adam_translation --fixed_path ... --moving_path ... --moving_output_path ...` like the reference."""
from convexadam_amd.convex_adam_translation import (apply_translation, convex_adam_translation,  # noqa: F401
                                                    convex_adam_translation_from_file, index_translation_to_world_translation, main)

if __name__ == "__main__":
    import sys
    sys.exit(main())
