"""Import shim: `from convexAdam.convex_adam_translation import ...` (reference module name) -> convexadam_amd implementation."""
from convexadam_amd.convex_adam_translation import (apply_translation, convex_adam_translation,  # noqa: F401
                                                    convex_adam_translation_from_file, index_translation_to_world_translation)
