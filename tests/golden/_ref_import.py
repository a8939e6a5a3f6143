"""Import harness for the upstream reference (ONLY usable in the build container).

/root/reference never travels to the GPU box, so nothing under ``tests/`` that runs with
``-m gpu`` (nor ``bench.py`` / ``smoke()``) may import this module.  It is used by
``make_golden.py`` to regenerate the committed fixtures and by the optional
``tests/test_oracle_vs_reference_live.py`` cross-check, which skips when the reference is absent.

nibabel / SimpleITK are not installed in the image; the reference only needs their names for
annotations and isinstance checks (convex_adam_utils.py:268-279), so empty stub modules suffice.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("CONVEXADAM_REFERENCE", "/root/reference")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "src", "convexAdam"))


def import_reference():
    """Returns (convex_adam_utils, convex_adam_MIND) modules of the upstream reference."""
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    for m in ("nibabel", "SimpleITK"):
        if m not in sys.modules:
            sys.modules[m] = types.ModuleType(m)
    if not hasattr(sys.modules["nibabel"], "Nifti1Image"):
        sys.modules["nibabel"].Nifti1Image = type("Nifti1Image", (), {})
    if not hasattr(sys.modules["SimpleITK"], "Image"):
        sys.modules["SimpleITK"].Image = type("Image", (), {})
    src = os.path.join(REF_ROOT, "src")
    if src not in sys.path:
        sys.path.insert(0, src)
    import importlib
    utils = importlib.import_module("convexAdam.convex_adam_utils")
    mind = importlib.import_module("convexAdam.convex_adam_MIND")
    return utils, mind
