"""convexadam_amd -- MI355X-native engine for the convexAdam registration hot path.

Drop-in mirror of the reference's operator interface (same names, arguments and output formats):
    convexadam_amd.convex_adam_utils : MINDSSC, correlate, coupled_convex, inverse_consistency
    convexadam_amd.convex_adam_MIND  : extract_features, convex_adam_pt, convex_adam
    convexadam_amd.convex_adam_nnUNet: extract_features, convex_adam
All arithmetic runs in hand-written HIP kernels (convexadam_amd/csrc, C ABI in include/).
"""
__version__ = "0.1.0"
