"""Import-name shim: lets callers written for the upstream package keep
`from convexAdam.convex_adam_utils import MINDSSC, correlate, ...` (tests/test_convex_adam_mind.py:10-14,
self_configuring/convex_adam_MIND.py:10-12 of the reference) while running on convexadam_amd."""
