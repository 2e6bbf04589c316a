"""Stand-in for the ``nose2`` package, which the reference's test files import for their parametrisation helpers
(``nose2.tools.params``, ``nose2.tools.such``) and which is not installed here.  Test infrastructure: only on
PYTHONPATH of the child process in which tests/test_reference_tests_verbatim.py runs the reference's own, unmodified
test files under pytest."""
from nose2 import tools  # noqa: F401
