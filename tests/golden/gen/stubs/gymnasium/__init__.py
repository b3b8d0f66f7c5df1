"""Test-only stand-in for the handful of `gymnasium` names the reference's planning
modules touch at import/run time (gymnasium itself is not installed in this image).

Used ONLY by tests/golden/gen/make_golden.py to import the unmodified reference from
/root/reference and generate golden vectors.  Not part of the product.
Surface follows SURVEY.md Appendix C.
"""
from . import core, spaces, error, logger, utils  # noqa: F401
from .core import Env, Wrapper  # noqa: F401


def make(*args, **kwargs):
    raise error.Error("gymnasium stub: make() is not available")
