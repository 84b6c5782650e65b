"""rlap_amd -- MI355X-native drop-in for rlap.ops.approximate_cholesky.

Mirrors the reference package layout (rlap/__init__.py:6-8): `ops` + VERSION.
"""
from . import ops  # noqa: F401
from . import graphs  # noqa: F401

VERSION = "0.0.1"
