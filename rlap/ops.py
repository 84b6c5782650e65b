"""rlap.ops -> rlap_amd.ops (same names as the reference's rlap/ops.py)."""
from rlap_amd.ops import approximate_cholesky, identity  # noqa: F401
