"""CPU oracle for rLap's approximate_cholesky -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package.  rlap_amd/ (the product) never does.
"""
from .pyoracle import (  # noqa: F401
    approximate_cholesky,
    build,
    stdsort_perm,
    heapsort_perm,
    keyed_order,
    uniforms,
    O_V,
    O_N,
)
