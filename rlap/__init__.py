"""`import rlap` shim: the reference's package name (rlap/__init__.py) bound to the
MI355X implementation, so scripts written against kvignesh1420/rlap run unchanged."""
from rlap_amd import ops  # noqa: F401
from rlap_amd import VERSION  # noqa: F401
