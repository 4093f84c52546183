"""CPU oracle for the cgs-vmc hot path.  Test infrastructure: see vmc_oracle.py header."""
from . import vmc_oracle  # noqa: F401
