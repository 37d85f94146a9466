"""CPU oracle (TEST INFRASTRUCTURE ONLY) -- see oracle/ronk_oracle.h.

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
"""
from .oracle import *  # noqa: F401,F403
