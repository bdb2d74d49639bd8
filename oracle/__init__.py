"""CPU oracle of the DrudeTGNHIntegrator step path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  PARITY UNPINNED: see oracle/tgnh_oracle.h and DESIGN.md "Oracle".
"""
from .binding import Oracle, build_oracle, MODE_DUALNH, MODE_TGNH  # noqa: F401
