"""CPU oracle of the DrudeTGNHIntegrator step path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  Pin status: statistical only (the reference's testWater passes on it; no golden vectors exist):
see oracle/tgnh_oracle.h and DESIGN.md section 6.
"""
from .binding import Oracle, build_oracle, water_forces, MODE_DUALNH, MODE_TGNH  # noqa: F401
