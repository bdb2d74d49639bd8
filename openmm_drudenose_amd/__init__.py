"""MI355X-native (gfx950 HIP) DrudeTGNHIntegrator step path behind a C ABI.

Product code only: nothing here imports the CPU oracle (``oracle/``).  The HIP
library must be present; there is no CPU fallback.
"""
from .system import DrudeSystem  # noqa: F401
from .drudetgnhplugin import DrudeTGNHIntegrator, HipContext, HostTopology, TgnhError  # noqa: F401
from . import synth  # noqa: F401

__all__ = ["DrudeSystem", "DrudeTGNHIntegrator", "HipContext", "HostTopology", "TgnhError", "synth"]
