"""`from drudetgnhplugin import *` -- the module name of the reference's SWIG wrapper (python/drudetgnhplugin.i:1),
so that a script written for it (example/nacl_tg.py: `from drudetgnhplugin import *`) finds `DrudeTGNHIntegrator`
under the same name.  Here the class drives the MI355X HIP library instead of an OpenMM platform."""
from openmm_drudenose_amd.drudetgnhplugin import DrudeTGNHIntegrator, HipContext, HostTopology, TgnhError  # noqa: F401

__all__ = ["DrudeTGNHIntegrator", "HipContext", "HostTopology", "TgnhError"]
