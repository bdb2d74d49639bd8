"""The driver's entry points on the GPU box: build() and smoke() of __graft_entry__.py in ONE interpreter.  build() loads the HIP
library to check that it imports; smoke() then needs torch for device memory -- and a process holds exactly one HIP runtime: the
library loaded ahead of torch brought the system's libamdhip64 in beside the copy torch ships, and whichever touched the device
second found none (round 4).  _lib.load() imports torch first now; this is the sequence that failed."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
@pytest.mark.parametrize("first", ["import __graft_entry__ as g; g.build(); g.smoke()",
                                   "from openmm_drudenose_amd import _lib; _lib.load(); import __graft_entry__ as g; g.smoke()"])
def test_library_then_torch_in_one_process(first):
    p = subprocess.run([sys.executable, "-c", first], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-2000:]
    assert "smoke: 20 steps" in p.stdout
