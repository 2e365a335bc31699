"""Static hazard audit of the compiled fused bilinear kernel (tools/diag/audit_flash_isa.py): the kernel issues its
MFMAs from inline asm, where hipcc's hazard recognizer cannot protect their destinations.  Runs on the CPU (hipcc
cross-compiles for gfx950)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("hipcc") is None, reason="hipcc not available")
def test_flash_kernel_isa_is_hazard_free():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "diag", "audit_flash_isa.py")], capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
