"""The public autograd path under torch.cuda.graph (no test captured it).  Each case runs in a child process:
a capture gone wrong aborts the process rather than raising.  python -m pytest tests -m gpu"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("kind,precision", [("bilinear", "bf16"), ("bilinear", "f32"), ("bilinear", "f32_exact"), ("concat_mlp", "f32"),
                                            ("concat_mlp", "bf16")])
def test_fused_mi_bound_forward_backward_capture(kind, precision):
    r = subprocess.run([sys.executable, os.path.join(HERE, "capture_worker.py"), kind, precision], capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, f"child exit {r.returncode}\n{r.stdout[-2000:]}\n{r.stderr[-4000:]}"
    assert "capture ok" in r.stdout
