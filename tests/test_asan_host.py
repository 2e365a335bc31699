"""The HOST side of the C-ABI (argument / shape / enum / workspace checks, the planners, the error strings) under
AddressSanitizer: tools/asan_host/build_and_run.sh builds the library's host code instrumented (device code as usual: a GPU
sanitizer build is not available on the pool), tools/asan_host/asan_host_driver.cpp sweeps the queries over valid, ragged and
degenerate shapes and calls every entry point with arguments it has to refuse.  No GPU is touched.  (First run: the sweep
found a division by zero in the workspace planner for a zero-width critic; the queries now return 0 for non-positive
sizes.)  CPU suite only: this file and tools/asan_host/ are in .gpurunignore."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_side_under_the_address_sanitizer(tmp_path):
    if not (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        pytest.skip("no hipcc")
    script = os.path.join(ROOT, "tools", "asan_host", "build_and_run.sh")
    env = dict(os.environ, PATH=os.environ.get("PATH", "") + ":/opt/rocm/bin")
    r = subprocess.run(["bash", script, str(tmp_path)], capture_output=True, text=True, timeout=1500, env=env)
    assert r.returncode == 0 and "asan driver ok" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]
    assert "AddressSanitizer" not in r.stderr
