"""Build-time checks that need the compiler but no GPU."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_async_residual_loads_are_not_touched_before_their_wait():
    """tools/check_async_loads.py: the inline-asm residual loads of the ping-pong bf16 GEMM must not have their
    destination registers read or written between the load and the counted wait (exit code 1 on a hazard)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_async_loads.py")], capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "asynchronous loads" in r.stdout
