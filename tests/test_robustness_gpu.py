"""GPU robustness checks: device memory does not grow with repeated use of the whole C-ABI."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_device_memory_growth():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "leak_check.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.strip().endswith("ok")
