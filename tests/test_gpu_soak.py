"""Short runs of the randomised differential soaks (tools/soak_scan.py, tools/soak_api.py): fixed base seeds, a few seconds
each; the long runs of the same tools are how two latent build bugs were found (see their docstrings)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("tool", ["soak_scan.py", "soak_api.py"])
def test_short_soak(tool):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool), "8"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "soak ok" in p.stdout, p.stdout[-2000:] + p.stderr[-2000:]
