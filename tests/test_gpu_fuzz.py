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


def test_fuzz_refinement_rounds_on_small_texts():
    """the builder re-sorts only the still-tied suffixes once few are left (k_refine_*), from n = 2^19 on in the shipped
    library; the measurement build takes the threshold from the environment, so that the random / repetitive / run-heavy
    texts of the fuzz driver go through those rounds"""
    lib = os.path.join(ROOT, "fm_index_amd", "libfmx_measure.so")
    # no skip: this is the only coverage of the refinement rounds on small texts -- a measurement library that did
    # not build (build_library() only warns about it) must turn the suite red, not quietly shrink it
    assert os.path.exists(lib), "fm_index_amd/libfmx_measure.so is missing: run `make -C fm_index_amd/csrc measure`"
    env = dict(os.environ, FMX_LIB=lib, FMX_REFINE_MIN_N="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz_gpu_vs_oracle.py"), "30", "13"],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "fuzz ok" in out.stdout
