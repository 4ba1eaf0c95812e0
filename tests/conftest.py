import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


def pytest_collection_modifyitems(config, items):
    """A `-m gpu` session needs the measurement-only libraries too (bench.py's census, the forced code paths of the
    fuzz test): when one is missing -- fm_index_amd._lib.build_library() only warns when their build fails -- the
    session fails at collection instead of losing tests to skips."""
    if not any(it.get_closest_marker("gpu") for it in items) or "gpu" not in (config.getoption("-m") or ""):
        return
    if "not gpu" in (config.getoption("-m") or ""):
        return
    missing = [n for n in ("libfmx_census.so", "libfmx_measure.so")
               if not os.path.exists(os.path.join(ROOT, "fm_index_amd", n))]
    if missing:
        raise pytest.UsageError("GPU test session without %s: run `python -c 'import __graft_entry__ as g; g.build()'` "
                                "(or `make -C fm_index_amd/csrc census measure`)" % ", ".join(missing))


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "reference_known_answers.json")) as f:
        return json.load(f)
