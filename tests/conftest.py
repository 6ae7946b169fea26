import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_addoption(parser):
    parser.addoption("--parity-tol", default=None, metavar="TOL",
                     help="A/B runs of builds with relaxed arithmetic ONLY (tools/r05_contract.sh): replace the bit-for-bit "
                          "requirement of the parity tests by this relative-L1 bar.  Never used by the driver's runs.")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the tolerance mode is an explicit command-line choice: a stray AKMI_PARITY_TOL in the environment must not weaken
    # the suite silently
    tol = config.getoption("--parity-tol")
    if tol is None and os.environ.get("AKMI_PARITY_TOL"):
        raise pytest.UsageError("AKMI_PARITY_TOL is set in the environment but --parity-tol was not given: the parity tests "
                                "assert bit equality; unset the variable (or pass --parity-tol for an A/B run)")
    import parity_util
    parity_util.RELAXED_TOL = float(tol) if tol is not None else None
    if tol is not None:
        os.environ["AKMI_PARITY_TOL"] = str(tol)        # tests that compare in a process of their own
        os.environ["AKMI_PARITY_RELAXED_BY_OPTION"] = "1"


def pytest_report_header(config):
    tol = config.getoption("--parity-tol")
    if tol is not None:
        return "RELAXED PARITY MODE: bit equality replaced by relative L1 <= %s (--parity-tol) -- not a parity run" % tol


def pytest_collection_modifyitems(config, items):
    """a plain `pytest tests/` on a machine without a GPU skips the gpu-marked tests instead of failing them"""
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this process (marked gpu)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _built():
    """make sure the oracle (and, when hipcc is present, the HIP library) are built"""
    from oracle import akref
    akref.build()
    yield
