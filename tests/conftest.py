import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))
if str(ROOT / "tests") not in sys.path:
    sys.path.insert(0, str(ROOT / "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def engine():
    """One CorrField on cuda:0 for the whole GPU test session (a single process uses the card)."""
    import correrender_amd as ca
    eng = ca.CorrField(0)
    yield eng
    eng.close()


def pytest_sessionfinish(session, exitstatus):
    """GPU sessions: the relative errors the floating-point parity checks saw (parity.REPORT), without the absolute
    floor, for values above 1e-3 -- copied into profiles/ by the builder."""
    try:
        import json
        import parity
        if parity.REPORT:
            out = ROOT / "gpurun_out"
            out.mkdir(exist_ok=True)
            worst = max(parity.REPORT, key=lambda r: r["max_rel_err_above_1e-3"])
            (out / "parity_relative_errors.json").write_text(json.dumps(
                {"checks": len(parity.REPORT), "worst": worst, "all": parity.REPORT}, indent=1))
    except Exception:
        pass
