import os
import sys

import pytest

# MIOpen's exhaustive kernel search for the CLIP patch-embedding convolution takes minutes on a
# fresh machine; the heuristic mode is enough for tests.  Must be set before MIOpen initialises.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")
AUDIT_LINES = []     # what tests/test_attack_gpu.py's audits of `check_against_golden` found, printed again at the very end


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    # -m gpu tests are skipped (not failed) when no device is visible, so a plain
    # `pytest tests/` in the CPU container stays green.
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """The golden-comparison audit (how many reference trajectories were compared to their LAST step on this device) at the
    end of the report, where the tail of a driver's log keeps it."""
    for line in AUDIT_LINES:
        terminalreporter.write_line("golden audit: " + line)
