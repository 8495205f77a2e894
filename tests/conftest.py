import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_terminal_summary(terminalreporter):
    """Which strict() comparisons ran at a tolerance raised to the measured CPU-vs-CPU floor
    (tests/parity.py); also written to gpurun_out/parity_raised.json when that directory exists."""
    import json
    import sys
    import parity
    try:  # which HIP runtime this process ended up with (PyTorch's bundled copy, or the system's with FVB_NO_TORCH=1)
        libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l})
        if libs:
            terminalreporter.write_line("HIP runtime of this process: %s (torch imported: %s)" % (", ".join(libs), "torch" in sys.modules))
    except OSError:
        pass
    if not parity.RAISED:
        return
    terminalreporter.write_line("parity.strict ran at a raised tolerance in %d comparison(s):" % len(parity.RAISED))
    for what, tol, err in parity.RAISED:
        terminalreporter.write_line("  %-40s applied %s observed %s" % (what, json.dumps(tol), json.dumps(err)))
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "parity_raised.json"), "w") as fh:
            json.dump([dict(what=w, applied=t, observed=e) for w, t, e in parity.RAISED], fh, indent=1)
