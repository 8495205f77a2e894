"""Engine adapter so tests/cases.py can drive the HIP path through its C ABI."""
from fabber_core_amd import hiplib


def run(holder, data):
    return hiplib.run_host(holder, data)
