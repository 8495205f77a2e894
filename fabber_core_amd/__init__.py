"""MI355X-native voxelwise Variational Bayes (the hot path of fabber_core), Python side."""
import importlib.util
import sys


def single_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64. If the system copy is loaded first (by
    the engine library) and torch's later, torch reports "No HIP GPUs are available"; the other
    way round both share torch's copy. So: when torch is installed, import it before the engine
    library is opened. Called by every loader in this package; C / C++ callers are not concerned."""
    import os
    if os.environ.get("FVB_NO_TORCH") == "1":
        # a process that stands for a C / C++ caller (bench.py's e2e child): the system's HIP runtime, no torch at all
        return
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
