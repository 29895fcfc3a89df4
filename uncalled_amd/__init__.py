"""uncalled_amd: MI355X-native UNCALLED map path (see README.md / DESIGN.md).

PyTorch ROCm wheels bundle their own libamdhip64.so.7 / libhsa-runtime64 under torch/lib, with the same SONAME as the
system ROCm runtime that libuncalled_hip.so links to.  Whichever copy is loaded first serves the whole process, and a
process that loads the system copy (through this package) and imports torch afterwards aborts inside the mixed
runtimes.  So when torch is installed it is imported first; set UNCALLED_AMD_NO_TORCH=1 for a torch-free process.
"""
import importlib.util
import os
import sys

if "torch" not in sys.modules and not os.environ.get("UNCALLED_AMD_NO_TORCH") and importlib.util.find_spec("torch") is not None:
    import torch  # noqa: F401  (HIP runtime load order, see above)
