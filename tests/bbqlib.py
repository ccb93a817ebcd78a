"""imports the product's Python plumbing (better-binary-quantization_amd/python/bbq_amd)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = os.path.join(ROOT, "better-binary-quantization_amd", "python")
if PY not in sys.path:
    sys.path.insert(0, PY)

# torch bundles its own copy of the HIP runtime (same soname as /opt/rocm's): when a test uses torch and libbbq in one
# process, torch has to be imported first so that both share ONE runtime - the order bench.py uses as well
try:
    import torch  # noqa: F401,E402
except Exception:  # pragma: no cover - torch is optional for the library itself
    pass

import bbq_amd  # noqa: E402
from bbq_amd import capi  # noqa: E402,F401
