"""imports the product's Python plumbing (better-binary-quantization_amd/python/bbq_amd)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PY = os.path.join(ROOT, "better-binary-quantization_amd", "python")
if PY not in sys.path:
    sys.path.insert(0, PY)

import bbq_amd  # noqa: E402
from bbq_amd import capi  # noqa: E402,F401
