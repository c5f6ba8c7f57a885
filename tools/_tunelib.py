"""Tools only: bind pips_amd to the library named by PIPS_LIB_PATH (tuning / trace / ablation builds) before anything
loads the product one.  Imported first by every script under tools/ that imports pips_amd; the product package itself
reads no environment variable (pips_amd/_lib.py::use_library)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pips_amd import _lib  # noqa: E402

if os.environ.get("PIPS_LIB_PATH"):
    _lib.use_library(os.environ["PIPS_LIB_PATH"])
