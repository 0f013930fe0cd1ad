#!/usr/bin/env python3
"""Diagnostic: run bench.py against an alternative build of the library (A/B of kernel variants).
    python tools/ab.py <path/to/lib.so> [bench.py flags]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import qpn_amd  # noqa: E402,F401
from qpn_amd import _lib  # noqa: E402
_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
