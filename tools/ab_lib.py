"""Run a probe with an alternative build of the library: python tools/ab_lib.py <lib.so> <script> [args]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
