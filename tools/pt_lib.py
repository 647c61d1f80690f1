import sys, os, runpy
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rdst_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, sys.argv[1])
sys.argv = ["pass_times.py"] + sys.argv[2:]
runpy.run_path(os.path.join(ROOT, "tools", "pass_times.py"), run_name="__main__")
