"""A/B of two builds of libkvhip.so on the same box: python tools/ab_bench.py old.so [bench args].
Symbols the older build lacks are dropped from the binding table (bench.py does not call them)."""
import ctypes, os, runpy, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime of torch first)
from tfplus_amd import _lib
so = os.path.abspath(sys.argv[1])
_lib.SO_PATH = so
probe = ctypes.CDLL(so)
for name in list(_lib.SIGNATURES):
  if not hasattr(probe, name):
    del _lib.SIGNATURES[name]
sys.argv = ["bench.py"] + sys.argv[2:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
