"""Soak run of the randomised differential tests with seeds outside the test suite's fixed set:
python tools/soak.py FIRST LAST [occ]  (on the GPU box).  Prints one line per seed; exits non-zero on a mismatch.
occ: the fuzz programs in occurrence-order mode (raw repeated ids to the optimizer ops; tests/test_gpu_fuzz.py)."""
import os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from tfplus_amd.kv_variable.python.ops import gen_kv_variable_ops as ops
import test_gpu_fuzz, test_gpu_delta_export
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
occ = len(sys.argv) > 3 and sys.argv[3] == "occ"
progs = ((("fuzz-occ", lambda o, sd: test_gpu_fuzz._random_program(o, sd, True)),) if occ else
         (("fuzz", test_gpu_fuzz.test_random_program_matches_oracle), ("delta", test_gpu_delta_export.test_delta_lists_match_oracle)))
for seed in range(first, last):
  for name, fn in progs:
    try:
      fn(ops, seed)
      print("seed %d %s ok" % (seed, name), flush=True)
    except Exception:
      bad += 1
      print("seed %d %s FAILED" % (seed, name), flush=True)
      traceback.print_exc()
sys.exit(1 if bad else 0)
