"""Run pytest against a variant build of libkvhip.so: python tools/ab_pytest.py build/ab/<name>.so [pytest args]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (HIP runtime of torch first)
from tfplus_amd import _lib
_lib.SO_PATH = os.path.abspath(sys.argv[1])
import pytest
sys.exit(pytest.main(sys.argv[2:]))
