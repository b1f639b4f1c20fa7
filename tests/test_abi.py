"""CPU checks of the drop-in boundary: libkvhip.so builds for gfx950, loads, and exports
every symbol include/kvhip.h declares.  No compute is issued (there is no GPU here)."""
import ctypes
import os
import re

import pytest

from tfplus_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
  text = open(os.path.join(ROOT, "include", "kvhip.h")).read()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(kv_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def so():
  return ctypes.CDLL(_lib.build())


def test_header_symbols_are_exported(so):
  names = _declared()
  assert len(names) >= 20
  for n in names:
    assert hasattr(so, n), "libkvhip.so does not export %s" % n


def test_binding_table_matches_header():
  assert sorted(_lib.SIGNATURES) == _declared()


def test_code_object_targets_gfx950():
  blob = open(_lib.build(), "rb").read()
  assert b"gfx950" in blob
  for other in (b"gfx942", b"gfx90a", b"sm_80"):
    assert other not in blob


def test_status_codes_match_header():
  text = open(os.path.join(ROOT, "include", "kvhip.h")).read()
  for name in ("KV_OK", "KV_INVALID_ARGUMENT", "KV_RESOURCE_EXHAUSTED", "KV_FAILED_PRECONDITION",
               "KV_UNIMPLEMENTED", "KV_INTERNAL", "KV_DT_FLOAT", "KV_DT_INT32", "KV_DT_INT64",
               "KV_DT_UINT64"):
    m = re.search(r"#define\s+%s\s+(\d+)" % name, text)
    assert m and int(m.group(1)) == getattr(_lib, name)


def test_product_never_imports_the_oracle():
  bad = []
  for d, _, files in os.walk(os.path.join(ROOT, "tfplus_amd")):
    for f in files:
      if f.endswith((".py", ".hip", ".h", ".cc")):
        s = open(os.path.join(d, f), errors="replace").read()
        if re.search(r"^\s*(from|import)\s+oracle\b", s, flags=re.M) or "kv_oracle.py" in s:
          bad.append(os.path.join(d, f))
  assert not bad, bad


@pytest.mark.gpu
def test_c_host_program_over_the_abi(tmp_path):
  """A C++ program that includes only include/kvhip.h + HIP (no Python, no torch) drives the table."""
  import shutil
  import subprocess
  torch = pytest.importorskip("torch")
  if not torch.cuda.is_available():
    pytest.skip("needs a GPU")
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  exe = str(tmp_path / "c_abi_smoke")
  csrc = os.path.join(ROOT, "tfplus_amd", "csrc")
  subprocess.check_call([hipcc, "-std=c++17", "-I", os.path.join(ROOT, "include"),
                         os.path.join(ROOT, "tests", "c_abi", "c_abi_smoke.cc"), "-L", csrc, "-lkvhip",
                         "-Wl,-rpath," + csrc, "-o", exe])
  r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
  assert r.returncode == 0 and "C ABI OK" in r.stdout, (r.stdout, r.stderr)
