"""The TF custom-op shim (tfplus_amd/tf_shim/kv_variable_ops_hip.cc) cannot be compiled in this image (no
TensorFlow headers), so its op SCHEMAS — the drop-in interface: op name, inputs and outputs in order, attr names,
types and defaults, statefulness — are checked here as text against the reference's own REGISTER_OP blocks
(tfplus/kv_variable/ops/*.cc), read at test time; nothing of the reference is stored in this repository.
Skipped where the reference tree is absent (the GPU box)."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "tfplus_amd", "tf_shim", "kv_variable_ops_hip.cc")
REF_OPS = "/root/reference/tfplus/kv_variable/ops"

# every op of SURVEY.md §8(b), the "next" rows, and what the reference's Python layer needs to build a graph with a
# Saver (KvVariableSaveable) and its hygiene calls
REQUIRED = ["KvVariable", "KvVariableV2", "KvVariableV3", "KvVariableV4", "InitKvVariableV2", "KvVariableIsInitializedV2",
            "KvVariableShapeV2", "KvVariableSizeV2", "KvVariableFrequency", "ReadKvVariableOpV2", "DestroyKvVariableOpV2",
            "KvVariableGatherOrInsertV2", "KvVariableGatherOrInsertWithCounts", "KvVariableGatherOrZerosV2",
            "KvVariableGroupSparseApplyAdamV4", "KvVariableGroupSparseApplyAdamV3", "KvVariableSparseApplyAdagrad",
            "KvVariableSparseGroupSparseApplyFtrlV2", "KvVariableInsertV2", "KvVariableScatterUpdateV2",
            "KvVariableScatterAddV2", "KvVariableScatterSubV2", "KvVariableScatterMulV2", "KvVariableScatterDivV2",
            "KvVariableScatterMinV2", "KvVariableScatterMaxV2",
            "KvVariableExport", "KvVariableImport", "KvVariableFullOrDeltaImport", "KvVariableFullOrDeltaImportV2",
            "KvVariableFullOrDeltaExport", "KvVariableSizeV3", "KvVariableGetCountV2", "KvVariableGetTimeStamp",
            "KvVariableDelete", "KvVariableDeleteWithTimestamp", "BatchKvVariableGatherOrZerosV2",
            "KvVariableIncreaseCountV2"]

# ops the reference's Python layer also names but that are outside the hot path and its neighbours:
# multi-hash variables, snapshots / remote tables / the dynamic restore and save_v3 checkpoint plumbing, and the
# separate-slot GroupAdam variants that only non-default kv_options reach
EXCLUDED = {"append_kv_variable_for_multi_hash", "kv_variable_export_for_multi_hash", "kv_variable_apply_snapshot",
            "kv_variable_apply_snapshot_v2", "kv_variable_dynamic_restore", "kv_variable_load_remote_table", "save_v3",
            "kv_variable_group_sparse_apply_adam_v2", "kv_variable_group_sparse_apply_adam_new_v2"}
REF_PY = "/root/reference/tfplus/kv_variable/python"


def _snake(name):
  """TF's op-name -> python wrapper name rule (KvVariableV4 -> kv_variable_v4)."""
  out = re.sub(r"(?<=[a-z0-9])(?=[A-Z])|(?<=[A-Z])(?=[A-Z][a-z])", "_", name).lower()
  return out


@pytest.mark.skipif(not os.path.isdir(REF_PY), reason="reference tree not present")
def test_required_covers_what_the_reference_python_layer_calls():
  files = [os.path.join(REF_PY, "ops", f) for f in ("kv_variable_ops.py", "embedding_ops.py", "variable_scope.py")] + \
          [os.path.join(REF_PY, "training", f) for f in ("group_adam.py", "adagrad.py", "sparse_group_ftrl.py", "adam.py",
                                                         "gradient_descent.py")]
  called = set()
  for f in files:
    if os.path.exists(f):
      called |= set(re.findall(r"gen_kv_variable_ops\.([a-z_0-9]+)", open(f).read()))
  have = {_snake(n) for n in REQUIRED}
  missing = sorted(called - have - EXCLUDED)
  assert not missing, missing


def _strip_comments(text):
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return re.sub(r"//[^\n]*", "", text)


def _schemas(text):
  """{op name: [(kind, spec), ...]} with kind in Input / Output / Attr / SetIsStateful, in source order."""
  text = _strip_comments(text)
  out = {}
  for m in re.finditer(r'REGISTER_OP\(\s*"(\w+)"\s*\)', text):
    i, depth, end = m.end(), 0, None
    while i < len(text):          # the statement ends at the first ';' outside parentheses / braces
      c = text[i]
      if c in "({":
        depth += 1
      elif c in ")}":
        depth -= 1
      elif c == ";" and depth == 0:
        end = i
        break
      i += 1
    body = text[m.end():end]
    items = []
    for k in re.finditer(r'\.(Input|Output|Attr)\(\s*((?:"[^"]*"\s*)+)\)|\.(SetIsStateful)\(\)', body):
      if k.group(3):
        items.append(("SetIsStateful", ""))
      else:
        spec = "".join(re.findall(r'"([^"]*)"', k.group(2)))     # adjacent string literals concatenate
        items.append((k.group(1), re.sub(r"\s+", " ", spec.strip())))
    out[m.group(1)] = items
  return out


def test_shim_registers_every_hot_path_op():
  ours = _schemas(open(SHIM).read())
  missing = [n for n in REQUIRED if n not in ours]
  assert not missing, missing
  # every REGISTER_OP of the shim has at least one kernel registered for it
  text = _strip_comments(open(SHIM).read())
  built = set(re.findall(r'Name\(\s*"(\w+)"\s*\)', text)) | set(re.findall(r'KV_REGISTER_\w+\(\s*"(\w+)"', text))
  assert set(ours) <= built, sorted(set(ours) - built)


def _expand_macros(text):
  """Kernel registrations as the preprocessor would leave them: the shim's own registration macros expanded (they are
  simple token pastes), so that every registration reads Name("Op").Device(DEVICE_X)...; one string per registration."""
  text = _strip_comments(text)
  # object-like / function-like macros of the shim, as (name, params, body)
  macros = {}
  for m in re.finditer(r'#define\s+(KV_\w+)(\(([^)]*)\))?((?:[^\n\\]|\\\n|\\.)*)', text):
    params = [a.strip() for a in m.group(3).split(",")] if m.group(3) else None
    macros[m.group(1)] = (params, m.group(4).replace("\\\n", " "))
  body = re.sub(r'#define\s+KV_\w+(?:[^\n\\]|\\\n|\\.)*', "", text)

  def split_args(a):
    out, depth, cur = [], 0, ""
    for ch in a:
      if ch in "(<": depth += 1
      if ch in ")>": depth -= 1
      if ch == "," and depth == 0:
        out.append(cur.strip()); cur = ""
      else:
        cur += ch
    out.append(cur.strip())
    return out

  def expand(t, depth=0):
    if depth > 6:
      return t
    changed = True
    while changed:
      changed = False
      for name, (params, mbody) in macros.items():
        if params is None:
          if re.search(r'\b%s\b' % name, t):
            t = re.sub(r'\b%s\b' % name, mbody, t); changed = True
          continue
        m = re.search(r'\b%s\(' % name, t)
        if not m:
          continue
        i, d = m.end(), 1
        while d:
          d += {"(": 1, ")": -1}.get(t[i], 0); i += 1
        args = split_args(t[m.end():i - 1])
        b = mbody
        if params and params[-1] == "...":
          fixed = params[:-1]
          va = ", ".join(args[len(fixed):])
          for p_, a_ in zip(fixed, args):
            b = re.sub(r'\b%s\b' % p_, a_, b)
          b = b.replace("__VA_ARGS__", va)
        else:
          for p_, a_ in zip(params, args):
            b = re.sub(r'\b%s\b' % p_, a_, b)
        t = t[:m.start()] + b + t[i:]
        changed = True
    return t
  flat = expand(body)
  regs = []
  for m in re.finditer(r'REGISTER_KERNEL_BUILDER\(\s*Name\(\s*"(\w+)"\s*\)', flat):
    i, d = m.start() + len("REGISTER_KERNEL_BUILDER("), 1
    while d:
      d += {"(": 1, ")": -1}.get(flat[i], 0); i += 1
    regs.append((m.group(1), re.sub(r"\s+", "", flat[m.start():i])))
  return regs


def test_shim_registers_device_gpu_kernels():
  """the measured path must be reachable from a TF graph — the lookups and the optimizer ops take
  device-resident indices / grad / output on DEVICE_GPU (TensorFlow-ROCm's device), the resource handle and the scalar
  hyper-parameters in host memory; every other op has a DEVICE_GPU registration too (a resource is visible to kernels
  of its own device only)."""
  text = open(SHIM).read()
  regs = _expand_macros(text)
  by_dev = {"DEVICE_CPU": {}, "DEVICE_GPU": {}}
  for name, r in regs:
    for dev in by_dev:
      if ".Device(%s)" % dev in r:
        by_dev[dev].setdefault(name, []).append(r)
  ours = _schemas(text)
  for dev in by_dev:
    assert set(ours) <= set(by_dev[dev]), (dev, sorted(set(ours) - set(by_dev[dev])))
  gpu = by_dev["DEVICE_GPU"]
  for op in ("KvVariableGatherOrZerosV2", "KvVariableGatherOrInsertV2", "KvVariableGatherOrInsertWithCounts"):
    assert len(gpu[op]) == 3, op                                   # int32 / int64 / uint64 indices
    for r in gpu[op]:
      assert 'HostMemory("table_handle")' in r and "KvGatherGpuOp" in r, r
      assert 'HostMemory("indices")' not in r and 'HostMemory("output")' not in r, r    # device-resident
  scalars = {"KvVariableGroupSparseApplyAdamV4": ("var", "m_v_linear", "lr", "beta1_power", "beta2_power", "beat1", "beta2", "epsilon", "l1", "l2", "l21"),
             "KvVariableGroupSparseApplyAdamV3": ("var", "m_v_linear", "lr", "beta1_power", "beta2_power", "beat1", "beta2", "epsilon", "l1", "l2", "l21"),
             "KvVariableSparseApplyAdagrad": ("var", "accum", "lr"),
             "KvVariableSparseGroupSparseApplyFtrlV2": ("var", "accum", "linear", "lr", "l1", "l2", "l21", "l2_shrinkage", "lr_power")}
  for op, host in scalars.items():
    assert len(gpu[op]) == 3, op
    for r in gpu[op]:
      assert "GpuOp" in r, r
      for h in host:
        assert 'HostMemory("%s")' % h in r, (op, h)
      assert 'HostMemory("grad")' not in r and 'HostMemory("indices")' not in r, r
    # ... and the host-memory names are inputs of the op
    names = [spec.split(":")[0].strip() for kind, spec in ours[op] if kind == "Input"]
    assert set(host) <= set(names), (op, set(host) - set(names))
  # every HostMemory argument of every registration names an input or output of its op
  for dev in by_dev:
    for op, rs in by_dev[dev].items():
      names = {spec.split(":")[0].strip() for kind, spec in ours[op] if kind in ("Input", "Output")}
      for r in rs:
        for h in re.findall(r'HostMemory\("(\w+)"\)', r):
          assert h in names, "%s: HostMemory(%s) is not an argument of the op (%s)" % (op, h, sorted(names))
  for op in ("KvVariable", "KvVariableV2", "KvVariableV3", "KvVariableV4", "InitKvVariableV2"):
    assert any('HostMemory("table_handle")' in r for r in gpu[op]), op


@pytest.mark.skipif(not os.path.isdir(REF_OPS), reason="reference tree not present")
def test_shim_schemas_equal_the_reference():
  ref = {}
  for f in sorted(glob.glob(os.path.join(REF_OPS, "*.cc"))):
    ref.update(_schemas(open(f).read()))
  ours = _schemas(open(SHIM).read())
  assert len(ours) >= len(REQUIRED)
  for name, items in ours.items():
    assert name in ref, "the reference registers no op named %s" % name
    assert items == ref[name], "%s:\n  shim      %s\n  reference %s" % (name, items, ref[name])


def test_shim_type_checks_against_a_mock_of_the_tf_api():
  """TensorFlow's headers are not in this image, so the shim cannot be built here.  tests/tf_mock/ declares the slice
  of TF 2.13's op API the shim uses (OpKernel, OpKernelContext, Tensor, ResourceMgr, shape inference, the
  registration macros); `g++ -fsyntax-only` against it catches everything but a mismatch between the mock and the
  real headers (INTEGRATION.md §2 has the real build line)."""
  import shutil
  import subprocess
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  if shutil.which("g++") is None or not os.path.isdir("/opt/rocm/include/hip"):
    pytest.skip("needs g++ and the HIP headers")
  r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I", os.path.join(root, "include"), "-I",
                      os.path.join(root, "tests", "tf_mock"), "-I", "/opt/rocm/include", "-D__HIP_PLATFORM_AMD__",
                      os.path.join(root, "tfplus_amd", "tf_shim", "kv_variable_ops_hip.cc")],
                     capture_output=True, text=True, timeout=300)
  assert r.returncode == 0, r.stderr[-3000:]


def test_unique_producer_rule(tmp_path):
  """the shim takes the racy-unless-unique one-launch apply when `indices` is output 0 of a tf.unique
  node.  The rule (tfplus_amd/tf_shim/unique_input.h, plain strings) is compiled and run here: the exact leaf Unique /
  UniqueV2 with TensorFlow's _<n> suffix, slot 0 only — ':1' is the inverse index vector, UniqueWithCounts and a
  placeholder called UniqueIds are not it, a control edge carries no tensor."""
  import shutil
  import subprocess
  if shutil.which("g++") is None:
    pytest.skip("needs g++")
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  cases = [("Unique", 1), ("Unique:0", 1), ("Unique_1:0", 1), ("tower/gradients/Unique_12", 1), ("a/b/UniqueV2:0", 1),
           ("UniqueV2_3", 1), ("Unique:1", 0), ("scope/Unique:1", 0), ("UniqueWithCounts", 0), ("UniqueWithCounts:0", 0),
           ("UniqueWithCounts:2", 0), ("UniqueIds", 0), ("UniqueIds:0", 0), ("Unique_", 0), ("Unique_x1", 0),
           ("Unique_1a", 0), ("^Unique", 0), ("scope/^Unique", 0), ("NotUnique", 0), ("unique", 0), ("", 0),
           ("Unique/read:0", 0), ("Unique:00", 0), ("UniqueV2:1", 0), ("UniqueV3", 0)]
  src = tmp_path / "u.cc"
  src.write_text('#include "unique_input.h"\n#include <cstdio>\nint main(int c, char** v) { for (int i = 1; i < c; ++i) '
                 'std::printf("%d\\n", kv_shim::InputIsUniqueValues(v[i]) ? 1 : 0); return 0; }\n')
  exe = tmp_path / "u"
  subprocess.check_call(["g++", "-std=c++17", "-I", os.path.join(root, "tfplus_amd", "tf_shim"), "-o", str(exe), str(src)])
  # (an empty argv string is passed through as is)
  out = subprocess.run([str(exe)] + [c for c, _ in cases], capture_output=True, text=True, check=True).stdout.split()
  got = [int(x) for x in out]
  assert got == [w for _, w in cases], [(c, g, w) for (c, w), g in zip(cases, got) if g != w]
