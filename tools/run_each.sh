#!/bin/bash
# run every GPU test file separately under a hard timeout; one line per file
cd ${GRAFT_REPO_ROOT:-/root/repo}
for f in tests/test_gpu_parity.py tests/test_gpu_sharded_two_ranks.py tests/test_gpu_fuzz.py tests/test_gpu_multi_table.py tests/test_gpu_lookup_sparse.py tests/test_next_rows.py tests/test_hygiene.py tests/test_gpu_delta_export.py tests/test_gpu_python_api.py tests/test_gpu_configs.py tests/test_gpu_graph_capture.py tests/test_gpu_config_sizes.py tests/test_gpu_bench_cli.py; do
  timeout -k 5 400 python -m pytest $f -m gpu -q -x --timeout 350 > gpurun_out/each_$(basename $f .py).log 2>&1
  rc=$?
  echo "$f rc=$rc $(tail -1 gpurun_out/each_$(basename $f .py).log)" >> gpurun_out/each_summary.log
  echo "$f rc=$rc"
  if [ $rc -ge 124 ]; then echo "HUNG: stopping"; break; fi
done
