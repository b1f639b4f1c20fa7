#!/bin/bash
# usage (on the GPU box, via gpurun): scripts_prof.sh <tag> [bench args]  -> gpurun_out/prof_<tag>/
# kernel trace + stats in one run, PMC counters in separate runs (never combined with tracing).
tag=$1; shift
repo=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
out=$repo/gpurun_out/prof_$tag
mkdir -p $out
# (the trace and the two HBM passes run bench.py WITH its extras, so that k_uapply — the op boundary — and k_part2 are in them)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $repo/bench.py --steps 20 --warmup 3 --no-cpu-baseline "$@" > $out/bench_under_trace.json 2> $out/trace.err
for c in "FETCH_SIZE" "WRITE_SIZE"; do
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$c -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_$c.err
done
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$n -- python3 $repo/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras "$@" > /dev/null 2> $out/pmc_$n.err
done
python3 $repo/tools/prof_summary.py $out > $out/summary.txt
cat $out/summary.txt
