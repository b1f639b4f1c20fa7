#!/bin/bash
# usage: scripts_prof.sh <tag> [bench args]   -> gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --no-cpu-baseline "$@" > $out/bench.json 2> $out/trace.err
for c in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_ATOMIC_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
  n=$(echo $c | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $c --output-format csv -d $out/pmc_$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2> $out/pmc_$n.err
done
find $out -name "*.csv" | head -50
