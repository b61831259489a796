#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline numbers are checked against.
# Usage (on the GPU box, from the repo root):  bash profiles/run_rocprof.sh <tag>
# Kernel trace + stats first, then PMC counters in their own passes (never combined with
# sys/hip/hsa tracing — see the task notes).  Outputs land in gpurun_out/prof_<tag>/ and the
# summaries are copied to profiles/<tag>_*.
set -u
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- $CMD > $OUT/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- $CMD > $OUT/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $OUT/pmc_valu -o valu -- $CMD > $OUT/pmc_valu.log 2>&1
find $OUT -name "*.csv" | head -50
python3 $ROOT/profiles/summarize.py $OUT $TAG
