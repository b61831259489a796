#!/bin/bash
# Round-5 GPU sessions (through gpurun, from the repo root):  bash tools/gpu_r5.sh <tag> <step> [<step> ...]
# Everything lands in gpurun_out/<tag>/ ; what is worth judging is copied into profiles/ afterwards.
set -u
TAG=${1:-r07a}; shift || true
STEPS=${*:-"quick sweep3"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
export DADA2HIP_WAIT_TIMEOUT_S=${DADA2HIP_WAIT_TIMEOUT_S:-60}
for s in $STEPS; do
  t0=$(date +%s)
  case $s in
    quick)   timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -x --durations=8 -k "${QUICK_K:-abort_hook or entry_barrier or tenant or golden}" > $OUT/gputests_quick.log 2>&1; echo "quick rc=$?" >> $OUT/steps.log; tail -12 $OUT/gputests_quick.log ;;
    tests)   timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=20 > $OUT/gputests.log 2>&1; echo "tests rc=$?" >> $OUT/steps.log; tail -8 $OUT/gputests.log ;;
    sweep3)  timeout 600 python tools/sweep_env.py --config 3 --reps ${REPS3:-3} --list "${SWEEP3:-DADA2HIP_V3_BLOCK=512;DADA2HIP_V2_TAIL=chain}" > $OUT/sweep_cfg3.jsonl 2> $OUT/sweep_cfg3.err; echo "sweep3 rc=$?" >> $OUT/steps.log; cut -c1-600 $OUT/sweep_cfg3.jsonl; tail -3 $OUT/sweep_cfg3.err ;;
    sweep2)  timeout 300 python tools/sweep_env.py --config 2 --reps 5 --list "${SWEEP2:-DADA2HIP_V3_BLOCK=512;DADA2HIP_V2_TAIL=chain}" > $OUT/sweep_cfg2.jsonl 2> $OUT/sweep_cfg2.err; echo "sweep2 rc=$?" >> $OUT/steps.log; cut -c1-600 $OUT/sweep_cfg2.jsonl; tail -3 $OUT/sweep_cfg2.err ;;
    sweepu)  timeout 600 python tools/sweep_env.py --config 3 --uniques ${SWEEP_UNIQUES:-250000} --reps ${REPS3:-5} --list "${SWEEPU:-DADA2HIP_V3_XBAR=0}" > $OUT/sweep_u${SWEEP_UNIQUES:-250000}.jsonl 2> $OUT/sweep_u.err; echo "sweepu rc=$?" >> $OUT/steps.log; cut -c1-400 $OUT/sweep_u${SWEEP_UNIQUES:-250000}.jsonl; tail -3 $OUT/sweep_u.err ;;
    bench3q) timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras ${BENCH_ARGS:-} > $OUT/bench_cfg3_quick.json 2> $OUT/bench_cfg3_quick.err; echo "bench3q rc=$?" >> $OUT/steps.log; cut -c1-1500 $OUT/bench_cfg3_quick.json; tail -3 $OUT/bench_cfg3_quick.err ;;
    bench3)  timeout 1200 python bench.py ${BENCH_ARGS:-} > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench3 rc=$?" >> $OUT/steps.log; cut -c1-1500 $OUT/bench_cfg3.json; tail -3 $OUT/bench_cfg3.err ;;
    bench2|bench4|bench5)
             CFG=${s#bench}; timeout 1200 python bench.py --config $CFG --steps ${BSTEPS:-3} --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > $OUT/bench_cfg$CFG.json 2> $OUT/bench_cfg$CFG.err; echo "$s rc=$?" >> $OUT/steps.log; cut -c1-900 $OUT/bench_cfg$CFG.json; tail -3 $OUT/bench_cfg$CFG.err ;;
    prof3|prof2|prof5)
             CFG=${s#prof}; P=$OUT/prof$CFG; mkdir -p $P
             CMD="python $ROOT/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras ${PROF_ARGS:-}"
             ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- $CMD > $P/trace.log 2>&1 ); echo "$s rc=$?" >> $OUT/steps.log
             python3 profiles/summarize.py $P ${TAG}_cfg$CFG $OUT/summaries "python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras" > $P/summarize.log 2>&1; tail -3 $P/summarize.log ;;
    pmc3|pmc2|pmc5)
             CFG=${s#pmc}; P=$OUT/prof$CFG; mkdir -p $P
             CMD="python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras ${PROF_ARGS:-}"
             ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o fetch -- $CMD > $P/pmc_fetch.log 2>&1 )
             ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o write -- $CMD > $P/pmc_write.log 2>&1 )
             python3 profiles/summarize.py $P ${TAG}_cfg$CFG $OUT/summaries "python bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras" > $P/summarize.log 2>&1
             echo "$s done" >> $OUT/steps.log ;;
    variants) for v in ${VARIANTS:-v0 v1 v2}; do echo "== $v" >> $OUT/variants.jsonl; L=$ROOT/dada2_amd/libdada2hip_$v.so; [ "$v" = base ] && L=$ROOT/dada2_amd/libdada2hip.so; DADA2HIP_LIB=$L timeout 300 python tools/sweep_env.py --config 3 --reps 3 --list "${SWEEPV:-}" >> $OUT/variants.jsonl 2>> $OUT/variants.err; done; echo "variants rc=$?" >> $OUT/steps.log; cut -c1-900 $OUT/variants.jsonl ;;
    summary3) DADA2HIP_V2_SUMMARY=1 timeout 600 python tools/sweep_env.py --config 3 --reps 2 --list "${SWEEP3:-DADA2HIP_V3_OVERLAP=0}" > $OUT/summary_cfg3.jsonl 2> $OUT/summary_cfg3.err; echo "summary3 rc=$?" >> $OUT/steps.log; cut -c1-1100 $OUT/summary_cfg3.jsonl; grep "^\[v3\]" $OUT/summary_cfg3.err | cut -c1-400 | tail -24 ;;
    derep)   DADA2HIP_DEREP_TIMES=1 timeout 600 python tools/bench_derep.py ${DEREP_N:-1200000} 250 > $OUT/derep.json 2> $OUT/derep.err; echo "derep rc=$?" >> $OUT/steps.log; cut -c1-700 $OUT/derep.json; grep "^\[derep\]" $OUT/derep.err ;;
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/steps.log; tail -3 $OUT/smoke.log ;;
  esac
  echo "$s took $(( $(date +%s) - t0 )) s" >> $OUT/steps.log
done
find $OUT -name "*.db" -size +4M -delete 2>/dev/null
du -sh $OUT
cat $OUT/steps.log
