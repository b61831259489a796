#!/bin/bash
# One GPU-box session: parity tests, peak micro-benchmark, headline bench, rocprofv3 summaries.
# Usage (through gpurun, from the repo root):  bash tools/gpu_round.sh <tag> [steps...]
# Everything lands in gpurun_out/<tag>/ ; the summaries worth judging are copied into profiles/ afterwards.
set -u
TAG=${1:-r02a}; shift || true
STEPS=${*:-"tests peaks bench3 bench2 prof3"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
export DADA2HIP_WAIT_TIMEOUT_S=${DADA2HIP_WAIT_TIMEOUT_S:-90}
for s in $STEPS; do
  t0=$(date +%s)
  case $s in
    tests)   timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $OUT/gputests.log 2>&1; echo "tests rc=$?" >> $OUT/steps.log; tail -5 $OUT/gputests.log ;;
    tests_fast) timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "not 1M and not config5 and not config2" --durations=10 > $OUT/gputests_fast.log 2>&1; echo "tests_fast rc=$?" >> $OUT/steps.log; tail -5 $OUT/gputests_fast.log ;;
    tests_new) timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "${TESTS_K:-user_alignment or longer_than or 4500}" --durations=10 > $OUT/gputests_new.log 2>&1; echo "tests_new rc=$?" >> $OUT/steps.log; tail -15 $OUT/gputests_new.log ;;
    nwphases5) timeout 900 python tools/nw_phases.py --config 5 --uniques ${NWP_UNIQUES:-60000} --sizes 2000,8000,20000,40000 --reps 3 > $OUT/nw_phases_cfg5.jsonl 2> $OUT/nw_phases_cfg5.err; echo "nwphases5 rc=$?" >> $OUT/steps.log; cat $OUT/nw_phases_cfg5.jsonl; tail -3 $OUT/nw_phases_cfg5.err ;;
    smoke)   timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/steps.log; tail -3 $OUT/smoke.log ;;
    bench3qchain) DADA2HIP_V2_TAIL=chain timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_cfg3_quick_chain.json 2> $OUT/bench_cfg3_quick_chain.err; echo "bench3qchain rc=$?" >> $OUT/steps.log; python3 -c "import json;b=json.load(open('$OUT/bench_cfg3_quick_chain.json'));print(b['ms_per_step'], b['resident']);print(b['phases_ms_last_step'])" ;;
    bench2q) timeout 600 python bench.py --config 2 --steps 10 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_cfg2_quick.json 2> $OUT/bench_cfg2_quick.err; echo "bench2q rc=$?" >> $OUT/steps.log; python3 -c "import json;b=json.load(open('$OUT/bench_cfg2_quick.json'));print(b['ms_per_step'], b['resident']);print(b['phases_ms_last_step'])" ;;
    subphases) for c in 2 3; do DADA2HIP_V2_SUMMARY=1 timeout 600 python bench.py --config $c --steps 1 --warmup 1 --no-cpu-baseline --no-extras 2> $OUT/subphases_cfg$c.err > $OUT/subphases_cfg$c.json; grep "sub-phase\|\[v3\] blocks" $OUT/subphases_cfg$c.err | tail -2; done; echo "subphases rc=$?" >> $OUT/steps.log ;;
    bench4sweep) for k in 1 2 3 4; do timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --inflight $k > $OUT/bench_cfg4_inflight$k.json 2> $OUT/bench_cfg4_inflight$k.err; python3 -c "import json;b=json.load(open('$OUT/bench_cfg4_inflight$k.json'));print('inflight $k', round(b['ms_per_step'],1), round(b['value']))"; done; echo "bench4sweep rc=$?" >> $OUT/steps.log ;;
    bimera)  P=$OUT/prof_bimera; mkdir -p $P
             DADA2HIP_BIMERA_TIMES=1 timeout 300 python tools/bench_bimera.py > $OUT/bench_bimera.json 2> $OUT/bench_bimera.err; echo "bimera rc=$?" >> $OUT/steps.log; cat $OUT/bench_bimera.json; grep "bimera" $OUT/bench_bimera.err
             ( cd /tmp && BIMERA_NO_REF=1 timeout 300 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- python $ROOT/tools/bench_bimera.py > $P/trace.log 2>&1 )
             python3 -c "
import sqlite3,glob
c=sqlite3.connect(glob.glob('$P/trace/**/*.db', recursive=True)[0])
for r in c.execute('select name,total_calls,total_duration,average,percentage from top_kernels limit 8'): print(r[0][:80], r[1], 'total_us', r[2], 'avg_us', r[3], 'pct', round(r[4],2))
" | tee $OUT/bimera_kernels.txt ;;
    occ)     timeout 300 tools/microbench occ > $OUT/occ.json 2> $OUT/occ.err; echo "occ rc=$?" >> $OUT/steps.log; cat $OUT/occ.json ;;
    launch)  timeout 300 tools/microbench launch > $OUT/launch.json 2> $OUT/launch.err; echo "launch rc=$?" >> $OUT/steps.log; cat $OUT/launch.json ;;
    tests_iter) timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -rf -p no:cacheprovider -k "not at_size and not 1M" --durations=10 > $OUT/gputests_iter.log 2>&1; echo "tests_iter rc=$?" >> $OUT/steps.log; tail -5 $OUT/gputests_iter.log ;;
    peaks)   timeout 300 tools/microbench peaks > $OUT/peaks.json 2> $OUT/peaks.err; echo "peaks rc=$?" >> $OUT/steps.log; cat $OUT/peaks.json ;;
    bench3)  timeout 1200 python bench.py --steps 5 --warmup 2 > $OUT/bench_cfg3.json 2> $OUT/bench_cfg3.err; echo "bench3 rc=$?" >> $OUT/steps.log; cut -c1-600 $OUT/bench_cfg3.json ;;
    bench3q) timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_cfg3_quick.json 2> $OUT/bench_cfg3_quick.err; echo "bench3q rc=$?" >> $OUT/steps.log; cut -c1-400 $OUT/bench_cfg3_quick.json; python3 -c "import json;b=json.load(open('$OUT/bench_cfg3_quick.json'));print(b['resident']);print(b['phases_ms_last_step'])" ;;
    nwphases) timeout 600 python tools/nw_phases.py --sizes 4000,8700,18000,36000 > $OUT/nw_phases.jsonl 2> $OUT/nw_phases.err; echo "nwphases rc=$?" >> $OUT/steps.log; cat $OUT/nw_phases.jsonl ;;
    trace)   timeout 600 python tools/trace_round.py --seqs ${TRACE_SEQS:-60,300,600} > $OUT/round_trace.jsonl 2> $OUT/round_trace.err; echo "trace rc=$?" >> $OUT/steps.log; cat $OUT/round_trace.jsonl; tail -3 $OUT/round_trace.err ;;
    bench3shard) timeout 900 python bench.py --shard --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > $OUT/bench_cfg3_shard1.json 2> $OUT/bench_cfg3_shard1.err; echo "bench3shard rc=$?" >> $OUT/steps.log; cut -c1-500 $OUT/bench_cfg3_shard1.json; tail -3 $OUT/bench_cfg3_shard1.err ;;
    tests_shard) timeout 900 python -m pytest tests/test_shard.py -m gpu -q -p no:cacheprovider > $OUT/gputests_shard.log 2>&1; echo "tests_shard rc=$?" >> $OUT/steps.log; tail -15 $OUT/gputests_shard.log ;;
    bench3full) timeout 1200 python bench.py --steps 5 --warmup 2 --cpu-full > $OUT/bench_cfg3_cpufull.json 2> $OUT/bench_cfg3_cpufull.err; echo "bench3full rc=$?" >> $OUT/steps.log; cut -c1-600 $OUT/bench_cfg3_cpufull.json ;;
    bench3sc) timeout 900 python bench.py --steps 2 --warmup 1 --selfconsist > $OUT/bench_cfg3_selfconsist.json 2> $OUT/bench_cfg3_selfconsist.err; echo "bench3sc rc=$?" >> $OUT/steps.log; cut -c1-600 $OUT/bench_cfg3_selfconsist.json ;;
    bench2)  timeout 600 python bench.py --config 2 --steps 10 --warmup 2 > $OUT/bench_cfg2.json 2> $OUT/bench_cfg2.err; echo "bench2 rc=$?" >> $OUT/steps.log; cut -c1-400 $OUT/bench_cfg2.json ;;
    bench2deep) timeout 1200 python bench.py --config 2 --deep --steps 3 --warmup 1 > $OUT/bench_cfg2_deep.json 2> $OUT/bench_cfg2_deep.err; echo "bench2deep rc=$?" >> $OUT/steps.log; cut -c1-400 $OUT/bench_cfg2_deep.json ;;
    bench4)  timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline ${BENCH4_ARGS:-} > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "bench4 rc=$?" >> $OUT/steps.log; cut -c1-400 $OUT/bench_cfg4.json ;;
    bench5)  timeout 1200 python bench.py --config 5 --steps 2 --warmup 1 ${BENCH5_ARGS:-} > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; echo "bench5 rc=$?" >> $OUT/steps.log; cut -c1-400 $OUT/bench_cfg5.json ;;
    prof3|prof2|prof5)
             CFG=${s#prof}; P=$OUT/prof$CFG; mkdir -p $P
             CMD="python $ROOT/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras ${PROF_ARGS:-}"
             ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- $CMD > $P/trace.log 2>&1 ); echo "$s rc=$?" >> $OUT/steps.log
             python3 profiles/summarize.py $P ${TAG}_cfg$CFG $OUT/summaries "python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras" > $P/summarize.log 2>&1 ;;
    pmc3|pmc2|pmc5)
             CFG=${s#pmc}; P=$OUT/prof$CFG; mkdir -p $P
             CMD="python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras ${PROF_ARGS:-}"
             ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o fetch -- $CMD > $P/pmc_fetch.log 2>&1 )
             ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o write -- $CMD > $P/pmc_write.log 2>&1 )
             ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $P/pmc_valu -o valu -- $CMD > $P/pmc_valu.log 2>&1 )
             python3 profiles/summarize.py $P ${TAG}_cfg$CFG $OUT/summaries "python bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras" > $P/summarize.log 2>&1
             echo "$s done" >> $OUT/steps.log ;;
    bench3t16|bench3t64|bench3t128) T=${s#bench3t}; DADA2HIP_HOST_THREADS=$T timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-profile-pass > $OUT/bench_cfg3_threads$T.json 2> $OUT/bench_cfg3_threads$T.err; echo "$s rc=$?" >> $OUT/steps.log; cut -c1-300 $OUT/bench_cfg3_threads$T.json ;;
    sweep)   timeout 600 python tools/sweep_env.py --config 3 --reps 3 --list "${SWEEP:-}" > $OUT/sweep.jsonl 2> $OUT/sweep.err; echo "sweep rc=$?" >> $OUT/steps.log; cut -c1-420 $OUT/sweep.jsonl ;;
    wcal)    for m in store_bytes store_wide read_wide; do ( cd /tmp && timeout 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/wcal_${m}_W -o pmc -- $ROOT/tools/microbench $m 268435456 > $OUT/wcal_${m}_W.log 2>&1 )
               ( cd /tmp && timeout 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/wcal_${m}_F -o pmc -- $ROOT/tools/microbench $m 268435456 > $OUT/wcal_${m}_F.log 2>&1 ); done
             echo "wcal done" >> $OUT/steps.log ;;
  esac
  echo "$s took $(( $(date +%s) - t0 )) s" >> $OUT/steps.log
done
# keep the pulled directory small: rocprof's raw databases are large, the CSVs are what is summarised
find $OUT -name "*.db" -size +4M -delete 2>/dev/null
du -sh $OUT
cat $OUT/steps.log
