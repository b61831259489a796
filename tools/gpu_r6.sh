#!/bin/bash
# Round 6 GPU sessions (through gpurun, from the repo root):  bash tools/gpu_r6.sh <tag> <steps...>
# Results land in gpurun_out/<tag>/ ; what is worth judging is copied into profiles/ afterwards.
set -u
TAG=${1:-r09a}; shift || true
STEPS=${*:-"sub3 sweep3 sweep2deep sc"}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
export TMPDIR=/tmp
export DADA2HIP_WAIT_TIMEOUT_S=${DADA2HIP_WAIT_TIMEOUT_S:-90}
(rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4; rocm-smi --showpower 2>/dev/null | grep -i "power" | head -2) > $OUT/clocks.txt 2>&1
for s in $STEPS; do
  t0=$(date +%s)
  case $s in
    sub3)   DADA2HIP_V2_SUMMARY=1 timeout 600 python bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras 2> $OUT/sub3.err > $OUT/sub3.json; echo "sub3 rc=$?" >> $OUT/steps.log
            grep "sub-phase" $OUT/sub3.err | tail -1; python3 -c "import json;b=json.load(open('$OUT/sub3.json'));print(b['ms_per_step'], b['resident']['ms_per_pass']);print(json.dumps(b['phases_ms_last_step']))" ;;
    sweep3) timeout 900 python tools/sweep_env.py --config 3 --reps ${REPS:-3} --list "${SWEEP3:-DADA2HIP_V3_PF_LOWREG=0;DADA2HIP_V3_OVERLAP=0}" > $OUT/sweep3.jsonl 2> $OUT/sweep3.err; echo "sweep3 rc=$?" >> $OUT/steps.log; cut -c1-700 $OUT/sweep3.jsonl ;;
    sweep2deep) timeout 600 python tools/sweep_env.py --config 2 --deep --reps ${REPS:-3} --list "${SWEEP2:-DADA2HIP_V3_OVERLAP=0;DADA2HIP_V3_PF_LOWREG=0}" > $OUT/sweep2deep.jsonl 2> $OUT/sweep2deep.err; echo "sweep2deep rc=$?" >> $OUT/steps.log; cut -c1-700 $OUT/sweep2deep.jsonl ;;
    sweep2) timeout 600 python tools/sweep_env.py --config 2 --reps ${REPS:-3} --list "${SWEEP2P:-DADA2HIP_V3_OVERLAP=0}" > $OUT/sweep2.jsonl 2> $OUT/sweep2.err; echo "sweep2 rc=$?" >> $OUT/steps.log; cut -c1-700 $OUT/sweep2.jsonl ;;
    sc)     DADA2HIP_V2_SUMMARY=1 timeout 600 python bench.py --selfconsist --steps 1 --warmup 1 --no-cpu-baseline > $OUT/sc.json 2> $OUT/sc.err; echo "sc rc=$?" >> $OUT/steps.log
            grep "^\[v3\] blocks\|^\[v2\]\|overlap:" $OUT/sc.err | tail -24; python3 -c "import json;b=json.load(open('$OUT/sc.json'));print(b['selfconsist'])" ;;
    scprof) DADA2HIP_V2_SUMMARY=1 DADA2HIP_PROFILE=1 timeout 600 python bench.py --selfconsist --steps 1 --warmup 0 --no-cpu-baseline > $OUT/scprof.json 2> $OUT/scprof.err; echo "scprof rc=$?" >> $OUT/steps.log
            grep "sub-phase\|^\[v3\] blocks" $OUT/scprof.err | tail -14 ;;
    tests)  timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $OUT/gputests.log 2>&1; echo "tests rc=$?" >> $OUT/steps.log; tail -5 $OUT/gputests.log ;;
    tests_iter) timeout 1200 python -X faulthandler -m pytest tests -m gpu -q -rf -p no:cacheprovider -k "not at_size and not 1M" --durations=10 > $OUT/gputests_iter.log 2>&1; echo "tests_iter rc=$?" >> $OUT/steps.log; tail -5 $OUT/gputests_iter.log ;;
    tests_k) timeout 1500 python -m pytest tests -m gpu -q -rf -p no:cacheprovider -k "${TESTS_K}" --durations=10 > $OUT/gputests_k.log 2>&1; echo "tests_k rc=$?" >> $OUT/steps.log; tail -15 $OUT/gputests_k.log ;;
    smoke)  timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?" >> $OUT/steps.log; tail -3 $OUT/smoke.log ;;
    bench)  timeout 1500 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err; echo "bench rc=$?" >> $OUT/steps.log; cut -c1-600 $OUT/bench_default.json ;;
    bench3q) timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $OUT/bench_cfg3_quick.json 2> $OUT/bench_cfg3_quick.err; echo "bench3q rc=$?" >> $OUT/steps.log; python3 -c "import json;b=json.load(open('$OUT/bench_cfg3_quick.json'));print(b['ms_per_step'], b['resident']);print(b['phases_ms_last_step'])" ;;
    bench5) timeout 1200 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline ${BENCH5_ARGS:-} > $OUT/bench_cfg5.json 2> $OUT/bench_cfg5.err; echo "bench5 rc=$?" >> $OUT/steps.log; python3 -c "import json;b=json.load(open('$OUT/bench_cfg5.json'));print(b['ms_per_step'], b['roofline']);print(b['phases_ms_last_step'])" ;;
    bench4) timeout 900 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass > $OUT/bench_cfg4.json 2> $OUT/bench_cfg4.err; echo "bench4 rc=$?" >> $OUT/steps.log; cut -c1-300 $OUT/bench_cfg4.json ;;
    bench2deep) timeout 600 python bench.py --config 2 --deep --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_cfg2_deep.json 2> $OUT/bench_cfg2_deep.err; echo "bench2deep rc=$?" >> $OUT/steps.log; cut -c1-300 $OUT/bench_cfg2_deep.json ;;
    nwphases) timeout 600 python tools/nw_phases.py --sizes 8700,36000 > $OUT/nw_phases.jsonl 2> $OUT/nw_phases.err; echo "nwphases rc=$?" >> $OUT/steps.log; cat $OUT/nw_phases.jsonl ;;
    nwphases5) timeout 900 python tools/nw_phases.py --config 5 --uniques ${NWP_UNIQUES:-60000} --sizes 8000,40000 --reps 3 > $OUT/nw_phases_cfg5.jsonl 2> $OUT/nw_phases_cfg5.err; echo "nwphases5 rc=$?" >> $OUT/steps.log; cat $OUT/nw_phases_cfg5.jsonl; tail -3 $OUT/nw_phases_cfg5.err ;;
    prof3|prof2|prof5)
            CFG=${s#prof}; P=$OUT/prof$CFG; mkdir -p $P
            CMD="python $ROOT/bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras ${PROF_ARGS:-}"
            ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- $CMD > $P/trace.log 2>&1 ); echo "$s rc=$?" >> $OUT/steps.log
            python3 profiles/summarize.py $P ${TAG}_cfg$CFG $OUT/summaries "python bench.py --config $CFG --steps 2 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras" > $P/summarize.log 2>&1 ;;
    profsc) P=$OUT/profsc; mkdir -p $P
            CMD="python $ROOT/bench.py --selfconsist --steps 1 --warmup 0 --no-cpu-baseline"
            ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $P/trace -o trace -- $CMD > $P/trace.log 2>&1 ); echo "$s rc=$?" >> $OUT/steps.log
            python3 profiles/summarize.py $P ${TAG}_selfconsist $OUT/summaries "python bench.py --selfconsist --steps 1 --warmup 0 --no-cpu-baseline" > $P/summarize.log 2>&1 ;;
    pmc3|pmc2|pmc5)
            CFG=${s#pmc}; P=$OUT/prof$CFG; mkdir -p $P
            CMD="python $ROOT/bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras ${PROF_ARGS:-}"
            ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $P/pmc_fetch -o fetch -- $CMD > $P/pmc_fetch.log 2>&1 )
            ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $P/pmc_write -o write -- $CMD > $P/pmc_write.log 2>&1 )
            ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES -d $P/pmc_valu -o valu -- $CMD > $P/pmc_valu.log 2>&1 )
            ( cd /tmp && timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS -d $P/pmc_lds -o lds -- $CMD > $P/pmc_lds.log 2>&1 )
            python3 profiles/summarize.py $P ${TAG}_cfg$CFG $OUT/summaries "python bench.py --config $CFG --steps 1 --warmup 1 --no-cpu-baseline --no-profile-pass --no-extras" > $P/summarize.log 2>&1
            echo "$s done" >> $OUT/steps.log ;;
  esac
  echo "$s took $(( $(date +%s) - t0 )) s" >> $OUT/steps.log
done
find $OUT -name "*.db" -size +4M -delete 2>/dev/null
du -sh $OUT
cat $OUT/steps.log
