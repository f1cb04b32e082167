#!/bin/bash
# Round-5 pipe-utilisation counters of bench.py's headline workload (1 024 instances of the one-public-input verifier plan; a kernel
# dispatch = one WINDOW of the session's schedule).  What binds the wide windows: the LDS pipe of the T-table AES, the VALU, or waiting?
# Separate `rocprofv3 --pmc` passes (SQ has 8 slots per pass; never combined with a trace domain; the program itself stands after `--`),
# each over ONE full verifier pass.  The counter names are filtered against `rocprofv3 -L` of the box, so a renamed counter drops out
# instead of failing the pass.  Outputs under gpurun_out/prof_<tag>/; tools/pipe_util.py turns them into pipe_util.json.
TAG=${1:-r05_pipe}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAGS="--no-cpu-baseline --no-e2e --no-rate-by-instances --no-mode-rates --no-cc16 --no-headline-ct-check"
cp $R/profiles/r05_final/counters_available.txt $OUT/counters_available.txt  # (`rocprofv3 -L` of the first run of the round: listing counters initialises the GPU and execs a helper, which the box refuses)
python3 $R/bench.py --steps 1 --warmup 0 $FLAGS > $OUT/bench_plan_build.log 2> $OUT/bench_plan_build.err   # builds + saves the plan file
pass() {  # pass <name> <counters...>
  local name=$1; shift
  local have=""
  for c in "$@"; do if grep -qw "$c" $OUT/counters_available.txt; then have="$have $c"; else echo "counter $c not on this box" >> $OUT/skipped_counters.txt; fi; done
  [ -z "$have" ] && return
  echo "pass $name:$have" >> $OUT/passes.txt
  timeout 900 rocprofv3 --pmc $have --output-format csv -d $OUT/$name -- python3 $R/bench.py --steps 10 --warmup 0 $FLAGS > $OUT/bench_$name.log 2> $OUT/bench_$name.err
  echo "pass $name rc=$?" >> $OUT/passes.txt
}
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
pass sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM
pass sq3 SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_WAVES SQ_INST_CYCLES_VMEM
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $R
python3 $R/tools/pipe_util.py $OUT > $OUT/pipe_util.log 2>&1
find $OUT -name "*.csv" -size +6M -delete
find $OUT -name "*.db" -delete
tail -n 40 $OUT/pipe_util.log; cat $OUT/passes.txt; tail -n 3 $OUT/bench_sq1.err
