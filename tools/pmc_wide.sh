#!/bin/bash
# Vector-memory-path counters of the wide shape (tools/pmc_wide.py), one rocprofv3 --pmc pass per small counter group; the program
# itself stands after `--`.  Output: gpurun_out/prof_<tag>/<group>/, summary by tools/pipe_util.py.
TAG=${1:-r05_vmem}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cp $R/profiles/r05_final/counters_available.txt $OUT/counters_available.txt  # (`rocprofv3 -L` of the first run of the round: listing counters initialises the GPU and execs a helper, which the box refuses)
pass() {
  local name=$1; shift
  local have=""
  for c in "$@"; do if grep -qw "$c" $OUT/counters_available.txt; then have="$have $c"; else echo "counter $c not on this box" >> $OUT/skipped_counters.txt; fi; done
  [ -z "$have" ] && return
  echo "pass $name:$have" >> $OUT/passes.txt
  timeout 600 rocprofv3 --pmc $have --output-format csv -d $OUT/$name -- python3 $R/tools/pmc_wide.py > $OUT/run_$name.log 2> $OUT/run_$name.err
  echo "pass $name rc=$? $(tail -1 $OUT/run_$name.log)" >> $OUT/passes.txt
}
pass sq  SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU
pass sq2 SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU
pass ta1 TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum
pass ta2 TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum
pass ta3 TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum
pass tcp2 TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcp3 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TOTAL_ACCESSES_sum
pass tcp4 TCP_GATE_EN1_sum TCP_TA_TCP_STATE_READ_sum
pass grbm GRBM_GUI_ACTIVE GRBM_COUNT
cd $R
python3 $R/tools/pipe_util.py $OUT > $OUT/pipe_util.log 2>&1
find $OUT -name "*.csv" -size +6M -delete
find $OUT -name "*.db" -delete
cat $OUT/passes.txt; grep -v "^   ->" $OUT/pipe_util.log | head -80
