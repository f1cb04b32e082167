#!/bin/bash
# SQ counters of the two latency-bound shapes (tools/pmc_wide.py with PMC_SHAPE=ladder / inverse: 1 024 instances, four per workgroup), one
# rocprofv3 --pmc pass per counter group; the program itself stands after `--` (PMC_SHAPE is exported before rocprofv3 starts).
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for shape in ladder inverse; do
  export PMC_SHAPE=$shape
  OUT=$R/gpurun_out/prof_r05_narrow_$shape
  mkdir -p $OUT
  echo "pass sq: SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU" >> $OUT/passes.txt
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VALU --output-format csv -d $OUT/sq -- python3 $R/tools/pmc_wide.py > $OUT/run_sq.log 2> $OUT/run_sq.err
  echo "pass sq rc=$? $(tail -1 $OUT/run_sq.log)" >> $OUT/passes.txt
  echo "pass sq2: SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA" >> $OUT/passes.txt
  timeout 600 rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/sq2 -- python3 $R/tools/pmc_wide.py > $OUT/run_sq2.log 2> $OUT/run_sq2.err
  echo "pass sq2 rc=$? $(tail -1 $OUT/run_sq2.log)" >> $OUT/passes.txt
  (cd $R && python3 $R/tools/pipe_util.py $OUT > $OUT/pipe_util.log 2>&1)
  find $OUT -name "*.csv" -size +6M -delete
  find $OUT -name "*.db" -delete
  cat $OUT/passes.txt; grep -A40 "run_program_kernel" $OUT/pipe_util.log | head -60
done
