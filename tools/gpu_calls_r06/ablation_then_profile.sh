#!/bin/bash
# Round 6: timing ablations of the current kernel (diag library), then the rocprofv3 set of the headline workload (stats, FETCH / WRITE, pipe counters)
tools/gpu_calls_r06/diag_ablation.sh
unset GSV_ENGINE_SO GSV_DIAG AB_SHAPES
tools/profile_r06.sh
