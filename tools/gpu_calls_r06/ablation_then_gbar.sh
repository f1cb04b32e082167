#!/bin/bash
# Round 6: timing ablations of the current kernel (diag library; window launches honour GSV_DIAG now), then the per-group step barrier A/B
tools/gpu_calls_r06/diag_ablation.sh
unset GSV_ENGINE_SO GSV_DIAG
AB_SHAPES=wide,ladder,inv_grp tools/kernel_ab_r06.sh gbar base gbar1 gbar2
