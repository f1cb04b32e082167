#!/bin/bash
# Round 6: the production library with the per-group step barrier in the FW instantiations on the A/B shapes, then the full GPU suite
AB_NO_PARITY=1 AB_SHAPES=wide,ladder,inv_grp tools/kernel_ab_r06.sh gbar_adopted base
tools/gpu_calls_r06/gpu_suite.sh
