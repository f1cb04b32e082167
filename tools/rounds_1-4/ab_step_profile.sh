#!/bin/bash
# step_profile.py of one circuit (one instance) under several engine builds: usage ab_step_profile.sh <tag> <circuit> <variant> [variant ...]
TAG=$1; SPEC=$2; shift; shift
mkdir -p gpurun_out/r04_kernel
out=gpurun_out/r04_kernel/step_profile_$TAG.log
: > $out
for v in "$@"; do
  if [ "$v" = base ]; then unset GSV_ENGINE_SO; else export GSV_ENGINE_SO=$PWD/garbled_snark_verifier_amd/libgsv_engine_$v.so; fi
  echo "== $v" >> $out
  python tools/step_profile.py $SPEC 1 2>&1 | head -12 >> $out
done
cat $out
