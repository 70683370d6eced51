#!/bin/bash
# Runs on the GPU box (via gpurun): configurations through tools/configs.py for several builds of the library
# (librabft_simulator_amd/liblbft_hip_<tag>.so, made with build.build_variant) -- A/B measurements of tuning switches.
#   bash tools/gpu_variants.sh "v0 v1 v2" "c4_16384x64_longtail_equivocators c5_8192x100_weighted_epochs"
set -u
TAGS=${1:-""}
CONFIGS=${2:-"c4_16384x64_longtail_equivocators"}
export TMPDIR=/tmp
mkdir -p gpurun_out
for tag in $TAGS; do
  lib=$PWD/librabft_simulator_amd/liblbft_hip_$tag.so
  [ "$tag" = "prod" ] && lib=$PWD/librabft_simulator_amd/liblbft_hip.so
  for lean in 0 1; do
    LBFT_NO_LEAN=$lean LBFT_HIP_LIB=$lib timeout 200 python tools/configs.py $CONFIGS 2>> gpurun_out/variants.err | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print('$tag', 'no_lean=$lean', d['config'][:14], round(d['kernel_ms'], 1), 'faulted', d['faulted_instances'], 'lpw', d['layout']['lanes_per_wavefront'], 'class', d['layout']['kernel_class'], flush=True)
"
  done
done
