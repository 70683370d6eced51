#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/live
mkdir -p $O; rm -f $O/configs.txt
run() { tag=$1; lib=$2; shift 2; LBFT_HIP_LIB=$PWD/librabft_simulator_amd/$lib timeout 900 python tools/configs.py --reps 2 "$@" 2>> $O/configs.err | python -c "
import json, sys
for l in sys.stdin:
    d = json.loads(l)
    print('$tag', d['config'][:16], round(d['kernel_ms'], 2), 'faulted', d['faulted_instances'], 'rounds', d['rounds'], 'commits', d['commits'], 'events', d['events'], flush=True)
" | tee -a $O/configs.txt; }
C4L=c4live_16384x64_longtail_equivocators_fixed; C5L=c5live_8192x100_rotating_rights_epochs_fixed
for v in $VARIANTS; do run $v liblbft_hip_$v.so $C4L $C5L; done
for v in $VARIANTS; do run $v liblbft_hip_$v.so $C4L; done
